# kernel durations of the MLP step loop (tools/bench_fc.py) under rocprofv3 --kernel-trace --stats
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/fcprof; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fcprof -- python tools/bench_fc.py > gpurun_out/fcprof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/fcprof/*/*_kernel_stats.csv')
for r in list(csv.DictReader(open(f[0])))[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
grep "us" gpurun_out/fcprof.log | grep -v amdgpu
rm -rf gpurun_out/fcprof
