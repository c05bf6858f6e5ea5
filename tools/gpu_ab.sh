# A/B of two builds of the library: bash tools/gpu_ab.sh <libA.so> <libB.so>   (paths relative to the repo root)
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for lib in "$@"; do
  tag=$(basename $lib .so)
  IDQN_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err && python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print('$tag', round(d['value'],1), round(d['ms_per_step'],4))"
  rm -rf gpurun_out/prof_$tag; IDQN_HIP_LIB=$PWD/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_$tag.log 2>&1
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_$tag/*/*_kernel_stats.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:14]:
        print(f"   {r['Name'][:50]:50s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
done
