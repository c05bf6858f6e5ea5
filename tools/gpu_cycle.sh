mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests -q -m gpu > gpurun_out/gpu_all.log 2>&1; tail -4 gpurun_out/gpu_all.log
timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err; python - <<'PY'
import json
try:
    d=json.load(open('gpurun_out/bench_cur.json')); print("BENCH value %.1f steps/s  ms/step %.3f  dom %.1f us  %.0f GB/s frac %.3f"%(d['value'],d['ms_per_step'],d['roofline']['launch_ms']*1e3,d['roofline']['achieved'],d['roofline']['frac']))
except Exception as e: print("bench failed", e); print(open('gpurun_out/bench_cur.err').read()[-2000:])
PY
rm -rf gpurun_out/prof_cur; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cur -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_cur.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_cur/*/*_kernel_stats.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    tot=sum(float(r['TotalDurationNs']) for r in rows if int(r['Calls'])>=100)
    for r in rows[:18]:
        print(f"{r['Name'][:52]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
    print("sum of kernel time per step (us): %.1f"%(tot/120/1e3))
PY
