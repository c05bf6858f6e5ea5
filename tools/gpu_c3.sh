mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in f32 bf16x3-forward bf16x3 f32 bf16x3-forward; do
IDQN_CONV=$mode timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/bench_$mode.json 2> gpurun_out/bench_$mode.err && python -c "
import json; d=json.load(open('gpurun_out/bench_$mode.json')); print('$mode', round(d['value'],1), round(d['ms_per_step'],4), d['final_losses'][:2])"
done
rm -rf gpurun_out/prof_c3; IDQN_CONV=bf16x3-forward timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_c3.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_c3/*/*_kernel_stats.csv')
if f:
    rows=list(csv.DictReader(open(f[0])))
    for r in rows[:18]:
        print(f"{r['Name'][:52]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
