mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IDQN_MIX=1 timeout -k 10 300 python -m pytest tests/test_gpu_fp_path.py -q -x -k "goldens or every_stage or ragged" 2>&1 | tail -2
for m in 0 1 0 1; do
IDQN_MIX=$m timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mix=$m', round(d['value'],1), round(d['ms_per_step'],4), 'dom', round(d['roofline']['launch_ms']*1e3,1))"
done
rm -rf gpurun_out/prof_mix; IDQN_MIX=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mix -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_mix.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_mix/*/*_kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
for r in rows[:8]:
    print(f"{r['Name'][:52]:52s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
