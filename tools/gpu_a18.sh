#!/bin/bash
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_a18; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_a18 -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --actions 18 > gpurun_out/prof_a18.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_a18/*/*_kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
for r in rows[:16]: print(f"{r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
