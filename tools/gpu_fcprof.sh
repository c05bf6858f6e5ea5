#!/bin/bash
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/prof_fc; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fc -- python tools/bench_fc.py > gpurun_out/prof_fc.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_fc/*/*_kernel_stats.csv')
rows=list(csv.DictReader(open(f[0])))
for r in rows[:8]: print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
grep "us" gpurun_out/prof_fc.log | head
