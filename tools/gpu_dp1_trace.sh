#!/bin/bash
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
rm -rf gpurun_out/trace_dp1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_dp1 -- python bench.py --gpus 1 --force-dp --no-cpu-baseline --repeats 1 --steps 60 --warmup 20 > gpurun_out/trace_dp1.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/trace_dp1/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find the step boundaries: k_stage launches
idx=[i for i,r in enumerate(rows) if 'k_stage' in r['Kernel_Name']]
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
prev_end=t0
for r in rows[a:b]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    print("%8.1f  +%6.1f  dur %6.1f  q%-3s %s" % ((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:70]))
    prev_end=max(prev_end,e)
print("step span %.1f us" % ((int(rows[b]['Start_Timestamp'])-t0)/1e3))
PY
