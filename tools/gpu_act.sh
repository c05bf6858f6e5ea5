# acting path: parity tests that use it, latency, kernel profile
timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -q -x -k "q_values or entry or heads" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rm -rf gpurun_out/prof_act && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_act -- python tools/probes/act_loop.py > gpurun_out/prof_act.log 2>&1; python - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_act/*/*_kernel_stats.csv")
rows=list(csv.DictReader(open(f[0])))
for r in rows[:6]: print("%-60s calls %5s avg %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
grep "us per call" gpurun_out/prof_act.log
timeout -k 10 100 python tools/probes/act_loop.py 2>&1 | grep "us per"
