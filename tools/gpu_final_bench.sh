cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"; O=gpurun_out/final3; mkdir -p $O
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --actions 18 --no-cpu-baseline > $O/bench_a18.json 2> $O/bench_a18.err; echo "a18 rc=$?"
rm -rf $O/prof; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline > $O/prof.log 2>&1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv && echo "kernel stats ok"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_short.json 2> $O/bench_short.err; echo "short rc=$?"
IDQN_D0_PAIR=0 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_pair0.json 2> $O/bench_pair0.err; echo "pair0 rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_again.json 2> $O/bench_again.err; echo "again rc=$?"
python - <<'PY'
import json
for f in ("bench","bench_a18","bench_short","bench_pair0","bench_again"):
    d=json.load(open("gpurun_out/final3/%s.json"%f)); r=d["roofline"]
    print(f, "%.1f steps/s %.4f ms | dominant %.1f us frac %.3f n=%d | regions %s" % (d["value"], d["ms_per_step"], r["launch_ms"]*1e3, r["frac"], r["launches_timed"], [round(x,4) for x in d["timing"]["ms_per_step_all"]]))
PY
