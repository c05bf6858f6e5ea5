# End-of-round measurement set (one GPU call): default bench line, A = 18 repeat, the driver's short run, rocprofv3 kernel stats, PMC
# passes, the configuration-4 / 5 single-device lines, the one-rank RCCL rehearsals of the data-parallel step (native C call with both
# stream modes, the Python schedule, the all-reduce variant), head-parallel line, emulated-rank lines, the i-IQN line + its kernel
# stats, the learner loop (rb.sample + step), trainer loop, MLP step, and the shipped switches against the default.  Outputs: gpurun_out/final/.
mkdir -p gpurun_out/final && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/final
PART=${PART:-all}   # a: headline set + PMC, b: other configurations, c: rehearsals / switches / loops (one gpurun call each: 1100 s limit)
if [ $PART = all ] || [ $PART = a ]; then
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench steps20 rc=$?"
timeout -k 10 300 python bench.py --actions 18 --no-cpu-baseline > $O/bench_a18.json 2> $O/bench_a18.err; echo "bench a18 rc=$?"
rm -rf $O/prof; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline --no-side-legs > $O/prof.log 2>&1
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv && echo "kernel stats ok"; rm -rf $O/prof
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -5 $O/pmc.log
fi
if [ $PART = all ] || [ $PART = b ]; then
timeout -k 10 300 python bench.py --batch 256 --steps 200 --warmup 20 --repeats 3 > $O/bench_b256.json 2> $O/bench_b256.err; echo "b256 rc=$?"
timeout -k 10 300 python bench.py --heads 64 --steps 60 --warmup 10 --repeats 3 > $O/bench_k64.json 2> $O/bench_k64.err; echo "k64 rc=$?"
fi
if [ $PART = all ] || [ $PART = c ]; then
for st in inline side; do timeout -k 10 300 python bench.py --gpus 1 --force-dp --dp-streams $st > $O/bench_dp1_native_$st.json 2> $O/dp1_native_$st.err; echo "dp native $st rc=$?"; done
IDQN_DP_MODE=factored timeout -k 10 300 python bench.py --gpus 1 --force-dp > $O/bench_dp1_python.json 2> $O/dp1_python.err; echo "dp python factored rc=$?"
IDQN_DP_MODE=allreduce timeout -k 10 300 python bench.py --gpus 1 --force-dp > $O/bench_dp1_allreduce.json 2> $O/dp1_allreduce.err; echo "dp allreduce rc=$?"
timeout -k 10 300 python bench.py --gpus 1 --heads-per-gpu 8 --steps 300 --warmup 30 > $O/bench_hp8.json 2> $O/hp8.err; echo "hp rc=$?"
for n in 1 2 4 8; do timeout -k 10 200 python bench.py --emulate-ranks $n --steps 200 --repeats 3 > $O/bench_emulate$n.json 2> $O/emulate$n.err; echo "emulate $n rc=$?"; done
fi
if [ $PART = all ] || [ $PART = b ]; then
timeout -k 10 400 python bench.py --algo iiqn --steps 30 --warmup 5 --repeats 3 > $O/bench_iiqn.json 2> $O/bench_iiqn.err; echo "iiqn rc=$?"
rm -rf $O/prof_iiqn; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_iiqn -- python bench.py --algo iiqn --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline > $O/prof_iiqn.log 2>&1
cp $O/prof_iiqn/*/*_kernel_stats.csv $O/iiqn_kernel_stats.csv && echo "iiqn kernel stats ok"; rm -rf $O/prof_iiqn
fi
if [ $PART = all ] || [ $PART = c ]; then
# the shipped library's second ways of issuing the same arithmetic, each against the default on THIS box
for sw in IDQN_NONE=1 IDQN_STEP_GRAPH=1 IDQN_CONV=f32 IDQN_NONE=2; do
  env $sw timeout -k 10 200 python bench.py --no-cpu-baseline --no-side-legs --steps 300 --repeats 3 > $O/bench_$sw.json 2> $O/bench_$sw.err; echo "$sw rc=$?"
done
for sw in IDQN_NONE=1 IDQN_CONV_PP=0; do
  env $sw timeout -k 10 200 python bench.py --batch 256 --no-cpu-baseline --no-side-legs --steps 100 --repeats 3 > $O/bench_b256_$sw.json 2> $O/bench_b256_$sw.err; echo "b256 $sw rc=$?"
  env $sw timeout -k 10 200 python bench.py --heads 64 --no-cpu-baseline --no-side-legs --steps 60 --repeats 3 > $O/bench_k64_$sw.json 2> $O/bench_k64_$sw.err; echo "k64 $sw rc=$?"
done
timeout -k 10 400 python bench.py --learner --steps 200 > $O/bench_learner.json 2> $O/bench_learner.err; echo "learner rc=$?"
timeout -k 10 300 python tools/bench_loop.py > $O/loop_all.log 2>&1; grep -E "us per|env steps" $O/loop_all.log > $O/loop.txt
timeout -k 10 200 python tools/bench_fc.py 2>/dev/null | grep -E "^fc " >> $O/loop.txt; cat $O/loop.txt
fi
python - <<'PY'
import json, os
if not os.path.exists("gpurun_out/final/bench.json"): raise SystemExit
d=json.load(open("gpurun_out/final/bench.json"))
print("BENCH %.1f steps/s  %.4f ms/step  dominant %.1f us %.0f GB/s frac %.3f  step frac_mfma %.3f frac_hbm %.3f  cpu %.2f steps/s on %d cores  x%.0f  jax: %s" % (
    d["value"], d["ms_per_step"], d["roofline"]["launch_ms"]*1e3, d["roofline"]["achieved"], d["roofline"]["frac"],
    d["step_roofline"]["frac_mfma"], d["step_roofline"]["frac_hbm"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["gpu_over_cpu"],
    d["cpu_baseline"]["jax"]["status"]))
for k in d["kernels"]: print("  %-36s %7.1f us" % (k["launch"], k["us"]))
print("heads_fit", d.get("heads_fit"))
print("sampling", json.dumps(d.get("sampling"))[:600])
import glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.json")):
    try:
        x=json.load(open(f)); print(f.split("/")[-1], "%.1f %s  %.4f ms/step" % (x["value"], x["unit"], x["ms_per_step"]))
    except Exception as e: print(f, "failed", e)
PY
