# round 5, first GPU call: the GPU suite at the pruned build + the new bench lines
mkdir -p gpurun_out/r5a && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 200 python bench.py --batch 256 --steps 200 --warmup 20 --repeats 3 > $O/bench_b256.json 2> $O/bench_b256.err; echo "b256 rc=$?"
timeout -k 10 300 python bench.py --heads 64 --steps 60 --warmup 10 --repeats 3 > $O/bench_k64.json 2> $O/bench_k64.err; echo "k64 rc=$?"
for st in side inline; do timeout -k 10 200 python bench.py --gpus 1 --force-dp --dp-streams $st --steps 300 --repeats 3 > $O/bench_dp1_native_$st.json 2> $O/dp1_native_$st.err; echo "dp native $st rc=$?"; done
IDQN_DP_MODE=factored timeout -k 10 200 python bench.py --gpus 1 --force-dp --steps 300 --repeats 3 > $O/bench_dp1_factored.json 2> $O/dp1_factored.err; echo "dp factored rc=$?"
timeout -k 10 200 python bench.py --emulate-ranks 1 --steps 300 --repeats 3 > $O/bench_emulate1.json 2> $O/emulate1.err; echo "emulate1 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5a/bench*.json")):
    try:
        x=json.load(open(f)); print(f.split("/")[-1], "%.1f %s  %.4f ms/step" % (x["value"], x["unit"], x["ms_per_step"]), x.get("roofline",{}) and ("dom %.1f us frac %.3f" % (x["roofline"]["launch_ms"]*1e3, x["roofline"]["frac"])), x.get("step_roofline",{}) and ("step frac_mfma %.3f" % x["step_roofline"]["frac_mfma"]), x.get("heads_fit",""))
        for k in x.get("kernels",[]): print("    %-40s %7.1f us" % (k["launch"], k["us"]))
    except Exception as e: print(f, "failed", e)
PY
