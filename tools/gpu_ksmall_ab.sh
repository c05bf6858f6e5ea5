# round 5: K = 1 .. 4 with the conv data-gradient | weight-gradient pairs built for their tile counts (default) against two launches (IDQN_NO_PAIR=1, variants build)
mkdir -p gpurun_out/r5k1 && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5k1
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
timeout -k 10 900 python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_switches.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
for K in 1 2 3 4; do
for cfg in "IDQN_HIP_LIB=$V" "IDQN_HIP_LIB=$V IDQN_NO_PAIR=1" "IDQN_HIP_LIB=$V" "IDQN_HIP_LIB=$V IDQN_NO_PAIR=1"; do
  env $cfg timeout -k 10 200 python bench.py --heads $K --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$cfg" $K <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5k1/ab.json"))
k = [(x["launch"], x["us"]) for x in d["kernels"] if x["launch"].startswith(("conv2 d", "conv2 w", "conv1 d", "conv1 w"))]
print("K=%s %-22s %.4f ms | %s" % (sys.argv[2], "two launches" if "NO_PAIR" in sys.argv[1] else "pairs (default)", d["ms_per_step"], "  ".join("%s %.1f" % x for x in k)))
PY
done
done
