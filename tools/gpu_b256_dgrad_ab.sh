# round 5: B = 256 on one device: the Dense_0 data gradient of 8 sample blocks as the tiled bf16x3 GEMM (default) against one
# f32-MFMA pass per block (IDQN_NB_DGRAD_F32=1, variants build); parity of the touched paths first
mkdir -p gpurun_out/r5h && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5h
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
timeout -k 10 900 python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_configs.py tests/test_gpu_dp_native.py tests/test_gpu_switches.py -x -q -m gpu > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
for cfg in "" "IDQN_HIP_LIB=$V IDQN_NB_DGRAD_FIN=1" "" "IDQN_HIP_LIB=$V IDQN_NB_DGRAD_FIN=1" "IDQN_HIP_LIB=$V IDQN_NB_DGRAD_F32=1"; do
  env $cfg timeout -k 10 200 python bench.py --batch 256 --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline > $O/b256.json 2> $O/b256.err || { echo "[$cfg] failed"; tail -5 $O/b256.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5h/b256.json"))
print("%-60s %7.1f steps/s %.4f ms | step frac_mfma %.3f" % (sys.argv[1] or "(default)", d["value"], d["ms_per_step"], d["step_roofline"]["frac_mfma"]))
for k in d["kernels"]: print("      %-40s %7.1f us" % (k["launch"], k["us"]))
PY
done
