# conv launch plans and per-launch times when the conv launches are planned for fewer CUs (IDQN_CUS)
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for cus in "$@"; do
  echo "== IDQN_CUS=$cus"
  IDQN_CUS=$cus IDQN_PLAN_PRINT=1 timeout -k 10 200 python bench.py --steps 200 --warmup 30 --repeats 3 --no-cpu-baseline > gpurun_out/cus_$cus.json 2> gpurun_out/cus_$cus.err || { echo "failed"; tail -5 gpurun_out/cus_$cus.err; exit 1; }
  grep "^\[plan\]" gpurun_out/cus_$cus.err | sort -u
  python - <<PY
import json
d = json.load(open("gpurun_out/cus_$cus.json"))
print("  %.1f steps/s  %.4f ms/step" % (d["value"], d["ms_per_step"]))
for k in d["kernels"]: print("    %-40s %7.1f us" % (k["launch"], k["us"]))
PY
done
