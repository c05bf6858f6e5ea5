# round 5: few heads: every stream of the fused Dense_0 update default-policy (IDQN_D0_KEEP_ALL = 1: theta, m, v stay in the memory-side cache) against
# m / v non-temporal (= 0), HEADS="1 2 3 4 5"; variants build, interleaved on one box
mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
CFG=()
for K in ${HEADS:-1 2 3 4 5}; do for r in 1 2; do CFG+=("--heads $K|IDQN_D0_KEEP_ALL=0" "--heads $K|IDQN_D0_KEEP_ALL=1"); done; done
for cfg in "${CFG[@]}"; do
  args=${cfg%%|*}; envs=${cfg##*|}
  env IDQN_HIP_LIB=$V $envs timeout -k 10 200 python bench.py $args --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$args" "$envs" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5pol/ab.json"))
k = {x["launch"]: x["us"] for x in d["kernels"]}
print("%-10s %-22s %.4f ms | dense0 fwd %.1f  update %.1f" % (sys.argv[1], sys.argv[2], d["ms_per_step"], k["dense0 fwd"], k["dense0 wgrad + dgrad + adam"]))
PY
done
