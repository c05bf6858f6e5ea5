# A/B of the factored data-parallel update's contraction on ONE box: bench.py --emulate-ranks N under IDQN_DP_ALDS = 0 (registers),
# 1 (a3 fragments through LDS, 32 x 256 tiles), 2 (the same on 64 x 256 tiles); prints step time and final losses (bit-identical)
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
run() { # label, N, env...
  lbl="$1"; n="$2"; shift 2
  env "$@" timeout -k 10 200 python bench.py --emulate-ranks $n --steps 200 --warmup 50 --no-cpu-baseline > gpurun_out/emu_tmp.json 2> gpurun_out/emu_tmp.err || { echo "[$lbl N=$n] failed"; tail -5 gpurun_out/emu_tmp.err; return 1; }
  python - "$lbl" "$n" <<PY
import json, sys
d = json.load(open("gpurun_out/emu_tmp.json"))
print("%-28s N=%s  %.1f us/step  losses %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"] * 1e3, d["final_losses"]))
PY
}
for n in ${@:-8 4 2}; do
  for rep in 1 2; do
    run "register (ALDS=0)" $n IDQN_DP_ALDS=0 && run "alds 32 x 256" $n IDQN_DP_ALDS=1 && run "alds 64 x 256" $n IDQN_DP_ALDS=2 || exit 1
  done
done
