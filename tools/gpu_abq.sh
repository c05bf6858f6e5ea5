# quick A/B of builds of the library on ONE box: bash tools/gpu_abq.sh <lib.so>...   (paths relative to the repo root)
# per build: two bench runs (median-of-3 x 300 steps) interleaved over the builds, printing steps/s and the per-launch table
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
for round in ${ROUNDS:-1 2}; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  IDQN_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --steps 300 --warmup 50 --repeats 3 --no-cpu-baseline > gpurun_out/abq_${tag}_$round.json 2> gpurun_out/abq_${tag}_$round.err || { echo "$tag failed"; tail -5 gpurun_out/abq_${tag}_$round.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/abq_${tag}_$round.json"))
print("$tag round $round: %.1f steps/s  %.4f ms/step  dominant %.1f us" % (d["value"], d["ms_per_step"], d["roofline"]["launch_ms"] * 1e3))
if $round == ${LAST:-2}:
    for k in d["kernels"]: print("    %-40s %7.1f us" % (k["launch"], k["us"]))
PY
done
done
