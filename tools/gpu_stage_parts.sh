# round 5: what the kernel packing inside k_stage costs: the launch with pixels only (IDQN_STAGE_PART=1) and packs only (=2), variants build,
# timing only (the step's results are wrong with a part missing) -- the upper bound of caching the target packs / packing in k_adam
mkdir -p gpurun_out/r5st && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5st
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for p in 0 1 2 0 1 2; do
IDQN_HIP_LIB=$V IDQN_STAGE_PART=$p timeout -k 10 200 python bench.py --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/p$p.json 2> $O/p$p.err || { echo "part=$p failed"; tail -5 $O/p$p.err; continue; }
python - $p <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5st/p%s.json" % sys.argv[1]))
st = [k["us"] for k in d["kernels"] if k["launch"].startswith("stage")]
print("IDQN_STAGE_PART=%s  %.4f ms/step   stage launch %.1f us (timeline)" % (sys.argv[1], d["ms_per_step"], st[0] if st else -1))
PY
done
