# env-knob sweep on ONE box: bash tools/gpu_knobs.sh "NAME=VAL NAME2=VAL2" "..." ; an empty string = defaults.  IDQN_HIP_LIB may be one of the knobs.
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python bench.py --steps 300 --warmup 50 --repeats 3 --no-cpu-baseline > gpurun_out/knob_$i.json 2> gpurun_out/knob_$i.err || { echo "[$cfg] failed"; tail -5 gpurun_out/knob_$i.err; continue; }
  python - "$cfg" <<PY
import json, sys
d = json.load(open("gpurun_out/knob_$i.json"))
k = {x["launch"]: x["us"] for x in d["kernels"]}
pick = lambda s: sum(v for n, v in k.items() if s in n)
print("%-44s %7.1f steps/s %.4f ms | d0fwd %5.1f fused %5.1f (ev %5.1f) convf %5.1f convb %5.1f small %5.1f" % (
    sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], pick("dense0 fwd"), pick("dense0 wgrad"), d["roofline"]["launch_ms"] * 1e3,
    pick("fwd") - pick("dense0 fwd"), pick("dgrad + wgrad") + pick("conv0 wgrad") + pick("conv1 ") + pick("conv2 dgrad") * 0,
    pick("stage") + pick("hidden") + pick("td +") + pick("finalize") + pick("adam (")))
PY
done
