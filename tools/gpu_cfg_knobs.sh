# env-knob sweep of one bench configuration on ONE box: bash tools/gpu_cfg_knobs.sh "<bench args>" "NAME=VAL …" "…" ; "" = defaults
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
args=$1; shift
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg IDQN_PLAN_PRINT=1 timeout -k 10 200 python bench.py --steps 100 --warmup 20 --repeats 3 --no-cpu-baseline $args > gpurun_out/cknob_$i.json 2> gpurun_out/cknob_$i.err || { echo "[$cfg] failed"; tail -5 gpurun_out/cknob_$i.err; continue; }
  python - "$cfg" <<PY
import json, sys
d = json.load(open("gpurun_out/cknob_$i.json"))
print("%-40s %7.1f steps/s %.4f ms | " % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"]) + " ".join("%s %.1f" % (x["launch"].split(" (")[0].replace(" ", "_"), x["us"]) for x in d["kernels"]))
PY
  grep "^\[plan\]" gpurun_out/cknob_$i.err | grep -v "parts [0-9]*:" | sort -u | head -24
done
