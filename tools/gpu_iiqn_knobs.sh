# i-IQN bench under env knobs on one box: bash tools/gpu_iiqn_knobs.sh "ENV=VAL ..." ...
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg timeout -k 10 200 python bench.py --algo iiqn --steps 30 --warmup 5 --repeats 3 --no-cpu-baseline > gpurun_out/iq_$i.json 2> gpurun_out/iq_$i.err || { echo "[$cfg] failed"; tail -5 gpurun_out/iq_$i.err; continue; }
  python - "$cfg" <<PY
import json, sys
d = json.load(open("gpurun_out/iq_$i.json"))
print("%-40s %7.1f steps/s %.4f ms |" % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"]), " ".join("%s %.0f" % (k["launch"].replace("iqn ", "").replace(" ", "_")[:18], k["us"]) for k in d["kernels"] if k["us"] > 60))
PY
done
