mkdir -p gpurun_out/r5k1 && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5k1
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for K in 1 2; do
for w in 0 32 64 128 0 64; do
  cfg="IDQN_HIP_LIB=$V"; [ $w != 0 ] && cfg="$cfg IDQN_WCHUNKS=$w"
  env $cfg timeout -k 10 200 python bench.py --heads $K --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$w" $K <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5k1/ab.json"))
k = [(x["launch"], x["us"]) for x in d["kernels"] if x["launch"].startswith(("conv2 d", "conv1 d", "conv0 w", "adam"))]
print("K=%s WCHUNKS=%-4s %.4f ms | %s" % (sys.argv[2], sys.argv[1], d["ms_per_step"], "  ".join("%s %.1f" % x for x in k)))
PY
done
done
