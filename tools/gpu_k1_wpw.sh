mkdir -p gpurun_out/r5k1 && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5k1
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu -k "fewer_heads or dqn" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -2 $O/parity.log
for w in 0 4 1 0 4 1; do
  cfg="IDQN_HIP_LIB=$V"; [ $w != 0 ] && cfg="$cfg IDQN_D0_FWD_WPW=$w"
  env $cfg timeout -k 10 200 python bench.py --heads 1 --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$w" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5k1/ab.json"))
k = [(x["launch"], x["us"]) for x in d["kernels"] if x["launch"].startswith(("dense0 fwd", "hidden"))]
print("K=1 waves per workgroup %-10s %.4f ms | %s" % (sys.argv[1] if sys.argv[1] != "0" else "2 (default)", d["ms_per_step"], "  ".join("%s %.1f" % x for x in k)))
PY
done
