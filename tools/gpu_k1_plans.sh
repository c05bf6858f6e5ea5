mkdir -p gpurun_out/r5k1 && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5k1
for K in 1 2 5; do
IDQN_PLAN_PRINT=1 timeout -k 10 200 python bench.py --heads $K --steps 300 --warmup 30 --repeats 3 --no-cpu-baseline > $O/k$K.json 2> $O/k$K.err || { echo "K=$K failed"; tail -5 $O/k$K.err; }
grep "^\[plan\]" $O/k$K.err | sort -u
python - $K <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5k1/k%s.json" % sys.argv[1]))
print("K=%s %7.1f steps/s %.4f ms" % (sys.argv[1], d["value"], d["ms_per_step"]))
for k in d["kernels"]: print("      %-40s %7.1f us" % (k["launch"], k["us"]))
PY
done
