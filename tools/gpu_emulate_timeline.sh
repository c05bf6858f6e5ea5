mkdir -p gpurun_out/r5em && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5em
for N in 1 8; do
timeout -k 10 300 python bench.py --emulate-ranks $N --steps 300 --warmup 30 --repeats 3 --no-cpu-baseline > $O/em$N.json 2> $O/em$N.err || { echo "N=$N failed"; tail -5 $O/em$N.err; }
python - $N <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5em/em%s.json" % sys.argv[1]))
print("N=%s %.4f ms" % (sys.argv[1], d["ms_per_step"]))
for k in d.get("kernels", []): print("      %-40s %7.1f us" % (k["launch"], k["us"]))
print({k: v for k, v in d.items() if k not in ("kernels", "config", "roofline", "step_roofline", "timing")})
PY
done
