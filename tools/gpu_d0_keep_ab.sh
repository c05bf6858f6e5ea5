# round 5: the pieces of "keep the online Dense_0 kernels on chip" at K = 5 (variants build, interleaved on one box; CFGS="A=1 B=2,C=3 ..." overrides the list):
# default (forward: online nets default-policy loads, target nets nt; update: theta_new default-policy store) / target nets default-policy too / every
# stream nt (rounds 3-4) / only the forward's online loads default-policy
mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for round in ${ROUNDS:-1 2}; do
for cfg in ${CFGS:-IDQN_NONE=1 IDQN_D0_FWD_NT_FROM=10 IDQN_D0_KEEP=0 IDQN_D0_KEEP=0,IDQN_D0_FWD_NT_FROM=5}; do
  env IDQN_HIP_LIB=$V $(echo $cfg | tr "," " ") timeout -k 10 200 python bench.py --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5pol/ab.json"))
k = {x["launch"]: x["us"] for x in d["kernels"]}
conv = sum(v for n, v in k.items() if n.startswith("conv"))
print("%-44s %.4f ms | dense0 fwd %.1f  update %.1f  conv launches %.1f  others %.1f" % (sys.argv[1], d["ms_per_step"], k["dense0 fwd"], k["dense0 wgrad + dgrad + adam"], conv, sum(k.values()) - conv - k["dense0 fwd"] - k["dense0 wgrad + dgrad + adam"]))
PY
done
done
