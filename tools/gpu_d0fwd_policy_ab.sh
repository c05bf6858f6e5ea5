# round 5: load policy of the Dense_0 forward's weight stream per net (k_dense0_fwd3, DenseFwdArgs::nt_from): online nets default-policy + target nets
# non-temporal (default) against every net non-temporal (IDQN_D0_FWD_NT_FROM=0, rounds 3-4) and every net default-policy (=10); variants build,
# interleaved on one box; the losses must be identical (a load policy changes no arithmetic)
mkdir -p gpurun_out/r5pol && cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}"
O=gpurun_out/r5pol
V=$PWD/i-dqn_amd/libidqn_hip_variants.so
for round in 1 2; do
for f in -1 0 10; do
  cfg="IDQN_HIP_LIB=$V"; [ $f != -1 ] && cfg="$cfg IDQN_D0_FWD_NT_FROM=$f"
  env $cfg timeout -k 10 200 python bench.py --steps 400 --warmup 30 --repeats 3 --no-cpu-baseline > $O/ab.json 2> $O/ab.err || { echo "[$cfg] failed"; tail -5 $O/ab.err; continue; }
  python - "$f" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r5pol/ab.json"))
k = {x["launch"]: x["us"] for x in d["kernels"]}
conv = sum(v for n, v in k.items() if n.startswith("conv"))
name = {"-1": "online default, target nt (default)", "0": "every net nt", "10": "every net default-policy"}[sys.argv[1]]
print("%-38s %.4f ms | dense0 fwd %.1f  update %.1f (events %.1f)  conv launches %.1f  others %.1f | %s" % (name, d["ms_per_step"], k["dense0 fwd"], k["dense0 wgrad + dgrad + adam"],
      d["roofline"]["launch_ms"] * 1e3, conv, sum(k.values()) - conv - k["dense0 fwd"] - k["dense0 wgrad + dgrad + adam"], " ".join("%.9e" % v for v in d["final_losses"][:2])))
PY
done
done
