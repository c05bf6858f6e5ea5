#!/bin/bash
# the i-IQN bench line under a list of environment settings ("A=1 B=2" per argument; "" = defaults): step time + the big launches
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout -k 10 300 python bench.py --algo iiqn --no-cpu-baseline --steps 15 --warmup 3 --repeats 1 > gpurun_out/ie_$i.json 2> gpurun_out/ie_$i.err || { echo "[$e] failed"; tail -3 gpurun_out/ie_$i.err; exit 1; }
  python3 - "$e" $i <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ie_{sys.argv[2]}.json"))
k = {r["launch"]: r["us"] for r in d["kernels"]}
wg = k.get('dense0 wgrad + adam', 0) + k.get('factor planes', 0) + k.get('iqn dense0 wgrad', 0) + k.get('iqn dense0 adam', 0)
print(f"{sys.argv[1] or '(defaults)':40s} {d['ms_per_step']:.4f} ms | fwd {k['iqn dense0 fwd']:7.1f} dgrad {k.get('iqn dense0 dgrad', 0) + k.get('iqn dense0 dgrad + wgrad', 0):7.1f} wgrad+adam {wg:7.1f} embed {k['iqn embedding x features']:6.1f} embed_bwd {k['iqn embedding backward']:6.1f}")
PY
done
