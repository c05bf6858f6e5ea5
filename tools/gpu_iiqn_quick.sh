#!/bin/bash
# i-IQN parity tests, then the bench line with its per-launch table (one GPU call while iterating on the i-IQN kernels)
cd "${GRAFT_REPO_ROOT:?}"
timeout -k 10 600 python -m pytest tests/test_gpu_iqn.py -m gpu -x -q > gpurun_out/iqn_tests.log 2>&1 || { tail -30 gpurun_out/iqn_tests.log; exit 1; }
tail -2 gpurun_out/iqn_tests.log
timeout -k 10 300 python bench.py --algo iiqn --no-cpu-baseline "$@" > gpurun_out/iiqn_bench.json 2> gpurun_out/iiqn_bench.err || { tail -5 gpurun_out/iiqn_bench.err; exit 1; }
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/iiqn_bench.json"))
print(f"{d['ms_per_step']:.4f} ms/step  {d['value']:.1f} steps/s   roofline {d['roofline']['kernel']} frac {d['roofline']['frac']:.3f}")
for r in d["kernels"]:
    if r["us"] > 25: print(f"  {r['launch']:40s} {r['us']:9.1f} us")
PY
