#!/bin/bash
# phase stamps of the plane conv forward / data-gradient roles at another configuration: CPROF_BATCH=256 / CPROF_HEADS=64
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_debug.so}
mkdir -p gpurun_out
for r in ${ROLES:-0 1 2 3 4}; do
  IDQN_CONV_PROF=$r timeout -k 10 120 python tools/probes/conv_prof.py > gpurun_out/cprof_${TAG:-cfg}_$r.txt 2>&1 || { tail -5 gpurun_out/cprof_${TAG:-cfg}_$r.txt; exit 1; }
done
cat gpurun_out/cprof_${TAG:-cfg}_*.txt
