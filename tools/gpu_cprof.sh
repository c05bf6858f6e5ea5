#!/bin/bash
# phase timing of every plane conv forward / data-gradient role (0..4) + the frame-ring tests
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_debug.so}  # the switches below exist in the debug build (__graft_entry__.build_debug()) only
mkdir -p gpurun_out
python -m pytest tests/test_gpu_int_path.py -x -q -m gpu > gpurun_out/int.log 2>&1 || { tail -20 gpurun_out/int.log; exit 1; }
tail -2 gpurun_out/int.log
for r in 0 1 2 3 4; do
  IDQN_CONV_PROF=$r IDQN_PLAN_PRINT=1 timeout -k 10 120 python tools/probes/conv_prof.py > gpurun_out/cprof_$r.txt 2>&1 || { tail -5 gpurun_out/cprof_$r.txt; exit 1; }
done
cat gpurun_out/cprof_*.txt
