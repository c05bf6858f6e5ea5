#!/bin/bash
# experiment: every conv launch planned for half the chip (what one role of a two-role launch would get)
export IDQN_HIP_LIB=${IDQN_HIP_LIB:-${GRAFT_REPO_ROOT:-$PWD}/i-dqn_amd/libidqn_hip_variants.so}  # the switches below exist in the variants build only
mkdir -p gpurun_out
echo "== default"; bash tools/gpu_prof.sh base | grep -E "k_cfwd|k_cwgrad" &&
echo "== half" && IDQN_PLAN_PRINT=1 IDQN_CONV_WGS=128 IDQN_WCHUNKS_DIV=2 bash tools/gpu_prof.sh alt | grep -E "k_cfwd|k_cwgrad"; grep "\[plan\]" gpurun_out/prof_alt.log | sort -u
