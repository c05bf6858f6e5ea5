#!/bin/bash
# A/B of one environment switch: kernel-time profiles (rocprofv3 --kernel-trace --stats) with and without it, same box
# usage: tools/gpu_ab_env.sh VAR=value [pytest files...]
mkdir -p gpurun_out
sw=$1; shift
python -m pytest ${@:-tests/test_gpu_fp_path.py} -x -q -m gpu > gpurun_out/fp.log 2>&1 || { tail -40 gpurun_out/fp.log; exit 1; }
tail -1 gpurun_out/fp.log
echo "== default"; bash tools/gpu_prof.sh base | head -17 &&
echo "== $sw" && env "$sw" bash tools/gpu_prof.sh alt | head -17
