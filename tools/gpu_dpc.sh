#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/dp_tests.log 2>&1 || { tail -40 gpurun_out/dp_tests.log; exit 1; }
tail -1 gpurun_out/dp_tests.log
echo "== bf16 planes"; timeout -k 10 300 python tools/probes/dp_compute.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dp_compute.txt
echo "== IDQN_DP_F32=1"; IDQN_DP_F32=1 timeout -k 10 300 python tools/probes/dp_compute.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/dp_compute_f32.txt
