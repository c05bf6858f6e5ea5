#!/bin/bash
# quick cycle: fp parity (+ the full-size configurations) + kernel profile
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_configs.py tests/test_gpu_learning_sanity.py -x -q -m gpu > gpurun_out/fp.log 2>&1 || { tail -40 gpurun_out/fp.log; exit 1; }
tail -1 gpurun_out/fp.log
bash tools/gpu_prof.sh base | head -16
