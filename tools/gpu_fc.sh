#!/bin/bash
# MLP ("fc") path: parity, step time and acting latency, LunarLander-shaped loop
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_learning_sanity.py tests/test_gpu_per_extension.py -x -q -m gpu > gpurun_out/fc_tests.log 2>&1 || { tail -40 gpurun_out/fc_tests.log; exit 1; }
tail -1 gpurun_out/fc_tests.log
timeout -k 10 200 python tools/bench_fc.py 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/bench_loop.py 2>&1 | grep -E "us per|env steps"
