#!/bin/bash
# acting path: tests that use it, latency with / without the polled mailbox, trainer-loop rates
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_learning_sanity.py -q -x -m gpu > gpurun_out/act_tests.log 2>&1 || { tail -30 gpurun_out/act_tests.log; exit 1; }
tail -1 gpurun_out/act_tests.log
echo "== mailbox"; timeout -k 10 100 python tools/probes/act_loop.py 2>&1 | grep "us per" &&
echo "== copy + sync" && IDQN_ACT_POLL=0 timeout -k 10 100 python tools/probes/act_loop.py 2>&1 | grep "us per" &&
echo "== loop (mailbox)" && timeout -k 10 300 python tools/bench_loop.py 2>&1 | grep "env steps" &&
echo "== loop (copy + sync)" && IDQN_ACT_POLL=0 timeout -k 10 300 python tools/bench_loop.py 2>&1 | grep "env steps"
