#!/bin/bash
# opt-in step graph (IDQN_STEP_GRAPH=1): parity, bench, trainer loop, against the eager launches on the same box
mkdir -p gpurun_out
IDQN_STEP_GRAPH=1 timeout -k 10 600 python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_learning_sanity.py -q -x -m gpu > gpurun_out/graph_tests.log 2>&1 || { tail -30 gpurun_out/graph_tests.log; exit 1; }
tail -1 gpurun_out/graph_tests.log
echo "== eager"; timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/b_eager.json 2> gpurun_out/b_eager.err && python -c "import json; d=json.load(open('gpurun_out/b_eager.json')); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'])" &&
echo "== graph" && IDQN_STEP_GRAPH=1 timeout -k 10 200 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/b_graph.json 2> gpurun_out/b_graph.err && python -c "import json; d=json.load(open('gpurun_out/b_graph.json')); print(d['value'], d['ms_per_step'], d['timing']['ms_per_step_all'])" &&
echo "== loop eager" && timeout -k 10 300 python tools/bench_loop.py 2>&1 | grep "env steps" &&
echo "== loop graph" && IDQN_STEP_GRAPH=1 timeout -k 10 300 python tools/bench_loop.py 2>&1 | grep "env steps"
