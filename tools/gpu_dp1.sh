#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dp_two_ranks.py tests/test_gpu_configs.py tests/test_gpu_fp_path.py -x -q -m gpu > gpurun_out/dp_tests.log 2>&1 || { tail -40 gpurun_out/dp_tests.log; exit 1; }
tail -1 gpurun_out/dp_tests.log
run() { env "$@" python bench.py --gpus 1 --force-dp --no-cpu-baseline --repeats 3 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(d['ms_per_step']*1e3,1))" || exit 1; }
run A=0
run A=1
python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', round(d['value']), round(d['ms_per_step']*1e3,1))"
