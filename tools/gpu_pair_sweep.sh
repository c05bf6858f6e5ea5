#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py -x -q -m gpu > gpurun_out/fp.log 2>&1 || { tail -40 gpurun_out/fp.log; exit 1; }
tail -1 gpurun_out/fp.log
run() { env "$@" python bench.py --no-cpu-baseline --repeats 1 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), [ (k['launch'][:12],round(k['us'],1)) for k in d['kernels'] if 'grad' in k['launch'] and 'dense' not in k['launch']])" || exit 1; }
run A=0
run IDQN_PAIR_ROLE_XCDS=1
run A=1
run IDQN_PAIR_ROLE_XCDS=1
