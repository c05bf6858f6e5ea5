#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --no-cpu-baseline --repeats 1 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), [ (k['launch'][:12],round(k['us'],1)) for k in d['kernels'] if 'grad' in k['launch'] and 'dense' not in k['launch']])" || exit 1; }
run A=0
run IDQN_PAIR_SKEW=100
run IDQN_PAIR_SKEW=200
run IDQN_PAIR_SKEW=300
run IDQN_PAIR_SKEW=-100
run IDQN_PAIR_SKEW=-200
run A=1
