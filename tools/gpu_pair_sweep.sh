#!/bin/bash
# sweep of the split of a conv backward pair (IDQN_PAIR_D<layer>: workgroups of the data gradient, IDQN_PAIR_C<layer>:
# position chunks of the weight gradient); prints the step rate and the event spans of the backward launches
mkdir -p gpurun_out
run() { env "$@" python bench.py --no-cpu-baseline --repeats 1 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), [ (k['launch'][:12],round(k['us'],1)) for k in d['kernels'] if 'grad' in k['launch'] and 'dense' not in k['launch']])" || exit 1; }
run A=0
run IDQN_PAIR_D2=112
run IDQN_PAIR_D2=144
run IDQN_PAIR_C2=7
run IDQN_PAIR_D2=144 IDQN_PAIR_C2=7
run IDQN_PAIR_C1=4
run A=1
