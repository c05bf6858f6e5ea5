#!/bin/bash
mkdir -p gpurun_out
run() { env "$@" python bench.py --gpus 1 --force-dp --no-cpu-baseline --repeats 3 --steps 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['value']), round(d['ms_per_step']*1e3,1), [ (k['launch'][:14],round(k['us'],1)) for k in d.get('kernels',[])])" || exit 1; }
run A=0
run NCCL_MAX_NCHANNELS=4
run NCCL_MAX_NCHANNELS=16
run IDQN_DP_MODE=allreduce
