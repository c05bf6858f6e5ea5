mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4))"; }
run X=0
run IDQN_CONV=bf16x3-forward
run IDQN_MIX=2
run IDQN_CONV=bf16x3-forward IDQN_MIX=2
run X=0
run IDQN_CONV=bf16x3-forward IDQN_MIX=2
IDQN_CONV=bf16x3-forward IDQN_MIX=2 timeout -k 10 300 python -m pytest tests/test_gpu_fp_path.py -q -x -k "goldens and f32" 2>&1 | tail -1
