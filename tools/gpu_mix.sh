# IDQN_MIX modes (mixed launches of the fused Dense_0 update and conv gradient kernels): parity + step rate
mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
IDQN_MIX=2 timeout -k 10 300 python -m pytest tests/test_gpu_fp_path.py -q -x -k "mixed or goldens" 2>&1 | tail -2
run() { env "$@" timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4))"; }
run IDQN_MIX=0
run IDQN_MIX=1
run IDQN_MIX=2
run IDQN_MIX=2 IDQN_MIX_A=100 IDQN_MIX_B=0
run IDQN_MIX=0
