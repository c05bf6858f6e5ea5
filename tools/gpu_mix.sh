mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4))"; }
run IDQN_MIX=0
run IDQN_MIX=2
run IDQN_MIX=2 IDQN_MIX_A=50 IDQN_MIX_B=30
run IDQN_MIX=2 IDQN_MIX_A=60 IDQN_MIX_B=25
run IDQN_MIX=2 IDQN_MIX_A=70 IDQN_MIX_B=20
run IDQN_MIX=2 IDQN_MIX_A=45 IDQN_MIX_B=45
run IDQN_MIX=2 IDQN_MIX_A=30 IDQN_MIX_B=40
run IDQN_MIX=2 IDQN_MIX_A=80 IDQN_MIX_B=20
run IDQN_MIX=2 IDQN_MIX_A=100 IDQN_MIX_B=0
