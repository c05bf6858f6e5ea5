mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 200 python bench.py --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['value'],1), round(d['ms_per_step'],4))"; }
run IDQN_MIX=0
run IDQN_MIX=2
run IDQN_MIX=2 IDQN_HIP_LIB=$PWD/i-dqn_amd/libidqn_hip_mw4.so
run IDQN_MIX=2
run IDQN_MIX=2 IDQN_HIP_LIB=$PWD/i-dqn_amd/libidqn_hip_mw4.so
