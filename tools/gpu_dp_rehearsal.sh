mkdir -p gpurun_out && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -q -m gpu > gpurun_out/gpu_all.log 2>&1; tail -6 gpurun_out/gpu_all.log
for mode in factored allreduce; do
IDQN_DP_MODE=$mode MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 timeout -k 10 200 python bench.py --gpus 1 --force-dp --steps 300 --warmup 50 > gpurun_out/bench_dp_$mode.json 2> gpurun_out/bench_dp_$mode.err && python -c "
import json; d=json.load(open('gpurun_out/bench_dp_$mode.json')); print('DP1 $mode', round(d['value'],1), round(d['ms_per_step'],4), d['roofline']['launch_ms'])"
done
timeout -k 10 200 python bench.py --steps 300 --warmup 50 --no-cpu-baseline > gpurun_out/bench_cur.json 2> gpurun_out/bench_cur.err && python -c "
import json; d=json.load(open('gpurun_out/bench_cur.json')); print('fused N=1', round(d['value'],1), round(d['ms_per_step'],4))"
