#!/bin/bash
# conv backward pairs (one launch for a layer's data + weight gradient): parity, then kernel profile with / without
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_learning_sanity.py -x -q -m gpu > gpurun_out/pair_tests.log 2>&1 || { tail -40 gpurun_out/pair_tests.log; exit 1; }
tail -1 gpurun_out/pair_tests.log
echo "== pairs"; IDQN_PLAN_PRINT=1 bash tools/gpu_prof.sh base | head -16; grep "\[plan\]" gpurun_out/prof_base.log | sort -u
echo "== IDQN_NO_PAIR=1"; IDQN_NO_PAIR=1 bash tools/gpu_prof.sh alt | head -16
for v in "" IDQN_NO_PAIR=1; do env $v python bench.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], [ (k['launch'],k['us']) for k in d['kernels'] if 'grad' in k['launch']])"; done
