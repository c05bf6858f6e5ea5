#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fp_path.py tests/test_gpu_per_extension.py -x -q -m gpu > gpurun_out/fp.log 2>&1 || { tail -40 gpurun_out/fp.log; exit 1; }
tail -1 gpurun_out/fp.log
for a in 6 18; do python bench.py --no-cpu-baseline --repeats 3 --actions $a 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A=$a', round(d['value']), d['ms_per_step'], [ (k['launch'][:10],round(k['us'],1)) for k in d['kernels'] if k['launch'][:2] in ('td','hi')])"; done
