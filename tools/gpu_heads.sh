#!/bin/bash
mkdir -p gpurun_out
echo "== pairs"; IDQN_PLAN_PRINT=1 timeout -k 10 400 python tools/bench_heads.py 2> gpurun_out/heads_plans.txt | tee gpurun_out/heads_pairs.txt
echo "== IDQN_NO_PAIR=1"; IDQN_NO_PAIR=1 timeout -k 10 400 python tools/bench_heads.py 2>/dev/null | tee gpurun_out/heads_nopair.txt
