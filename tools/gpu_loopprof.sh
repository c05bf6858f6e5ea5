#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/probes/loop_profile.py > gpurun_out/loop_profile.txt 2>&1 || { tail -20 gpurun_out/loop_profile.txt; exit 1; }
grep -v "^$" gpurun_out/loop_profile.txt | tail -50
