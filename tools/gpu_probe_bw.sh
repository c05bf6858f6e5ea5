#!/bin/bash
mkdir -p gpurun_out && hipcc --offload-arch=gfx950 -O3 tools/probes/read_bw_probe.hip -o /tmp/read_bw_probe || exit 1
for mb in 160 480; do echo "== $mb MB"; MB=$mb timeout -k 10 60 /tmp/read_bw_probe || exit 1; done | tee gpurun_out/read_bw_small.txt
