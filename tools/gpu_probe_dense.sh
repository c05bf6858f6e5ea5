#!/bin/bash
mkdir -p gpurun_out && hipcc --offload-arch=gfx950 -O3 -w tools/probes/dense_read_probe.hip -o /tmp/dense_read_probe || exit 1
timeout -k 10 120 /tmp/dense_read_probe | tee gpurun_out/dense_read_probe.txt
