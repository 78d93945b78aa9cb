#!/bin/bash
# r03: phase stamps of the tap-reuse kernel on the Detect / Bottleneck shapes (32 frames)
set -e
O=gpurun_out/probe; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -Iinclude -o /tmp/kernel_probe tools/probes/kernel_probe.hip 2> $O/build.log
/tmp/kernel_probe rows > $O/rows_stamps.txt 2>&1
cat $O/rows_stamps.txt
