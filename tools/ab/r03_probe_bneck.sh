#!/bin/bash
# r03: phase stamps of the fused Bottleneck (c = 32, 160 x 160, 32 frames; with and without the C2f.cv2 tail)
set -e
O=gpurun_out/probe; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -Iinclude -o /tmp/kernel_probe tools/probes/kernel_probe.hip 2> $O/build.log
/tmp/kernel_probe > $O/bneck_stamps.txt 2>&1
cat $O/bneck_stamps.txt
