#!/bin/bash
# r04: random-shape parity of the ping-pong kernels against the general 64x64 tile, with guard zones around every output (tools/probes/pp_fuzz.hip)
O=gpurun_out/r04/fuzz; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pp_fuzz tools/probes/pp_fuzz.hip 2> /dev/null || exit 1
timeout -k 10 500 /tmp/pp_fuzz ${1:-150} ${2:-1} $3 > $O/fuzz_$2$3.txt 2>&1; echo "rc=$?" >> $O/fuzz_$2$3.txt
tail -25 $O/fuzz_$2$3.txt | cut -c1-400
