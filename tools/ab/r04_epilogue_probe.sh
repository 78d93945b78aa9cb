#!/bin/bash
# r04: the ping-pong kernels' epilogue after a change: stamped build (cycle counts), plain build (times + values against the tap-reuse reference tile)
O=gpurun_out/r04/epi_probe; mkdir -p $O; T=${1:-run}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -DRTMODT_DIAG -o /tmp/pp_probe_stamp tools/probes/pp_probe.hip 2> /dev/null || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pp_probe tools/probes/pp_probe.hip 2> /dev/null || exit 1
for f in "6.m.0.cv1" "6.m.0.cv2" "4.m.0.cv1" "22.s0 P3" "small"; do timeout -k 10 120 /tmp/pp_probe_stamp "$f" 3 >> $O/${T}_stamp.txt 2>&1 || exit 1; done
timeout -k 10 300 /tmp/pp_probe "" 20 > $O/${T}.txt 2>&1 || exit 1
grep -n "pp:\|epilogue of the first tile\|k-loop of the first" $O/${T}_stamp.txt
grep -n "MISMATCH" $O/${T}.txt | head; grep -c "outside tol: 0 " $O/${T}.txt
