#!/bin/bash
# r04: what the ping-pong kernels' epilogue is made of, on the device: stamped build of tools/probes/pp_probe.hip with the diagnostic ablation bits
# (RTMODT_EPI_PRIO bit 0 = no SiLU, bit 1 = no global stores; results wrong, times only) on the 128 -> 128 @ 40 and 64 -> 64 @ 80 shapes.
O=gpurun_out/r04/epi_ablate; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DRTMODT_STAMP -DRTMODT_DIAG -o /tmp/pp_probe_stamp tools/probes/pp_probe.hip || exit 1
for a in 0 1 2 3; do
  for f in "6.m.0.cv1" "4.m.0.cv1" "22.s0 P3"; do
    echo "== ablate $a, $f" >> $O/ablate.txt
    RTMODT_EPI_PRIO=$a timeout -k 10 120 /tmp/pp_probe_stamp "$f" 3 >> $O/ablate.txt 2>&1 || exit 1
  done
done
grep -n "== ablate\|pp:\|epilogue of the first tile\|k-loop of the first" $O/ablate.txt
