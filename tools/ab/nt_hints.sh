#!/bin/bash
# non-temporal hints in the persistent tile kernel: RTMODT_WT bit 2 = A operand of single-slice 1x1 convs by nt DMA, bit 3 = streaming stores
mkdir -p gpurun_out/nt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2 3; do
  for w in 0 4 8 12; do
    RTMODT_WT=$w timeout -k 10 200 python bench.py --steps 200 --warmup 20 $Q > gpurun_out/nt/w${w}_$rep.json 2>/dev/null || exit 1
  done
done
for w in 0 4 8; do RTMODT_WT=$w RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/nt/layers_w$w.txt 2>&1 || exit 1; done
for f in gpurun_out/nt/w*_?.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
for w in 0 4 8; do tail -1 gpurun_out/nt/layers_w$w.txt; done
