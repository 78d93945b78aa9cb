#!/bin/bash
mkdir -p gpurun_out/s
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2; do
  timeout -k 10 200 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/s/wt0_$rep.json 2>/dev/null || exit 1
  RTMODT_WT=1 timeout -k 10 200 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/s/wt1_$rep.json 2>/dev/null || exit 1
done
timeout -k 10 200 python bench.py --steps 150 --warmup 20 $Q --frames-per-stream 6 > gpurun_out/s/f6_1.json 2>/dev/null || exit 1
for f in gpurun_out/s/*.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
