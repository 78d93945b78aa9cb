#!/bin/bash
mkdir -p gpurun_out/u
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2 3; do
  for z in 0 1; do
    RTMODT_ZIGZAG=$z timeout -k 10 200 python bench.py --steps 200 --warmup 20 $Q > gpurun_out/u/F4_z${z}_$rep.json 2>/dev/null || exit 1
  done
done
for rep in 1 2; do
  for z in 0 1; do
    RTMODT_ZIGZAG=$z timeout -k 10 200 python bench.py --steps 200 --warmup 20 $Q --frames-per-stream 2 > gpurun_out/u/F2_z${z}_$rep.json 2>/dev/null || exit 1
  done
done
for f in gpurun_out/u/F*.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
