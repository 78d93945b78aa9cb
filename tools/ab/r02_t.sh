#!/bin/bash
# A/B of the alternating tile order (RTMODT_ZIGZAG=0: every launch ascending)
mkdir -p gpurun_out/t
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "bottleneck or benchmarked or batch_equals or tap_reuse" > gpurun_out/t/tests.txt 2>&1 || { tail -5 gpurun_out/t/tests.txt; exit 1; }
tail -1 gpurun_out/t/tests.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for F in 4 2 1; do
  for z in 0 1; do
    RTMODT_ZIGZAG=$z timeout -k 10 200 python bench.py --steps 150 --warmup 20 $Q --frames-per-stream $F > gpurun_out/t/F${F}_z$z.json 2>/dev/null || exit 1
  done
done
for z in 0 1; do
  RTMODT_ZIGZAG=$z RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/t/layers_F4_z$z.txt 2>&1 || exit 1
  RTMODT_ZIGZAG=$z RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 1 > gpurun_out/t/layers_F1_z$z.txt 2>&1 || exit 1
done
for f in gpurun_out/t/F*.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
tail -1 gpurun_out/t/layers_F4_z0.txt; tail -1 gpurun_out/t/layers_F4_z1.txt; tail -1 gpurun_out/t/layers_F1_z0.txt; tail -1 gpurun_out/t/layers_F1_z1.txt
