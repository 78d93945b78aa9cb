#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "stem_and_layer1" > gpurun_out/t_sl1.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/t_sl1.log
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 2 > gpurun_out/layers_k.txt 2> gpurun_out/layers_k.err
grep "stem + layer" gpurun_out/layers_k.err; head -8 gpurun_out/layers_k.txt; tail -1 gpurun_out/layers_k.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for m in 0 1; do RTMODT_STEM_L1=$m python bench.py --steps 150 --warmup 20 $Q > gpurun_out/k_sl$m.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/k_sl$m.json').read().strip().splitlines()[-1]); print('stem_l1=$m', j['value'], j['ms_per_step'])"; done
