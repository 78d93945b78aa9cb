#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "bottleneck or forward_layers or benchmarked" > gpurun_out/t_bn.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_bn.log
RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 2 > gpurun_out/layers_g.txt 2> /dev/null
grep -E "m\.[01] \(cv1|^total" gpurun_out/layers_g.txt
RTMODT_BNECK=1 RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 2 > gpurun_out/layers_g_fused.txt 2> /dev/null
grep -E "m\.[01] \(cv1|^total" gpurun_out/layers_g_fused.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python bench.py --steps 300 --warmup 30 $Q > gpurun_out/g_bench.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/g_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
