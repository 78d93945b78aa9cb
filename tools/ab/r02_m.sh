#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "weight_stationary" > gpurun_out/t_ws.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/t_ws.log
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/layers_m.txt 2> gpurun_out/layers_m.err
grep -E "ws:" gpurun_out/layers_m.err | head -40
grep -E "ws:|^total" gpurun_out/layers_m.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python bench.py --steps 150 --warmup 20 $Q > gpurun_out/m_bench.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/m_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
