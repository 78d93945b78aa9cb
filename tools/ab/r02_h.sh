#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "benchmarked" > gpurun_out/t_bs.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/t_bs.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/h20.json 2> gpurun_out/h20.err; echo "rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/h20.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['timing'], j.get('host_frames'), j.get('verified',{}).get('ok'), j.get('two_frames_per_stream_per_step'), j.get('one_frame_per_stream_per_step'), j.get('latency_single_stream_ms',{}).get('p50'), j.get('cpu_baseline',{}).get('value'))
PY
