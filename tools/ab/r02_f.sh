#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -q -m gpu -s > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/t_gpu.log
grep -E "free-running|end to end|rank flips|passed|failed|Error|error" gpurun_out/t_gpu.log | tail -30
python tools/run_pipeline_synth.py > gpurun_out/pipeline_640.json 2> /dev/null; cat gpurun_out/pipeline_640.json
python tools/run_pipeline_synth.py --source 1920x1080 > gpurun_out/pipeline_1080p.json 2> /dev/null; cat gpurun_out/pipeline_1080p.json
