#!/bin/bash
# full GPU test suite, two-rank gloo rehearsal on one device, host-thread time per step
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu -s > gpurun_out/t_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/t_gpu.log
grep -E "free-running|end to end|passed|failed|error" gpurun_out/t_gpu.log | tail -20
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 50 --warmup 10 --backend gloo --one-device --no-cpu-baseline --no-latency > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err; echo "gloo2 rc=$?"
tail -c 600 gpurun_out/bench_gloo2.json
timeout -k 10 200 python tools/probes/host_time.py > gpurun_out/host_time.txt 2>&1; cat gpurun_out/host_time.txt; nproc
