#!/bin/bash
# A/B on one box: lib/librtmodt_hip_prev.so (HEAD) against the working tree's library
mkdir -p gpurun_out/v
L="real-time-multi-object-detection---tracking-system_amd/lib"
cp $L/librtmodt_hip.so $L/new.so.keep
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "forward_layers or benchmarked or tile or batch_equals or persistent" > gpurun_out/v/tests.txt 2>&1 || { tail -5 gpurun_out/v/tests.txt; exit 1; }
tail -1 gpurun_out/v/tests.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so; else cp $L/new.so.keep $L/librtmodt_hip.so; fi
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 $Q > gpurun_out/v/${v}_$rep.json 2>/dev/null || exit 1
  done
done
cp $L/new.so.keep $L/librtmodt_hip.so
RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/v/layers_new.txt 2>&1 || exit 1
cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so
RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/v/layers_prev.txt 2>&1 || exit 1
for f in gpurun_out/v/*_?.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
tail -1 gpurun_out/v/layers_new.txt; tail -1 gpurun_out/v/layers_prev.txt
