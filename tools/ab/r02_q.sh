#!/bin/bash
# A/B on one box: previous commit's library (lib/librtmodt_hip_prev.so, built by hand) against the working tree's
mkdir -p gpurun_out/q
L="real-time-multi-object-detection---tracking-system_amd/lib"
cp $L/librtmodt_hip.so $L/new.so.keep
timeout -k 10 120 tools/probes/bin/kernel_probe l1 > gpurun_out/q/probe_l1.txt 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_detector.py -q -m gpu -x > gpurun_out/q/tests.txt 2>&1 || { tail -5 gpurun_out/q/tests.txt; exit 1; }
tail -2 gpurun_out/q/tests.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so; else cp $L/new.so.keep $L/librtmodt_hip.so; fi
    timeout -k 10 200 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/q/${v}_$rep.json 2>/dev/null || exit 1
  done
done
cp $L/new.so.keep $L/librtmodt_hip.so
for f in gpurun_out/q/*_?.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/q/layers_new.txt 2>&1 || exit 1
cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so
RTMODT_CHAINS=1 timeout -k 10 200 python tools/profile_layers.py --frames-per-stream 4 > gpurun_out/q/layers_prev.txt 2>&1 || exit 1
tail -1 gpurun_out/q/layers_new.txt; tail -1 gpurun_out/q/layers_prev.txt
