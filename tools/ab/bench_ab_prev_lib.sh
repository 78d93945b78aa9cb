#!/bin/bash
# A/B on one box, 4 alternating pairs: lib/librtmodt_hip_prev.so (HEAD) against the working tree's library
mkdir -p gpurun_out/w
L="real-time-multi-object-detection---tracking-system_amd/lib"
cp $L/librtmodt_hip.so $L/new.so.keep
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2 3 4; do
  for v in prev new; do
    if [ $v = prev ]; then cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so; else cp $L/new.so.keep $L/librtmodt_hip.so; fi
    timeout -k 10 200 python bench.py --steps 300 --warmup 20 $Q > gpurun_out/w/${v}_$rep.json 2>/dev/null || exit 1
  done
done
cp $L/new.so.keep $L/librtmodt_hip.so
for f in gpurun_out/w/*_?.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
