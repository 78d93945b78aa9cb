#!/bin/bash
# stage streams at different HSA queue priorities (diagnostic library tools/ab/lib_diagprio.so, RTMODT_STAGE_PRIO = one letter per stage: h n l)
O=${1:-gpurun_out/r05/stage_prio}; mkdir -p $O
L=real-time-multi-object-detection---tracking-system_amd/lib/librtmodt_hip.so
cp $L /tmp/lib_keep.so && cp tools/ab/lib_diagprio.so $L || exit 1
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --prewarm 0.2 --steps 300 --warmup 30"
for r in 1 2; do
  for v in none nnn hnl lnh hhl lhh nhn hln; do
    if [ $v = none ]; then unset RTMODT_STAGE_PRIO; else export RTMODT_STAGE_PRIO=$v; fi
    timeout -k 10 200 $B > $O/${v}_$r.json 2> $O/${v}_$r.err || { echo "$v failed"; tail -3 $O/${v}_$r.err; cp /tmp/lib_keep.so $L; exit 1; }
    python3 -c "import json; d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1]); print('$v', $r, d['value'], d['ms_per_step'])"
  done
done
cp /tmp/lib_keep.so $L
