#!/bin/bash
# is reading the upsampled half of the neck concats where it was produced (UP_READ) still the better choice?  RTMODT_UP_READ=0 materialises the upsample; product library, alternating runs
O=${1:-gpurun_out/r05/upread_ab}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --prewarm 0.2 --steps 300 --warmup 30"
for r in 1 2 3; do
  for v in default up0 up1; do
    unset RTMODT_UP_READ; [ $v = up0 ] && export RTMODT_UP_READ=0; [ $v = up1 ] && export RTMODT_UP_READ=1
    timeout -k 10 200 $B > $O/${v}_$r.json 2> $O/${v}_$r.err || { echo "$v failed"; tail -3 $O/${v}_$r.err; exit 1; }
    python3 -c "import json; d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1]); print('$v', $r, d['value'], d['ms_per_step'])"
  done
done
