#!/bin/bash
# is the fused 1x1 tail (layer 3 + 4.cv1) still the better choice with round 5's kernels?  RTMODT_TAIL=0 forces separate launches; product library, alternating runs
O=${1:-gpurun_out/r05/tail_ab}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --no-tracker-stress --long 0 --prewarm 0.2 --steps 300 --warmup 30"
for r in 1 2 3; do
  for v in default tail0 tail1; do
    unset RTMODT_TAIL; [ $v = tail0 ] && export RTMODT_TAIL=0; [ $v = tail1 ] && export RTMODT_TAIL=1
    timeout -k 10 200 $B > $O/${v}_$r.json 2> $O/${v}_$r.err || { echo "$v failed"; tail -3 $O/${v}_$r.err; exit 1; }
    python3 -c "import json; d=json.loads(open('$O/${v}_$r.json').read().strip().splitlines()[-1]); print('$v', $r, d['value'], d['ms_per_step'])"
  done
done
