#!/bin/bash
# r03: post-processing stream (NMS, tracker) at high priority against default, staged bench, same box, alternating
set -e
O=gpurun_out/postprio; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 200 --warmup 20"
export RTMODT_TUNE_CACHE=/tmp/tune_pp.txt
$B > $O/warm.json 2>/dev/null
for i in 1 2 3; do
  $B > $O/default_$i.json 2>/dev/null
  RTMODT_POST_PRIO=1 $B > $O/high_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/postprio/*_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"], d["roofline"]["stages"], d["roofline"]["batch_latency_ms"])
PY
