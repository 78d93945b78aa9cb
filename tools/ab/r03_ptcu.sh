#!/bin/bash
# r03: persistent tiles with ONE workgroup per CU (the other slot left to the other stages' launches) against two, staged bench, same box, alternating
set -e
O=gpurun_out/ptcu; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 200 --warmup 20"
export RTMODT_TUNE_CACHE=/tmp/tune_ptcu.txt
$B > $O/warm.json 2>/dev/null
for i in 1 2 3; do
  $B > $O/two_$i.json 2>/dev/null
  RTMODT_PT_PER_CU=1 $B > $O/one_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ptcu/*_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"])
PY
