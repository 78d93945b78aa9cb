#!/bin/bash
# r03: the staged bench with nms_kernel at 1024 threads (product) and at 256 (rounds 1-2), alternating; then the default bench line
set -e
O=gpurun_out/nms_bench; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-compare --no-host-leg --steps 200 --warmup 20"
export RTMODT_TUNE_CACHE=/tmp/tune_nmsb.txt
$B --no-latency > $O/warm.json 2>/dev/null
for i in 1 2; do
  $B > $O/t1024_$i.json 2>/dev/null
  RTMODT_NMS_THREADS=256 $B > $O/t256_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/nms_bench/t*_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"], {k:v for k,v in d.get("latency",{}).items() if "p50" in k or "p99" in k} if isinstance(d.get("latency"),dict) else d.get("latency"))
PY
