#!/bin/bash
# r03: wave issue priority around the conv epilogues (RTMODT_EPI_PRIO: 0 = none, 1 = epilogue raised, 2 = main loops raised), staged bench, alternating
set -e
O=gpurun_out/epi_prio; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --steps 200 --warmup 20"
export RTMODT_TUNE_CACHE=/tmp/tune_prio.txt
$B > $O/warm.json 2>/dev/null
for i in 1 2 3; do
  for m in 0 1 2; do RTMODT_EPI_PRIO=$m $B > $O/m${m}_$i.json 2>/dev/null; done
done
python3 - <<'PY' | tee gpurun_out/epi_prio/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/epi_prio/m*_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"], d["roofline"]["in_kernel_clock"]["ghz_mean"], d.get("verified"))
PY
