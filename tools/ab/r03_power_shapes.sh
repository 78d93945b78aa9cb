#!/bin/bash
# r03: in-kernel clock and package power (hwmon, where readable) for the engine shapes: three stages (default), two, plain engine, one frame per stream
set -e
O=gpurun_out/power_shapes; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --steps 200 --warmup 20 --long 2"
export RTMODT_TUNE_CACHE=/tmp/tune_ps.txt
$B > $O/warm.json 2>/dev/null
$B > $O/stages3.json 2>/dev/null
$B --stages 2 > $O/stages2.json 2>/dev/null
RTMODT_CHAINS=1 $B > $O/plain.json 2>/dev/null
RTMODT_CHAINS=2 $B > $O/chains2.json 2>/dev/null
$B --frames-per-stream 1 > $O/F1.json 2>/dev/null
RTMODT_CHAINS=1 $B --frames-per-stream 1 > $O/plain_F1.json 2>/dev/null
python3 - <<'PY' | tee gpurun_out/power_shapes/summary.txt
import json,glob
for n in ("stages3","stages2","chains2","plain","F1","plain_F1"):
    d=json.load(open(f"gpurun_out/power_shapes/{n}.json")); c=d["roofline"].get("in_kernel_clock") or {}
    pc=(d.get("timing",{}).get("long_window") or {}).get("power_clock")
    print(n, d["value"], "frames/s  frac", d["roofline"]["frac"], " in-kernel GHz", c.get("ghz_mean"), " frac at clock", c.get("frac_at_clock"), " hwmon", pc)
PY
