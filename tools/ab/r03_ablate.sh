#!/bin/bash
# r03: ABLATION (diagnostic build): every SiLU evaluated TWICE (common.h, -DRTMODT_ABLATE_SILU2; same values to an ulp, so the same data
# through the MFMAs and the same NMS work) -- what one SiLU per output costs the staged bench in time AND in clock (power).
# (Taking the activation AWAY changes the data: without it the calibrated net overflows; with a hard-swish the NMS candidate count explodes.)
# Rebuilds the library inside the GPU box's scratch copy only.
set -e
O=gpurun_out/ablate; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 200 --warmup 20"
C=real-time-multi-object-detection---tracking-system_amd/csrc
export RTMODT_TUNE_CACHE=/tmp/tune_abl.txt
$B > $O/warm.json 2>/dev/null
for i in 1 2; do $B > $O/silu_$i.json 2>/dev/null; done
echo "rebuilding with the doubled activation"
make -C $C clean > /dev/null; make -C $C -j16 EXTRA=-DRTMODT_ABLATE_SILU2 > $O/build.log 2>&1
rm -f /tmp/tune_abl.txt; $B > $O/warm2.json 2>/dev/null
for i in 1 2; do $B > $O/silu2_$i.json 2>/dev/null; done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/ablate/*_[12].json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"], d["roofline"].get("in_kernel_clock",{}).get("ghz_mean"))
PY
