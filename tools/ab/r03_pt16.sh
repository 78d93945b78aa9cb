#!/bin/bash
# r03: persistent tile kernel on 256x128 with 16 waves: parity, tuner timings, layer table, bench A/B (on / off through the tile table: RTMODT_NO_PT16)
set -e
O=gpurun_out/pt16; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_persistent_tile_kernel" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 timeout -k 10 300 python3 tools/profile_layers.py > $O/layers.txt 2> $O/tune.log
grep -E "pt:256x128s3/16w" $O/tune.log | head -40
grep -E "16w|^total" $O/layers.txt
# same-box A/B of the one-workgroup-per-CU tile families in the STAGED bench (the tuner times launches alone; 144-KiB workgroups keep the other stages off their CU)
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 200 --warmup 20"
for i in 1 2; do
  RTMODT_TUNE_SKIP="48,49,53,54" RTMODT_TUNE_CACHE=/tmp/tune_small.txt $B > $O/bench_small_$i.json 2>/dev/null
  RTMODT_TUNE_SKIP="48,49,54" RTMODT_TUNE_CACHE=/tmp/tune_w16.txt $B > $O/bench_w16_$i.json 2>/dev/null
  RTMODT_TUNE_CACHE=/tmp/tune_all.txt $B > $O/bench_all_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/pt16/bench_*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["roofline"]["frac"], d["roofline"].get("in_kernel_clock",{}).get("ghz_mean"))
PY
