#!/bin/bash
# r03: nms_kernel alone on the device at the benchmarked shape, noise frames (113-1 531 candidates per image) and structured frames
# (a saturated random head: median ~3 000, up to ~7 900 candidates per image); then the staged bench on both kinds of frames
set -o pipefail
O=gpurun_out/nms_dense; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for K in noise structured; do
  KIND=$K python3 tools/nms_alone.py | tee -a $O/summary.txt
  KIND=$K rocprofv3 --kernel-trace --stats --output-format csv -d $O/x -- python3 tools/nms_alone.py > /dev/null 2> $O/x.log || { echo failed; continue; }
  f=$(ls $O/x/*/*_kernel_stats.csv | head -1)
  python3 - $f $K <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "nms_kernel" in r["Name"] or "tracker_update" in r["Name"]:
        print("%s frames: %-16s calls %s avg %.1f us min %.1f max %.1f" % (sys.argv[2], r["Name"].split("(")[0].split("::")[-1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $O/x
done
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --steps 200 --warmup 20"
export RTMODT_TUNE_CACHE=/tmp/tk.txt
for i in 1 2; do
  $B > $O/noise_$i.json 2>/dev/null && $B --frames-kind structured > $O/struct_$i.json 2>/dev/null
done
python3 - <<'PY' | tee -a gpurun_out/nms_dense/summary.txt
import json,glob
for f in sorted(glob.glob("gpurun_out/nms_dense/*_[12].json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["value"], "frames/s  frac", d["roofline"]["frac"], " clock", d["roofline"]["in_kernel_clock"]["ghz_mean"], " detections/frame", d.get("detections_per_frame"), " verified", d.get("verified",{}).get("ok", d.get("verified")) if isinstance(d.get("verified"),dict) else d.get("verified"))
PY
