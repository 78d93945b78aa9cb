#!/bin/bash
# r03: where nms_kernel's time goes at the benchmarked shape: the kernel cut short after each phase (RTMODT_NMS_STOP=1..5, timing-only),
# ALONE on the device (tools/nms_alone.py fetches every batch before it enqueues the next) under rocprofv3 --kernel-trace --stats.
# phases: 1 = candidate count + compaction offsets, 2 = keys written, 3 = sorted, 4 = boxes gathered + masks cleared, 5 = IoU walk, 0 = all
set -o pipefail
O=gpurun_out/nms_phases; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/nms_alone.py | tee $O/summary.txt
for T in 1024 256; do
export RTMODT_NMS_THREADS=$T
echo "== nms_kernel with $T threads" | tee -a $O/summary.txt
for s in 1 2 3 4 5 0; do
  RTMODT_NMS_STOP=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O/t${T}_s$s -- python3 tools/nms_alone.py > /dev/null 2> $O/t${T}_s$s.log || { echo "stop $s failed"; continue; }
  f=$(ls $O/t${T}_s$s/*/*_kernel_stats.csv | head -1)
  python3 - $f $s <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "nms_kernel" in r["Name"] or "tracker_update" in r["Name"]:
        print("stop=%s %-16s calls %s avg %.1f us min %.1f max %.1f" % (sys.argv[2], r["Name"].split("(")[0].split("::")[-1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $O/t${T}_s$s
done
done
