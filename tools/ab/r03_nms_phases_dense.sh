#!/bin/bash
# r03: nms_kernel's phases (RTMODT_NMS_STOP) alone on the device on STRUCTURED frames (median ~3 000, up to ~7 900 candidates per image)
set -o pipefail
O=gpurun_out/nms_phases_dense; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for s in 2 3 4 0; do
  KIND=structured RTMODT_NMS_STOP=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O/x -- python3 tools/nms_alone.py > /dev/null 2> $O/x.log || { echo failed; continue; }
  f=$(ls $O/x/*/*_kernel_stats.csv | head -1)
  python3 - $f $s <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "nms_kernel" in r["Name"]:
        print("stop=%s %-16s calls %s avg %.1f us min %.1f max %.1f" % (sys.argv[2], r["Name"].split("(")[0].split("::")[-1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $O/x
done
