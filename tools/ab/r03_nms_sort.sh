#!/bin/bash
# r03: nms_kernel (1024 threads) alone on the device as a function of the candidate count: LDS rank sort up to RTMODT_NMS_RANK_MAX candidates,
# bitonic network above; stop=3: up to and including the sort, stop=0: the whole kernel
set -o pipefail
O=gpurun_out/nms_sort; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for N in 60 120 250 500 1000 2000; do
for RM in 2048 0; do
for s in 3 0; do
  RTMODT_NMS_RANK_MAX=$RM RTMODT_NMS_STOP=$s rocprofv3 --kernel-trace --stats --output-format csv -d $O/x -- python3 tools/nms_one.py $N > /dev/null 2> $O/x.log || { echo "failed"; continue; }
  f=$(ls $O/x/*/*_kernel_stats.csv | head -1)
  python3 - $f $s $RM $N <<'PY' | tee -a $O/summary.txt
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "nms_kernel" in r["Name"]:
        print("n=%s rank_max=%s stop=%s avg %.1f us min %.1f max %.1f" % (sys.argv[4], sys.argv[3], sys.argv[2], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
  rm -rf $O/x
done
done
done
