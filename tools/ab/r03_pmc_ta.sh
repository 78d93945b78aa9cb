#!/bin/bash
# r03: texture-addresser / vector-L1 counters per kernel over the staged bench (two separate --pmc passes, kernel serialised by the counter collection):
# how busy the vector-memory path is in the conv kernels, how long it is stalled by the L2 side, the mean L2 read latency a CU sees
set -o pipefail
O=gpurun_out/pmc_ta; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export RTMODT_TUNE_CACHE=/tmp/tune_ta.txt
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0"
$B --steps 20 --warmup 5 > /dev/null 2>&1
# few counters per pass: the TA block holds two per instance ("Request exceeds the capabilities of the hardware to collect" aborts rocprofv3 otherwise, and
# the aborted process then hangs on its incomplete dispatch -- hence the timeouts)
pass() { n=$1; shift; RTMODT_CHAIN_PROBE=0 timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $O/$n -- $B --steps 10 --warmup 3 > /dev/null 2> $O/$n.log || { echo "pass $n failed"; grep -i "error code" $O/$n.log | head -2; return 1; }; echo "pass $n done"; }
pass p1 TA_TA_BUSY_sum GRBM_GUI_ACTIVE || exit 1
pass p2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum || exit 1
pass p3 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum || exit 1
pass p4 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum || exit 1
python3 - <<'PY' | tee gpurun_out/pmc_ta/summary.txt
import csv,glob,collections,re
def load(d):
    f=glob.glob(d+"/*/*counter_collection.csv")[0]
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").replace("rtmodt::","")[:44]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        key=(r["Dispatch_Id"]); 
        if (k,key) not in seen: seen.add((k,key)); n[k]+=1
    return acc,n
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for d in ("p1","p2","p3","p4"):
    a,c=load("gpurun_out/pmc_ta/"+d)
    for k in a:
        for m,v in a[k].items(): acc[k][m]+=v
        n[k]=max(n[k],c[k])
print("%-44s %6s %9s %12s %12s %11s %11s %11s"%("kernel","calls","TA busy%","addr stall%","data stall%","L2 lat clk","lines/call","pend stall%"))
inst=32.0      # the _sum counters add the 256 CUs' instances and GRBM_GUI_ACTIVE adds the 8 XCDs' (it is 8 x the launch's cycles): 256 / 8
for k in sorted(acc, key=lambda k:-acc[k].get("GRBM_GUI_ACTIVE",0))[:18]:
    a=acc[k]; g=max(a.get("GRBM_GUI_ACTIVE",1.0),1.0); c=max(n[k],1)
    rq=a.get("TCP_TCC_READ_REQ_sum",0)
    print("%-44s %6d %9.1f %12.1f %12.1f %11.0f %11.0f %11.1f"%(k,c,100*a.get("TA_TA_BUSY_sum",0)/(g*inst),100*a.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum",0)/(g*inst),
        100*a.get("TA_DATA_STALLED_BY_TC_CYCLES_sum",0)/(g*inst),a.get("TCP_TCC_READ_REQ_LATENCY_sum",0)/max(rq,1),rq/c,100*a.get("TCP_PENDING_STALL_CYCLES_sum",0)/(g*inst)))
PY
