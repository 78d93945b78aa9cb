#!/bin/bash
# r04: the plain (one-graph) engine under rocprofv3 --kernel-trace at --steps 100, ONCE, with the Python fault handler on, and the
# same 100 steps without graphs -- which of the two the segfault follows.  Logs are kept whichever way they go (profiles/r04/rocprof_plain100/).
O=gpurun_out/r04/rocprof_plain100; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 -X faulthandler bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 100 --warmup 10"
export RTMODT_TUNE_CACHE=/tmp/tune_rp100.txt
python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 5 --warmup 2 > /dev/null 2>&1      # tunes, writes the cache
RTMODT_CHAINS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- $B > $O/graph.json 2> $O/graph.log; echo "graph, 100 steps: rc=$?" | tee -a $O/summary.txt
RTMODT_CHAINS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- $B --no-graph > $O/eager.json 2> $O/eager.log; echo "eager, 100 steps: rc=$?" | tee -a $O/summary.txt
ls $O/graph/*/ 2>/dev/null | head -5 >> $O/summary.txt
rm -f $O/*/*/*_kernel_trace.csv
tail -30 $O/graph.log >> $O/summary.txt
