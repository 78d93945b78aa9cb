#!/bin/bash
# r03: does rocprofv3 --kernel-trace crash on the plain (one-graph) engine only with graphs?
O=gpurun_out/rp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 30 --warmup 5"
export RTMODT_TUNE_CACHE=/tmp/tune_rp.txt
$B > /dev/null 2>&1
RTMODT_CHAINS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -- $B --no-graph > $O/eager.json 2> $O/eager.log; echo "eager rc=$?"
RTMODT_CHAINS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -- $B > $O/graph.json 2> $O/graph.log; echo "graph rc=$?"
RTMODT_CHAINS=1 RTMODT_ONE_EXEC=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph1 -- $B > $O/graph1.json 2> $O/graph1.log; echo "graph one-exec rc=$?"
RTMODT_CHAINS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/chains2 -- $B > $O/chains2.json 2> $O/chains2.log; echo "chains2 rc=$?"
rm -rf $O/*/*/*_kernel_trace.csv
