#!/bin/bash
# r04 (second and last run): a DIAGNOSTIC build of the library (-DRTMODT_DIAG, built on the box into /tmp) prints the launch index and can
# re-instantiate the two graph executables every N launches.  Which launch the segfault comes at with two executables / with one; whether
# re-instantiating every 64 launches gets past it.
O=gpurun_out/r04/rocprof_plain100; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P=real-time-multi-object-detection---tracking-system_amd
cp $P/lib/librtmodt_hip.so /tmp/librtmodt_product.so
(cd $P/csrc && rm -f build/engine.o && make EXTRA=-DRTMODT_DIAG > /tmp/diag_build.log 2>&1) || { tail -5 /tmp/diag_build.log; exit 1; }
B="python3 -X faulthandler bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 100 --warmup 10"
export RTMODT_TUNE_CACHE=/tmp/tune_rp100.txt
python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 5 --warmup 2 > /dev/null 2>&1
run() { name=$1; shift; env RTMODT_CHAINS=1 RTMODT_DEBUG_GRAPH_COUNT=1 "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- $B > $O/$name.json 2> $O/$name.log; echo "$name: rc=$? last launch printed: $(grep -a '\[graph\] launch' $O/$name.log | tail -1)" | tee -a $O/summary_b.txt; rm -f $O/$name/*/*_kernel_trace.csv; }
run two_execs A=1
run one_exec RTMODT_ONE_EXEC=1
run reinst64 RTMODT_DEBUG_REINST=64
cp /tmp/librtmodt_product.so $P/lib/librtmodt_hip.so
