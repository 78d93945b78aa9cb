#!/bin/bash
# r04: pin rocprofv3 --kernel-trace's segfault inside hipGraphLaunch to the profiler or to the engine.
#  (1) tools/probes/graph_trace_repro.hip: no engine code, one graph of 45 spinning kernels, 330 launches, 4 in flight -- under the same profiler command;
#  (2) the plain engine under the profiler with the HIP runtime's AQL packet capture for graphs switched off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0);
#  (3) the plain engine under the profiler as it is (control; crashed in tools/collect_profiles.sh on this tree).
O=gpurun_out/r04/graph_trace_repro; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_trace_repro tools/probes/graph_trace_repro.hip || exit 1
rp() { name=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.out 2> $O/$name.log; echo "$name: rc=$?" | tee -a $O/summary.txt; rm -f $O/$name/*/*_kernel_trace.csv; }
rp repro_45x330_d4_e1 /tmp/graph_trace_repro 45 330 4 1 40
rp repro_45x1000_d8_e1 /tmp/graph_trace_repro 45 1000 8 1 10
rp repro_90x400_d4_e1 /tmp/graph_trace_repro 90 400 4 1 20
export RTMODT_TUNE_CACHE=/tmp/tune_gtr.txt
B="bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 100 --warmup 10"
python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2 --steps 5 --warmup 2 > /dev/null 2>&1
export RTMODT_CHAINS=1
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rp engine_plain_nocapture python3 $B
unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
rp engine_plain_control python3 $B
cat $O/summary.txt
