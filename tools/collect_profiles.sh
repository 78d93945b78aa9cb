#!/bin/bash
# Everything profiles/rNN/ is distilled from, in one go on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r02'   then   python tools/summarize_profiles.py gpurun_out/r02 profiles/r02
# PMC counters are collected in their own passes (never together with a trace domain other than kernel-trace); counter
# collection serialises every kernel, so the engine's stream-overlap probe is switched off there (RTMODT_CHAIN_PROBE=0) to
# keep the two-stream configuration the bench runs.
set -o pipefail
R=${1:-r02}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the first run tunes and records its choices; every later process (profilers included) replays exactly that configuration
export RTMODT_TUNE_CACHE=/tmp/rtmodt_tune_$R.txt
rm -f $RTMODT_TUNE_CACHE
python bench.py > $O/bench_default.json 2> $O/bench_default.log || exit 1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> /dev/null || exit 1
echo "[collect] bench done"
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0.2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B --steps 100 --warmup 10 > $O/stats_bench.json 2> $O/stats.log || exit 1
python tools/trace_gaps.py $O/stats/*/*_kernel_trace.csv 50 > $O/step_gaps.txt 2>&1
rm -f $O/stats/*/*_kernel_trace.csv
# the same bench on the PLAIN engine (one stream, one graph): every forward-pass kernel alone on the device, so the per-kernel
# averages of the trace can be set against the HIP-event times of tools/profile_layers.py and bench.py (in the staged default
# the kernels of the two stages overlap and each one's duration in the trace includes the time it shared the CUs)
# (100 steps.  rocprofv3 --kernel-trace segfaults inside hipGraphLaunch after 200-300 launches of ONE big captured graph -- the plain engine's, or
# the 45-node graph of tools/probes/graph_trace_repro.hip, which holds no engine code -- when the HIP runtime submits the graph's pre-built AQL
# packets; with that path off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: the runtime enqueues the nodes one by one) the trace completes.  The variable
# is set for THIS profiler pass only; profiles/r04/graph_trace_repro/ holds the five runs)
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 RTMODT_CHAINS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_chains1 -- $B --steps 100 --warmup 10 > $O/stats_chains1_bench.json 2> $O/stats_chains1.log || exit 1
python tools/trace_gaps.py $O/stats_chains1/*/*_kernel_trace.csv 50 > $O/step_gaps_chains1.txt 2>&1
rm -f $O/stats_chains1/*/*_kernel_trace.csv
echo "[collect] kernel trace done"
RTMODT_CHAIN_PROBE=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B --prewarm 0 --steps 20 --warmup 5 > /dev/null 2> $O/pmc_fetch.log || exit 1
RTMODT_CHAIN_PROBE=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B --prewarm 0 --steps 20 --warmup 5 > /dev/null 2> $O/pmc_write.log || exit 1
RTMODT_CHAIN_PROBE=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA \
    --output-format csv -d $O/pmc_sq -- $B --prewarm 0 --steps 20 --warmup 5 > /dev/null 2> $O/pmc_sq.log || exit 1
echo "[collect] pmc done"
python tools/profile_layers.py > $O/layers.txt 2> /dev/null
RTMODT_CHAINS=1 python tools/profile_layers.py > $O/layers_chains1.txt 2> /dev/null
RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 2 > $O/layers_chains1_F2.txt 2> /dev/null
RTMODT_CHAINS=1 python tools/profile_layers.py --frames-per-stream 8 > $O/layers_chains1_F8.txt 2> /dev/null
RTMODT_CHAINS=1 $B --steps 300 --warmup 30 > $O/bench_chains1.json 2> /dev/null
RTMODT_CHAINS=2 $B --steps 300 --warmup 30 > $O/bench_chains2.json 2> /dev/null
$B --steps 300 --warmup 30 --stages 2 > $O/bench_stages2.json 2> /dev/null
$B --steps 300 --warmup 30 --frames-per-stream 1 > $O/bench_frames1.json 2> /dev/null
$B --steps 300 --warmup 30 --frames-per-stream 2 > $O/bench_frames2.json 2> /dev/null
$B --steps 150 --warmup 20 --frames-per-stream 8 > $O/bench_frames8.json 2> /dev/null
$B --steps 300 --warmup 30 --streams 16 --frames-per-stream 1 > $O/bench_streams16.json 2> /dev/null
$B --steps 200 --warmup 20 --streams 32 --frames-per-stream 1 > $O/bench_streams32.json 2> /dev/null
$B --steps 300 --warmup 30 --host-frames > $O/bench_host_frames.json 2> /dev/null
$B --steps 300 --warmup 30 --host-frames --frames-per-stream 2 > $O/bench_host_frames_F2.json 2> /dev/null
$B --steps 200 --warmup 30 --host-frames --pageable > $O/bench_host_frames_pageable.json 2> /dev/null
python tools/run_pipeline_synth.py > $O/pipeline_640.json 2> /dev/null
python tools/run_pipeline_synth.py --source 1920x1080 > $O/pipeline_1080p.json 2> /dev/null
python tests/perf/tracker_modes.py > $O/tracker_modes.json 2> /dev/null
python tools/bw_probe.py > $O/bandwidth_probe.txt 2> /dev/null
echo "[collect] all done"
# soak: 3 000 steps of the default engine with the line's self-check (NMS bit-equal to the oracle, every stored layer of two images) at the end
python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --long 0 --prewarm 0.2 --steps 3000 --warmup 30 > $O/bench_soak3000.json 2> /dev/null || exit 1
echo "[collect] soak done"
