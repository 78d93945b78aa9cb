#!/bin/bash
# one-off: per-kernel instruction mix on the plain engine (every kernel alone on the device)
set -o pipefail
O=gpurun_out/r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L > $O/counters.txt 2>&1
export RTMODT_TUNE_CACHE=/tmp/rtmodt_tune_r.txt
B="python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0 --prewarm 0"
RTMODT_CHAINS=1 $B --steps 20 --warmup 5 > $O/bench0.json 2>/dev/null || exit 1
RTMODT_CHAINS=1 RTMODT_CHAIN_PROBE=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
    --output-format csv -d $O/pmc_a -- $B --steps 10 --warmup 3 > /dev/null 2> $O/pmc_a.log || { tail -5 $O/pmc_a.log; exit 1; }
RTMODT_CHAINS=1 RTMODT_CHAIN_PROBE=0 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d $O/pmc_b -- $B --steps 10 --warmup 3 > /dev/null 2> $O/pmc_b.log || { tail -5 $O/pmc_b.log; exit 1; }
ls $O/pmc_a/* $O/pmc_b/* | head
