#!/bin/bash
# write-through on the 16-byte stores only; frames-per-stream sweep
set -o pipefail
mkdir -p gpurun_out
export RTMODT_TUNE_CACHE=/tmp/tune_c.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python bench.py --steps 100 --warmup 10 $Q > /dev/null 2>&1
for rep in 1 2; do
for wt in 0 1; do
  RTMODT_WT=$wt python bench.py --steps 300 --warmup 30 $Q > gpurun_out/c_wt${wt}_$rep.json 2> /dev/null
done
done
for F in 1 3 4 6 8; do
  RTMODT_WT=1 python bench.py --steps 150 --warmup 20 $Q --frames-per-stream $F > gpurun_out/c_F$F.json 2> gpurun_out/c_F$F.err
done
for f in gpurun_out/c_*.json; do echo -n "$f "; python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'])"; done
