#!/bin/bash
# A/B: write-through (sc1) output stores vs plain stores -- staged bench, plain engine bench, per-layer times; same tune cache
set -o pipefail
mkdir -p gpurun_out
export RTMODT_TUNE_CACHE=/tmp/tune_b.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python bench.py --steps 100 --warmup 10 $Q > /dev/null 2>&1     # tunes once
for rep in 1 2; do
for wt in 0 1; do
  RTMODT_WT=$wt python bench.py --steps 300 --warmup 30 $Q > gpurun_out/wt${wt}_staged_$rep.json 2> /dev/null
  RTMODT_WT=$wt RTMODT_CHAINS=1 python bench.py --steps 200 --warmup 30 $Q > gpurun_out/wt${wt}_plain_$rep.json 2> /dev/null
done
done
RTMODT_WT=0 RTMODT_CHAINS=1 python tools/profile_layers.py > gpurun_out/layers_wt0.txt 2> /dev/null
RTMODT_WT=1 RTMODT_CHAINS=1 python tools/profile_layers.py > gpurun_out/layers_wt1.txt 2> /dev/null
for f in gpurun_out/wt*_*.json; do echo -n "$f "; python -c "import json,sys; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
paste <(cut -c1-60,73-82 gpurun_out/layers_wt0.txt) <(cut -c73-82 gpurun_out/layers_wt1.txt)
