#!/bin/bash
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2; do
python bench.py --steps 150 --warmup 20 $Q > gpurun_out/p_off_$rep.json 2>/dev/null
RTMODT_TUNE_WS=1 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/p_ws1_$rep.json 2>/dev/null
RTMODT_TUNE_WS=1 RTMODT_TUNE_WS_BIAS=0.85 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/p_ws085_$rep.json 2>/dev/null
done
for f in gpurun_out/p_*.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
