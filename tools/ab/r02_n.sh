#!/bin/bash
mkdir -p gpurun_out
export RTMODT_TUNE_CACHE=/tmp/tune_n.txt
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python bench.py --steps 100 --warmup 10 $Q > gpurun_out/n_hbm.json 2>/dev/null
for rep in 1 2; do
python bench.py --steps 200 --warmup 20 $Q > gpurun_out/n_hbm3_$rep.json 2>/dev/null
python bench.py --steps 200 --warmup 20 $Q --stages 2 > gpurun_out/n_hbm2_$rep.json 2>/dev/null
python bench.py --steps 200 --warmup 20 $Q --host-frames > gpurun_out/n_host3_$rep.json 2>/dev/null
python bench.py --steps 200 --warmup 20 $Q --host-frames --stages 2 > gpurun_out/n_host2_$rep.json 2>/dev/null
done
for f in gpurun_out/n_*_*.json; do echo -n "$f "; python -c "import json; j=json.loads(open('$f').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; done
