#!/bin/bash
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
export RTMODT_TUNE_CACHE=/tmp/tune_l.txt
python bench.py --steps 100 --warmup 10 $Q > /dev/null 2>&1
for g in 512 1024 2048 0; do RTMODT_SL1_GRID=$g RTMODT_STEM_L1=1 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/l_$g.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/l_$g.json').read().strip().splitlines()[-1]); print('grid $g', j['value'], j['ms_per_step'], [x for x in j['roofline']['slowest_launches'] if x['op'].startswith('0 ')])"; done
RTMODT_STEM_L1=0 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/l_off.json 2>/dev/null; python -c "import json; j=json.loads(open('gpurun_out/l_off.json').read().strip().splitlines()[-1]); print('off', j['value'], j['ms_per_step'])"
