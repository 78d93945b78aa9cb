#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for rep in 1 2; do
for pen in 0 0.05 0.15 1.0; do
  RTMODT_TUNE_LDS_PENALTY=$pen python bench.py --steps 150 --warmup 20 $Q > gpurun_out/i_pen${pen}_$rep.json 2> /dev/null
  echo -n "penalty $pen rep $rep: "; python -c "import json; j=json.loads(open('gpurun_out/i_pen${pen}_$rep.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"
done
done
RTMODT_TUNE_LDS_PENALTY=1.0 RTMODT_TUNE_LDS_CAP=64 python bench.py --steps 150 --warmup 20 $Q > gpurun_out/i_cap64.json 2> /dev/null; python -c "import json; j=json.loads(open('gpurun_out/i_cap64.json').read().strip().splitlines()[-1]); print('cap64', j['value'], j['ms_per_step'])"
