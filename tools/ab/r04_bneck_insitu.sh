#!/bin/bash
# r04: the tuner times a Bottleneck's two forms ALONE (fused kernel vs two ping-pong launches: 59.7 vs 50.2 us at c = 64).  In the staged bench the launches
# share the device with two other stages: does forcing the fused form (fewer bytes, smaller LDS footprint) change the step?  Same box, alternating.
O=gpurun_out/r04/bneck_insitu; mkdir -p $O
Q="--steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
for i in 1 2; do
  for m in tuner 1 0; do
    if [ $m = tuner ]; then unset RTMODT_BNECK; else export RTMODT_BNECK=$m; fi
    timeout -k 10 200 python3 bench.py $Q > $O/${m}_$i.json 2> /dev/null || exit 1
    python3 - <<PY
import json
d=json.loads(open("$O/${m}_$i.json").read().strip().splitlines()[-1]); print("bneck=$m run $i:", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["conv_kernels_ms_eager"])
PY
  done
done
