#!/bin/bash
# r04: do the ping-pong launches (one persistent workgroup per CU, the CU then exclusively theirs) better leave some CUs to the other stages' HBM-bound launches?
# Diagnostic build in the box's scratch copy; RTMODT_PP_RESERVE = CUs left free by every conv3x3_pp / conv_tile_pp launch.  Same box, one tune cache.
O=gpurun_out/r04/pp_reserve; mkdir -p $O
P=real-time-multi-object-detection---tracking-system_amd
(cd $P/csrc && rm -f build/conv_pp.o build/conv.o build/engine.o build/postprocess.o && make DIAG=1 > /tmp/diag_build.log 2>&1) || { tail -5 /tmp/diag_build.log; exit 1; }
export RTMODT_TUNE_CACHE=/tmp/tune_ppres.txt
Q="--steps 100 --warmup 10 --no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 0"
python3 bench.py $Q > /dev/null 2>&1
for r in 0 16 32 48 0 24; do RTMODT_PP_RESERVE=$r timeout -k 10 200 python3 bench.py $Q > $O/r$r.json 2> /dev/null || exit 1; python3 - <<PY
import json
d=json.loads(open("$O/r$r.json").read().strip().splitlines()[-1]); print("reserve $r:", d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
done
