#!/bin/bash
# clocks and package power while the bench runs: staged default, plain engine, 1 frame per stream
mkdir -p gpurun_out/x
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 300 --warmup 20"
sample() {  # $1 = tag, rest = env + command
  tag=$1; shift
  env "$@" > gpurun_out/x/$tag.json 2>/dev/null &
  pid=$!
  sleep 12
  for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|Graphics Package" >> gpurun_out/x/$tag.smi; sleep 1; done
  wait $pid
}
sample staged RTMODT_X=1 python bench.py $Q --long 14
sample chains1 RTMODT_CHAINS=1 python bench.py $Q --long 14
sample frames1 RTMODT_X=1 python bench.py $Q --long 14 --frames-per-stream 1
for t in staged chains1 frames1; do echo "== $t"; python -c "import json; j=json.loads(open('gpurun_out/x/$t.json').read().strip().splitlines()[-1]); print(j['value'], j['timing']['long_window']['value'])"; grep -E "sclk" gpurun_out/x/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; grep -E "Power" gpurun_out/x/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; done
