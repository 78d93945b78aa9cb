#!/bin/bash
# clocks and package power, same box: previous library vs working tree's (alternating, 2 pairs)
mkdir -p gpurun_out/y; rm -f gpurun_out/y/*.smi
L="real-time-multi-object-detection---tracking-system_amd/lib"
cp $L/librtmodt_hip.so $L/new.so.keep
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 300 --warmup 20 --long 10"
for rep in 1 2; do
for v in prev new; do
  if [ $v = prev ]; then cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so; else cp $L/new.so.keep $L/librtmodt_hip.so; fi
  python bench.py $Q > gpurun_out/y/${v}_$rep.json 2>/dev/null &
  pid=$!
  sleep 11
  for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|Graphics Package" >> gpurun_out/y/${v}_$rep.smi; sleep 1; done
  wait $pid
done
done
cp $L/new.so.keep $L/librtmodt_hip.so
for t in prev_1 new_1 prev_2 new_2; do echo "== $t"; python -c "import json; j=json.loads(open('gpurun_out/y/$t.json').read().strip().splitlines()[-1]); print(j['value'], j['timing']['long_window']['value'])"; grep -E "sclk" gpurun_out/y/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; grep -E "Power" gpurun_out/y/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; done
