#!/bin/bash
# how much of the staged bench's rate is the chip's power limit answering the DATA: white-noise frames against smooth ones
mkdir -p gpurun_out/fk; rm -f gpurun_out/fk/*.smi
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --steps 300 --warmup 20 --long 10"
for rep in 1 2; do
for k in noise structured; do
  python bench.py $Q --frames-kind $k > gpurun_out/fk/${k}_$rep.json 2>/dev/null &
  pid=$!
  sleep 25
  for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|Graphics Package" >> gpurun_out/fk/${k}_$rep.smi; sleep 1; done
  wait $pid
done
done
for t in noise_1 structured_1 noise_2 structured_2; do echo "== $t"; python -c "import json; j=json.loads(open('gpurun_out/fk/$t.json').read().strip().splitlines()[-1]); print(j['value'], j['timing']['long_window']['value'], j.get('detections_per_frame'))"; grep -E "sclk" gpurun_out/fk/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; grep -E "Power" gpurun_out/fk/$t.smi | awk '{print $NF}' | tr '\n' ' '; echo; done
