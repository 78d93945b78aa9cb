#!/bin/bash
mkdir -p gpurun_out
Q="--no-cpu-baseline --no-latency --no-compare --no-host-leg --no-verify --long 8"
python bench.py --steps 300 --warmup 20 $Q > gpurun_out/j_bench.json 2> /dev/null &
BP=$!
sleep 6
for i in 1 2 3 4; do rocm-smi --showclocks --showpower --showuse 2>&1 | grep -E "sclk|mclk|fclk|Power|GPU use|busy" | head -8; echo ---; sleep 1; done
wait $BP
python -c "import json; j=json.loads(open('gpurun_out/j_bench.json').read().strip().splitlines()[-1]); print(j['value'], j['timing']['long_window'])"
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | head -4
