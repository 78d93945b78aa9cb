#!/bin/bash
# r03: tap-reuse kernels no longer enumerate the top / bottom border rows: parity, layer table, bench
set -e
O=gpurun_out/rowskip; mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_persistent_tap_reuse_kernel or test_tap_reuse_conv_tiles or test_benchmarked_shape_parity or test_forward_layers or test_config5 or test_epilogue_variants" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_CHAINS=1 timeout -k 10 300 python3 tools/profile_layers.py > $O/layers.txt 2> /dev/null
grep -E "rows|^total" $O/layers.txt
python3 bench.py --no-cpu-baseline --no-latency --no-compare --no-host-leg --steps 200 --warmup 20 > $O/bench.json 2>/dev/null
python3 -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['roofline']['frac'], d['verified']['layers_ok'])"
