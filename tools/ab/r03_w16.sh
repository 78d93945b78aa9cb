#!/bin/bash
# r03: the big tiles with 16 waves per workgroup: parity, then the tuner's timings
set -e
O=gpurun_out/w16; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_eight_wave_tiles" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 timeout -k 10 300 python3 tools/profile_layers.py > $O/layers.txt 2> $O/tune.log
grep -E "16w" $O/tune.log | head -60
tail -n 1 $O/layers.txt
grep -c "16w" $O/layers.txt || true
