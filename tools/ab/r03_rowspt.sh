#!/bin/bash
# r03: persistent tap-reuse kernel (conv3x3_rows_stream): parity, then the tuner's per-layer timings and the layer table at 32 frames
set -e
O=gpurun_out/rowspt; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_persistent_tap_reuse_kernel or test_tap_reuse_conv_tiles" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 timeout -k 10 300 python3 tools/profile_layers.py > $O/layers.txt 2> $O/tune.log
grep -E "rows|^total|22\." $O/layers.txt
grep -E "rows-pt|rows64-pt" $O/tune.log | head -40
