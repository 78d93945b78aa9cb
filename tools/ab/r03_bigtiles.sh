#!/bin/bash
# r03: big tiles (256x128 s3, 256x256 s2; one workgroup per CU): parity, then the tuner's per-layer timings at 32 frames
set -e
O=gpurun_out/bigtiles; mkdir -p $O
python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_eight_wave_tiles" > $O/tests.txt 2>&1 || { tail -n 30 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python3 tools/profile_layers.py > $O/layers.txt 2> $O/tune.log
tail -n 50 $O/layers.txt
