#!/bin/bash
# r03: layer 1 in pixel-pair form: parity tests, then the layer table at 32 frames with and without it (same box)
set -e
O=gpurun_out/pair; mkdir -p $O
python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_layer1_pixel_pair_form or test_conv_with_fused_1x1_tail or test_stem_and_layer1 or test_forward_layers_yolov8s_640 or test_benchmarked_shape_parity" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python3 tools/profile_layers.py > $O/layers_pair.txt 2> $O/tune_pair.log
RTMODT_L1_PAIR=0 RTMODT_CHAINS=1 python3 tools/profile_layers.py > $O/layers_plain.txt 2> /dev/null
head -n 6 $O/layers_pair.txt; tail -n 1 $O/layers_pair.txt; head -n 6 $O/layers_plain.txt; tail -n 1 $O/layers_plain.txt
grep -E "^\[tune\] 1 " $O/tune_pair.log | sort -k4 -n | head -n 12
