#!/bin/bash
# r03: fused Bottleneck with the any-row conflict-free LDS swizzle: parity tests, then the layer table at 32 frames
set -e
O=gpurun_out/bneck; mkdir -p $O
python3 -m pytest tests/test_gpu_detector.py -m gpu -x -q -k "test_fused_bottleneck_kernel or test_bottleneck_with_c2f_cv2_tail or test_forward_layers or test_benchmarked_shape_parity or test_config5" > $O/tests.txt 2>&1 || { tail -n 40 $O/tests.txt; exit 1; }
tail -n 3 $O/tests.txt
RTMODT_TUNE_LOG=1 RTMODT_CHAINS=1 python3 tools/profile_layers.py > $O/layers.txt 2> $O/tune.log
grep -E "bottleneck|two launches|^total" $O/layers.txt
grep -E "fused +[0-9.]+ us vs two|with cv2 tail" $O/tune.log
