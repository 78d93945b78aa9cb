#!/bin/bash
L="real-time-multi-object-detection---tracking-system_amd/lib"
cp $L/librtmodt_hip.so $L/new.so.keep
timeout -k 10 300 python -m pytest tests/test_gpu_detector.py -q -m gpu -x -k "nms" 2>&1 | tail -2 || exit 1
echo "== new"; timeout -k 10 120 python tools/nms_time.py || exit 1
cp $L/librtmodt_hip_prev.so $L/librtmodt_hip.so
echo "== prev"; timeout -k 10 200 python tools/nms_time.py
cp $L/new.so.keep $L/librtmodt_hip.so
