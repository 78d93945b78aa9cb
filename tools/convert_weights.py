#!/usr/bin/env python3
"""yolov8*.pt (Ultralytics checkpoint) -> RTMODTW1 fused-weight file, without Ultralytics.
    python tools/convert_weights.py yolov8s.pt weights/yolov8s.rtw
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
if len(sys.argv) < 3:
    sys.exit(__doc__)
scale, nc = pkg.weights.convert_pt(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
print(f"wrote {sys.argv[2]}: YOLOv8{scale}, nc={nc}")
