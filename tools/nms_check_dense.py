"""Every image of STEPS batches at the benchmarked shape on structured frames (a saturated random head: thousands of candidates per image):
fetched detections against oracle NMS on the engine's own pre-NMS tensor, bit for bit.  (Checker use of oracle/: a tool, not the product.)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401
from oracle import yolo_oracle as Y  # noqa: E402

pkg = sys.modules["rtmodt_amd"]
S, F, size, steps = 8, 4, 640, int(os.environ.get("STEPS", "8"))
B = S * F
path = os.path.join(tempfile.gettempdir(), "nms_alone_yolov8s_640.rtw")
if not os.path.exists(path):
    pkg.weights.save(path, pkg.weights.synthetic("s", input_size=size), "s")
det = pkg.Detector(path, input_size=(size, size), warmup=False, batch=B, autotune=False, chains=1, max_det=100)
gen = pkg.synth.structured_frames if os.environ.get("KIND", "structured") == "structured" else pkg.synth.frames
frames = np.stack([gen(F * steps, size, size, seed=1234 + s) for s in range(S)], 1).reshape(steps, F, S, size, size, 3)
buf = pkg._ffi.DeviceBuffer(frames.nbytes)
buf.upload(frames)
per = size * size * 3
bad = tot = 0
for t in range(steps):
    det.enqueue([buf.ptr + ((t * F + f) * S + s) * per for f in range(F) for s in range(S)], height=size, width=size)
    got = det.fetch()
    for i in range(B):
        _, _, pred = det.debug_fetch(i, want_input=False, want_heads=False)
        dets, _ = Y.non_max_suppression(pred, det.confidence, det.iou, det.classes, det.agnostic_nms, 100)
        ref = Y.scale_boxes(dets[:, :4], size, size, size, size) if len(dets) else np.empty((0, 4), np.float32)
        d = got[i]
        ok = len(d) == len(dets) and np.array_equal(d.xyxy.view(np.int32), ref.view(np.int32)) and \
            np.array_equal(d.confidence.view(np.int32), dets[:, 4].astype(np.float32).view(np.int32)) and d.class_id.tolist() == dets[:, 5].astype(np.int32).tolist()
        tot += 1
        if not ok:
            bad += 1
            n = int((pred[4:].max(0) > det.confidence).sum())
            print(f"MISMATCH step {t} image {i}: {n} candidates, engine {len(d)} vs oracle {len(dets)} detections")
print(f"{tot} images checked, {bad} mismatches")
sys.exit(1 if bad else 0)
