#!/usr/bin/env python3
"""Wall time of one synchronous NMS call (C ABI, pred tensor uploaded inside the call) against the candidate count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rtmodt_amd  # noqa
pkg = sys.modules["rtmodt_amd"]
for n_anchors, frac in ((8400, 0.02), (8400, 0.3), (8400, 0.6), (8400, 1.0), (33600, 0.3), (33600, 1.0)):
    rng = np.random.default_rng(n_anchors + int(frac * 100))
    pred = np.zeros((84, n_anchors), np.float32)
    pred[0] = rng.uniform(0, 640, n_anchors); pred[1] = rng.uniform(0, 640, n_anchors)
    pred[2] = rng.uniform(10, 200, n_anchors); pred[3] = rng.uniform(10, 200, n_anchors)
    hot = rng.uniform(size=n_anchors) < frac
    cls = rng.integers(0, 80, n_anchors)
    sc = rng.uniform(0.36, 0.99, n_anchors).astype(np.float32)
    pred[4 + cls[hot], np.nonzero(hot)[0]] = sc[hot]
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        xy, cf, ci, an = pkg._ffi.nms_pred(pred, conf=0.35, iou=0.45, classes=None, agnostic=False, max_det=100)
        ts.append(time.perf_counter() - t0)
    print(f"{n_anchors} anchors, {int(hot.sum())} candidates, {len(an)} kept: {min(ts[1:]) * 1e3:.3f} ms")
