"""One image through rtmodt_nms_pred with N candidates (uniform boxes, 6 classes, tied scores), 20 times: run under rocprofv3 --kernel-trace --stats
(tools/ab/r03_nms_sort.sh) to time nms_kernel alone as a function of the candidate count."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
n = int(sys.argv[1])
A = 8400
rng = np.random.default_rng(n)
pred = np.zeros((84, A), np.float32)
pred[0] = rng.uniform(0, 640, A); pred[1] = rng.uniform(0, 640, A)
pred[2] = rng.uniform(10, 120, A); pred[3] = rng.uniform(10, 120, A)
hot = rng.permutation(A)[:n]
pred[4 + rng.integers(0, 6, n), hot] = np.round(rng.uniform(0.36, 0.99, n), 3).astype(np.float32)
for _ in range(20):
    out = pkg._ffi.nms_pred(pred, max_det=100)
print(n, "candidates ->", len(out[3]), "kept")
