"""Autotuned engine at the benchmarked shape: launch list, detection counts and the first stored layer that leaves the fp32
oracle's tolerance (teacher-forced).  python3 tools/diag/autotuned_layers.py [batch] [size] [autotune 0/1] [scale n/s/m]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

pkg = importlib.import_module("real-time-multi-object-detection---tracking-system_amd")
from oracle import yolo_oracle as Y

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 640
tune = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
scale = sys.argv[4] if len(sys.argv) > 4 else "s"
wdir = tempfile.mkdtemp()
path = os.path.join(wdir, scale + ".rtw")
w = pkg.weights.synthetic(scale, input_size=size, calibrate="noise")
pkg.weights.save(path, w, scale)
w, _, _, _ = pkg.weights.load(path)
det = pkg.Detector(path, input_size=(size, size), warmup=False, batch=B, autotune=tune, chains=-2, max_det=100)
frames = list(pkg.synth.frames(B, size, size, seed=1234))
res = det.detect_batch(frames)
print("detections per image:", [len(r) for r in res])
for n, ms, _ in det.profile(1):
    print("   ", n, f"{ms * 1e3:.1f} us")
inp, _, _ = det.debug_fetch(0, want_heads=False, want_pred=False)
names = [x.name for x in pkg.weights.spec(scale)]
gpu = {}
for n in names:
    try:
        gpu[n] = det.debug_layer(n, 0).astype(np.float32)
    except pkg._ffi.RtmodtError as e:
        print("   not stored:", n, str(e)[:80])
taps = {}
Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
for n in names:
    if n in gpu:
        err = float(np.abs(taps[n] - gpu[n]).max())
        tol = 4e-3 * np.abs(taps[n]).max() + 2e-3
        print(f"{'BAD ' if err > tol else 'ok  '} {n:14s} max|d| {err:.4g}  tol {tol:.4g}  max|ref| {np.abs(taps[n]).max():.3g}")
