#!/usr/bin/env python3
"""How much of a bench step the HOST needs: time inside submit (enqueue + tracker calls) and inside fetch
(wait for the previous batch + conversion to Detections), per step."""
import os, sys, tempfile, time, gc
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rtmodt_amd  # noqa
pkg = sys.modules["rtmodt_amd"]
from importlib import import_module
core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
size, R, S = 640, 8, 8
wpath = os.path.join(tempfile.gettempdir(), "rtmodt_bench_yolov8s_640.rtw")
if not os.path.exists(wpath):
    pkg.weights.save(wpath, pkg.weights.synthetic("s"), "s")
per = size * size * 3
for F in (1, 2, 4):
    ring = pkg._ffi.DeviceBuffer(S * R * per)
    for s in range(S):
        ring.upload(pkg.synth.frames(R, size, size, seed=1234 + s), offset=s * R * per)
    ptrs = [[ring.ptr + (s * R + r) * per for s in range(S)] for r in range(R)]
    det = pkg.Detector(wpath, batch=S * F, warmup=False, max_det=100)
    trk = core_cls(device=0, n_streams=S, max_dets=128, max_tracks=2048)
    def submit(t):
        det.enqueue([pt for f in range(F) for pt in ptrs[(t * F + f) % R]], height=size, width=size)
        for f in range(F):
            trk.update_from_detector(det, f * S, S)
    gc.collect(); gc.freeze(); gc.disable()
    submit(0)
    for t in range(1, 30):
        submit(t); det.fetch()
    ts = tf = 0.0
    n = 300
    t0 = time.perf_counter()
    for t in range(n):
        a = time.perf_counter(); submit(30 + t); b = time.perf_counter(); det.fetch(); c = time.perf_counter()
        ts += b - a; tf += c - b
    det.synchronize()
    el = time.perf_counter() - t0
    gc.enable()
    print(f"F={F}: step {el / n * 1e3:.3f} ms; host in submit {ts / n * 1e3:.3f} ms, in fetch (incl. waiting for the GPU) {tf / n * 1e3:.3f} ms", flush=True)
    det.fetch(); det.close(); trk.close(); ring.free()
