#!/usr/bin/env python3
"""Experiment: do two independent detector handles (own streams, own graphs) overlap on one GPU?"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa
pkg = sys.modules["rtmodt_amd"]
size, R = 640, 8
wpath = os.path.join(tempfile.gettempdir(), "rtmodt_bench_yolov8s_640.rtw")
if not os.path.exists(wpath):
    pkg.weights.save(wpath, pkg.weights.synthetic("s"), "s")
per = size * size * 3
def run(n_det, batch, steps=200):
    ring = pkg._ffi.DeviceBuffer(batch * R * per)
    for s in range(batch):
        ring.upload(pkg.synth.frames(R, size, size, seed=1234 + s), offset=s * R * per)
    dets = [pkg.Detector(wpath, batch=batch, warmup=False) for _ in range(n_det)]
    ptrs = [[ring.ptr + (s * R + r) * per for s in range(batch)] for r in range(R)]
    for d in dets: d.enqueue(ptrs[0], height=size, width=size)      # 2-deep pipeline per handle, as bench.py runs it
    for t in range(1, 20):
        for d in dets: d.enqueue(ptrs[t % R], height=size, width=size)
        for d in dets: d.fetch()
    dets[0].synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        for d in dets: d.enqueue(ptrs[t % R], height=size, width=size)
        for d in dets: d.fetch()
    dets[0].synchronize()
    dt = time.perf_counter() - t0
    for d in dets:
        d.fetch(); d.close()
    ring.free()
    return n_det * batch * steps / dt
for n_det, batch in ((1, 8), (1, 16), (2, 8), (1, 16), (2, 8), (3, 8), (2, 16), (1, 32)):
    print(f"{n_det} detector(s) x batch {batch}: {run(n_det, batch):8.0f} fps", flush=True)
