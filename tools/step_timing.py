#!/usr/bin/env python3
"""Where a bench step's wall time goes on the host: submit (enqueue + tracker launch) vs fetch."""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa
pkg = sys.modules["rtmodt_amd"]
from importlib import import_module
core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
S, R, size = 8, 16, 640
wpath = os.path.join(tempfile.gettempdir(), "rtmodt_bench_yolov8s_640.rtw")
if not os.path.exists(wpath):
    pkg.weights.save(wpath, pkg.weights.synthetic("s"), "s")
per = size * size * 3
ring = pkg._ffi.DeviceBuffer(S * R * per)
for s in range(S):
    ring.upload(pkg.synth.frames(R, size, size, seed=1234 + s), offset=s * R * per)
det = pkg.Detector(wpath, batch=S, warmup=False)
trk = core_cls(n_streams=S, max_dets=128, max_tracks=2048)
ptrs = [[ring.ptr + (s * R + r) * per for s in range(S)] for r in range(R)]
def submit(t):
    det.enqueue(ptrs[t % R], height=size, width=size); trk.update_from_detector(det)
submit(0)
for t in range(1, 40):
    submit(t); det.fetch()
ts, tf, tt = [], [], []
t_all = time.perf_counter()
for t in range(40, 340):
    a = time.perf_counter(); submit(t); b = time.perf_counter(); det.fetch(); c = time.perf_counter(); det.last_timing(); d = time.perf_counter()
    ts.append(b - a); tf.append(c - b); tt.append(d - c)
det.synchronize()
wall = (time.perf_counter() - t_all) / 300
print(f"step wall {wall*1e3:.3f} ms | host submit {np.mean(ts)*1e3:.3f} ms, fetch (incl. wait) {np.mean(tf)*1e3:.3f} ms, last_timing {np.mean(tt)*1e3:.3f} ms")
per = np.asarray(ts) + np.asarray(tf) + np.asarray(tt)
for i in range(0, 300, 50):
    print(f"  steps {i:3d}-{i+49:3d}: {per[i:i+50].mean()*1e3:.3f} ms/step  live tracks stream0 = ?")
idx = np.argsort(-per)[:6]
print("slowest steps:", [(int(i), round(float(per[i])*1e3, 2), round(float(ts[i])*1e3,2), round(float(tf[i])*1e3,2)) for i in idx])
print("median step", float(np.median(per))*1e3)
print("tracks per stream now:", [len(trk.snapshot(s)["ids"]) for s in range(S)])
# no tracker / no fetch-wait variants
for t in range(340, 360):
    submit(t); det.fetch()
