#!/usr/bin/env python3
"""Per-launch device time of the forward pass (HIP events around every eager launch).
    python tools/profile_layers.py [--model s] [--size 640] [--streams 8] [--no-autotune]
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s")
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--streams", type=int, default=8)
ap.add_argument("--frames-per-stream", type=int, default=4)        # bench.py default: 32 frames per launch set
ap.add_argument("--no-autotune", action="store_true")
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
wpath = os.path.join(tempfile.gettempdir(), f"rtmodt_bench_yolov8{a.model}_{a.size}.rtw")
if not os.path.exists(wpath):
    pkg.weights.save(wpath, pkg.weights.synthetic(a.model, input_size=a.size), a.model)
B = a.streams * a.frames_per_stream
det = pkg.Detector(wpath, input_size=(a.size, a.size), batch=B, warmup=False, autotune=not a.no_autotune)
frames = list(pkg.synth.frames(B, a.size, a.size, seed=1))
det.detect_batch(frames)
rows = det.profile(a.iters)
tot = sum(ms for _, ms, _ in rows)
conv = sum(ms for _, ms, fl in rows if fl > 0)
flops = sum(fl for _, _, fl in rows)
print(f"{'launch':72s} {'us':>8s} {'TFLOP/s':>8s} {'%':>5s}")
for n, ms, fl in rows:
    print(f"{n:72s} {ms * 1e3:8.1f} {fl / (ms * 1e-3) / 1e12 if ms > 0 else 0:8.1f} {100 * ms / tot:5.1f}")
print(f"total {tot * 1e3:.1f} us; conv launches {conv * 1e3:.1f} us = {flops / (conv * 1e-3) / 1e12:.1f} TFLOP/s over {flops / 1e9:.1f} GFLOP")
