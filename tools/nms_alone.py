"""nms_kernel and tracker_update ALONE on the device at the benchmarked shape (YOLOv8s 640, 32 images per batch, bench weights and frames):
every batch is fetched before the next one is enqueued, so nothing else runs beside the post-processing kernels.  Run under
`rocprofv3 --kernel-trace --stats` (tools/ab/r03_nms_phases.sh), with RTMODT_NMS_STOP=1..5 to cut the kernel short after each phase.
Prints the candidate counts the kernel sees (score > conf) per image."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa: E402,F401

pkg = sys.modules["rtmodt_amd"]
from importlib import import_module  # noqa: E402

S, F, size, steps = 8, 4, 640, int(os.environ.get("STEPS", "12"))
B = S * F
path = os.path.join(tempfile.gettempdir(), "nms_alone_yolov8s_640.rtw")
if not os.path.exists(path):
    pkg.weights.save(path, pkg.weights.synthetic("s", input_size=size), "s")
det = pkg.Detector(path, input_size=(size, size), warmup=False, batch=B, autotune=False, chains=1, max_det=100)
core = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore(n_streams=S, max_dets=128, max_tracks=2048)
gen = pkg.synth.structured_frames if os.environ.get("KIND") == "structured" else pkg.synth.frames
frames = np.stack([gen(F * steps, size, size, seed=1234 + s) for s in range(S)], 1).reshape(steps, F, S, size, size, 3)
buf = pkg._ffi.DeviceBuffer(frames.nbytes)
buf.upload(frames)
per = size * size * 3
for t in range(steps):
    det.enqueue([buf.ptr + ((t * F + f) * S + s) * per for f in range(F) for s in range(S)], height=size, width=size)
    core.update_from_detector(det, 0, S, frames_per_stream=F)
    out = det.fetch()
    det.synchronize()
if not os.environ.get("RTMODT_NMS_STOP"):
    cands = []
    for i in range(B):
        _, _, pred = det.debug_fetch(i, want_input=False, want_heads=False)
        cands.append(int((pred[4:].max(0) > det.confidence).sum()))
    print("candidates per image: min %d  median %d  max %d;  detections per image: %.1f;  live tracks per stream: %.0f" % (
        min(cands), int(np.median(cands)), max(cands), np.mean([len(d) for d in out]), np.mean([len(core.snapshot(s)["ids"]) for s in range(S)])))
