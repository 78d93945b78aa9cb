#!/usr/bin/env python3
"""The reference's per-frame loop (tools/run_pipeline.py:121-158) on synthetic frames, with the
stage table its profiler prints -- single stream, batch 1, sync per stage like the reference."""
import argparse, json, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rtmodt_amd  # noqa
pkg = sys.modules["rtmodt_amd"]
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="s"); ap.add_argument("--size", type=int, default=640)
ap.add_argument("--frames", type=int, default=1050); ap.add_argument("--source", default="640x640")
a = ap.parse_args()
w, h = (int(v) for v in a.source.split("x"))
wpath = os.path.join(tempfile.gettempdir(), f"rtmodt_bench_yolov8{a.model}_{a.size}.rtw")
if not os.path.exists(wpath):
    pkg.weights.save(wpath, pkg.weights.synthetic(a.model, input_size=a.size), a.model)
det = pkg.Detector(wpath, input_size=(a.size, a.size), classes=[0, 1, 2, 3, 5, 7], max_source_size=(max(w, a.size), max(h, a.size)))
trk = pkg.MultiObjectTracker("bytetrack", track_thresh=0.5, track_buffer=30, match_thresh=0.8, mot20=False)
trk.report = "matched"                                  # feed the zone engine the tracks matched / spawned this frame
zones = [{"name": "restricted_area_1", "polygon": [[100, 200], [400, 200], [400, 600], [100, 600]], "trigger": "intrusion",
          "dwell_time_sec": 2.0, "cooldown_sec": 10.0},
         {"name": "exit_gate", "polygon": [[800, 400], [1200, 400], [1200, 700], [800, 700]], "trigger": "crossing", "cooldown_sec": 5.0}]
eng = pkg.events.ZoneEventEngine(zones, log_path=os.path.join(tempfile.gettempdir(), "rtmodt_events.jsonl"))   # config/default.yaml:67-77
prof = pkg.profiling.LatencyProfiler(gpu_sync=True, warmup_frames=50, log_interval=100)   # config/default.yaml:86-90
out = pkg.pipeline.run(pkg.pipeline.SyntheticSource(pkg.synth.frames(16, h, w, seed=1234)), det, trk, prof, max_frames=a.frames, event_engine=eng)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}))
