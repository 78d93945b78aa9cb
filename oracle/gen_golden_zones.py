"""Generate tests/golden/zones_g1.json.gz by RUNNING THE REFERENCE zone event engine.

Runs only in the build container (needs /root/reference).  ``src/events/zone_engine.py`` is loaded
by path.  Two of its module-level imports are absent here and get in-memory stand-ins:

* ``loguru``  -- a no-op ``logger`` (used for two log lines, ``zone_engine.py:78,155``);
* ``cv2``     -- ONE function, ``pointPolygonTest`` (``zone_engine.py:95``), bound to
  ``oracle.zone_oracle.point_polygon_test``.  That makes the point-in-polygon arithmetic the
  restatement's, not OpenCV's: the fixture pins the engine's occupancy / dwell / cooldown / purge
  logic and its event records, and says nothing about ``cv::pointPolygonTest`` (PARITY UNPINNED).

``time.time`` is scripted for the duration of each ``process`` call (the reference reads it once,
``zone_engine.py:84``).  The fixture holds DATA only: zone configs, the per-frame clock and track
lists fed in, the events and the two private ledgers that came out.

    python oracle/gen_golden_zones.py            # rewrites tests/golden/zones_g1.json.gz
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import tempfile
import types
from dataclasses import asdict, dataclass

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle.zone_oracle import point_polygon_test  # noqa: E402

REF = "/root/reference/src/events/zone_engine.py"
OUT = os.path.join(ROOT, "tests", "golden", "zones_g1.json.gz")


def load_reference():
    if "loguru" not in sys.modules:
        stub = types.ModuleType("loguru")

        class _L:
            def __getattr__(self, _):
                return lambda *a, **k: None

        stub.logger = _L()
        sys.modules["loguru"] = stub
    if "cv2" not in sys.modules:
        cv = types.ModuleType("cv2")
        cv.pointPolygonTest = lambda contour, pt, measure: float(point_polygon_test(contour, pt[0], pt[1]))
        sys.modules["cv2"] = cv
    spec = importlib.util.spec_from_file_location("ref_zone_engine", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_zone_engine"] = mod
    spec.loader.exec_module(mod)
    return mod


@dataclass
class T:                           # what zone_engine.py reads from a track (:89-92, :109-112)
    track_id: int
    xyxy: np.ndarray
    class_id: int
    class_name: str = ""


ZONES = [
    {"name": "restricted_area_1", "polygon": [[100, 200], [400, 200], [400, 600], [100, 600]], "trigger": "intrusion",
     "dwell_time_sec": 2.0, "cooldown_sec": 10.0},                                  # config/default.yaml:68-72
    {"name": "exit_gate", "polygon": [[800, 400], [1200, 400], [1200, 700], [800, 700]], "trigger": "crossing",
     "direction": "left_to_right", "cooldown_sec": 5.0},                           # config/default.yaml:73-77 (dwell defaults to 2.0)
    {"name": "notch", "polygon": [[450, 100], [750, 100], [750, 400], [600, 250], [450, 400]], "trigger": "intrusion",
     "dwell_time_sec": 0.0, "cooldown_sec": 0.7},                                   # concave, fires at once, short cooldown
    {"name": "exit_gate", "polygon": [[900, 350], [1300, 350], [1300, 650]], "trigger": "intrusion",
     "dwell_time_sec": 1.0, "cooldown_sec": 3.0},                                   # SAME NAME as zone 1: shares its ledger keys
]


def scenario(seed: int, n_tracks: int, n_frames: int, t0: float):
    """Boxes drifting over a 1400 x 800 field; every track is withheld for a few frames now and then
    (the purge at zone_engine.py:126-128) and the clock ticks unevenly."""
    rng = np.random.default_rng(seed)
    c = np.stack([rng.uniform(50, 1350, n_tracks), rng.uniform(50, 750, n_tracks)], 1)
    v = rng.uniform(-9, 9, size=(n_tracks, 2))
    wh = rng.uniform(30, 90, size=(n_tracks, 2))
    cls = rng.integers(0, 8, n_tracks)
    hidden_until = np.zeros(n_tracks, int)
    now = t0
    frames = []
    for f in range(n_frames):
        now += float(rng.choice([0.04, 0.1, 0.25, 0.5, 1.0]))
        c += v + rng.normal(0, 1.5, size=c.shape)
        for a, hi in ((0, 1400), (1, 800)):
            out = (c[:, a] < 0) | (c[:, a] > hi)
            v[out, a] *= -1
            c[:, a] = np.clip(c[:, a], 0, hi)
        hide = rng.random(n_tracks) < 0.04
        hidden_until[hide] = f + rng.integers(1, 5, int(hide.sum()))
        vis = np.nonzero(hidden_until <= f)[0]
        vis = vis[rng.permutation(len(vis))] if f % 7 == 3 else vis              # list order is the caller's business
        xyxy = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
        frames.append({"frame_id": f, "now": now, "ids": [int(i) + 1 for i in vis],
                       "xyxy": [[float(x) for x in xyxy[i]] for i in vis], "cls": [int(cls[i]) for i in vis]})
    return frames


def run(ref, frames, tmp):
    import time as _time
    eng = ref.ZoneEventEngine(ZONES, log_path=os.path.join(tmp, "events.jsonl"))
    real = _time.time
    out = []
    try:
        for fr in frames:
            _time.time = lambda now=fr["now"]: now
            tracks = [T(i, np.asarray(b, np.float32), k) for i, b, k in zip(fr["ids"], fr["xyxy"], fr["cls"])]
            evs = eng.process(tracks, fr["frame_id"])
            recs = []
            for e in evs:
                d = asdict(e)
                d.pop("timestamp_utc")                      # wall-clock string of the generating run
                d.pop("metadata")
                recs.append(d)
            occ = sorted([int(t), str(n), float(x)] for t, dd in eng._occupancy.items() for n, x in dd.items())
            cd = sorted([int(t), str(n), float(x)] for (t, n), x in eng._cooldown.items())
            out.append({"events": recs, "occupancy": occ, "cooldown": cd})
    finally:
        _time.time = real
    with open(os.path.join(tmp, "events.jsonl")) as f:
        n_lines = sum(1 for _ in f)
    return out, n_lines


def main():
    ref = load_reference()
    cases = {}
    for name, (seed, n_tracks, n_frames, t0) in {"epoch_clock": (1, 30, 120, 1.7e9), "small_clock": (2, 20, 90, 3.0)}.items():
        frames = scenario(seed, n_tracks, n_frames, t0)
        with tempfile.TemporaryDirectory() as tmp:
            expect, n_lines = run(ref, frames, tmp)
        assert n_lines == sum(len(e["events"]) for e in expect)
        cases[name] = {"frames": frames, "expect": expect}
        print(name, "frames", n_frames, "events", n_lines, "by zone",
              {z: sum(1 for e in expect for r in e["events"] if r["zone_name"] == z) for z in sorted({z["name"] for z in ZONES})})
    import gzip
    with gzip.GzipFile(OUT, "wb", mtime=0) as f:
        f.write(json.dumps({"zones": ZONES, "cases": cases}, separators=(",", ":")).encode())
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
