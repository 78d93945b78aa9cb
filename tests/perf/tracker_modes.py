#!/usr/bin/env python3
"""(Lives under tests/ because it times the CPU oracle next to the kernels; only tests, smoke() and the bench's CPU baseline may touch oracle/.)
Device time per frame of the single-launch tracker in its two assignment modes and of the zone kernel, on the
BASELINE config-3 / config-5 sequences (200 boxes at 640, 500 boxes at 1280), beside the CPU oracle (one thread).
HIP-event-free: wall clock around synchronous C-ABI calls, so PCIe round trips of the host-array path are included;
the device-resident path (update_from_detector) hides all of it behind the forward pass."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import rtmodt_amd  # noqa
from types import SimpleNamespace
pkg = sys.modules["rtmodt_amd"]
from oracle import tracker_oracle as T  # noqa: E402  (CPU baseline only)

out = {}
for name, (n, size) in {"200 boxes @640": (200, 640), "500 boxes @1280": (500, 1280)}.items():
    xy, cf, cl = pkg.synth.box_sequence(n, size, 120, seed=1234)
    row = {}
    for mode, am in (("greedy", pkg._ffi.ASSIGN_GREEDY), ("lapjv", pkg._ffi.ASSIGN_LAPJV), ("greedy_kalman", pkg._ffi.ASSIGN_GREEDY)):
        core = pkg.tracking.tracker._ByteTrackCore(assign_mode=am, kalman=mode.endswith("kalman"))
        ts = []
        for rep in range(3):
            if rep and core.kalman:
                break                                             # (reset keeps the filter arrays: one pass is enough for the timing)
            core.reset()
            for f in range(120):
                t0 = time.perf_counter()
                core.update(xy[f], cf, cl)
                ts.append(time.perf_counter() - t0)
        row[f"gpu_{mode}_p50_us"] = round(float(np.median(ts)) * 1e6, 1)
        row[f"{mode}_live_tracks"] = len(core.tracks(0))
        if mode == "greedy":
            zones = [{"name": "left", "polygon": [[0, 0], [size // 2, 0], [size // 2, size], [0, size]], "dwell_time_sec": 0.5, "cooldown_sec": 2.0},
                     {"name": "mid", "polygon": [[size // 4, size // 4], [3 * size // 4, size // 4], [3 * size // 4, 3 * size // 4], [size // 4, 3 * size // 4]]}]
            eng = pkg.events.ZoneEventEngine(zones, log_path="/tmp/rtmodt_modes_events.jsonl", max_tracks=2048)
            tz, ne = [], 0
            for f in range(120):
                core.update(xy[f], cf, cl)
                t0 = time.perf_counter()
                ne += sum(len(e) for e in eng.process_tracker(SimpleNamespace(_core=core, report="matched"), f, now=100.0 + 0.04 * f))
                tz.append(time.perf_counter() - t0)
            row["gpu_zones_p50_us"] = round(float(np.median(tz)) * 1e6, 1)
            row["zone_events"] = ne
            eng.close()
        core.close()
    for mode in ("greedy", "lapjv"):
        o = T.TrackerOracle(assign=mode)
        ts = []
        for f in range(120):
            t0 = time.perf_counter()
            o.update(xy[f], cf, cl)
            ts.append(time.perf_counter() - t0)
        row[f"cpu_oracle_{mode}_p50_us"] = round(float(np.median(ts)) * 1e6, 1)
    out[name] = row
print(json.dumps(out))
