"""Generate tests/golden/profiler_g1.json by RUNNING THE REFERENCE LatencyProfiler
(/root/reference/src/profiling/latency_profiler.py) under a scripted clock.

Build-container only.  ``loguru`` (absent) is replaced by an in-memory no-op; the clock the
reference reads (``time.perf_counter`` inside its module) is replaced by a deterministic
sequence, so the statistics it returns are pure functions of the scripted durations.
The fixture holds the script parameters and the summaries -- data only.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src/profiling/latency_profiler.py"


class ScriptedClock:
    """perf_counter() stand-in: advances by the next scripted increment on every call."""

    def __init__(self, seed, n):
        rng = np.random.default_rng(seed)
        self.inc = rng.uniform(0.0002, 0.004, size=n)          # seconds between consecutive clock reads
        self.inc[::37] += 0.02                                 # occasional slow frame
        self.t = 100.0
        self.i = 0

    def __call__(self):
        self.t += float(self.inc[self.i % len(self.inc)])
        self.i += 1
        return self.t


def scripted_run(cls, module, seed=7, frames=260, warmup=50, log_interval=100):
    clock = ScriptedClock(seed, 4096)
    module.time = types.SimpleNamespace(perf_counter=clock)
    prof = cls(gpu_sync=False, warmup_frames=warmup, log_interval=log_interval)
    logged = []
    for f in range(frames):
        for stage in ("decode", "inference", "tracking", "events", "visualization"):
            if stage == "events" and f % 3 == 0:
                continue                                       # a stage may be skipped in a frame
            prof.tick(stage)
            prof.tock(stage)
        s = prof.end_frame()
        if s is not None:
            logged.append({"frame": f + 1, "summary": s})
    return {"logged": logged, "final": prof.summary(), "current_fps": prof.current_fps}


def main():
    stub = types.ModuleType("loguru")
    stub.logger = types.SimpleNamespace(info=lambda *a, **k: None)
    sys.modules.setdefault("loguru", stub)
    spec = importlib.util.spec_from_file_location("ref_profiler", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {"seed": 7, "frames": 260, "warmup": 50, "log_interval": 100, "result": scripted_run(mod.LatencyProfiler, mod)}
    with open(os.path.join(ROOT, "tests", "golden", "profiler_g1.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(len(out["result"]["logged"]), "logged summaries;", len(out["result"]["final"]), "keys")


if __name__ == "__main__":
    main()
