"""CPU: the LatencyProfiler mirror computes exactly what the reference's class computes
(fixture produced by running the reference under a scripted clock, oracle/gen_golden_profiler.py)."""
import json
import os
from importlib import import_module

from conftest import GOLDEN


def test_profiler_matches_reference_under_scripted_clock(pkg):
    from oracle.gen_golden_profiler import scripted_run
    z = json.load(open(os.path.join(GOLDEN, "profiler_g1.json")))
    mod = import_module(pkg.__name__ + ".profiling.latency_profiler")
    saved = mod.time
    try:
        got = scripted_run(mod.LatencyProfiler, mod, z["seed"], z["frames"], z["warmup"], z["log_interval"])
    finally:
        mod.time = saved
    ref = z["result"]
    assert [g["frame"] for g in got["logged"]] == [r["frame"] for r in ref["logged"]] == [150, 250]
    for g, r in zip(got["logged"], ref["logged"]):
        assert g["summary"] == r["summary"]                    # bit-identical floats, same keys
    assert got["final"] == ref["final"]
    assert got["current_fps"] == ref["current_fps"]
    assert "preprocess_mean_ms" not in ref["final"] and "total_p99_ms" in ref["final"]


def test_profiler_extensions_do_not_change_reference_keys(pkg):
    P = pkg.profiling.LatencyProfiler
    assert P.STAGE_ORDER == ["decode", "preprocess", "inference", "nms", "tracking", "events", "visualization", "total"]
    p = P(gpu_sync=False, warmup_frames=0, log_interval=2)
    for f in range(4):
        p.tick("inference"); p.tock("inference")
        p.record("preprocess", 0.25); p.record("nms", 0.5)
        s = p.end_frame()
        assert (s is not None) == (f % 2 == 1)
    s = p.summary(p50=True)
    assert s["preprocess_mean_ms"] == 0.25 and s["nms_p50_ms"] == 0.5 and "total_p50_ms" in s
    assert "nms_p50_ms" not in p.summary()
    p.reset()
    assert p.summary() == {} and p.current_fps == 0.0
