"""GPU: the zone event kernel (csrc/zones.hip) against the fixture written by the reference's own
zone_engine.py and against the oracle; everything through the C ABI."""
import gzip
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import tracker_oracle as T
from oracle import zone_oracle as Z
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def load_cases():
    with gzip.open(os.path.join(GOLDEN, "zones_g1.json.gz"), "rt") as f:
        return json.load(f)


def as_dict(e):
    d = {k: getattr(e, k) for k in ("event_type", "zone_name", "track_id", "class_id", "dwell_time_sec", "bbox_xyxy", "centroid", "frame_id")}
    return d


@pytest.mark.parametrize("case", ["epoch_clock", "small_clock"])
def test_zone_engine_matches_reference_fixture(pkg, tmp_path, case):
    g = load_cases()
    eng = pkg.events.ZoneEventEngine(g["zones"], log_path=str(tmp_path / "ev" / "events.jsonl"), max_tracks=64)
    c = g["cases"][case]
    total = 0
    for fr, want in zip(c["frames"], c["expect"]):
        tracks = [SimpleNamespace(track_id=i, xyxy=np.asarray(b, np.float32), class_id=k, class_name="")
                  for i, b, k in zip(fr["ids"], fr["xyxy"], fr["cls"])]
        got = [as_dict(e) for e in eng.process(tracks, fr["frame_id"], now=fr["now"])]
        ref = [{k: v for k, v in e.items() if k != "class_name"} for e in want["events"]]
        assert got == ref, f"frame {fr['frame_id']}"
        snap = eng.snapshot()
        assert snap["occupancy"] == want["occupancy"], f"frame {fr['frame_id']}"
        assert snap["cooldown"] == want["cooldown"], f"frame {fr['frame_id']}"
        total += len(got)
    with open(tmp_path / "ev" / "events.jsonl") as f:          # the alert log (zone_engine.py:153-157)
        lines = [json.loads(x) for x in f]
    assert len(lines) == total > 30 and set(lines[0]) >= {"timestamp_utc", "event_type", "zone_name", "track_id", "dwell_time_sec", "centroid"}
    eng.close()


def test_point_in_polygon_vs_oracle_via_zero_dwell_zones(pkg, tmp_path):
    """Every zone with dwell 0 / cooldown 0 fires exactly when the centroid is inside or on the edge: sweeps a
    grid of integer centroids (vertices and edge points included) over concave and degenerate polygons."""
    polys = [[[10, 10], [60, 10], [60, 60], [35, 30], [10, 60]], [[0, 0], [8, 8], [16, 0], [16, 16], [0, 16]],
             [[20, 5], [40, 5], [40, 5], [55, 25], [30, 50], [5, 25]], [[70, 70], [90, 70]], [[3, 3]]]
    zones = [{"name": f"z{i}", "polygon": p, "dwell_time_sec": 0.0, "cooldown_sec": 0.0} for i, p in enumerate(polys)]
    eng = pkg.events.ZoneEventEngine(zones, log_path=str(tmp_path / "e.jsonl"), max_tracks=4096, max_events=8192)
    pts = [(x, y) for x in range(-2, 95, 1) for y in range(-2, 75, 2)][:4000]
    tracks = [SimpleNamespace(track_id=i + 1, xyxy=np.array([x - 3, y - 2, x + 3, y + 2], np.float32), class_id=0) for i, (x, y) in enumerate(pts)]
    got = {(e.track_id, e.zone_name) for e in eng.process(tracks, 0, now=100.0)}
    want = {(i + 1, f"z{k}") for i, (x, y) in enumerate(pts) for k, p in enumerate(polys) if Z.point_polygon_test(np.array(p, np.int32), x, y) >= 0}
    assert got == want and len(want) > 1000
    eng.close()


def test_zone_engine_on_device_resident_tracker(pkg, tmp_path):
    """process_tracker: zones evaluated on the tracker's device state (tracks matched/spawned this frame) for several
    streams in one launch, against the oracle fed from the tracker oracle's state."""
    zones = [{"name": "left", "polygon": [[0, 0], [320, 0], [320, 640], [0, 640]], "dwell_time_sec": 0.5, "cooldown_sec": 2.0},
             {"name": "mid", "polygon": [[200, 200], [440, 200], [440, 440], [200, 440]], "dwell_time_sec": 0.0, "cooldown_sec": 1.0}]
    S = 3
    core = pkg.tracking.tracker._ByteTrackCore(n_streams=S, max_tracks=256, max_dets=128)
    eng = pkg.events.ZoneEventEngine(zones, log_path=str(tmp_path / "e.jsonl"), n_streams=S, max_tracks=256, max_events=512)
    oras = [(T.TrackerOracle(), Z.ZoneOracle(zones)) for _ in range(S)]
    seqs = [pkg.synth.box_sequence(40 + 10 * s, 640, 50, seed=20 + s) for s in range(S)]
    rng = np.random.default_rng(0)
    now, n_total = 50.0, 0
    for f in range(50):
        now += 0.2
        xyxy = np.zeros((S, 128, 4), np.float32); conf = np.zeros((S, 128), np.float32); cls = np.zeros((S, 128), np.int32)
        cnt = np.zeros(S, np.int32)
        per = []
        for s in range(S):
            xy, cf, cl = seqs[s]
            keep = rng.random(len(cf)) > 0.15                       # detections drop out: tracks go unmatched, return later
            b, c, k = xy[f][keep], cf[keep], cl[keep]
            xyxy[s, :len(c)], conf[s, :len(c)], cls[s, :len(c)], cnt[s] = b, c, k, len(c)
            per.append((b, c, k))
        core.update_batch(xyxy, conf, cls, cnt)
        got = eng.process_tracker(SimpleNamespace(_core=core, report="matched"), f, now=now)
        for s in range(S):
            tor, zor = oras[s]
            tor.update(*per[s])
            st = tor.snapshot()
            passed = [(int(i), st["xyxy"][j], int(st["cls"][j])) for j, i in enumerate(st["ids"]) if st["tsu"][j] == 1]
            want = zor.process(passed, f, now)
            assert [as_dict(e) for e in got[s]] == want, f"frame {f} stream {s}"
            snap = eng.snapshot(s)
            live = set(int(i) for i in st["ids"])
            ref = zor.snapshot()
            assert snap["occupancy"] == ref["occupancy"], f"frame {f} stream {s}"
            assert snap["cooldown"] == [r for r in ref["cooldown"] if r[0] in live], f"frame {f} stream {s}"   # dead ids are dropped
            n_total += len(want)
    assert n_total > 50
    core.close(); eng.close()


def test_zone_engine_limits(pkg, tmp_path):
    zones = [{"name": "all", "polygon": [[0, 0], [1000, 0], [1000, 1000], [0, 1000]], "dwell_time_sec": 0.0, "cooldown_sec": 0.0}]
    eng = pkg.events.ZoneEventEngine(zones, log_path=str(tmp_path / "e.jsonl"), max_tracks=8, max_events=4)
    mk = lambda ids: [SimpleNamespace(track_id=i, xyxy=np.array([10, 10, 20, 20], np.float32), class_id=1) for i in ids]
    assert len(eng.process(mk([5, 3]), 0, now=1.0)) == 2
    assert [e.track_id for e in eng.process(mk([9, 3, 7]), 1, now=2.0)] == [9, 3, 7]            # caller's order, not id order
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        eng.process(mk([1, 1]), 2, now=3.0)                                                    # duplicate id
    assert e.value.code == pkg._ffi.E_INVALID
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        eng.process(mk(range(10, 19)), 3, now=4.0)                                             # 9 tracks > max_tracks
    assert e.value.code == pkg._ffi.E_CAPACITY
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        eng.process(mk(range(20, 26)), 4, now=5.0)                                             # 6 events > max_events (sticky)
    assert e.value.code == pkg._ffi.E_CAPACITY
    eng.close()
    eng = pkg.events.ZoneEventEngine([], log_path=str(tmp_path / "e2.jsonl"), max_tracks=8)    # no zones: nothing to do
    assert eng.process(mk([1, 2]), 0, now=1.0) == [] and eng.get_zone_polygons() == []
    eng.close()


def test_pipeline_loop_with_event_stage(pkg, tmp_path_factory, tmp_path):
    """tools/run_pipeline.py:121-158 incl. the `events` stage: detector -> tracker (report="matched", the opt-in
    corrected filter) -> zone engine; a whole-frame zero-dwell zone must fire once per reported track."""
    wdir = tmp_path_factory.mktemp("w")
    path = str(wdir / "n320.rtw")
    pkg.weights.save(path, pkg.weights.synthetic("n", input_size=320), "n")
    det = pkg.Detector(path, input_size=(320, 320), warmup=False, autotune=False)
    trk = pkg.MultiObjectTracker("bytetrack")
    trk.report = "matched"
    zones = [{"name": "frame", "polygon": [[0, 0], [320, 0], [320, 320], [0, 320]], "dwell_time_sec": 0.0, "cooldown_sec": 1e9}]
    eng = pkg.events.ZoneEventEngine(zones, log_path=str(tmp_path / "events.jsonl"))
    prof = pkg.profiling.LatencyProfiler(gpu_sync=True, warmup_frames=2, log_interval=1000)
    frames = pkg.synth.frames(4, 320, 320, seed=3)
    out = pkg.pipeline.run(pkg.pipeline.SyntheticSource(frames), det, trk, prof, max_frames=12, event_engine=eng)
    assert out["events_mean_ms"] > 0 and out["events_p50_ms"] > 0
    reported = set()
    trk2 = pkg.MultiObjectTracker("bytetrack")
    trk2.report = "matched"
    for i in range(12):
        reported |= {t.track_id for t in trk2.update(det.detect(frames[i % 4]))}
    assert out["events"] == len(reported) > 0              # cooldown 1e9: exactly one event per track id ever reported
    det.close(); eng.close()
