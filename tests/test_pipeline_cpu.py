"""pipeline.run's hand-over rules on the host (no GPU: scripted detector / tracker / event engines).
Reference loop: tools/run_pipeline.py:121-158."""
import numpy as np


class _Det:
    model = type("M", (), {"names": {0: "person"}})()

    def detect(self, frame):
        return type("D", (), {"xyxy": np.zeros((1, 4), np.float32), "confidence": np.ones(1, np.float32), "class_id": np.zeros(1, np.int32),
                              "__len__": lambda self: 1})()


class _Trk:
    def __init__(self):
        self.calls = []

    def update_from_detector(self, det, materialize=True):
        self.calls.append(materialize)
        return [("track", len(self.calls))] if materialize else []

    def update(self, detections):
        self.calls.append("host")
        return [("track", len(self.calls))]


class _HostEvents:                       # the reference's API only (src/events/zone_engine.py:82)
    def __init__(self):
        self.seen = []

    def process(self, tracks, fid):
        self.seen.append(list(tracks))
        return [("event", fid)] if tracks else []


class _DeviceEvents(_HostEvents):
    def process_tracker(self, tracker, fid, class_names=None):
        self.seen.append("device")
        return [("event", fid)], None


def _run(pkg, events, handoff=True):
    frames = np.zeros((2, 8, 8, 3), np.uint8)
    trk = _Trk()
    prof = pkg.profiling.LatencyProfiler(gpu_sync=False, warmup_frames=0, log_interval=1000)
    out = pkg.pipeline.run(pkg.pipeline.SyntheticSource(frames), _Det(), trk, prof, max_frames=4, device_stages=False,
                           event_engine=events, device_handoff=handoff)
    return out, trk


def test_host_only_event_engine_gets_the_materialised_tracks(pkg):
    """with the device hand-off on and an event engine that only has `process(tracks, fid)`, the loop used to
    pass materialize=False and feed the engine an empty list -- zero events, silently."""
    ev = _HostEvents()
    out, trk = _run(pkg, ev)
    assert trk.calls == [True] * 4                       # materialised: the host engine needs the list
    assert all(len(t) == 1 for t in ev.seen) and out["events"] == 4


def test_device_event_engine_keeps_tracks_on_the_device(pkg):
    ev = _DeviceEvents()
    out, trk = _run(pkg, ev)
    assert trk.calls == [False] * 4 and ev.seen == ["device"] * 4 and out["events"] == 4


def test_no_event_engine_and_no_handoff(pkg):
    out, trk = _run(pkg, None)
    assert trk.calls == [True] * 4 and out["events"] == 0
    ev = _DeviceEvents()
    out, trk = _run(pkg, ev, handoff=False)
    assert trk.calls == ["host"] * 4 and all(isinstance(t, list) for t in ev.seen) and out["events"] == 4
