"""The hot loop of the reference's ``tools/run_pipeline.py:121-158`` around the native
detector and tracker: read frame -> ``detector.detect`` ->
``tracker.update`` -> ``profiler.end_frame``, every stage bracketed by the sync-ing
profiler exactly as the reference does.  RTSP ingestion, the zone engine and the renderer
are out of scope; their stages are simply absent from the table.

On top of the reference's three wall-clock stages (``decode``, ``inference``, ``tracking``)
the loop records what the engine measured with HIP events inside ``inference``:
``preprocess`` (letterbox), ``nms`` and the forward pass itself -- the sub-stages the
reference's ``STAGE_ORDER`` names but its loop never ticks.
"""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np

from .profiling import LatencyProfiler


class SyntheticSource:
    """Stand-in for ``RTSPReader.read()`` (src/ingestion/rtsp_reader.py:74-79): returns
    ``(ok, frame, frame_id)`` from a pre-generated ring of frames (copied, like the reader)."""

    def __init__(self, frames: np.ndarray):
        self._frames = frames
        self._i = 0

    def read(self):
        f = self._frames[self._i % len(self._frames)].copy()
        self._i += 1
        return True, f, self._i


class PinnedFrameRing:
    """The frame ring between a decoder and the detector: ``slots`` page-locked H x W x 3
    uint8 images.  ``write(i, frame)`` is what a capture thread does with a decoded frame; ``frame(i)`` is the
    view handed to ``Detector.enqueue`` -- its upload is then a true asynchronous DMA on the engine's copy stream."""

    def __init__(self, slots: int, height: int, width: int, device=0):
        from . import _ffi
        self._mem = _ffi.PinnedArray((slots, height, width, 3), np.uint8, _ffi.device_ordinal(device))
        self.slots = slots

    def write(self, i: int, frame: np.ndarray) -> np.ndarray:
        dst = self._mem.array[i % self.slots]
        np.copyto(dst, frame)
        return dst

    def frame(self, i: int) -> np.ndarray:
        return self._mem.array[i % self.slots]

    def close(self) -> None:
        self._mem.free()


def run(source, detector, tracker, profiler: Optional[LatencyProfiler] = None, max_frames: int = 200,
        device_stages: bool = True, event_engine=None, device_handoff: bool = True) -> dict:
    """Runs ``max_frames`` iterations of the reference loop; returns ``profiler.summary(p50=True)``
    plus the last frame's detections and tracks.

    ``device_handoff`` (default): the tracker consumes the detector's detections where they are, on the device, and the
    zone engine the tracker's device-resident state (``tracker.update_from_detector`` / ``event_engine.process_tracker``);
    the host receives the detections (as the reference's ``Detector._parse`` does) and the events, never the track arrays.
    ``False``: the reference's literal data flow -- ``tracker.update(detections)`` on host arrays, ``process(tracks)``."""
    profiler = profiler or LatencyProfiler(gpu_sync=True, warmup_frames=50, log_interval=100)
    detections = tracks = None
    n_events = 0
    for _ in range(max_frames):
        profiler.tick("decode")
        ok, frame, fid = source.read()
        profiler.tock("decode")
        if not ok or frame is None:
            continue
        profiler.tick("inference")
        detections = detector.detect(frame)
        total_inf = profiler.tock("inference")
        if device_stages and hasattr(detector, "stage_times"):
            pre, fwd, nms = detector.stage_times()
            # split the wall-clock "inference" stage the way the reference's STAGE_ORDER intends:
            # preprocess + inference (forward+decode, host overhead included) + nms == the bracketed time
            profiler.record("preprocess", pre)
            profiler.record("nms", nms)
            profiler.record("inference", max(total_inf - pre - nms, 0.0))
        handoff = device_handoff and hasattr(tracker, "update_from_detector") and hasattr(detector, "model")
        # the track list stays on the device only when the event stage can read it there; an engine with the reference's
        # host API alone (`process(tracks, fid)`) must be handed the materialised list, trails included
        events_on_device = handoff and event_engine is not None and hasattr(event_engine, "process_tracker")
        profiler.tick("tracking")
        tracks = tracker.update_from_detector(detector, materialize=not events_on_device) if handoff else tracker.update(detections)
        profiler.tock("tracking")
        if event_engine is not None:                       # tools/run_pipeline.py:141-146
            profiler.tick("events")
            if events_on_device:
                n_events += len(event_engine.process_tracker(tracker, fid, class_names=getattr(detector.model, "names", None))[0])
            else:
                n_events += len(event_engine.process(tracks, fid))
            profiler.tock("events")
        profiler.end_frame()
    out = profiler.summary(p50=True)
    out["frames"] = max_frames
    out["last_detections"] = 0 if detections is None else len(detections)
    if device_handoff and hasattr(tracker, "_core") and getattr(tracker, "report", "") == "matched":
        out["last_tracks"] = int((tracker._core.snapshot(0)["tsu"] == 1).sum())      # read back once, after the loop
    else:
        out["last_tracks"] = 0 if tracks is None else len(tracks)
    out["events"] = n_events
    return out
