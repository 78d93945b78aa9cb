"""MI355X-native detect + track hot path (drop-in for the reference's
``src/detection/detector.py`` and ``src/tracking/tracker.py``).

The directory name is not a Python identifier; import it through the alias
module at the repo root (``import rtmodt_amd``) or with
``importlib.import_module("real-time-multi-object-detection---tracking-system_amd")``.

Sub-modules (imported lazily so that ``synth``/``weights`` work without the HIP
library being built):

* ``detection.detector`` -- ``Detector`` / ``Detections``  (reference: src/detection/detector.py:29-135)
* ``tracking.tracker``   -- ``MultiObjectTracker`` / ``Track`` (reference: src/tracking/tracker.py:27-259)
* ``_ffi``               -- ctypes binding of ``include/rtmodt.h`` (librtmodt_hip.so)
* ``weights``            -- flat fused-conv weight format, synthetic weights, BN folding
* ``synth``              -- deterministic synthetic frames / box sequences
* ``streams``            -- stream sharding across GPUs + barrier / max-time / stats reduce (RCCL or gloo)
* ``profiling``          -- ``LatencyProfiler`` (reference: src/profiling/latency_profiler.py:35-143)
* ``events``             -- ``ZoneEventEngine`` on device-resident tracks (reference: src/events/zone_engine.py:64-157)
* ``pipeline``           -- the reference's per-frame loop (tools/run_pipeline.py:121-158) around the native classes
* ``ingestion``          -- ``FrameReader`` / ``RTSPReader``: latest-frame reader thread with pluggable capture back-ends,
                            decoding into a page-locked ring (reference: src/ingestion/rtsp_reader.py:27-158)
"""
import importlib as _importlib

__all__ = ["Detector", "Detections", "MultiObjectTracker", "Track"]

_LAZY = {
    "Detector": ".detection.detector",
    "Detections": ".detection.detector",
    "MultiObjectTracker": ".tracking.tracker",
    "Track": ".tracking.tracker",
    "ZoneEventEngine": ".events.zone_engine",
    "FrameReader": ".ingestion.reader",
    "RTSPReader": ".ingestion.reader",
}


def __getattr__(name):
    if name in _LAZY:
        mod = _importlib.import_module(_LAZY[name], __name__)
        return getattr(mod, name)
    if name in ("synth", "weights", "_ffi", "detection", "tracking", "yolo_spec", "streams", "profiling", "pipeline", "events", "ingestion"):
        return _importlib.import_module("." + name, __name__)
    raise AttributeError(name)
