"""Per-frame stage profiler -- mirror of the reference's
``src/profiling/latency_profiler.py:35-143``.

Same contract: ``tick(stage)`` / ``tock(stage)`` bracket a stage with a device
synchronisation before each timestamp when ``gpu_sync`` (reference lines 60-72; here the
sync is ``rtmodt_synchronize`` through the C ABI instead of ``torch.cuda.synchronize``),
history is kept only after ``warmup_frames`` (line 85), ``end_frame()`` returns a summary
every ``log_interval`` frames (lines 97-103) and ``summary()`` reports mean / p95 / p99 per
stage plus ``fps_mean`` / ``fps_p5`` (lines 106-120).  The arithmetic is identical to the
reference's (pinned by ``tests/golden/profiler_g1.json``, produced by running the reference
class under a scripted clock).

Two additions, both opt-in so the reference's keys and values are untouched:

* ``record(stage, ms)`` injects a duration measured elsewhere -- the engine's HIP-event
  times for the sub-stages the reference lists in ``STAGE_ORDER`` but never measures
  (``preprocess``, ``nms``: its ``inference`` stage is one opaque ``model.predict`` call);
* ``summary(p50=True)`` adds ``<stage>_p50_ms`` (BASELINE.json's metric asks for p50).
"""
from __future__ import annotations

import logging
import time
from collections import defaultdict
from typing import Optional

import numpy as np

log = logging.getLogger("rtmodt.profiler")


class LatencyProfiler:
    STAGE_ORDER = ["decode", "preprocess", "inference", "nms", "tracking", "events", "visualization", "total"]

    def __init__(self, gpu_sync: bool = True, warmup_frames: int = 50, log_interval: int = 100, *, device=0) -> None:
        self._sync = None
        if gpu_sync:
            try:
                from .. import _ffi
                import ctypes as C
                L = _ffi.lib()
                n = C.c_int(0)
                if L.rtmodt_device_count(C.byref(n)) == 0 and n.value > 0:
                    ordinal = _ffi.device_ordinal(device)
                    self._sync = lambda: L.rtmodt_synchronize(ordinal)
            except Exception:                      # no library / no GPU: behave like the reference without CUDA
                self._sync = None
        self.gpu_sync = self._sync is not None
        self.warmup = warmup_frames
        self.log_interval = log_interval
        self._starts = {}
        self._frame_times = {}
        self._history = defaultdict(list)
        self._frame_count = 0
        self._fps_t0 = time.perf_counter()
        self._fps_history = []

    # ------------------------------------------------------------------
    def tick(self, stage: str) -> None:
        if self.gpu_sync:
            self._sync()
        self._starts[stage] = time.perf_counter()

    def tock(self, stage: str) -> float:
        if self.gpu_sync:
            self._sync()
        elapsed = (time.perf_counter() - self._starts[stage]) * 1000.0
        self._frame_times[stage] = elapsed
        return elapsed

    def record(self, stage: str, elapsed_ms: float) -> None:
        """Account a duration measured on the device (HIP events) to ``stage`` of this frame."""
        self._frame_times[stage] = float(elapsed_ms)

    def end_frame(self) -> Optional[dict]:
        self._frame_count += 1
        self._frame_times["total"] = sum(self._frame_times.values())
        if self._frame_count > self.warmup:
            for k, v in self._frame_times.items():
                self._history[k].append(v)
            now = time.perf_counter()
            dt = now - self._fps_t0
            if dt > 0:
                self._fps_history.append(1.0 / dt)
            self._fps_t0 = now
        self._frame_times.clear()
        if self._frame_count > self.warmup and (self._frame_count - self.warmup) % self.log_interval == 0:
            s = self.summary()
            self._log_summary(s)
            return s
        return None

    def summary(self, p50: bool = False) -> dict:
        out = {}
        for stage in self.STAGE_ORDER:
            arr = self._history.get(stage, [])
            if arr:
                a = np.array(arr)
                out[f"{stage}_mean_ms"] = float(np.mean(a))
                if p50:
                    out[f"{stage}_p50_ms"] = float(np.percentile(a, 50))
                out[f"{stage}_p95_ms"] = float(np.percentile(a, 95))
                out[f"{stage}_p99_ms"] = float(np.percentile(a, 99))
        if self._fps_history:
            fps = np.array(self._fps_history)
            out["fps_mean"] = float(np.mean(fps))
            out["fps_p5"] = float(np.percentile(fps, 5))
        return out

    def reset(self) -> None:
        self._history.clear()
        self._fps_history.clear()
        self._frame_count = 0

    @property
    def current_fps(self) -> float:
        if len(self._fps_history) < 2:
            return 0.0
        return float(np.mean(self._fps_history[-30:]))

    def _log_summary(self, s: dict) -> None:
        parts = [f"{st}={s[st + '_mean_ms']:.1f}ms" for st in self.STAGE_ORDER if f"{st}_mean_ms" in s]
        log.info("PROFILE [FPS=%.1f] | %s", s.get("fps_mean", 0), " | ".join(parts))
