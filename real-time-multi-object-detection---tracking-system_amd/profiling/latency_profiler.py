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
from typing import Dict, Optional

import numpy as np

log = logging.getLogger("rtmodt.profiler")


class _Series:
    """Append-only float64 samples in a doubling buffer (one per stage, one for the frame periods)."""

    __slots__ = ("buf", "n")

    def __init__(self) -> None:
        self.buf = np.empty(256, np.float64)
        self.n = 0

    def push(self, v: float) -> None:
        if self.n == len(self.buf):
            self.buf = np.concatenate([self.buf, np.empty(len(self.buf), np.float64)])
        self.buf[self.n] = v
        self.n += 1

    def values(self) -> np.ndarray:
        return self.buf[:self.n]

    def clear(self) -> None:
        self.n = 0


class LatencyProfiler:
    STAGE_ORDER = ["decode", "preprocess", "inference", "nms", "tracking", "events", "visualization", "total"]

    def __init__(self, gpu_sync: bool = True, warmup_frames: int = 50, log_interval: int = 100, *, device=0) -> None:
        self._device_sync = self._find_device_sync(device) if gpu_sync else None
        self.gpu_sync = self._device_sync is not None
        self.warmup = warmup_frames
        self.log_interval = log_interval
        self._opened: Dict[str, float] = {}            # stage -> clock at tick()
        self._frame_ms: Dict[str, float] = {}          # this frame's stages, in the order they were first timed
        self._samples: Dict[str, _Series] = {}         # stage -> durations of the frames after the warm-up
        self._periods = _Series()                      # seconds between consecutive end_frame() calls after the warm-up
        self._frames_seen = 0
        self._last_frame_end = time.perf_counter()

    @staticmethod
    def _find_device_sync(device):
        """``rtmodt_synchronize`` of the device through the C ABI, or None without the library / a GPU (the reference
        behaves the same way without CUDA: latency_profiler.py:43)."""
        try:
            import ctypes as C

            from .. import _ffi
            L = _ffi.lib()
            n = C.c_int(0)
            if L.rtmodt_device_count(C.byref(n)) != 0 or n.value <= 0:
                return None
            ordinal = _ffi.device_ordinal(device)
            return lambda: L.rtmodt_synchronize(ordinal)
        except Exception:
            return None

    # ---- stage brackets (reference lines 60-72) ----
    def tick(self, stage: str) -> None:
        if self._device_sync is not None:
            self._device_sync()
        self._opened[stage] = time.perf_counter()

    def tock(self, stage: str) -> float:
        if self._device_sync is not None:
            self._device_sync()
        ms = (time.perf_counter() - self._opened[stage]) * 1000.0
        self._frame_ms[stage] = ms
        return ms

    def record(self, stage: str, elapsed_ms: float) -> None:
        """Account a duration measured on the device (HIP events) to ``stage`` of this frame."""
        self._frame_ms[stage] = float(elapsed_ms)

    # ---- frame boundary (reference lines 80-103) ----
    def end_frame(self) -> Optional[dict]:
        self._frames_seen += 1
        self._frame_ms["total"] = sum(self._frame_ms.values())
        counted = self._frames_seen > self.warmup
        if counted:
            for stage, ms in self._frame_ms.items():
                series = self._samples.get(stage)
                if series is None:
                    series = self._samples[stage] = _Series()
                series.push(ms)
            now = time.perf_counter()
            period = now - self._last_frame_end
            if period > 0:
                self._periods.push(period)
            self._last_frame_end = now
        self._frame_ms.clear()
        if counted and (self._frames_seen - self.warmup) % self.log_interval == 0:
            report = self.summary()
            self._log_summary(report)
            return report
        return None

    # ---- statistics (reference lines 106-120) ----
    def _rates(self) -> np.ndarray:
        return 1.0 / self._periods.values()

    def summary(self, p50: bool = False) -> dict:
        out = {}
        for stage in self.STAGE_ORDER:
            series = self._samples.get(stage)
            if series is None or series.n == 0:
                continue
            a = series.values()
            out[f"{stage}_mean_ms"] = float(np.mean(a))
            if p50:
                out[f"{stage}_p50_ms"] = float(np.percentile(a, 50))
            out[f"{stage}_p95_ms"] = float(np.percentile(a, 95))
            out[f"{stage}_p99_ms"] = float(np.percentile(a, 99))
        if self._periods.n:
            fps = self._rates()
            out["fps_mean"] = float(np.mean(fps))
            out["fps_p5"] = float(np.percentile(fps, 5))
        return out

    def reset(self) -> None:
        self._samples.clear()
        self._periods.clear()
        self._frames_seen = 0

    @property
    def current_fps(self) -> float:
        if self._periods.n < 2:
            return 0.0
        return float(np.mean(self._rates()[-30:]))

    def _log_summary(self, s: dict) -> None:
        parts = [f"{st}={s[st + '_mean_ms']:.1f}ms" for st in self.STAGE_ORDER if f"{st}_mean_ms" in s]
        log.info("PROFILE [FPS=%.1f] | %s", s.get("fps_mean", 0), " | ".join(parts))
