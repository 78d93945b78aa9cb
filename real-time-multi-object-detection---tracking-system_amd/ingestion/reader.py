"""Frame ingest in front of the hot path.

Mirrors the interface of the reference's ``RTSPReader`` (src/ingestion/rtsp_reader.py:27-158: constructor arguments,
``start() / read() -> (ok, frame, frame_id) / stop() / is_alive``, context manager, latest-frame-only hand-over, reconnects
with a capped linear back-off ``reconnect_delay * min(attempt, 5)``, give up after ``max_reconnects`` failures in a row)
so that ``pipeline.run`` takes either.  What is different, because the step after it is different:

* the reference decodes through ``cv2.VideoCapture`` (FFmpeg or a GStreamer string with NVIDIA's decoder element); neither
  OpenCV nor a video decoder is part of this build, so the capture device is a small protocol (``opened / grab() /
  retrieve() / release()``) with registered back-ends: ``"raw"`` -- BGR24 frames from a file, FIFO or pipe (what
  ``ffmpeg -i rtsp://... -f rawvideo -pix_fmt bgr24 -`` writes), ``"synthetic"`` -- a generated ring; ``register_backend``
  adds others (a ``cv2.VideoCapture`` wrapper is four lines and listed in INTEGRATION.md);
* decoded frames land directly in a PAGE-LOCKED ring (``pipeline.PinnedFrameRing``) when one is given: the detector's upload
  of such a frame is one asynchronous DMA, and ``read(copy=False)`` hands out the slot itself instead of a copy -- as a
  LEASE: the capture thread never writes into a leased slot, and the consumer gives it back with ``release(frame_id)`` once
  the batch the frame went into has been fetched (``Detector.fetch``; the detector keeps up to three batches in flight and
  may read page-locked frames in place).  The ring needs at least ``buffer_size + 2`` slots (the latest frame, one being
  written, ``buffer_size`` leased); size it by the consumer's in-flight depth: ``slots = frames in flight + 2``.
"""
from __future__ import annotations

import logging
import threading
import time
from typing import Callable, Dict, Optional, Tuple

import numpy as np

log = logging.getLogger("rtmodt.ingestion")


class SyntheticCapture:
    """Capture back-end that replays a pre-generated ring of frames at ``fps`` (0: as fast as asked)."""

    def __init__(self, source, resolution: Optional[Tuple[int, int]] = None, fps: float = 0.0, frames: Optional[np.ndarray] = None):
        from .. import synth
        w, h = resolution or (640, 640)
        self._frames = frames if frames is not None else synth.frames(8, h, w, seed=1234)
        self._period = 1.0 / fps if fps > 0 else 0.0
        self._next = time.perf_counter()
        self._i = 0
        self.opened = True

    def grab(self) -> bool:
        if not self.opened:
            return False
        if self._period:
            delay = self._next - time.perf_counter()
            if delay > 0:
                time.sleep(delay)
            self._next = max(self._next + self._period, time.perf_counter())
        self._i += 1
        return True

    def retrieve(self, out: Optional[np.ndarray] = None):
        src = self._frames[(self._i - 1) % len(self._frames)]
        if out is None:
            return True, src.copy()
        np.copyto(out, src)
        return True, out

    def release(self) -> None:
        self.opened = False


class RawVideoCapture:
    """BGR24 frames of a fixed ``resolution`` (width, height) read back to back from a file, FIFO or ``-`` (stdin)."""

    def __init__(self, source: str, resolution: Optional[Tuple[int, int]] = None, **_):
        if not resolution:
            raise ValueError("the raw back-end needs resolution=(width, height)")
        self._w, self._h = int(resolution[0]), int(resolution[1])
        self._nbytes = self._w * self._h * 3
        import sys
        try:
            self._f = sys.stdin.buffer if source == "-" else open(source, "rb", buffering=0)
        except OSError as e:
            raise ConnectionError(f"Cannot open stream: {source}") from e
        self._owns = source != "-"
        self._buf = bytearray(self._nbytes)
        self.opened = True

    def grab(self) -> bool:
        if not self.opened:
            return False
        view, got = memoryview(self._buf), 0
        while got < self._nbytes:
            n = self._f.readinto(view[got:])
            if not n:                                  # end of file / writer gone: the stream is down
                return False
            got += n
        return True

    def retrieve(self, out: Optional[np.ndarray] = None):
        src = np.frombuffer(self._buf, np.uint8).reshape(self._h, self._w, 3)
        if out is None:
            return True, src.copy()
        np.copyto(out, src)
        return True, out

    def release(self) -> None:
        if self.opened and self._owns:
            self._f.close()
        self.opened = False


_BACKENDS: Dict[str, Callable] = {"synthetic": SyntheticCapture, "raw": RawVideoCapture}


def register_backend(name: str, factory: Callable) -> None:
    """``factory(source, resolution=..., **kw)`` returns an object with ``opened``, ``grab()``, ``retrieve(out=None)``, ``release()``;
    it raises ``ConnectionError`` when the source cannot be opened."""
    _BACKENDS[name.lower()] = factory


class FrameReader:
    """Thread-safe, latest-frame-only frame provider (same constructor and methods as the reference's ``RTSPReader``)."""

    def __init__(self, source: str, backend: str = "raw", buffer_size: int = 1, target_fps: int = 30, reconnect_delay: float = 3.0,
                 max_reconnects: int = 10, resolution: Optional[Tuple[int, int]] = None, ring=None, **backend_kw) -> None:
        self.source = source
        self.backend = backend.lower()
        self.buffer_size = max(1, int(buffer_size))
        self.target_fps = target_fps
        self.reconnect_delay = reconnect_delay
        self.max_reconnects = max_reconnects
        self.resolution = resolution
        if self.backend not in _BACKENDS:
            raise ValueError(f"unknown ingest backend {backend!r}; registered: {sorted(_BACKENDS)}")
        self._backend_kw = backend_kw
        self._ring = ring                              # pipeline.PinnedFrameRing or None
        if ring is not None and ring.slots < self.buffer_size + 2:
            raise ValueError(f"a ring of {ring.slots} slots cannot hold buffer_size={self.buffer_size} leased frames plus the latest "
                             f"frame and the one being written: build it with >= {self.buffer_size + 2} slots")
        self._leases: Dict[int, int] = {}              # ring slot -> outstanding leases
        self.lease_misses = 0                          # read(copy=False) calls served by a copy because no slot could be leased
        self._lease_slot: Dict[int, list] = {}         # frame id -> [ring slot (-1: only copy-served holders so far), holders]
        self._latest_slot = -1
        self._cap = None
        self._latest: Optional[np.ndarray] = None
        self._slot = 0
        self._frame_id = 0
        self._lock = threading.Lock()
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.reconnects = 0                            # total re-opens so far
        self.dropped = 0                               # frames overwritten before anybody read them
        self._served_id = 0

    # ---- public API (reference: rtsp_reader.py:66-96) ----
    def start(self) -> "FrameReader":
        self._open()
        self._stop.clear()
        self._thread = threading.Thread(target=self._loop, name="rtmodt-ingest", daemon=True)
        self._thread.start()
        log.info("FrameReader started | source=%s backend=%s", self.source, self.backend)
        return self

    def read(self, copy: bool = True):
        """``(ok, frame, frame_id)``, non-blocking.  ``copy=False`` hands out the (page-locked) ring slot itself under a
        lease: it is not rewritten until ``release(frame_id)``.  With every slot but the latest and the write slot on
        lease the call degrades to a copy (counted in ``lease_misses``) instead of failing -- the reference's ``read`` never
        raises (rtsp_reader.py:117-149); release earlier frames after their batch's ``fetch``, or build a larger ring."""
        with self._lock:
            if self._latest is None:
                return False, None, self._frame_id
            self._served_id = self._frame_id
            if copy:
                return True, self._latest.copy(), self._frame_id
            if self._ring is not None:                 # (without a ring every frame is a fresh array: nothing to protect)
                slot = self._latest_slot
                ent = self._lease_slot.get(self._frame_id)        # [slot or -1, holders]: every read(copy=False) of a frame id is one HOLDER, whatever it was handed
                if (ent is None or ent[0] < 0) and slot not in self._leases and len(self._leases) >= self._ring.slots - 2:
                    # every leasable slot is out: hand a copy over.  The holder is still counted under this frame id, so that its release() can
                    # never be mistaken for the release of a REAL lease a later read of the same (still latest) id obtains
                    self.lease_misses += 1
                    if self.lease_misses == 1:
                        log.warning("FrameReader: every leasable ring slot is on lease -- read(copy=False) now returns pageable copies "
                                    "(release frames after their batch's fetch, or build a larger ring)")
                    if ent is None:
                        self._lease_slot[self._frame_id] = [-1, 1]
                    else:
                        ent[1] += 1
                    return True, self._latest.copy(), self._frame_id
                if ent is None or ent[0] < 0:          # first real lease of this frame id: the slot goes on lease
                    self._leases[slot] = self._leases.get(slot, 0) + 1
                    if ent is None:
                        self._lease_slot[self._frame_id] = [slot, 1]
                    else:
                        ent[0] = slot; ent[1] += 1
                else:
                    ent[1] += 1
            return True, self._latest, self._frame_id

    def release(self, frame_id: int) -> None:
        """One holder of ``frame_id`` (a ``read(copy=False)`` of it) is done.  The frame's ring slot comes off lease when its LAST holder has
        released -- copy-served holders included, so a slot is never rewritten under a holder that still uploads from it.  No-op for ids that
        hold nothing."""
        with self._lock:
            ent = self._lease_slot.get(frame_id)
            if ent is None:
                return
            ent[1] -= 1
            if ent[1] > 0:
                return
            del self._lease_slot[frame_id]
            slot = ent[0]
            if slot < 0:
                return
            left = self._leases.get(slot, 0) - 1
            if left > 0:
                self._leases[slot] = left
            else:
                self._leases.pop(slot, None)

    @property
    def leased(self) -> int:
        """Ring slots currently on lease."""
        with self._lock:
            return len(self._leases)

    def stop(self) -> None:
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=5.0)
        with self._lock:                               # leases do not outlive the reader
            self._leases.clear()
            self._lease_slot.clear()
        self._release()
        log.info("FrameReader stopped.")

    @property
    def is_alive(self) -> bool:
        return not self._stop.is_set() and self._thread is not None and self._thread.is_alive()

    def __enter__(self) -> "FrameReader":
        return self.start()

    def __exit__(self, *_) -> None:
        self.stop()

    # ---- internals ----
    def _open(self) -> None:
        cap = _BACKENDS[self.backend](self.source, resolution=self.resolution, **self._backend_kw)
        if not getattr(cap, "opened", False):
            raise ConnectionError(f"Cannot open stream: {self.source}")
        self._cap = cap

    def _release(self) -> None:
        if self._cap is not None:
            self._cap.release()
            self._cap = None

    def _loop(self) -> None:
        failures = 0
        while not self._stop.is_set():
            if self._cap is None or not self._cap.opened:
                if failures >= self.max_reconnects:
                    log.error("Max reconnect attempts reached. Stopping reader.")
                    self._stop.set()
                    break
                failures += 1
                self.reconnects += 1
                wait = self.reconnect_delay * min(failures, 5)
                log.warning("Reconnecting (%d/%d) in %.1fs", failures, self.max_reconnects, wait)
                if self._stop.wait(wait):
                    break
                try:
                    self._open()
                except ConnectionError:
                    pass
                continue
            if not self._cap.grab():
                self._release()
                continue
            dst, slot = None, -1
            if self._ring is not None:
                with self._lock:                         # next slot that is neither on lease nor the latest frame
                    n = self._ring.slots
                    slot = next((c for c in ((self._slot + k) % n for k in range(1, n + 1))
                                 if c not in self._leases and c != self._latest_slot), -1)
                if slot < 0:                             # cannot happen while leases <= slots - 2; never write over a lease
                    self._cap.retrieve(None)
                    self.dropped += 1
                    continue
                self._slot = slot
                dst = self._ring.frame(slot)
            ok, frame = self._cap.retrieve(dst)
            if not ok or frame is None:
                continue
            with self._lock:
                if self._frame_id != self._served_id:
                    self.dropped += 1
                self._latest = frame
                self._latest_slot = slot
                self._frame_id += 1
            failures = 0


RTSPReader = FrameReader        # the reference's name for it
