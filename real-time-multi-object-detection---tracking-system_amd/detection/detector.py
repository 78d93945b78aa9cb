"""Drop-in for the reference's ``src/detection/detector.py`` on MI355X.

Same public surface (``Detector(...)``, ``Detector.detect(frame) -> Detections``,
``Detections.filter_classes``; reference file lines 29-135), but every number is
produced by the hand-written HIP kernels behind ``include/rtmodt.h`` -- letterbox,
the YOLOv8 forward pass, DFL decode, per-class NMS and box rescale all run on the GPU,
one D2H copy of at most ``max_det`` rows per frame (what ``Detector._parse`` does at
reference lines 117-129).

Differences a maintainer should know (also in INTEGRATION.md):

* ``model_path`` / ``fallback_model`` name an ``RTMODTW1`` fused-weight file
  (``weights.py``), not an Ultralytics ``.pt`` / TensorRT ``.engine``.
* The engine has a static input shape like the reference's preferred TensorRT engine
  (``config/default.yaml:32``): frames are letterboxed to ``input_size[0]`` squared
  (only ``input_size[0]`` is used, as at reference line 102), never to a minimal rectangle.
* Activations are fp16 with fp32 accumulation regardless of ``half`` (``half=False`` is
  accepted for signature compatibility and reported in ``self.half``).
"""
from __future__ import annotations

import ctypes as C
import logging
from dataclasses import dataclass, field
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from .. import _ffi
from ..yolo_spec import COCO_NAMES

log = logging.getLogger("rtmodt.detector")


@dataclass
class Detections:
    """One frame's detections (reference: detector.py:29-48)."""
    xyxy: np.ndarray            # (N, 4) float32
    confidence: np.ndarray      # (N,)   float32
    class_id: np.ndarray        # (N,)   int32
    class_names: list = field(default_factory=list)

    def __len__(self) -> int:
        return len(self.confidence)

    def filter_classes(self, keep: Sequence[int]) -> "Detections":
        sel = np.isin(self.class_id, keep)
        names = [nm for nm, s in zip(self.class_names, sel) if s]
        return Detections(self.xyxy[sel], self.confidence[sel], self.class_id[sel], names)


def _empty() -> Detections:
    return Detections(np.empty((0, 4), dtype=np.float32), np.empty(0, dtype=np.float32), np.empty(0, dtype=np.int32))


class _NativeModel:
    """What ``Detector.model`` holds: the native engine handle plus the bits of the
    Ultralytics model object callers look at (``names``)."""

    def __init__(self, handle, path: str, scale_id: int, nc: int, n_anchors: int, n_convs: int, flops: int, arena: int):
        self.handle = handle
        self.path = path
        self.scale = "nsmlx"[scale_id]
        self.nc, self.n_anchors, self.n_convs = nc, n_anchors, n_convs
        self.conv_flops_per_frame, self.arena_bytes = flops, arena
        self.names = {i: (COCO_NAMES[i] if nc == 80 else str(i)) for i in range(nc)}

    def __repr__(self):
        return f"<rtmodt YOLOv8{self.scale} nc={self.nc} anchors={self.n_anchors} convs={self.n_convs} file={Path(self.path).name}>"


class Detector:
    """YOLOv8 detector on the native gfx950 engine (reference: detector.py:54-135).

    Keyword-only extensions (no reference counterpart): ``batch`` frames per launch set (streams batched per GPU);
    ``chains`` picks the engine (``rtmodt_det_cfg.chains``): 0 automatic (two stages for ``batch >= 2``, the plain
    engine for ``batch == 1``), 1 plain, n > 1 sub-batch chains, -1 / -2 the staged engine with two / three stages
    (keep stages + 1 batches in flight through :meth:`enqueue` / :meth:`fetch`; three stages only when the frames are
    already in device memory); ``rect`` the minimal-rectangle letterbox of ``predict`` on a ``.pt`` model."""

    _WARMUP_ITERATIONS = 10

    def __init__(
        self,
        model_path: str,
        fallback_model: Optional[str] = None,
        input_size: tuple = (640, 640),
        confidence: float = 0.35,
        iou: float = 0.45,
        classes: Optional[list] = None,
        half: bool = True,
        device: str = "cuda:0",
        max_det: int = 100,
        agnostic_nms: bool = False,
        *,
        batch: int = 1,
        max_source_size: Optional[tuple] = None,
        use_graph: bool = True,
        autotune: bool = True,
        chains: int = 0,
        warmup: bool = True,
        rect: bool = False,
    ) -> None:
        self.input_size = input_size
        self.confidence = confidence
        self.iou = iou
        self.classes = classes
        self.device = device
        self.max_det = max_det
        self.agnostic_nms = agnostic_nms
        self.batch = int(batch)
        self._ordinal = _ffi.device_ordinal(device)
        primary = Path(model_path)
        if primary.exists():                              # detector.py:82-90
            chosen = str(primary)
        elif fallback_model and Path(fallback_model).exists():
            chosen = str(fallback_model)
            log.warning("Primary model missing; loaded fallback: %s", fallback_model)
        else:
            raise FileNotFoundError(f"No model found at {model_path} or {fallback_model}")

        L = _ffi.lib()                                   # raises if the HIP library is not built
        ndev = C.c_int(0)
        _ffi.check(L.rtmodt_device_count(C.byref(ndev)))
        self.half = bool(half) and ndev.value > 0        # detector.py:76
        if not half:
            log.warning("half=False requested: the native engine stores activations in fp16 (fp32 accumulate) regardless")

        side = int(input_size[0])                         # detector.py:102 -- only input_size[0] is used
        self._side = side
        #: False -> every frame is letterboxed to side x side (what the reference's preferred TensorRT
        #: ``.engine`` does); True -> the minimal rectangle ``predict`` uses for a ``.pt`` model
        #: (LetterBox auto=True, 1080p -> 384x640): one native engine per rectangle, built on first use.
        self.rect = bool(rect)
        msw, msh = (max_source_size if max_source_size else (max(side, 1920), max(side, 1080)))
        cls_arr = None if classes is None else np.ascontiguousarray(classes, dtype=np.int32)
        self._cls_keepalive = cls_arr
        self._create_args = (chosen, cls_arr, int(msw), int(msh), int(bool(use_graph)), int(bool(autotune)), int(chains))
        self._models = {}
        self.model = self._model_for(side, side)
        self._xyxy = np.empty((self.batch, max_det, 4), np.float32)
        self._conf = np.empty((self.batch, max_det), np.float32)
        self._cls = np.empty((self.batch, max_det), np.int32)
        self._n = np.zeros(self.batch, np.int32)
        self._in_flight = []                              # frame counts of batches enqueued, not yet fetched
        self._keepalive = []                              # per batch in flight: the host arrays its upload / in-place read uses
        if warmup:
            self._warmup()

    def _model_for(self, in_h: int, in_w: int) -> "_NativeModel":
        """The native engine for an ``in_h x in_w`` network input (created and cached on first use)."""
        m = self._models.get((in_h, in_w))
        if m is None:
            chosen, cls_arr, msw, msh, use_graph, autotune, chains = self._create_args
            cfg = _ffi.DetCfg(chosen.encode(), in_w, in_h, float(self.confidence), float(self.iou),
                              None if cls_arr is None else cls_arr.ctypes.data_as(C.POINTER(C.c_int32)),
                              0 if cls_arr is None else len(cls_arr), 1, self._ordinal, int(self.max_det),
                              int(bool(self.agnostic_nms)), self.batch, msw, msh, use_graph, autotune, chains, int(self.rect))
            L = _ffi.lib()
            h = C.c_void_p()
            _ffi.check(L.rtmodt_detector_create(C.byref(cfg), C.byref(h)))
            sid, nc, na, ncv = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
            fl, ar = C.c_int64(), C.c_int64()
            _ffi.check(L.rtmodt_detector_info(h, C.byref(sid), C.byref(nc), C.byref(na), C.byref(ncv), C.byref(fl), C.byref(ar)))
            m = _NativeModel(h, chosen, sid.value, nc.value, na.value, ncv.value, fl.value, ar.value)
            m.input_hw = (in_h, in_w)
            nch = C.c_int32()
            _ffi.check(L.rtmodt_detector_chains(h, C.byref(nch)))
            m.chains = nch.value             # sub-batch chains the batch runs as (own streams, joined by the post-processing stream)
            _ffi.check(L.rtmodt_detector_stages(h, C.byref(nch)))
            m.stages = nch.value             # 2 / 3: staged engine (chains=-1 / -2), the stages of the net on their own streams
            self._models[(in_h, in_w)] = m
        return m

    @staticmethod
    def rect_shape(h: int, w: int, imgsz: int, stride: int = 32) -> tuple:
        """Network input (H, W) that ultralytics' ``LetterBox(auto=True)`` produces for an ``h x w`` frame."""
        r = min(imgsz / h, imgsz / w)
        nw, nh = int(round(w * r)), int(round(h * r))
        return nh + (imgsz - nh) % stride, nw + (imgsz - nw) % stride

    # ------------------------------------------------------------------
    def detect(self, frame: np.ndarray) -> Detections:
        """One BGR uint8 frame -> ``Detections`` (detector.py:98-112)."""
        return self.detect_batch([frame])[0]

    def detect_batch(self, frames: Sequence[np.ndarray]) -> list:
        """Up to ``batch`` same-sized frames in one pass (streams batched per GPU)."""
        while self._in_flight:                            # drain anything enqueued earlier
            self.fetch()
        self.enqueue(frames)
        return self.fetch()

    def enqueue(self, frames: Sequence, height: int = 0, width: int = 0, pitch: int = 0) -> None:
        """Asynchronous half of :meth:`detect_batch`.  ``frames``: NumPy BGR images, or raw
        device addresses (ints) of BGR images with ``height/width/pitch`` given.

        Host frames in page-locked memory (``_ffi.PinnedArray`` / ``pipeline.PinnedFrameRing``) that need no resize are
        read by the stem kernel in place, other host frames are copied asynchronously: either way the caller must not
        rewrite a frame before the batch it belongs to has been fetched (the arrays themselves are kept alive here)."""
        n = len(frames)
        if n < 1 or n > self.batch:
            raise ValueError(f"{n} frames for a detector built with batch={self.batch}")
        arr = (C.c_void_p * n)()
        keep = None
        if isinstance(frames[0], (int, np.integer)):
            kind, h, w, p = _ffi.MEM_DEVICE, height, width, pitch or width * 3
            for i, a in enumerate(frames):
                arr[i] = int(a)
        else:
            kind = _ffi.MEM_HOST
            keep = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames]
            h, w = keep[0].shape[:2]
            p = keep[0].strides[0]
            for i, a in enumerate(keep):
                if a.shape[:2] != (h, w) or a.ndim != 3 or a.shape[2] != 3:
                    raise ValueError("frames of one batch must share one H x W x 3 shape")
                arr[i] = a.ctypes.data
        if self.rect:
            want = self.rect_shape(int(h), int(w), self._side)
            if want != self.model.input_hw:
                if self._in_flight:                       # they live in the other rectangle's engine
                    raise RuntimeError("fetch() the pending results before enqueueing frames of another size")
                self.model = self._model_for(*want)
        _ffi.check(_ffi.lib().rtmodt_detector_enqueue_batch(self.model.handle, arr, n, int(h), int(w), int(p), kind))
        self._in_flight.append(n)
        self._keepalive.append(keep)

    def fetch(self) -> list:
        """Results of the OLDEST batch in flight (up to three may be: enqueue t+1 [, t+2], fetch t)."""
        if not self._in_flight:
            raise RuntimeError("fetch() without a pending enqueue()")
        n = self._in_flight[0]
        _ffi.check(_ffi.lib().rtmodt_detector_fetch(self.model.handle, _ffi.ptr(self._xyxy), _ffi.ptr(self._conf),
                                                    _ffi.ptr(self._cls), _ffi.ptr(self._n)))
        self._in_flight.pop(0)
        self._keepalive.pop(0)                            # the batch's frames have been consumed
        return [self._parse(i) for i in range(n)]

    def synchronize(self) -> None:
        _ffi.check(_ffi.lib().rtmodt_synchronize(self._ordinal))

    def close(self) -> None:
        for m in getattr(self, "_models", {}).values():
            if m.handle:
                _ffi.lib().rtmodt_detector_destroy(m.handle)
                m.handle = None
        self._models = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _parse(self, i: int) -> Detections:
        k = int(self._n[i])
        if k == 0:                                        # detector.py:119-124
            return _empty()
        cls = self._cls[i, :k].copy()
        names = [self.model.names.get(int(c), str(c)) for c in cls]
        return Detections(self._xyxy[i, :k].copy(), self._conf[i, :k].copy(), cls, names)

    def _warmup(self) -> None:
        dummy = np.zeros((*self.input_size[::-1], 3), dtype=np.uint8)     # detector.py:131-135
        for _ in range(self._WARMUP_ITERATIONS):
            self.detect(dummy)

    # ---- introspection for parity tests / bench (no reference counterpart) -------------
    def debug_fetch(self, img: int = 0, want_input=True, want_heads=True, want_pred=True):
        m = self.model
        ih, iw = m.input_hw
        inp = np.empty((ih, iw, 3), np.float16) if want_input else None
        heads = np.empty(m.n_anchors * (64 + m.nc), np.float16) if want_heads else None
        pred = np.empty((4 + m.nc, m.n_anchors), np.float32) if want_pred else None
        _ffi.check(_ffi.lib().rtmodt_detector_debug_fetch(m.handle, img, _ffi.ptr(inp), _ffi.ptr(heads), _ffi.ptr(pred)))
        return inp, heads, pred

    def debug_layer(self, name: str, img: int = 0) -> np.ndarray:
        hwc = (C.c_int32 * 3)()
        _ffi.check(_ffi.lib().rtmodt_detector_debug_layer(self.model.handle, name.encode(), img, None, hwc))
        out = np.empty((hwc[0], hwc[1], hwc[2]), np.float16)
        _ffi.check(_ffi.lib().rtmodt_detector_debug_layer(self.model.handle, name.encode(), img, _ffi.ptr(out), hwc))
        return out

    def profile(self, iters: int = 5):
        """Per-launch device time (ms, HIP events) of eager forwards: [(name, ms, flops)]."""
        cap = 256
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_int64 * cap)()
        n = C.c_int32()
        _ffi.check(_ffi.lib().rtmodt_detector_profile(self.model.handle, iters, cap, names, ms, fl, C.byref(n)))
        return [(names[i].decode(), float(ms[i]), int(fl[i])) for i in range(min(n.value, cap))]

    def stage_times(self):
        """Device ms of the last fetched batch: (preprocess, inference, nms) -- HIP events."""
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        _ffi.check(_ffi.lib().rtmodt_detector_stage_times(self.model.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def clock_sampling(self, on: bool) -> None:
        """Start / stop sampling the in-kernel shader clock (one wave behind every batch's NMS; include/rtmodt.h)."""
        _ffi.check(_ffi.lib().rtmodt_detector_clock_enable(self.model.handle, int(bool(on))))

    def clock_read(self):
        """(mean, min, max) GHz and the sample count since sampling was enabled / last read."""
        a, b, c, n = C.c_double(), C.c_double(), C.c_double(), C.c_int32()
        _ffi.check(_ffi.lib().rtmodt_detector_clock_read(self.model.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(n)))
        return a.value, b.value, c.value, n.value

    def last_timing(self):
        a, b = C.c_float(), C.c_float()
        _ffi.check(_ffi.lib().rtmodt_detector_last_timing(self.model.handle, C.byref(a), C.byref(b)))
        return a.value, b.value
