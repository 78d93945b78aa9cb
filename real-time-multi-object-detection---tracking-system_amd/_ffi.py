"""ctypes binding of ``include/rtmodt.h`` (``lib/librtmodt_hip.so``).

Thin by design: argument marshalling and error translation only.  There is NO CPU
fallback -- if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librtmodt_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "rtmodt.h")

OK, E_INVALID, E_IO, E_HIP, E_CAPACITY, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5
MEM_HOST, MEM_DEVICE = 0, 1
ASSIGN_GREEDY, ASSIGN_LAPJV = 0, 1


class RtmodtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[rtmodt {code}] {msg}")
        self.code = code
        self.msg = msg


class DetCfg(C.Structure):
    _fields_ = [("weight_path", C.c_char_p), ("in_w", C.c_int32), ("in_h", C.c_int32), ("conf", C.c_float),
                ("iou", C.c_float), ("classes", C.POINTER(C.c_int32)), ("n_classes", C.c_int32), ("half", C.c_int32),
                ("device", C.c_int32), ("max_det", C.c_int32), ("agnostic", C.c_int32), ("batch", C.c_int32),
                ("max_src_w", C.c_int32), ("max_src_h", C.c_int32), ("use_graph", C.c_int32), ("autotune", C.c_int32), ("chains", C.c_int32), ("rect", C.c_int32)]


class ZoneCfg(C.Structure):                          # struct rtmodt_zone_cfg
    _fields_ = [("polygon_xy", C.POINTER(C.c_int32)), ("n_points", C.c_int32), ("dwell_time_sec", C.c_double),
                ("cooldown_sec", C.c_double), ("key", C.c_int32)]


_lib = None


def header_symbols() -> list[str]:
    """Every function name ``include/rtmodt.h`` declares."""
    txt = open(HEADER_PATH).read()
    return sorted(set(re.findall(r"\b(rtmodt_[a-z0-9_]+)\s*\(", txt)))


def lib() -> C.CDLL:
    """Loads the library (once).  Raises ``RtmodtError`` if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtmodtError(E_IO, f"{LIB_PATH} not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i32, f32, i64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64
    sig = {
        "rtmodt_last_error": (C.c_char_p, []),
        "rtmodt_version": (C.c_char_p, []),
        "rtmodt_build_info": (C.c_char_p, []),
        "rtmodt_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_char_p)]),
        "rtmodt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
        "rtmodt_synchronize": (C.c_int, [C.c_int]),
        "rtmodt_device_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
        "rtmodt_device_free": (C.c_int, [C.c_int, vp]),
        "rtmodt_host_alloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(vp)]),
        "rtmodt_host_free": (C.c_int, [C.c_int, vp]),
        "rtmodt_memcpy_h2d": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
        "rtmodt_memcpy_d2h": (C.c_int, [C.c_int, vp, vp, C.c_size_t]),
        "rtmodt_detector_create": (C.c_int, [C.POINTER(DetCfg), C.POINTER(vp)]),
        "rtmodt_detector_destroy": (None, [vp]),
        "rtmodt_detector_detect": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
        "rtmodt_detector_detect_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
        "rtmodt_detector_enqueue_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "rtmodt_detector_fetch": (C.c_int, [vp, vp, vp, vp, vp]),
        "rtmodt_detector_info": (C.c_int, [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                           C.POINTER(i64), C.POINTER(i64)]),
        "rtmodt_detector_chains": (C.c_int, [vp, C.POINTER(i32)]),
        "rtmodt_detector_stages": (C.c_int, [vp, C.POINTER(i32)]),
        "rtmodt_detector_debug_fetch": (C.c_int, [vp, C.c_int, vp, vp, vp]),
        "rtmodt_detector_debug_layer": (C.c_int, [vp, C.c_char_p, C.c_int, vp, C.POINTER(i32)]),
        "rtmodt_detector_profile": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(f32),
                                              C.POINTER(i64), C.POINTER(i32)]),
        "rtmodt_detector_last_timing": (C.c_int, [vp, C.POINTER(f32), C.POINTER(f32)]),
        "rtmodt_detector_stage_times": (C.c_int, [vp, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32)]),
        "rtmodt_detector_clock_enable": (C.c_int, [vp, C.c_int]),
        "rtmodt_detector_clock_read": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
        "rtmodt_nms_pred": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, f32, f32, vp, C.c_int, C.c_int, C.c_int,
                                      vp, vp, vp, vp, C.POINTER(i32)]),
        "rtmodt_preprocess": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
        "rtmodt_tracker_create": (C.c_int, [C.c_int, f32, C.c_int, f32, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
        "rtmodt_tracker_destroy": (None, [vp]),
        "rtmodt_tracker_set_cost_limit": (C.c_int, [vp, C.c_double]),
        "rtmodt_tracker_update": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(i32)]),
        "rtmodt_tracker_update_batch": (C.c_int, [vp, vp, vp, vp, vp, vp]),
        "rtmodt_tracker_update_from_detector": (C.c_int, [vp, vp]),
        "rtmodt_tracker_update_from_detector_frames": (C.c_int, [vp, vp, C.c_int, C.c_int]),
        "rtmodt_tracker_update_from_detector_batch": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int]),
        "rtmodt_tracker_state": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i64)]),
        "rtmodt_tracker_reset": (C.c_int, [vp, C.c_int]),
        "rtmodt_tracker_enable_kalman": (C.c_int, [vp]),
        "rtmodt_tracker_kalman_state": (C.c_int, [vp, C.c_int, vp, vp, C.POINTER(i32)]),
        "rtmodt_iou_matrix": (C.c_int, [C.c_int, vp, C.c_int, vp, C.c_int, vp]),
        "rtmodt_assign_greedy": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, f32, vp, vp]),
        "rtmodt_assign_lapjv": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.c_double, vp, vp]),
        "rtmodt_zones_create": (C.c_int, [C.c_int, C.POINTER(ZoneCfg), C.c_int, C.c_int, C.c_int, C.c_int, i64, C.POINTER(vp)]),
        "rtmodt_zones_destroy": (None, [vp]),
        "rtmodt_zones_process": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_double, i64, vp, vp, vp, vp, C.POINTER(i32)]),
        "rtmodt_zones_process_tracker": (C.c_int, [vp, vp, C.c_double, i64, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
        "rtmodt_zones_state": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, C.POINTER(i32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here == header/library drift
        fn.restype = res
        fn.argtypes = args
    L._signatures = sig
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != OK:
        raise RtmodtError(rc, lib().rtmodt_last_error().decode(errors="replace"))


def ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    """GPUs visible to this process (``rtmodt_device_count``)."""
    n = C.c_int(0)
    check(lib().rtmodt_device_count(C.byref(n)))
    return n.value


def device_ordinal(device) -> int:
    """``"cuda:1"`` / ``"1"`` / ``1`` -> 1 (ROCm keeps the ``cuda`` spelling, detector.py:66)."""
    if isinstance(device, int):
        return device
    s = str(device)
    if s in ("cuda", "hip", "gpu", ""):
        return 0
    if ":" in s:
        s = s.split(":", 1)[1]
    return int(s)


# ---- small functional wrappers used by tests / bench ----------------------------------
def iou_matrix(a: np.ndarray, b: np.ndarray, device: int = 0) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(b, np.float32).reshape(-1, 4)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    check(lib().rtmodt_iou_matrix(device, ptr(a), a.shape[0], ptr(b), b.shape[0], ptr(out)))
    return out


def assign_greedy(iou: np.ndarray, thresh: float, device: int = 0):
    iou = np.ascontiguousarray(iou, np.float32)
    m, n = iou.shape
    r2c = np.empty(m, np.int32)
    used = np.empty(n, np.int32)
    check(lib().rtmodt_assign_greedy(device, ptr(iou), m, n, float(thresh), ptr(r2c), ptr(used)))
    mr = [int(i) for i in range(m) if r2c[i] >= 0]
    return mr, [int(r2c[i]) for i in mr], [int(i) for i in range(m) if r2c[i] < 0], [int(j) for j in range(n) if not used[j]]


def assign_lapjv(iou: np.ndarray, thresh: float, device: int = 0):
    """tracker.py:168-181: ``lap.lapjv(1 - iou, extend_cost=True, cost_limit=1 - thresh)`` -> the four lists."""
    iou = np.ascontiguousarray(iou, np.float32)
    m, n = iou.shape
    r2c = np.empty(m, np.int32)
    used = np.empty(n, np.int32)
    check(lib().rtmodt_assign_lapjv(device, ptr(iou), m, n, float(1 - thresh), ptr(r2c), ptr(used)))
    mr = [int(i) for i in range(m) if r2c[i] >= 0]
    return mr, [int(r2c[i]) for i in mr], [int(i) for i in range(m) if r2c[i] < 0], [int(j) for j in range(n) if not used[j]]


def nms_pred(pred: np.ndarray, conf=0.35, iou=0.45, classes=None, agnostic=False, max_det=100, device: int = 0):
    pred = np.ascontiguousarray(pred, np.float32)
    nc, A = pred.shape[0] - 4, pred.shape[1]
    cl = None if classes is None else np.ascontiguousarray(classes, np.int32)
    xy = np.empty((max_det, 4), np.float32)
    cf = np.empty(max_det, np.float32)
    ci = np.empty(max_det, np.int32)
    an = np.empty(max_det, np.int32)
    n = C.c_int32(0)
    check(lib().rtmodt_nms_pred(device, ptr(pred), nc, A, float(conf), float(iou), ptr(cl), 0 if cl is None else len(cl),
                                int(bool(agnostic)), int(max_det), ptr(xy), ptr(cf), ptr(ci), ptr(an), C.byref(n)))
    k = n.value
    return xy[:k].copy(), cf[:k].copy(), ci[:k].copy(), an[:k].copy()


def preprocess(frame: np.ndarray, in_w: int = 640, in_h: int = 640, device: int = 0) -> np.ndarray:
    frame = np.ascontiguousarray(frame, np.uint8)
    h, w = frame.shape[:2]
    out = np.empty((in_h, in_w, 3), np.float16)
    check(lib().rtmodt_preprocess(device, ptr(frame), h, w, frame.strides[0], in_w, in_h, ptr(out)))
    return out


class DeviceBuffer:
    """A raw device allocation owned by the library (frame rings for the bench)."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device, self.nbytes = device, nbytes
        p = C.c_void_p()
        check(lib().rtmodt_device_alloc(device, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, host: np.ndarray, offset: int = 0):
        host = np.ascontiguousarray(host)
        assert offset + host.nbytes <= self.nbytes
        check(lib().rtmodt_memcpy_h2d(self.device, C.c_void_p(self.ptr + offset), ptr(host), host.nbytes))

    def free(self):
        if self.ptr:
            lib().rtmodt_device_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A NumPy view over page-locked host memory owned by the library: frames written here (by a decoder,
    a capture thread, ...) reach the GPU by asynchronous DMA underneath the previous batch's compute."""

    def __init__(self, shape, dtype=np.uint8, device: int = 0):
        self.device = device
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(lib().rtmodt_host_alloc(device, max(self.nbytes, 1), C.byref(p)))
        self.ptr = p.value
        buf = (C.c_uint8 * self.nbytes).from_address(self.ptr)
        buf._owner = self                                 # every NumPy view of the memory keeps its owner (and so the allocation) alive
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            lib().rtmodt_host_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
