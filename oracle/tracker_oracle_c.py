"""ctypes wrapper of oracle/tracker_oracle.c (TEST INFRASTRUCTURE / CPU baseline)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libtracker_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "libtracker_oracle.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.tro_create.restype = C.c_void_p
        L.tro_create.argtypes = [C.c_float, C.c_int, C.c_float, C.c_int]
        L.tro_destroy.argtypes = [C.c_void_p]
        L.tro_update.restype = C.c_int
        L.tro_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.tro_count.restype = C.c_int
        L.tro_count.argtypes = [C.c_void_p]
        L.tro_next_id.restype = C.c_int64
        L.tro_next_id.argtypes = [C.c_void_p]
        L.tro_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.tro_batch_iou.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


class TrackerOracleC:
    def __init__(self, track_thresh=0.5, track_buffer=30, match_thresh=0.8, cap=8192):
        self._h = lib().tro_create(track_thresh, track_buffer, match_thresh, cap)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().tro_destroy(self._h)
            self._h = None

    def update(self, xyxy, conf, cls) -> int:
        xyxy = np.ascontiguousarray(xyxy, dtype=np.float32).reshape(-1, 4)
        conf = np.ascontiguousarray(conf, dtype=np.float32).reshape(-1)
        cls = np.ascontiguousarray(cls, dtype=np.int32).reshape(-1)
        r = lib().tro_update(self._h, xyxy.ctypes.data, conf.ctypes.data, cls.ctypes.data, conf.shape[0])
        if r < 0:
            raise RuntimeError("tracker oracle capacity exceeded")
        return r

    def snapshot(self) -> dict:
        n = lib().tro_count(self._h)
        s = {"ids": np.empty(n, np.int64), "xyxy": np.empty((n, 4), np.float32), "conf": np.empty(n, np.float32),
             "cls": np.empty(n, np.int32), "age": np.empty(n, np.int32), "tsu": np.empty(n, np.int32)}
        lib().tro_state(self._h, *[s[k].ctypes.data for k in ("ids", "xyxy", "conf", "cls", "age", "tsu")])
        s["next_id"] = int(lib().tro_next_id(self._h))
        return s


def batch_iou(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 4)
    b = np.ascontiguousarray(b, np.float32).reshape(-1, 4)
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().tro_batch_iou(a.ctypes.data, a.shape[0], b.ctypes.data, b.shape[0], out.ctypes.data)
    return out
