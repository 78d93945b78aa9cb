"""Fused-conv weight files for the native detector.

The reference loads an Ultralytics ``.pt``/``.engine`` by path
(``src/detection/detector.py:82-90``).  The native engine reads its own flat
format instead (no pickle, no Python objects): one record per BN-folded
convolution, weights fp16 ``w[cout][kh][kw][cin]`` (the NHWC implicit-GEMM
operand order), bias fp32.

``RTMODTW1`` layout (little endian)::

    0   char[8]  magic "RTMODTW1"
    8   u32      version (1), scale_id (n,s,m,l,x = 0..4), nc, reg_max, n_convs, 3 x reserved
    40  n_convs x 72-byte records:
            char[32] name ("2.m.0.cv1", "22.cv3.1.2", ...), u32 cin, cout, k, stride, act, reserved,
            u64 w_offset, u64 b_offset            (byte offsets from the start of the file, 64-B aligned)
    ...  payloads

Also here: deterministic synthetic weights (there is no checkpoint on the build
or GPU boxes; throughput does not depend on weight values) and BatchNorm folding
for converting a real ``state_dict``.
"""
from __future__ import annotations

import struct

import numpy as np

from .yolo_spec import SCALE_ID, ConvSpec, conv_table

MAGIC = b"RTMODTW1"
_HDR = struct.Struct("<8s8I")
_REC = struct.Struct("<32s6I2Q")


def synthetic(scale: str = "s", nc: int = 80, reg_max: int = 16, seed: int = 0,
              calibrate: str | None = "noise", cls_bias: float = -4.1, input_size: int = 640) -> dict:
    """Seeded synthetic weights.  ``w ~ N(0, 1/(k*k*cin))``, ``b ~ N(0, 0.05^2)``, then -- because a
    63-conv SiLU stack with random weights either collapses to its biases or overflows
    fp16 -- every conv is rescaled layer by layer (LSUV-style, :func:`calibrate_`) so its
    pre-activation has unit standard deviation on a calibration set: ``calibrate="noise"``
    -> white-noise frames (the BASELINE video, ``synth.frames``); ``"mixed"`` -> noise +
    smooth (``synth.structured_frames``) + zeros, normalising the largest response so the
    other inputs shrink rather than explode; ``None`` -> raw fan-in scaling.  The class branch's output bias ``cls_bias`` puts ~2 %
    of the anchors of a random frame above ``conf = 0.35``.  Values are rounded to fp16 so
    the fp32 oracle and the fp16 engine share the exact same parameters.
    Returns ``{name: (w f32 [cout,k,k,cin], b f32 [cout])}``."""
    rng = np.random.default_rng(seed)
    out = {}
    for c in conv_table(scale, nc, reg_max):
        std = np.sqrt(1.0 / (c.k * c.k * c.cin))
        w = rng.normal(0.0, std, size=(c.cout, c.k, c.k, c.cin)).astype(np.float32)
        b = rng.normal(0.0, 0.05, size=c.cout).astype(np.float32)
        if c.act == 0:
            b[:] = 0.0 if ".cv2." in c.name else cls_bias
            b += rng.normal(0.0, 0.1, size=c.cout).astype(np.float32)
        out[c.name] = (w, b)
    if calibrate:
        calibrate_(out, scale, nc, reg_max, input_size, mode=calibrate)
    return {k: (w.astype(np.float16).astype(np.float32), b) for k, (w, b) in out.items()}


def torch_forward(x, weights: dict, scale: str = "s", nc: int = 80, reg_max: int = 16, on_conv=None):
    """fp32 torch-CPU forward of the YOLOv8 graph (NCHW, ``F.conv2d``) -- offline tooling
    only: LSUV calibration of synthetic weights and the "reference CPU path" timed by
    bench.py's cpu_baseline.  ``x``: (B,3,H,W) float tensor in [0,1] RGB.  ``on_conv(name,
    pre_activation) -> scale`` may rescale a conv in place (calibration).  Returns the three
    Detect maps ``(B, 4*reg_max+nc, H_i, W_i)``."""
    import torch
    import torch.nn.functional as F

    from .yolo_spec import repeats
    cache = {}

    def cv(name, t):
        w, b = weights[name]
        k = w.shape[1]
        stride = 2 if (name.isdigit() and k == 3) else 1
        if name not in cache or on_conv is not None:
            cache[name] = (torch.from_numpy(np.ascontiguousarray(w.transpose(0, 3, 1, 2))), torch.from_numpy(b))
        tw, tb = cache[name]
        y = F.conv2d(t, tw, tb, stride=stride, padding=k // 2)
        if on_conv is not None:
            s = on_conv(name, y, tb)
            if s != 1.0:
                weights[name] = (w * np.float32(s), b)
                y = (y - tb[None, :, None, None]) * s + tb[None, :, None, None]
        return F.silu(y) if not name.endswith(".2") or not name.startswith("22.") else y

    def c2f(i, t, n, shortcut):
        y = cv(f"{i}.cv1", t)
        ys = list(y.chunk(2, 1))
        for j in range(n):
            u = cv(f"{i}.m.{j}.cv2", cv(f"{i}.m.{j}.cv1", ys[-1]))
            ys.append(ys[-1] + u if shortcut else u)
        return cv(f"{i}.cv2", torch.cat(ys, 1))

    R = lambda n: repeats(n, scale)  # noqa: E731
    t = cv("1", cv("0", x))
    t = c2f(2, t, R(3), True)
    p3 = c2f(4, cv("3", t), R(6), True)
    p4 = c2f(6, cv("5", p3), R(6), True)
    t = c2f(8, cv("7", p4), R(3), True)
    y = cv("9.cv1", t)
    m1 = F.max_pool2d(y, 5, 1, 2)
    m2 = F.max_pool2d(m1, 5, 1, 2)
    m3 = F.max_pool2d(m2, 5, 1, 2)
    p5 = cv("9.cv2", torch.cat([y, m1, m2, m3], 1))
    h4 = c2f(12, torch.cat([F.interpolate(p5, scale_factor=2, mode="nearest"), p4], 1), R(3), False)
    h3 = c2f(15, torch.cat([F.interpolate(h4, scale_factor=2, mode="nearest"), p3], 1), R(3), False)
    h4 = c2f(18, torch.cat([cv("16", h3), h4], 1), R(3), False)
    h5 = c2f(21, torch.cat([cv("19", h4), p5], 1), R(3), False)
    outs = []
    for lvl, f in enumerate((h3, h4, h5)):
        bx = cv(f"22.cv2.{lvl}.2", cv(f"22.cv2.{lvl}.1", cv(f"22.cv2.{lvl}.0", f)))
        cl = cv(f"22.cv3.{lvl}.2", cv(f"22.cv3.{lvl}.1", cv(f"22.cv3.{lvl}.0", f)))
        outs.append(torch.cat([bx, cl], 1))
    return outs


def calibrate_(weights: dict, scale="s", nc=80, reg_max=16, input_size=640, threads: int = 8,
               mode: str = "noise") -> None:
    """In-place layer-sequential unit-variance scaling (see :func:`synthetic`)."""
    import torch

    from . import synth
    torch.set_num_threads(threads)
    if mode == "noise":
        cal = synth.frames(2, input_size, input_size, 4321)
    elif mode == "mixed":
        cal = np.concatenate([synth.frames(1, input_size, input_size, 4321),
                              synth.structured_frames(1, input_size, input_size, 7),
                              np.zeros((1, input_size, input_size, 3), np.uint8)])
    else:
        raise ValueError(f"unknown calibration mode {mode!r}")
    x = torch.from_numpy(np.ascontiguousarray(cal[..., ::-1].transpose(0, 3, 1, 2))).float() / 255.0

    def on_conv(name, y, tb):
        pre = y - tb[None, :, None, None]
        sd = float(pre.flatten(1).std(dim=1).max())
        target = 1.5 if (name.startswith("22.cv2.") and name.endswith(".2")) else 1.0
        # the measured deviation depends, in its last bits, on the host's conv kernels (ISA, thread split): keep 5 mantissa
        # bits of the scale so that every box derives the SAME weights from the same seed (parity tests then see one
        # sample, not one per CPU model); unit variance to within 3 % is all the calibration is for
        m, e = np.frexp(target / max(sd, 1e-12))
        return float(np.ldexp(np.round(m * 32.0) / 32.0, e))

    with torch.no_grad():
        torch_forward(x, weights, scale, nc, reg_max, on_conv=on_conv)


def digest(weights: dict) -> str:
    """sha256 over (name, fp16 weights, fp32 bias) in name order: two boxes that print the same digest ran the same net."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(weights):
        w, b = weights[name]
        h.update(name.encode())
        h.update(np.ascontiguousarray(w, dtype=np.float16).tobytes())
        h.update(np.ascontiguousarray(b, dtype=np.float32).tobytes())
    return h.hexdigest()[:16]


def save(path: str, weights: dict, scale: str = "s", nc: int = 80, reg_max: int = 16) -> None:
    table = conv_table(scale, nc, reg_max)
    off = _HDR.size + _REC.size * len(table)
    recs, blobs = [], []

    def place(blob: bytes):
        nonlocal off
        pad = (-off) % 64
        blobs.append(b"\0" * pad)
        off += pad
        at = off
        blobs.append(blob)
        off += len(blob)
        return at

    for c in table:
        w, b = weights[c.name]
        assert w.shape == (c.cout, c.k, c.k, c.cin), (c.name, w.shape)
        assert b.shape == (c.cout,), (c.name, b.shape)
        w_at = place(np.ascontiguousarray(w, dtype=np.float16).tobytes())
        b_at = place(np.ascontiguousarray(b, dtype=np.float32).tobytes())
        recs.append(_REC.pack(c.name.encode(), c.cin, c.cout, c.k, c.stride, c.act, 0, w_at, b_at))
    with open(path, "wb") as f:
        f.write(_HDR.pack(MAGIC, 1, SCALE_ID[scale], nc, reg_max, len(table), 0, 0, 0))
        for r in recs:
            f.write(r)
        for b in blobs:
            f.write(b)


def load(path: str):
    """Returns ``(weights dict (fp32 views of the stored fp16), scale, nc, reg_max)``."""
    raw = open(path, "rb").read()
    magic, ver, sid, nc, reg_max, n, *_ = _HDR.unpack_from(raw, 0)
    if magic != MAGIC or ver != 1:
        raise ValueError(f"{path}: not an RTMODTW1 weight file")
    scale = {v: k for k, v in SCALE_ID.items()}[sid]
    out = {}
    for i in range(n):
        name, cin, cout, k, stride, act, _, w_at, b_at = _REC.unpack_from(raw, _HDR.size + i * _REC.size)
        name = name.rstrip(b"\0").decode()
        w = np.frombuffer(raw, dtype=np.float16, count=cout * k * k * cin, offset=w_at).reshape(cout, k, k, cin)
        b = np.frombuffer(raw, dtype=np.float32, count=cout, offset=b_at)
        out[name] = (w.astype(np.float32), b.copy())
    return out, scale, nc, reg_max


def fold_bn(conv_w_oihw: np.ndarray, gamma, beta, mean, var, eps: float = 1e-3):
    """Conv(bias=False)+BatchNorm2d(eps=1e-3) -> conv+bias; returns ``(w[cout,kh,kw,cin], b)``."""
    s = np.asarray(gamma, np.float64) / np.sqrt(np.asarray(var, np.float64) + eps)
    w = np.asarray(conv_w_oihw, np.float64) * s[:, None, None, None]
    b = np.asarray(beta, np.float64) - np.asarray(mean, np.float64) * s
    return np.ascontiguousarray(w.transpose(0, 2, 3, 1)).astype(np.float32), b.astype(np.float32)


def from_state_dict(sd: dict, scale: str = "s", nc: int = 80, reg_max: int = 16) -> dict:
    """Convert an Ultralytics-style ``state_dict`` (``model.<i>...conv.weight`` / ``...bn.*``;
    values anything ``np.asarray`` accepts) into the fused dict.  Detect's final 1x1
    convs are plain ``Conv2d`` with bias (``model.22.cv2.<l>.2.weight`` / ``.bias``)."""
    out = {}
    for c in conv_table(scale, nc, reg_max):
        p = "model." + c.name
        if c.act:
            out[c.name] = fold_bn(np.asarray(sd[p + ".conv.weight"]), np.asarray(sd[p + ".bn.weight"]),
                                  np.asarray(sd[p + ".bn.bias"]), np.asarray(sd[p + ".bn.running_mean"]),
                                  np.asarray(sd[p + ".bn.running_var"]))
        else:
            w = np.asarray(sd[p + ".weight"], np.float32).transpose(0, 2, 3, 1)
            out[c.name] = (np.ascontiguousarray(w), np.asarray(sd[p + ".bias"], np.float32))
    return out


# ---------------------------------------------------------------------------------------
# Reading an Ultralytics checkpoint without Ultralytics
# ---------------------------------------------------------------------------------------
class _Inert:
    """Stand-in for every class the restricted unpickler does not know (``ultralytics.*``,
    ``torch.nn.*`` modules, loss objects, ...): holds whatever state the pickle restores,
    runs no code of the original class."""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state

    def __call__(self, *a, **k):                     # objects rebuilt through REDUCE with odd callables
        return _Inert()


_STORAGE_DTYPES = {"FloatStorage": np.float32, "HalfStorage": np.float16, "DoubleStorage": np.float64,
                   "LongStorage": np.int64, "IntStorage": np.int32, "ShortStorage": np.int16, "CharStorage": np.int8,
                   "ByteStorage": np.uint8, "BoolStorage": np.bool_, "BFloat16Storage": "bf16"}


def read_pt(path: str) -> dict:
    """``state_dict``-like ``{dotted.name: np.ndarray}`` of the ``model`` (or ``ema``) entry of a
    ``torch.save`` zip checkpoint, obtained with a RESTRICTED unpickler: only tensors are
    materialised, every unknown class becomes an inert container, no code from the
    checkpoint's classes (and no ``ultralytics`` import) ever runs.  This replaces
    ``YOLO(path)`` at src/detection/detector.py:84."""
    import collections
    import pickle
    import zipfile

    zf = zipfile.ZipFile(path)
    names = zf.namelist()
    pkl = next(n for n in names if n.endswith("data.pkl"))
    root = pkl[: -len("data.pkl")]

    class _StorageType:
        def __init__(self, name):
            self.dtype = _STORAGE_DTYPES[name]

    def _rebuild_tensor_v2(storage, offset, size, stride, *unused):
        arr, dt = storage
        if len(size) == 0:
            return arr[offset:offset + 1].reshape(())
        a = np.lib.stride_tricks.as_strided(arr[offset:], shape=tuple(size), strides=tuple(s * arr.itemsize for s in stride))
        a = np.array(a)                                   # own, contiguous copy
        if dt == "bf16":
            a = (a.astype(np.uint32) << 16).view(np.float32)
        return a

    def _rebuild_parameter(data, requires_grad=False, backward_hooks=None, *a):
        return data

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "collections" and name == "OrderedDict":
                return collections.OrderedDict
            if module == "torch._utils" and name in ("_rebuild_tensor_v2", "_rebuild_tensor"):
                return _rebuild_tensor_v2
            if module == "torch._utils" and name in ("_rebuild_parameter", "_rebuild_parameter_with_state"):
                return _rebuild_parameter
            if module in ("torch", "torch.storage") and name in _STORAGE_DTYPES:
                return _StorageType(name)
            if module == "torch" and name == "Size":
                return tuple
            if module == "builtins" and name in ("set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "slice", "range", "complex", "bytes", "bytearray"):
                return getattr(__import__("builtins"), name)
            return type(name, (_Inert,), {"__module__": module})     # inert: never the real class

        def persistent_load(self, pid):
            # ('storage', storage_type, key, location, numel)
            kind, stype, key = pid[0], pid[1], pid[2]
            assert kind == "storage"
            dt = stype.dtype if isinstance(stype, _StorageType) else np.float32
            raw = zf.read(f"{root}data/{key}")
            arr = np.frombuffer(raw, dtype=np.uint16 if dt == "bf16" else dt)
            return (arr, dt)

    ckpt = U(zf.open(pkl)).load()
    model = ckpt
    if isinstance(ckpt, dict):
        model = ckpt.get("ema") or ckpt.get("model") or ckpt
    out = {}

    def walk(obj, prefix):
        d = getattr(obj, "__dict__", None)
        if isinstance(obj, dict) and not d:               # a plain state_dict
            for k, v in obj.items():
                if isinstance(v, np.ndarray):
                    out[prefix + str(k)] = v
            return
        if not d:
            return
        for group in ("_parameters", "_buffers"):
            for k, v in (d.get(group) or {}).items():
                if isinstance(v, np.ndarray):
                    out[prefix + k] = v
        for k, v in (d.get("_modules") or {}).items():
            walk(v, prefix + k + ".")

    walk(model, "")
    if not out:
        raise ValueError(f"{path}: no tensors found (not a torch zip checkpoint of a module or state_dict?)")
    return out


def infer_scale(sd: dict) -> str:
    """Model scale from the stem / backbone widths of a state_dict."""
    c1 = int(np.asarray(sd["model.0.conv.weight"]).shape[0])
    n_c2f = len([k for k in sd if k.startswith("model.2.m.") and k.endswith(".cv1.conv.weight")])
    for s in "nsmlx":
        from .yolo_spec import repeats, width
        if width(64, s) == c1 and repeats(3, s) == n_c2f:
            return s
    raise ValueError(f"cannot infer YOLOv8 scale from stem width {c1} / {n_c2f} bottlenecks")


def convert_pt(pt_path: str, out_path: str, scale: str | None = None) -> tuple:
    """``.pt`` -> ``RTMODTW1``: restricted read, BN folding, NHWC weight order, fp16.  Returns (scale, nc)."""
    sd = read_pt(pt_path)
    sd = {k: np.asarray(v, dtype=np.float32) if np.asarray(v).dtype.kind == "f" else np.asarray(v) for k, v in sd.items()}
    scale = scale or infer_scale(sd)
    nc = int(sd["model.22.cv3.0.2.weight"].shape[0])
    fused = from_state_dict(sd, scale, nc)
    save(out_path, fused, scale, nc)
    return scale, nc


def spec(scale="s", nc=80, reg_max=16) -> list[ConvSpec]:
    return conv_table(scale, nc, reg_max)
