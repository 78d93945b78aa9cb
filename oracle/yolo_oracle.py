"""CPU restatement (NumPy, float32) of what ``Detector.detect`` computes --
TEST INFRASTRUCTURE ONLY, **PARITY UNPINNED**.

The reference's ``src/detection/detector.py:98-129`` is a thin adapter around
``ultralytics.YOLO.predict``; every number is produced by third-party code that
is not vendored, not pinned (``ultralytics>=8.1.0``, ``torchvision>=0.16.0``,
``opencv-python-headless>=4.8.0``; /root/reference/requirements.txt:12-14,26) and not
installed here, and the reference holds no test or golden vector at that
boundary.  This file therefore restates the *published* algorithm of those
packages and is anchored on:

* the reference call site (detector.py:100-111) for parameters and dtypes,
* known-answer checks: fused parameter counts 3 151 904 / 11 156 544 /
  25 886 080 (n/s/m) and 2xMAC FLOPs 8.74 / 28.60 / 78.94 G at 640x640,
* independent arithmetic (torch CPU conv2d / max_pool2d / softmax) in
  tests/test_oracle_yolo.py.

Layout: activations are NHWC float32 ``(H, W, C)`` for one image; fused conv
weights are ``w[cout, kh, kw, cin]`` + ``b[cout]``.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}


# ---------------------------------------------------------------------------
# architecture
# ---------------------------------------------------------------------------
def _ch(c, width, max_ch):
    return int(math.ceil(min(c, max_ch) * width / 8) * 8)


def _rep(n, depth):
    return max(round(n * depth), 1)


def arch(scale: str = "s", nc: int = 80, reg_max: int = 16):
    """Returns the backbone/neck module list [(idx, kind, from, args...)] and head info."""
    d, w, mc = SCALES[scale]
    c = lambda x: _ch(x, w, mc)  # noqa: E731
    mods = [
        ("conv", -1, 3, c(64), 3, 2),            # 0
        ("conv", -1, c(64), c(128), 3, 2),       # 1
        ("c2f", -1, c(128), c(128), _rep(3, d), True),   # 2
        ("conv", -1, c(128), c(256), 3, 2),      # 3
        ("c2f", -1, c(256), c(256), _rep(6, d), True),   # 4  P3 skip
        ("conv", -1, c(256), c(512), 3, 2),      # 5
        ("c2f", -1, c(512), c(512), _rep(6, d), True),   # 6  P4 skip
        ("conv", -1, c(512), c(1024), 3, 2),     # 7
        ("c2f", -1, c(1024), c(1024), _rep(3, d), True),  # 8
        ("sppf", -1, c(1024), c(1024), 5),       # 9  P5
        ("up", -1),                              # 10
        ("cat", (-1, 6)),                        # 11
        ("c2f", -1, c(1024) + c(512), c(512), _rep(3, d), False),  # 12
        ("up", -1),                              # 13
        ("cat", (-1, 4)),                        # 14
        ("c2f", -1, c(512) + c(256), c(256), _rep(3, d), False),   # 15 -> P3
        ("conv", -1, c(256), c(256), 3, 2),      # 16
        ("cat", (-1, 12)),                       # 17
        ("c2f", -1, c(256) + c(512), c(512), _rep(3, d), False),   # 18 -> P4
        ("conv", -1, c(512), c(512), 3, 2),      # 19
        ("cat", (-1, 9)),                        # 20
        ("c2f", -1, c(512) + c(1024), c(1024), _rep(3, d), False),  # 21 -> P5
    ]
    ch_in = (c(256), c(512), c(1024))
    c2 = max(16, ch_in[0] // 4, reg_max * 4)
    c3 = max(ch_in[0], min(nc, 100))
    return mods, {"ch": ch_in, "c2": c2, "c3": c3, "nc": nc, "reg_max": reg_max}


def fused_convs(scale: str = "s", nc: int = 80, reg_max: int = 16):
    """Ordered list of fused convolutions ``(name, cin, cout, k, stride, act)``;
    ``act`` is 1 for SiLU, 0 for the plain (biased) Conv2d that ends each Detect branch."""
    mods, head = arch(scale, nc, reg_max)
    out = []
    for i, m in enumerate(mods):
        if m[0] == "conv":
            out.append((f"{i}", m[2], m[3], m[4], m[5], 1))
        elif m[0] == "c2f":
            c1, c2, n = m[2], m[3], m[4]
            c = c2 // 2
            out.append((f"{i}.cv1", c1, 2 * c, 1, 1, 1))
            for j in range(n):
                out.append((f"{i}.m.{j}.cv1", c, c, 3, 1, 1))
                out.append((f"{i}.m.{j}.cv2", c, c, 3, 1, 1))
            out.append((f"{i}.cv2", (2 + n) * c, c2, 1, 1, 1))
        elif m[0] == "sppf":
            c1, c2 = m[2], m[3]
            out.append((f"{i}.cv1", c1, c1 // 2, 1, 1, 1))
            out.append((f"{i}.cv2", c1 // 2 * 4, c2, 1, 1, 1))
    for lvl, ci in enumerate(head["ch"]):
        out.append((f"22.cv2.{lvl}.0", ci, head["c2"], 3, 1, 1))
        out.append((f"22.cv2.{lvl}.1", head["c2"], head["c2"], 3, 1, 1))
        out.append((f"22.cv2.{lvl}.2", head["c2"], 4 * reg_max, 1, 1, 0))
        out.append((f"22.cv3.{lvl}.0", ci, head["c3"], 3, 1, 1))
        out.append((f"22.cv3.{lvl}.1", head["c3"], head["c3"], 3, 1, 1))
        out.append((f"22.cv3.{lvl}.2", head["c3"], nc, 1, 1, 0))
    return out


def param_count(scale="s", nc=80, reg_max=16) -> int:
    """Fused parameters (weights + biases) + the ``reg_max`` DFL constants."""
    return sum(co * ci * k * k + co for _, ci, co, k, _, _ in fused_convs(scale, nc, reg_max)) + reg_max


def conv_flops(scale="s", h=640, w=640, nc=80, reg_max=16) -> int:
    """2 x MACs of every fused conv at input ``h x w`` (nothing else counted)."""
    res = _resolutions(scale, h, w, nc, reg_max)
    return sum(2 * res[name][0] * res[name][1] * co * ci * k * k for name, ci, co, k, _, _ in fused_convs(scale, nc, reg_max))


def _resolutions(scale, h, w, nc, reg_max):
    """Output (H, W) of each fused conv."""
    mods, head = arch(scale, nc, reg_max)
    cur = (h, w)
    saved = {}
    res = {}
    for i, m in enumerate(mods):
        if m[0] == "conv":
            cur = ((cur[0] + 1) // 2, (cur[1] + 1) // 2) if m[5] == 2 else cur
            res[f"{i}"] = cur
        elif m[0] == "c2f":
            res[f"{i}.cv1"] = cur
            res[f"{i}.cv2"] = cur
            for j in range(m[4]):
                res[f"{i}.m.{j}.cv1"] = cur
                res[f"{i}.m.{j}.cv2"] = cur
        elif m[0] == "sppf":
            res[f"{i}.cv1"] = cur
            res[f"{i}.cv2"] = cur
        elif m[0] == "up":
            cur = (cur[0] * 2, cur[1] * 2)
        saved[i] = cur
    for lvl, src in enumerate((15, 18, 21)):
        for br in ("cv2", "cv3"):
            for k in range(3):
                res[f"22.{br}.{lvl}.{k}"] = saved[src]
    return res


# ---------------------------------------------------------------------------
# primitive ops
# ---------------------------------------------------------------------------
def silu(x):
    x = x.astype(F32, copy=False)
    return (x / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def sigmoid(x):
    x = x.astype(F32, copy=False)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def conv2d_nhwc(x, w, b, stride=1, act=1):
    """x (H,W,Cin) f32; w (Cout,k,k,Cin); b (Cout,).  padding = k//2.  im2col + sgemm."""
    h, wd, cin = x.shape
    cout, k, _, cin_w = w.shape
    assert cin == cin_w, (cin, cin_w)
    p = k // 2
    ho = (h + 2 * p - k) // stride + 1
    wo = (wd + 2 * p - k) // stride + 1
    if k == 1 and stride == 1:
        cols = x.reshape(h * wd, cin)
    else:
        xp = np.zeros((h + 2 * p, wd + 2 * p, cin), dtype=F32)
        xp[p:p + h, p:p + wd] = x
        cols = np.empty((ho, wo, k, k, cin), dtype=F32)
        for kh in range(k):
            for kw in range(k):
                cols[:, :, kh, kw, :] = xp[kh:kh + stride * (ho - 1) + 1:stride, kw:kw + stride * (wo - 1) + 1:stride, :]
        cols = cols.reshape(ho * wo, k * k * cin)
    y = cols @ w.reshape(cout, -1).T.astype(F32)
    y = (y + b.astype(F32)[None, :]).astype(F32)
    if act:
        y = silu(y)
    return y.reshape(ho, wo, cout)


def maxpool5(x):
    """MaxPool2d(kernel 5, stride 1, padding 2) with -inf padding, NHWC."""
    h, w, c = x.shape
    xp = np.full((h + 4, w + 4, c), -np.inf, dtype=F32)
    xp[2:2 + h, 2:2 + w] = x
    out = xp[0:h, 0:w].copy()
    for dy in range(5):
        for dx in range(5):
            np.maximum(out, xp[dy:dy + h, dx:dx + w], out=out)
    return out


def upsample2(x):
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


# ---------------------------------------------------------------------------
# network forward
# ---------------------------------------------------------------------------
def forward(x, weights: dict, scale="s", nc=80, reg_max=16, taps: dict | None = None, force: dict | None = None, only=None):
    """x: (H,W,3) float32 RGB in [0,1].  ``weights[name] = (w, b)``.  Returns the three
    Detect maps ``[(H_i, W_i, 4*reg_max + nc)]`` (box logits first, then class logits).

    ``taps`` (optional dict) receives every fused conv's output by name (for a Bottleneck's
    second conv: the value AFTER the residual add, which is what the engine stores) and
    every module output by integer index.  ``force[name]`` (optional) replaces that conv's
    output before it is used downstream -- "teacher forcing" with the engine's own
    activations, so each layer can be checked in isolation with a tight tolerance.
    ``only`` (optional set of conv names): ONLY these convs are computed, every other conv's output is taken
    from ``force`` (which must then hold it) -- teacher-forced checks of a few layers of a large net (YOLOv8m
    at 1280 x 1280 is 316 GFLOP of NumPy) without running the rest; ``taps`` then holds the computed ones only
    (and the module outputs they feed, which are NOT meaningful: the Detect maps returned are not to be used)."""
    mods, head = arch(scale, nc, reg_max)
    force = force or {}

    def tap(name, y):
        if taps is not None:
            taps[name] = y
        f = force.get(name)
        return y if f is None else np.asarray(f, dtype=F32).reshape(y.shape)

    def cv(name, t, stride, act, res=None):
        w, b = weights[name]
        if only is not None and name not in only:          # not asked for: pass the engine's tensor on (or zeros of the right shape when it stored none)
            f = force.get(name)
            if f is not None:
                return np.asarray(f, dtype=F32)
            k = w.shape[1]
            return np.zeros(((t.shape[0] + 2 * (k // 2) - k) // stride + 1, (t.shape[1] + 2 * (k // 2) - k) // stride + 1, w.shape[0]), dtype=F32)
        y = conv2d_nhwc(t, w, b, stride=stride, act=act)
        if res is not None:
            y = (res + y).astype(F32)
        return tap(name, y)

    saved = {}
    cur = x.astype(F32)
    for i, m in enumerate(mods):
        kind = m[0]
        if kind == "conv":
            cur = cv(f"{i}", cur, m[5], 1)
        elif kind == "c2f":
            n, shortcut = m[4], m[5]
            y = cv(f"{i}.cv1", cur, 1, 1)
            c = y.shape[2] // 2
            ys = [y[..., :c], y[..., c:]]
            for j in range(n):
                t = cv(f"{i}.m.{j}.cv1", ys[-1], 1, 1)
                ys.append(cv(f"{i}.m.{j}.cv2", t, 1, 1, res=ys[-1] if shortcut else None))
            cur = cv(f"{i}.cv2", np.concatenate(ys, axis=2), 1, 1)
        elif kind == "sppf":
            y = cv(f"{i}.cv1", cur, 1, 1)
            p1 = maxpool5(y)
            p2 = maxpool5(p1)
            p3 = maxpool5(p2)
            cur = cv(f"{i}.cv2", np.concatenate([y, p1, p2, p3], axis=2), 1, 1)
        elif kind == "up":
            cur = upsample2(cur)
        elif kind == "cat":
            a, bidx = m[1]
            cur = np.concatenate([cur if a == -1 else saved[a], saved[bidx]], axis=2)
        saved[i] = cur
        if taps is not None:
            taps[i] = cur
    outs = []
    for lvl, src in enumerate((15, 18, 21)):
        f = saved[src]
        bx = cv(f"22.cv2.{lvl}.0", f, 1, 1)
        bx = cv(f"22.cv2.{lvl}.1", bx, 1, 1)
        bx = cv(f"22.cv2.{lvl}.2", bx, 1, 0)
        cl = cv(f"22.cv3.{lvl}.0", f, 1, 1)
        cl = cv(f"22.cv3.{lvl}.1", cl, 1, 1)
        cl = cv(f"22.cv3.{lvl}.2", cl, 1, 0)
        outs.append(np.concatenate([bx, cl], axis=2))
    return outs


# ---------------------------------------------------------------------------
# B.1 preprocess
# ---------------------------------------------------------------------------
def _round_half_even(v: float) -> int:
    return int(round(v))          # Python round == banker's, as in Ultralytics' LetterBox


def letterbox_params(h, w, new_h=640, new_w=640, auto=False, stride=32, scaleup=True):
    """Ultralytics LetterBox geometry: returns (resized_w, resized_h, top, bottom, left, right)."""
    r = min(new_h / h, new_w / w)
    if not scaleup:
        r = min(r, 1.0)
    uw, uh = _round_half_even(w * r), _round_half_even(h * r)
    dw, dh = new_w - uw, new_h - uh
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = _round_half_even(dh - 0.1), _round_half_even(dh + 0.1)
    left, right = _round_half_even(dw - 0.1), _round_half_even(dw + 0.1)
    return uw, uh, top, bottom, left, right


def _resize_coeffs(dst: int, src: int):
    """OpenCV INTER_LINEAR tables for 8-bit images: source index pair and 11-bit
    fixed-point weights (cv::resize generic path: INTER_RESIZE_COEF_BITS = 11)."""
    scale = 1.0 / (dst / src)                       # scale_x = 1./inv_scale_x, doubles
    idx = np.empty(dst, dtype=np.int32)
    a0 = np.empty(dst, dtype=np.int32)
    a1 = np.empty(dst, dtype=np.int32)
    for d in range(dst):
        f = F32((d + 0.5) * scale - 0.5)
        s = int(math.floor(f))
        f = F32(f - F32(s))
        if s < 0:
            s, f = 0, F32(0)
        if s >= src - 1:
            s, f = src - 1, F32(0)
        idx[d] = s
        # saturate_cast<short>(float * 2048) == cvRound (round-half-even)
        a0[d] = int(np.rint(F32(F32(1.0) - f) * F32(2048)))
        a1[d] = int(np.rint(f * F32(2048)))
    return idx, a0, a1


def resize_linear_u8(img: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(img, (dw, dh), interpolation=INTER_LINEAR) for uint8, generic
    fixed-point path: horizontal pass in int32 (x2048), vertical pass
    ``(((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2``."""
    sh, sw = img.shape[:2]
    xi, xa0, xa1 = _resize_coeffs(dw, sw)
    yi, yb0, yb1 = _resize_coeffs(dh, sh)
    src = img.astype(np.int32)
    x1 = np.minimum(xi + 1, sw - 1)
    hrow = src[:, xi, :] * xa0[None, :, None] + src[:, x1, :] * xa1[None, :, None]     # (sh, dw, c)
    y1 = np.minimum(yi + 1, sh - 1)
    r0 = hrow[yi] >> 4
    r1 = hrow[y1] >> 4
    out = (((yb0[:, None, None] * r0) >> 16) + ((yb1[:, None, None] * r1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(img: np.ndarray, new_h=640, new_w=640, auto=False, stride=32):
    """Returns (padded uint8 image, (uw, uh, top, left))."""
    h, w = img.shape[:2]
    uw, uh, top, bottom, left, right = letterbox_params(h, w, new_h, new_w, auto, stride)
    if (w, h) != (uw, uh):
        img = resize_linear_u8(img, uw, uh)
    out = np.full((uh + top + bottom, uw + left + right, 3), 114, dtype=np.uint8)
    out[top:top + uh, left:left + uw] = img
    return out, (uw, uh, top, left)


def preprocess(frame_bgr: np.ndarray, new_h=640, new_w=640, auto=False):
    """BGR uint8 HWC -> letterbox(114) -> RGB -> float32 /255, NHWC.  ``auto=True`` is what
    ``predict`` does for a ``.pt`` model (minimal rectangle: pads taken mod 32, App. B.1)."""
    lb, _ = letterbox(frame_bgr, new_h, new_w, auto=auto)
    return (lb[..., ::-1].astype(F32) / F32(255)).astype(F32)


# ---------------------------------------------------------------------------
# B.2 decode
# ---------------------------------------------------------------------------
def decode(head_maps, nc=80, reg_max=16, strides=(8, 16, 32)):
    """Detect maps -> ``pred[(4+nc), A]`` (cx,cy,w,h in input pixels; sigmoid class scores).
    Anchors: levels P3,P4,P5 concatenated, row-major (y outer, x inner), centres (x+.5,y+.5)."""
    cols = []
    proj = np.arange(reg_max, dtype=F32)
    for m, s in zip(head_maps, strides):
        h, w, _ = m.shape
        box = m[..., :4 * reg_max].reshape(h * w, 4, reg_max).astype(F32)
        e = np.exp(box - box.max(axis=2, keepdims=True), dtype=F32)
        p = (e / e.sum(axis=2, keepdims=True, dtype=F32)).astype(F32)
        dist = (p * proj[None, None, :]).sum(axis=2, dtype=F32)          # (A_i, 4) l,t,r,b
        yy, xx = np.mgrid[0:h, 0:w]
        ax = (xx.reshape(-1).astype(F32) + F32(0.5))
        ay = (yy.reshape(-1).astype(F32) + F32(0.5))
        x1, y1 = ax - dist[:, 0], ay - dist[:, 1]
        x2, y2 = ax + dist[:, 2], ay + dist[:, 3]
        cx, cy = (x1 + x2) / F32(2), (y1 + y2) / F32(2)
        bw, bh = x2 - x1, y2 - y1
        xywh = np.stack([cx, cy, bw, bh], axis=0) * F32(s)
        cls = sigmoid(m[..., 4 * reg_max:].reshape(h * w, nc)).T
        cols.append(np.concatenate([xywh.astype(F32), cls], axis=0))
    return np.concatenate(cols, axis=1).astype(F32)


# ---------------------------------------------------------------------------
# B.3 non_max_suppression  (multi_label=False, max_nms=30000, max_wh=7680)
# ---------------------------------------------------------------------------
def nms_indices(boxes: np.ndarray, scores: np.ndarray, iou_thres: float) -> np.ndarray:
    """torchvision.ops.nms: visit in stable descending-score order; a kept box
    suppresses every later box with ``inter/(area_i+area_j-inter) > iou_thres``
    (strict, no eps, float32 arithmetic, threshold compared as a double like the
    CPU kernel).  Returns kept indices in descending-score order."""
    n = boxes.shape[0]
    order = np.argsort(-scores.astype(F32), kind="stable")
    b = boxes.astype(F32)
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(F32)
    dead = np.zeros(n, dtype=bool)
    keep = []
    thr = float(iou_thres)
    for pos in range(n):
        i = order[pos]
        if dead[i]:
            continue
        keep.append(i)
        rest = order[pos + 1:]
        xx1 = np.maximum(b[i, 0], b[rest, 0])
        yy1 = np.maximum(b[i, 1], b[rest, 1])
        xx2 = np.minimum(b[i, 2], b[rest, 2])
        yy2 = np.minimum(b[i, 3], b[rest, 3])
        iw = np.maximum(F32(0), xx2 - xx1)
        ih = np.maximum(F32(0), yy2 - yy1)
        inter = (iw * ih).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = (inter / ((area[i] + area[rest]) - inter)).astype(F32)
        dead[rest[ovr.astype(np.float64) > thr]] = True
    return np.asarray(keep, dtype=np.int64)


def non_max_suppression(pred, conf_thres=0.35, iou_thres=0.45, classes=None, agnostic=False,
                        max_det=100, nc=80, max_nms=30000, max_wh=7680):
    """pred ``(4+nc, A)`` -> (dets ``(n,6)`` = x1,y1,x2,y2,conf,cls ; anchor index of each det)."""
    cls = pred[4:4 + nc]
    xc = cls.max(axis=0) > F32(conf_thres)                         # strict, float32
    cand = np.nonzero(xc)[0]                                       # anchor order
    p = pred[:, cand].T.astype(F32)                                # (n, 4+nc)
    xy, wh = p[:, 0:2], p[:, 2:4]
    half = wh / F32(2)
    box = np.concatenate([xy - half, xy + half], axis=1).astype(F32)
    j = np.argmax(p[:, 4:4 + nc], axis=1) if len(cand) else np.zeros(0, dtype=np.int64)   # first max
    conf = p[np.arange(len(cand)), 4 + j] if len(cand) else np.zeros(0, dtype=F32)
    ok = conf > F32(conf_thres)
    if classes is not None:
        ok &= np.isin(j, np.asarray(classes))
    box, conf, j, cand = box[ok], conf[ok], j[ok], cand[ok]
    if box.shape[0] > max_nms:
        top = np.argsort(-conf, kind="stable")[:max_nms]
        box, conf, j, cand = box[top], conf[top], j[top], cand[top]
    off = (j.astype(F32) * F32(0 if agnostic else max_wh))[:, None]
    keep = nms_indices((box + off).astype(F32), conf, iou_thres)[:max_det]
    dets = np.concatenate([box[keep], conf[keep, None], j[keep, None].astype(F32)], axis=1).astype(F32)
    return dets.reshape(-1, 6), cand[keep]


# ---------------------------------------------------------------------------
# B.4 rescale
# ---------------------------------------------------------------------------
def scale_boxes(boxes, in_h, in_w, orig_h, orig_w):
    """Undo the letterbox: subtract pad, divide by gain, clip to the original frame."""
    gain = min(in_h / orig_h, in_w / orig_w)
    pad_x = _round_half_even((in_w - orig_w * gain) / 2 - 0.1)
    pad_y = _round_half_even((in_h - orig_h * gain) / 2 - 0.1)
    b = boxes.astype(F32).copy()
    b[:, [0, 2]] -= F32(pad_x)
    b[:, [1, 3]] -= F32(pad_y)
    b[:, :4] /= F32(gain)
    b[:, [0, 2]] = np.clip(b[:, [0, 2]], F32(0), F32(orig_w))
    b[:, [1, 3]] = np.clip(b[:, [1, 3]], F32(0), F32(orig_h))
    return b


# ---------------------------------------------------------------------------
# Detector.detect end to end  (detector.py:98-129)
# ---------------------------------------------------------------------------
def detect(frame_bgr, weights, scale="s", input_size=(640, 640), confidence=0.35, iou=0.45,
           classes=None, max_det=100, agnostic_nms=False, nc=80, return_intermediate=False, rect=False):
    """Returns (xyxy (N,4) f32, confidence (N,) f32, class_id (N,) i32) like ``Detector._parse``.
    ``rect=True``: the minimal-rectangle letterbox ``predict`` applies to ``.pt`` models (1080p -> 384x640);
    the network and ``scale_boxes`` then see the rectangle's shape."""
    in_w, in_h = input_size[0], input_size[0]          # only input_size[0] is used (detector.py:102)
    x = preprocess(frame_bgr, in_h, in_w, auto=rect)
    in_h, in_w = x.shape[:2]
    heads = forward(x, weights, scale, nc)
    pred = decode(heads, nc)
    dets, anchors = non_max_suppression(pred, confidence, iou, classes, agnostic_nms, max_det, nc)
    xyxy = scale_boxes(dets[:, :4], in_h, in_w, frame_bgr.shape[0], frame_bgr.shape[1]) if len(dets) else np.empty((0, 4), F32)
    out = (xyxy.astype(F32), dets[:, 4].astype(F32), dets[:, 5].astype(np.int32))
    if return_intermediate:
        return out, {"heads": heads, "pred": pred, "anchors": anchors}
    return out
