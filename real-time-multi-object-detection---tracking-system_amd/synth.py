"""Deterministic synthetic workloads.

There is no network, dataset or checkpoint on the build or GPU boxes, so every
test and benchmark input is generated here from fixed seeds:

* :func:`box_sequence`  -- constant-velocity boxes for the tracker (BASELINE
  configs 3 and 5: 200 boxes on 640x640, 500 boxes on 1280x1280).
* :func:`frames`        -- uniform-random BGR uint8 video frames.
* :func:`planted_pred`  -- a pre-NMS ``(84, A)`` prediction tensor with planted
  box clusters whose survivors are known by construction (BASELINE config 2).
"""
from __future__ import annotations

import numpy as np


def box_sequence(n_boxes: int = 200, canvas: int = 640, n_frames: int = 120, seed: int = 1234):
    """Returns ``(xyxy[n_frames, n, 4] f32, conf[n] f32, cls[n] i32)``.

    wh ~ U(40,120), centres ~ U(60, canvas-60), velocity ~ U(-1.5,1.5) px/frame,
    per-frame centre jitter ~ N(0,0.3), conf ~ U(0.36,0.99) fixed per object,
    cls ~ U{0..79}; everything float32."""
    rng = np.random.default_rng(seed)
    wh = rng.uniform(40.0, 120.0, size=(n_boxes, 2))
    c0 = rng.uniform(60.0, canvas - 60.0, size=(n_boxes, 2))
    vel = rng.uniform(-1.5, 1.5, size=(n_boxes, 2))
    conf = rng.uniform(0.36, 0.99, size=n_boxes).astype(np.float32)
    cls = rng.integers(0, 80, size=n_boxes).astype(np.int32)
    out = np.empty((n_frames, n_boxes, 4), dtype=np.float32)
    for f in range(n_frames):
        c = c0 + vel * f + rng.normal(0.0, 0.3, size=(n_boxes, 2))
        out[f, :, 0:2] = (c - wh / 2).astype(np.float32)
        out[f, :, 2:4] = (c + wh / 2).astype(np.float32)
    return out, conf, cls


def frames(n: int, height: int = 640, width: int = 640, seed: int = 1234) -> np.ndarray:
    """``n`` BGR uint8 frames ``(n, H, W, 3)``; stream ``s`` uses ``seed = 1234 + s``."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, height, width, 3), dtype=np.uint8)


def structured_frames(n: int, height: int = 640, width: int = 640, seed: int = 7) -> np.ndarray:
    """Smooth blobs on a gradient -- exercises the bilinear letterbox far better
    than white noise (neighbouring pixels are correlated, rounding matters)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    out = np.empty((n, height, width, 3), dtype=np.uint8)
    for i in range(n):
        img = np.zeros((height, width, 3), dtype=np.float32)
        for c in range(3):
            img[..., c] = 40 + 60 * (xx / width) + 50 * (yy / height) * (c + 1) / 3
        for _ in range(12):
            cx, cy = rng.uniform(0, width), rng.uniform(0, height)
            r = rng.uniform(10, 90)
            col = rng.uniform(-120, 160, size=3)
            w = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))
            img += w[..., None] * col
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def planted_pred(n_anchors: int = 8400, n_classes: int = 80, n_clusters: int = 40,
                 per_cluster: int = 6, canvas: int = 640, seed: int = 99):
    """Pre-NMS tensor ``pred[(4+nc), A]`` float32 in the layout the Detect head
    emits (rows: cx, cy, w, h, then per-class scores) with ``n_clusters`` planted
    objects, each hit by ``per_cluster`` jittered near-duplicates with distinct
    scores.  Returns ``(pred, truth)`` where ``truth`` lists, per cluster, the
    anchor index of the highest-scoring member (the expected NMS survivor when
    clusters do not overlap each other above the IoU threshold)."""
    rng = np.random.default_rng(seed)
    pred = np.zeros((4 + n_classes, n_anchors), dtype=np.float32)
    pred[4:] = rng.uniform(0.0, 0.2, size=(n_classes, n_anchors)).astype(np.float32)
    pred[0] = rng.uniform(0, canvas, n_anchors)
    pred[1] = rng.uniform(0, canvas, n_anchors)
    pred[2] = rng.uniform(8, 64, n_anchors)
    pred[3] = rng.uniform(8, 64, n_anchors)
    slots = rng.permutation(n_anchors)[: n_clusters * per_cluster].reshape(n_clusters, per_cluster)
    grid = int(np.ceil(np.sqrt(n_clusters)))
    cell = canvas / grid
    truth = []
    for k in range(n_clusters):
        gx, gy = k % grid, k // grid
        cx, cy = (gx + 0.5) * cell, (gy + 0.5) * cell
        w, h = rng.uniform(0.35, 0.6, 2) * cell
        c = int(rng.integers(0, n_classes))
        scores = np.sort(rng.uniform(0.5, 0.98, per_cluster))[::-1]
        order = rng.permutation(per_cluster)
        for r, s in zip(order, scores):
            a = slots[k, r]
            jit = rng.normal(0, 0.01 * cell, 4)
            pred[0:4, a] = (cx + jit[0], cy + jit[1], w + jit[2], h + jit[3])
            pred[4 + c, a] = s
        truth.append(int(slots[k, order[0]]))
    return pred, np.asarray(truth, dtype=np.int64)
