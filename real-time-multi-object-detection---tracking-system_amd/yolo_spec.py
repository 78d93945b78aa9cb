"""YOLOv8 detection-network description used by the weight tools.

Mirrors what ``ultralytics`` builds for ``yolov8{n,s,m,l,x}.yaml`` (the model family
``src/detection/detector.py:82-90`` loads); the authoritative graph that actually
runs is built natively in ``csrc/yolo_graph.cpp`` -- this table only names the
fused convolutions (order, shapes) so weight files can be written and converted.
"""
from __future__ import annotations

import math
from typing import NamedTuple

SCALES = {"n": (0.33, 0.25, 1024), "s": (0.33, 0.50, 1024), "m": (0.67, 0.75, 768),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.25, 512)}
SCALE_ID = {"n": 0, "s": 1, "m": 2, "l": 3, "x": 4}


class ConvSpec(NamedTuple):
    name: str      # e.g. "2.m.0.cv1", "22.cv3.1.2"
    cin: int
    cout: int
    k: int
    stride: int
    act: int       # 1 = SiLU, 0 = linear (last conv of each Detect branch)


def width(c: int, scale: str) -> int:
    _, w, mx = SCALES[scale]
    return int(math.ceil(min(c, mx) * w / 8) * 8)


def repeats(n: int, scale: str) -> int:
    return max(round(n * SCALES[scale][0]), 1)


def conv_table(scale: str = "s", nc: int = 80, reg_max: int = 16) -> list[ConvSpec]:
    W = lambda c: width(c, scale)  # noqa: E731
    R = lambda n: repeats(n, scale)  # noqa: E731
    t: list[ConvSpec] = []

    def conv(i, ci, co, s):
        t.append(ConvSpec(f"{i}", ci, co, 3, s, 1))

    def c2f(i, ci, co, n):
        h = co // 2
        t.append(ConvSpec(f"{i}.cv1", ci, co, 1, 1, 1))
        for j in range(n):
            t.append(ConvSpec(f"{i}.m.{j}.cv1", h, h, 3, 1, 1))
            t.append(ConvSpec(f"{i}.m.{j}.cv2", h, h, 3, 1, 1))
        t.append(ConvSpec(f"{i}.cv2", (2 + n) * h, co, 1, 1, 1))

    p1, p2, p3, p4, p5 = W(64), W(128), W(256), W(512), W(1024)
    conv(0, 3, p1, 2)
    conv(1, p1, p2, 2)
    c2f(2, p2, p2, R(3))
    conv(3, p2, p3, 2)
    c2f(4, p3, p3, R(6))
    conv(5, p3, p4, 2)
    c2f(6, p4, p4, R(6))
    conv(7, p4, p5, 2)
    c2f(8, p5, p5, R(3))
    t.append(ConvSpec("9.cv1", p5, p5 // 2, 1, 1, 1))
    t.append(ConvSpec("9.cv2", 2 * p5, p5, 1, 1, 1))
    c2f(12, p5 + p4, p4, R(3))
    c2f(15, p4 + p3, p3, R(3))
    conv(16, p3, p3, 2)
    c2f(18, p3 + p4, p4, R(3))
    conv(19, p4, p4, 2)
    c2f(21, p4 + p5, p5, R(3))
    cbox = max(16, p3 // 4, 4 * reg_max)
    ccls = max(p3, min(nc, 100))
    for lvl, ci in enumerate((p3, p4, p5)):
        t.append(ConvSpec(f"22.cv2.{lvl}.0", ci, cbox, 3, 1, 1))
        t.append(ConvSpec(f"22.cv2.{lvl}.1", cbox, cbox, 3, 1, 1))
        t.append(ConvSpec(f"22.cv2.{lvl}.2", cbox, 4 * reg_max, 1, 1, 0))
        t.append(ConvSpec(f"22.cv3.{lvl}.0", ci, ccls, 3, 1, 1))
        t.append(ConvSpec(f"22.cv3.{lvl}.1", ccls, ccls, 3, 1, 1))
        t.append(ConvSpec(f"22.cv3.{lvl}.2", ccls, nc, 1, 1, 0))
    return t


COCO_NAMES = [
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light",
    "fire hydrant", "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow",
    "elephant", "bear", "zebra", "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee",
    "skis", "snowboard", "sports ball", "kite", "baseball bat", "baseball glove", "skateboard", "surfboard",
    "tennis racket", "bottle", "wine glass", "cup", "fork", "knife", "spoon", "bowl", "banana", "apple",
    "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut", "cake", "chair", "couch",
    "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote", "keyboard", "cell phone",
    "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors", "teddy bear",
    "hair drier", "toothbrush",
]
