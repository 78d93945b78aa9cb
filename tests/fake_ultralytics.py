"""A structurally identical stand-in for an Ultralytics YOLOv8 ``DetectionModel`` (test helper, not a test).

Ultralytics is not installed anywhere this runs, so the `.pt` side of SURVEY 8f rank 2 (``src/detection/detector.py:82-90``:
``YOLO(path)``) is exercised with a torch model whose module tree, parameter names and pickled class paths
(``ultralytics.nn.tasks.DetectionModel``, ``ultralytics.nn.modules.Conv`` ...) are those of the real thing [UPSTREAM, SURVEY
App. A]: Conv = Conv2d(bias=False) + BatchNorm2d(eps=1e-3) + SiLU, C2f, Bottleneck, SPPF, Detect with plain biased Conv2d ends.
``forward_heads`` runs the UNFOLDED graph (BatchNorm as BatchNorm) in torch: the reference a converted, BN-folded weight
file is checked against."""
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import yolo_oracle as Y


def build_fake_ultralytics():
    tasks = types.ModuleType("ultralytics.nn.tasks")
    mods = types.ModuleType("ultralytics.nn.modules")

    class Conv(nn.Module):
        def __init__(self, c1, c2, k=1, s=1):
            super().__init__()
            self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
            self.bn = nn.BatchNorm2d(c2, eps=1e-3)
            self.act = nn.SiLU()

        def forward(self, x):
            return self.act(self.bn(self.conv(x)))

    class Bottleneck(nn.Module):
        def __init__(self, c, shortcut):
            super().__init__()
            self.cv1, self.cv2, self.add = Conv(c, c, 3), Conv(c, c, 3), shortcut

        def forward(self, x):
            y = self.cv2(self.cv1(x))
            return x + y if self.add else y

    class C2f(nn.Module):
        def __init__(self, c1, c2, n, shortcut):
            super().__init__()
            self.c = c2 // 2
            self.cv1, self.cv2 = Conv(c1, 2 * self.c, 1), Conv((2 + n) * self.c, c2, 1)
            self.m = nn.ModuleList(Bottleneck(self.c, shortcut) for _ in range(n))

        def forward(self, x):
            y = list(self.cv1(x).chunk(2, 1))
            y.extend(m(y[-1]) for m in self.m)
            return self.cv2(torch.cat(y, 1))

    class SPPF(nn.Module):
        def __init__(self, c1, c2):
            super().__init__()
            self.cv1, self.cv2 = Conv(c1, c1 // 2, 1), Conv(c1 * 2, c2, 1)
            self.m = nn.MaxPool2d(5, 1, 2)

        def forward(self, x):
            y = [self.cv1(x)]
            y.extend(self.m(y[-1]) for _ in range(3))
            return self.cv2(torch.cat(y, 1))

    class Placeholder(nn.Module):        # Upsample / Concat slots (no parameters)
        pass

    class Detect(nn.Module):
        def __init__(self, nc, ch, c2, c3):
            super().__init__()
            self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 64, 1)) for x in ch)
            self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, nc, 1)) for x in ch)

    class DetectionModel(nn.Module):
        def __init__(self, scale, nc):
            super().__init__()
            arch, head = Y.arch(scale, nc)
            layers = []
            for m in arch:
                if m[0] == "conv":
                    layers.append(Conv(m[2], m[3], m[4], m[5]))
                elif m[0] == "c2f":
                    layers.append(C2f(m[2], m[3], m[4], m[5]))
                elif m[0] == "sppf":
                    layers.append(SPPF(m[2], m[3]))
                else:
                    layers.append(Placeholder())
            layers.append(Detect(nc, head["ch"], head["c2"], head["c3"]))
            self.model = nn.Sequential(*layers)
            self.scale, self.nc = scale, nc
            self.names = {i: str(i) for i in range(nc)}

    for c in (Conv, Bottleneck, C2f, SPPF, Placeholder, Detect):
        c.__module__ = "ultralytics.nn.modules"
        c.__qualname__ = c.__name__
        setattr(mods, c.__name__, c)
    DetectionModel.__module__ = "ultralytics.nn.tasks"
    DetectionModel.__qualname__ = "DetectionModel"
    tasks.DetectionModel = DetectionModel
    return {"ultralytics": types.ModuleType("ultralytics"), "ultralytics.nn": types.ModuleType("ultralytics.nn"),
            "ultralytics.nn.tasks": tasks, "ultralytics.nn.modules": mods}, DetectionModel



def forward_heads(model, x):
    """The three Detect maps ``(B, 64 + nc, H_i, W_i)`` of the unfolded model for ``x`` (B,3,H,W) in [0,1] RGB; the routing
    (Upsample / Concat sources) follows SURVEY App. A, as ``oracle.yolo_oracle.arch`` lists it."""
    mods, _ = Y.arch(model.scale, model.nc)
    saved, cur = [], x
    for i, m in enumerate(mods):
        if m[0] in ("conv", "c2f", "sppf"):
            cur = model.model[i](cur)
        elif m[0] == "up":
            cur = F.interpolate(cur, scale_factor=2, mode="nearest")
        else:
            a, b = m[1]
            cur = torch.cat([cur if a == -1 else saved[a], saved[b]], 1)
        saved.append(cur)
    det = model.model[len(mods)]
    return [torch.cat([det.cv2[l](saved[s]), det.cv3[l](saved[s])], 1) for l, s in enumerate((15, 18, 21))]


def randomise_and_calibrate(model, x, seed=0, cls_bias=-4.1):
    """Non-trivial BatchNorm statistics (running mean / variance / beta random) and, because a 60-conv SiLU stack with
    torch's default initialisation decays to zero long before the head (fp16 underflow would hide any conversion error),
    every BN gamma set layer by layer so that its pre-activation has about unit deviation on ``x`` (times a random
    factor in [0.8, 1.2] per channel); Detect's last plain convs are scaled the same way, class bias ``cls_bias``."""
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.bias.data.normal_(0, 0.2, generator=g)
            m.running_mean.normal_(0, 0.2, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
            m.weight.data.fill_(1.0)
    hooks = []

    def bn_pre(mod, inp):
        z = (inp[0] - mod.running_mean[None, :, None, None]) / torch.sqrt(mod.running_var[None, :, None, None] + mod.eps)
        jitter = torch.empty(mod.num_features).uniform_(0.8, 1.2, generator=g)
        mod.weight.data = jitter / z.std().clamp_min(1e-12)

    def last_pre(mod, inp):
        y = F.conv2d(inp[0], mod.weight, None)
        box = mod.out_channels == 64
        mod.weight.data = mod.weight.data * ((1.5 if box else 1.0) / y.std().clamp_min(1e-12))
        mod.bias.data = torch.full((mod.out_channels,), 0.0 if box else cls_bias) + torch.empty(mod.out_channels).normal_(0, 0.1, generator=g)

    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            hooks.append(m.register_forward_pre_hook(bn_pre))
        elif isinstance(m, nn.Conv2d) and m.bias is not None:
            hooks.append(m.register_forward_pre_hook(last_pre))
    with torch.no_grad():
        forward_heads(model, x)
    for h in hooks:
        h.remove()


def make_checkpoint(pkg, tmp_path, scale="n", size=320, seed=0):
    """A calibrated fake-Ultralytics checkpoint on disk (fp16 like the real ones) + its converted RTMODTW1 file.
    Returns (unfolded fp32 torch model holding exactly the checkpoint's fp16 values, .rtw path, calibration frame)."""
    fake, DetectionModel = build_fake_ultralytics()
    torch.manual_seed(seed)
    model = DetectionModel(scale, 80).eval()
    frame = pkg.synth.frames(1, size, size, seed=4321)[0]
    x = torch.from_numpy(np.ascontiguousarray(Y.preprocess(frame, size, size).transpose(2, 0, 1)))[None]
    randomise_and_calibrate(model, x, seed=seed, cls_bias=-3.3)
    pt = str(tmp_path / f"yolov8{scale}_fake.pt")
    sys.modules.update(fake)
    try:
        torch.save({"model": model.half(), "epoch": -1}, pt)
    finally:
        for k in fake:
            sys.modules.pop(k, None)
    out = str(tmp_path / f"yolov8{scale}_converted.rtw")
    assert pkg.weights.convert_pt(pt, out) == (scale, 80)
    return model.float(), out, frame
