"""CPU: `.pt` -> RTMODTW1 conversion.  No Ultralytics here, so the test
builds a structurally identical torch model under fake ``ultralytics.nn.*`` module paths,
saves it with torch.save (so the pickle names classes that do NOT exist when it is read back),
reads it with the restricted unpickler and checks (1) every tensor, (2) that BN folding +
layout conversion reproduce the torch model's own forward pass through the oracle."""
import sys
import types

import numpy as np
import torch
import torch.nn as nn

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


def test_convert_pt_without_ultralytics(pkg, tmp_path):
    fake, DetectionModel = build_fake_ultralytics()
    torch.manual_seed(0)
    model = DetectionModel("n", 80).eval()
    for m in model.modules():                      # non-trivial BN statistics
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
    ref_sd = {k: v.numpy() for k, v in model.state_dict().items()}
    pt = str(tmp_path / "yolov8n_fake.pt")
    sys.modules.update(fake)
    try:
        torch.save({"model": model.half(), "epoch": -1, "train_args": {"imgsz": 640}}, pt)     # Ultralytics stores the model in fp16
    finally:
        for k in fake:
            sys.modules.pop(k, None)
    assert "ultralytics" not in sys.modules
    sd = pkg.weights.read_pt(pt)                   # restricted: the fake classes are gone, nothing of theirs can run
    assert set(sd) == set(ref_sd)
    for k in ref_sd:
        assert sd[k].shape == ref_sd[k].shape, k
        if ref_sd[k].dtype.kind == "f":
            np.testing.assert_allclose(sd[k].astype(np.float32), ref_sd[k].astype(np.float16).astype(np.float32), rtol=0, atol=0, err_msg=k)
    assert pkg.weights.infer_scale(sd) == "n"
    out = str(tmp_path / "yolov8n.rtw")
    assert pkg.weights.convert_pt(pt, out) == ("n", 80)
    w, scale, nc, _ = pkg.weights.load(out)
    # folded weights reproduce the torch model's own Conv+BN+SiLU on a stem-to-layer-2 slice
    model = model.float()
    x = np.random.default_rng(0).uniform(size=(64, 64, 3)).astype(np.float32)
    with torch.no_grad():
        t = torch.from_numpy(x.transpose(2, 0, 1).copy())[None]
        ref = model.model[2](model.model[1](model.model[0](t)))[0].permute(1, 2, 0).numpy()
    taps = {}
    try:
        Y.forward(x, w, "n", taps=taps)
    except Exception:
        pass
    got = taps[2]
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2)     # weights went through fp16 twice


def test_read_pt_rejects_non_checkpoints(pkg, tmp_path):
    import pytest
    p = tmp_path / "x.pt"
    p.write_bytes(b"not a zip")
    with pytest.raises(Exception):
        pkg.weights.read_pt(str(p))
