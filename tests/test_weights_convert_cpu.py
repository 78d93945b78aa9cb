"""CPU: `.pt` -> RTMODTW1 conversion.  No Ultralytics here, so the test
builds a structurally identical torch model under fake ``ultralytics.nn.*`` module paths,
saves it with torch.save (so the pickle names classes that do NOT exist when it is read back),
reads it with the restricted unpickler and checks (1) every tensor, (2) that BN folding +
layout conversion reproduce the torch model's own forward pass through the oracle."""
import sys

import numpy as np
import torch
import torch.nn as nn

from oracle import yolo_oracle as Y


from fake_ultralytics import build_fake_ultralytics, make_checkpoint


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


def test_converted_checkpoint_reproduces_the_unfolded_model_end_to_end(pkg, tmp_path):
    """`.pt` -> restricted read -> BN fold -> NHWC fp16 -> RTMODTW1, then the WHOLE net: the oracle run on the converted file
    against torch running the checkpoint's own unfolded Conv + BatchNorm + SiLU graph -- all three Detect maps, and the
    detections after decode + NMS decision by decision (the only differences: folded weights are rounded to fp16 once more)."""
    import nms_audit as NA
    from fake_ultralytics import forward_heads
    model, rtw, _ = make_checkpoint(pkg, tmp_path)
    w, scale, nc, _ = pkg.weights.load(rtw)
    frame = pkg.synth.frames(1, 320, 320, seed=77)[0]
    x = Y.preprocess(frame, 320, 320)
    with torch.no_grad():
        ref = [h[0].permute(1, 2, 0).numpy() for h in forward_heads(model, torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None])]
    got = Y.forward(x, w, scale)
    for lvl in range(3):
        e = np.abs(got[lvl] - ref[lvl])
        assert np.isfinite(ref[lvl]).all() and ref[lvl][..., :64].std() > 0.5          # a live net, not one that decayed to its biases
        assert np.percentile(e, 99) < 0.02 and e.max() < 0.1, (lvl, float(np.percentile(e, 99)), float(e.max()))
    po, pr = Y.decode(got), Y.decode(ref)
    n_cand = int((pr[4:].max(0) > 0.35).sum())
    assert 10 < n_cand < 1500, n_cand
    res = NA.audit(po, pr, 0.35, 0.45, None, score_tol=0.001, iou_tol=0.002)
    assert not res["hard"], NA.describe(res)
