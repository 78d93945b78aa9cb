"""CPU: self-checks of the detector oracle (PARITY UNPINNED -- the reference holds no
fixture at this boundary; see oracle/yolo_oracle.py header).  Known answers from
SURVEY.md Appendix A + independent torch-CPU arithmetic."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import yolo_oracle as Y


def test_known_answer_param_and_flop_counts():
    assert [Y.param_count(s) for s in "nsm"] == [3151904, 11156544, 25886080]
    assert [Y.conv_flops(s) for s in "nsm"] == [8742912000, 28601548800, 78935654400]
    assert Y.conv_flops("m", 1280, 1280) == 315742617600
    assert [len(Y.fused_convs(s)) for s in "nsm"] == [63, 63, 83]


def test_spec_matches_package_table(pkg):
    for s in "nsmlx":
        assert sorted(tuple(c) for c in pkg.yolo_spec.conv_table(s)) == sorted(Y.fused_convs(s))


@pytest.mark.parametrize("k,stride,cin,cout,h,w", [(3, 1, 8, 16, 13, 17), (3, 2, 5, 7, 16, 16), (3, 2, 3, 8, 15, 21), (1, 1, 24, 8, 9, 9)])
def test_conv_matches_torch(k, stride, cin, cout, h, w):
    rng = np.random.default_rng(k * 100 + cin)
    x = rng.normal(size=(h, w, cin)).astype(np.float32)
    wt = rng.normal(size=(cout, k, k, cin)).astype(np.float32)
    b = rng.normal(size=cout).astype(np.float32)
    got = Y.conv2d_nhwc(x, wt, b, stride, act=1)
    ref = F.silu(F.conv2d(torch.from_numpy(x.transpose(2, 0, 1))[None], torch.from_numpy(wt.transpose(0, 3, 1, 2).copy()),
                          torch.from_numpy(b), stride=stride, padding=k // 2))[0].permute(1, 2, 0).numpy()
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)


def test_maxpool_chain_matches_torch():
    x = np.random.default_rng(1).normal(size=(20, 20, 6)).astype(np.float32)
    t = torch.from_numpy(x.transpose(2, 0, 1))[None]
    a = Y.maxpool5(x)
    b = Y.maxpool5(a)
    c = Y.maxpool5(b)
    ta = F.max_pool2d(t, 5, 1, 2); tb = F.max_pool2d(ta, 5, 1, 2); tc = F.max_pool2d(tb, 5, 1, 2)
    for got, ref in ((a, ta), (b, tb), (c, tc)):
        assert np.array_equal(got, ref[0].permute(1, 2, 0).numpy())
    # chained 5x5 pools == single 9x9 / 13x13 windows (what the HIP SPPF kernel computes)
    assert np.array_equal(b, F.max_pool2d(t, 9, 1, 4)[0].permute(1, 2, 0).numpy())
    assert np.array_equal(c, F.max_pool2d(t, 13, 1, 6)[0].permute(1, 2, 0).numpy())


def test_forward_matches_independent_torch_graph(pkg):
    """Oracle (NumPy NHWC im2col) vs weights.torch_forward (torch NCHW F.conv2d): two
    independently written graphs must agree -- catches wiring errors in either."""
    w = pkg.weights.synthetic("n", calibrate=None, seed=3)
    x = np.random.default_rng(0).uniform(size=(96, 128, 3)).astype(np.float32)
    heads = Y.forward(x, w, "n")
    with torch.no_grad():
        th = pkg.weights.torch_forward(torch.from_numpy(x.transpose(2, 0, 1).copy())[None], w, "n")
    for a, b in zip(heads, th):
        ref = b[0].permute(1, 2, 0).numpy()
        assert a.shape == ref.shape
        np.testing.assert_allclose(a, ref, rtol=1e-3, atol=1e-4)


def test_decode_known_answers():
    nc, rm = 80, 16
    m = np.zeros((2, 3, 4 * rm + nc), np.float32)     # uniform DFL -> expectation 7.5 bins each side
    m[..., 4 * rm:] = -20.0
    m[1, 2, 4 * rm + 5] = 3.0
    pred = Y.decode([m], nc, rm, strides=(8,))
    assert pred.shape == (84, 6)
    np.testing.assert_allclose(pred[2], 15 * 8, rtol=1e-6)      # w = (7.5+7.5)*stride
    np.testing.assert_allclose(pred[0], (np.arange(6) % 3 + 0.5) * 8, rtol=1e-6)
    np.testing.assert_allclose(pred[1], (np.arange(6) // 3 + 0.5) * 8, rtol=1e-6)
    assert abs(pred[4 + 5, 5] - 1 / (1 + np.exp(-3.0))) < 1e-6
    # one-hot DFL: all mass on bin k -> distance k
    m2 = np.full((1, 1, 4 * rm + nc), -30.0, np.float32)
    for side, k in enumerate((2, 5, 9, 14)):
        m2[0, 0, side * rm + k] = 30.0
    p = Y.decode([m2], nc, rm, strides=(16,))
    np.testing.assert_allclose(p[:4, 0], [(0.5 - 2 + 0.5 + 9) / 2 * 16, (0.5 - 5 + 0.5 + 14) / 2 * 16, 11 * 16, 19 * 16], rtol=1e-5)


def brute_nms(boxes, scores, thr):
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            x1 = max(boxes[i, 0], boxes[j, 0]); y1 = max(boxes[i, 1], boxes[j, 1])
            x2 = min(boxes[i, 2], boxes[j, 2]); y2 = min(boxes[i, 3], boxes[j, 3])
            inter = np.float32(max(np.float32(0), x2 - x1)) * np.float32(max(np.float32(0), y2 - y1))
            ai = (boxes[i, 2] - boxes[i, 0]) * (boxes[i, 3] - boxes[i, 1])
            aj = (boxes[j, 2] - boxes[j, 0]) * (boxes[j, 3] - boxes[j, 1])
            if float(np.float32(inter / np.float32(np.float32(aj + ai) - inter))) > thr:
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


def test_nms_matches_bruteforce_and_torch_sort_semantics():
    rng = np.random.default_rng(2)
    c = rng.uniform(50, 500, size=(300, 2)).astype(np.float32)
    wh = rng.uniform(20, 120, size=(300, 2)).astype(np.float32)
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    scores = rng.uniform(0.3, 1.0, 300).astype(np.float32)
    scores[10:20] = scores[5]                                   # ties -> stable order matters
    got = Y.nms_indices(boxes, scores, 0.45).tolist()
    assert got == brute_nms(boxes, scores, 0.45)


def test_planted_clusters_survivors(pkg):
    pred, truth = pkg.synth.planted_pred()
    dets, anchors = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 300)
    assert sorted(anchors.tolist()) == sorted(truth.tolist())
    dets2, anchors2 = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 10)
    assert len(dets2) == 10 and anchors2.tolist() == anchors[:10].tolist()    # keep[:max_det]
    assert np.all(np.diff(dets[:, 4]) <= 0)                                    # descending score


def test_nms_class_filter_and_offsets(pkg):
    pred = np.zeros((84, 4), np.float32)
    pred[:4] = np.array([[100, 100, 50, 50]] * 4, np.float32).T               # four identical boxes
    pred[4 + 0, 0] = 0.9; pred[4 + 0, 1] = 0.8                                # same class: second suppressed
    pred[4 + 3, 2] = 0.7                                                      # other class: survives (7680 offset)
    pred[4 + 9, 3] = 0.6
    d, a = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 100)
    assert a.tolist() == [0, 2, 3] and d[:, 5].tolist() == [0, 3, 9]
    d, a = Y.non_max_suppression(pred, 0.35, 0.45, None, True, 100)           # agnostic: one survivor
    assert a.tolist() == [0]
    d, a = Y.non_max_suppression(pred, 0.35, 0.45, [3, 9], False, 100)        # classes filter
    assert a.tolist() == [2, 3]
    d, a = Y.non_max_suppression(pred, 0.95, 0.45, None, False, 100)          # nothing passes
    assert d.shape == (0, 6)


def test_letterbox_geometry_and_identity():
    assert Y.letterbox_params(640, 640) == (640, 640, 0, 0, 0, 0)
    assert Y.letterbox_params(1080, 1920) == (640, 360, 140, 140, 0, 0)
    assert Y.letterbox_params(1080, 1920, auto=True) == (640, 360, 12, 12, 0, 0)   # -> 384x640
    assert Y.letterbox_params(480, 640) == (640, 480, 80, 80, 0, 0)
    img = np.random.default_rng(0).integers(0, 256, size=(640, 640, 3), dtype=np.uint8)
    out, _ = Y.letterbox(img)
    assert np.array_equal(out, img)
    img = np.random.default_rng(0).integers(0, 256, size=(90, 160, 3), dtype=np.uint8)
    out, (uw, uh, top, left) = Y.letterbox(img, 64, 64)
    assert out.shape == (64, 64, 3) and (uw, uh, top, left) == (64, 36, 14, 0)
    assert np.all(out[:14] == 114) and np.all(out[50:] == 114)


def test_resize_fixed_point_properties():
    flat = np.full((37, 53, 3), 201, np.uint8)
    assert np.all(Y.resize_linear_u8(flat, 80, 64) == 201)                   # weights sum to 2048 exactly
    img = np.random.default_rng(3).integers(0, 256, size=(48, 64, 3), dtype=np.uint8)
    up = Y.resize_linear_u8(img, 128, 96)
    ref = F.interpolate(torch.from_numpy(img.transpose(2, 0, 1)[None].astype(np.float32)), size=(96, 128), mode="bilinear", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(up.astype(np.float32) - ref).max() <= 1.0                   # same half-pixel convention, +-1 LSB fixed point
    assert np.array_equal(Y.resize_linear_u8(img, 64, 48), img)              # scale 1 is exact


def test_scale_boxes_roundtrip():
    b = np.array([[0, 140, 640, 500], [-5, 100, 700, 520]], np.float32)
    out = Y.scale_boxes(b, 640, 640, 1080, 1920)
    np.testing.assert_allclose(out[0], [0, 0, 1920, 1080], rtol=1e-6)
    np.testing.assert_allclose(out[1], [0, 0, 1920, 1080], rtol=1e-6)         # clipped


def test_weight_file_roundtrip(pkg, tmp_path):
    w = pkg.weights.synthetic("n", calibrate=None, seed=1)
    p = str(tmp_path / "n.rtw")
    pkg.weights.save(p, w, "n")
    w2, scale, nc, rm = pkg.weights.load(p)
    assert (scale, nc, rm) == ("n", 80, 16) and set(w2) == set(w)
    for k in w:
        assert np.array_equal(w[k][0], w2[k][0]) and np.array_equal(w[k][1], w2[k][1])


def test_bn_folding_matches_torch(pkg):
    torch.manual_seed(0)
    conv = torch.nn.Conv2d(6, 10, 3, 1, 1, bias=False)
    bn = torch.nn.BatchNorm2d(10, eps=1e-3)
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    bn.eval()
    x = torch.randn(1, 6, 9, 9)
    with torch.no_grad():
        ref = bn(conv(x))
    w, b = pkg.weights.fold_bn(conv.weight.detach().numpy(), bn.weight.detach().numpy(), bn.bias.detach().numpy(),
                               bn.running_mean.numpy(), bn.running_var.numpy())
    got = Y.conv2d_nhwc(x[0].permute(1, 2, 0).numpy(), w, b, 1, act=0)
    np.testing.assert_allclose(got, ref[0].permute(1, 2, 0).numpy(), rtol=1e-4, atol=1e-5)


def test_byte_scale_by_multiplication_matches_division_in_fp16():
    """The fused stem (conv.hip: stem_fused) turns a byte into fp16 as half(float(c) * (1/255)); the reference
    preprocess divides (detector.py's letterboxed tensor / 255 -> .half()).  The two differ in fp32 for about half
    of the byte values but round to the same fp16 for all 256 of them."""
    c = np.arange(256, dtype=np.float32)
    by_div = (c / np.float32(255.0)).astype(np.float16)
    by_mul = (c * np.float32(1.0 / 255.0)).astype(np.float16)
    assert np.array_equal(by_div.view(np.uint16), by_mul.view(np.uint16))
    frame = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    ref = Y.preprocess(frame, 16, 16).astype(np.float16)                # (3, 16, 16) or (16, 16, 3): every byte value once per channel
    assert np.array_equal(np.sort(ref.reshape(-1).view(np.uint16)), np.sort(np.repeat(by_mul, 3).view(np.uint16)))
