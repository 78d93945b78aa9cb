"""GPU: letterbox, forward, decode and NMS kernels against the CPU oracle, all through the
C ABI.  Bars (SURVEY.md section 7 "fp16 vs the fp32 CPU reference"):
  * integer work (letterbox fixed-point resize, NMS survivor indices): bit-exact;
  * one conv layer fed the engine's own fp16 inputs: |err| <= 2e-3 * max|ref| + 2e-3
    (fp16 output rounding + fp32 accumulation order);
  * decode on identical head logits: rtol 2e-4;
  * end to end vs the all-fp32 oracle: every confident detection matched with IoU >= 0.99.
"""
import os

import numpy as np
import pytest

from oracle import yolo_oracle as Y

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wdir(tmp_path_factory):
    return tmp_path_factory.mktemp("weights")


def make_detector(pkg, wdir, scale, size, calibrate="noise", **kw):
    path = os.path.join(str(wdir), f"yolov8{scale}_{size}_{calibrate}.rtw")
    if not os.path.exists(path):
        w = pkg.weights.synthetic(scale, input_size=size, calibrate=calibrate)
        pkg.weights.save(path, w, scale)
    w, _, _, _ = pkg.weights.load(path)
    det = pkg.Detector(path, input_size=(size, size), warmup=False, **kw)
    return det, w


# ------------------------------------------------------------------ preprocess

def _tile_ids():
    """ConvTile ids by name, parsed from csrc/kernels.h: the tests name tiles, they do not hard-code a table that a pruned library renumbers"""
    import re
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "real-time-multi-object-detection---tracking-system_amd", "csrc", "kernels.h")).read()
    body = src[src.index("enum ConvTile {"):]
    body = body[:body.index("};")]
    return {m.group(1): int(m.group(2)) for m in re.finditer(r"TILE_(\w+)\s*=\s*(\d+)", body)}


T = _tile_ids()

@pytest.mark.parametrize("h,w,size", [(640, 640, 640), (1080, 1920, 640), (480, 640, 640), (200, 300, 640), (720, 1280, 320), (37, 53, 64)])
def test_letterbox_bit_exact(pkg, h, w, size):
    img = pkg.synth.structured_frames(1, h, w, seed=h + w)[0]
    img[::7, ::5] = pkg.synth.frames(1, h, w, seed=3)[0][::7, ::5]
    got = pkg._ffi.preprocess(img, size, size)
    ref = Y.preprocess(img, size, size).astype(np.float16)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint16), ref.view(np.uint16))


# ------------------------------------------------------------------ NMS on a supplied pred tensor (BASELINE config 2)
def check_nms(pkg, pred, **kw):
    args = dict(conf=0.35, iou=0.45, classes=None, agnostic=False, max_det=100)
    args.update(kw)
    xy, cf, ci, an = pkg._ffi.nms_pred(pred, **args)
    dets, anchors = Y.non_max_suppression(pred, args["conf"], args["iou"], args["classes"], args["agnostic"], args["max_det"], nc=pred.shape[0] - 4)
    assert an.tolist() == anchors.tolist()
    assert np.array_equal(xy.view(np.int32), dets[:, :4].view(np.int32))
    assert np.array_equal(cf.view(np.int32), dets[:, 4].view(np.int32))
    assert ci.tolist() == dets[:, 5].astype(np.int32).tolist()
    return an


@pytest.fixture(params=["1024", "256"], ids=["1024-threads", "256-threads"])
def nms_threads(request, monkeypatch):
    """nms_kernel's two workgroup sizes (1024 = the product's; 256 = rounds 1-2, an A/B hook).  RTMODT_NMS_THREADS is resolved into an NmsPlan once per
    detector (at create) and once per rtmodt_nms_pred call (postprocess.hip: nms_plan_from_options) -- never per launch; check_nms goes through
    rtmodt_nms_pred, so setting the variable before the call is enough."""
    monkeypatch.setenv("RTMODT_NMS_THREADS", request.param)
    return request.param


def test_nms_planted_clusters(pkg, nms_threads):
    pred, truth = pkg.synth.planted_pred()
    an = check_nms(pkg, pred, max_det=300)
    assert sorted(an.tolist()) == sorted(truth.tolist())
    check_nms(pkg, pred, max_det=10)
    check_nms(pkg, pred, classes=[1, 5, 17, 63, 64, 79])
    check_nms(pkg, pred, agnostic=True)
    check_nms(pkg, pred, conf=0.6, iou=0.2)


@pytest.mark.parametrize("n_anchors,frac", [(8400, 0.05), (8400, 0.6), (2100, 1.0), (33600, 0.3), (8400, 1.0), (33600, 1.0)])
def test_nms_dense_random(pkg, nms_threads, n_anchors, frac):
    """Hundreds to ~10k candidates, heavy overlap, duplicated scores (stable order matters);
    the 33600-anchor case drives the > 8192-candidate global rank-sort path."""
    rng = np.random.default_rng(n_anchors + int(frac * 100))
    pred = np.zeros((84, n_anchors), np.float32)
    pred[0] = rng.uniform(0, 640, n_anchors); pred[1] = rng.uniform(0, 640, n_anchors)
    pred[2] = rng.uniform(10, 200, n_anchors); pred[3] = rng.uniform(10, 200, n_anchors)
    hot = rng.uniform(size=n_anchors) < frac
    cls = rng.integers(0, 6, n_anchors)
    sc = np.round(rng.uniform(0.36, 0.99, n_anchors), 2).astype(np.float32)      # many exact ties
    pred[4 + cls[hot], np.nonzero(hot)[0]] = sc[hot]
    check_nms(pkg, pred, max_det=300)
    check_nms(pkg, pred, max_det=100, agnostic=True)


def test_nms_empty_and_single(pkg, nms_threads):
    pred = np.zeros((84, 100), np.float32)
    xy, cf, ci, an = pkg._ffi.nms_pred(pred)
    assert len(an) == 0
    pred[:4, 7] = (50, 60, 20, 30); pred[4 + 3, 7] = 0.9
    check_nms(pkg, pred)


def test_nms_dense_scene_through_the_engine(pkg, wdir, nms_threads):
    """A saturated head (the random-weight net on STRUCTURED frames: thousands of candidates per image, as a dense scene gives a
    trained one) through the detector itself, 8 images per batch: the fetched detections must equal oracle NMS + scale_boxes on
    the engine's own pre-NMS tensor bit for bit -- the 8 192-key LDS sort, the walk over still-alive positions and the
    boxes-in-global-scratch path are what this input exercises (profiles/r03/nms_phases/)."""
    size, B = 640, 8
    det, _ = make_detector(pkg, wdir, "s", size, batch=B, autotune=False, max_det=100)
    frames = list(pkg.synth.structured_frames(B, size, size, seed=1234))
    got = det.detect_batch(frames)
    cands = []
    for i in range(B):
        _, _, pred = det.debug_fetch(i, want_input=False, want_heads=False)
        cands.append(int((pred[4:].max(0) > det.confidence).sum()))
        dets, _ = Y.non_max_suppression(pred, det.confidence, det.iou, det.classes, det.agnostic_nms, 100)
        ref = Y.scale_boxes(dets[:, :4], size, size, size, size) if len(dets) else np.empty((0, 4), np.float32)
        d = got[i]
        assert len(d) == len(dets), (i, cands[-1])
        assert np.array_equal(d.xyxy.view(np.int32), ref.view(np.int32)), i
        assert np.array_equal(d.confidence.view(np.int32), dets[:, 4].astype(np.float32).view(np.int32)), i
        assert d.class_id.tolist() == dets[:, 5].astype(np.int32).tolist(), i
    assert max(cands) > 2048, cands              # the input really is dense


# ------------------------------------------------------------------ forward pass, layer by layer
def fetch_layers(pkg, det, names, img=0):
    """Every fused conv's output as the engine stored it; the first conv of a Bottleneck that runs
    as ONE fused launch has no global image (it lives in LDS) and is skipped -- the oracle then
    recomputes it from the forced input, so the pair is still checked end to end."""
    out = {}
    for n in names:
        try:
            out[n] = det.debug_layer(n, img).astype(np.float32)
        except pkg._ffi.RtmodtError as e:
            assert e.code == pkg._ffi.E_UNSUPPORTED and (n.endswith(".cv1") or "fused into its launch" in str(e) or "runs as one launch" in str(e)), (n, str(e))
    return out


def layer_check(pkg, det, w, frame, scale, size):
    d = det.detect(frame)
    inp, heads, pred = det.debug_fetch(0)
    ref_in = Y.preprocess(frame, size, size).astype(np.float16)
    assert np.array_equal(inp.view(np.uint16), ref_in.view(np.uint16))
    names = [c.name for c in pkg.weights.spec(scale)]
    gpu = fetch_layers(pkg, det, names)
    taps = {}
    Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
    worst = ("", 0.0)
    for n in gpu:
        ref, got = taps[n], gpu[n]
        assert ref.shape == got.shape, n
        tol = 2e-3 * np.abs(ref).max() + 2e-3
        err = float(np.abs(ref - got).max())
        if err / tol > worst[1]:
            worst = (n, err / tol)
        assert err <= tol, f"layer {n}: max err {err:.4g} > tol {tol:.4g} (ref max {np.abs(ref).max():.3g}); launches: {launch_list(det)}"
    return d, heads, pred, worst


def launch_list(det):
    """which kernels ran (tile per launch): printed with every layer failure -- most tests run autotuned tiles, which differ from box to box"""
    try:
        return [n for n, _, _ in det.profile(1)]
    except Exception as e:      # (never hide the real failure behind the report)
        return f"<profile failed: {e}>"


def test_stored_layers_with_sub_batch_chains(pkg, wdir):
    """Two sub-batch chains, autotuned: the chains' ops are tuned on their own (half-batch GEMMs) and may fuse launches the
    whole-batch ops keep apart; which layers `debug_layer` reports as stored must follow the ops that RUN (round 3: it followed
    the whole-batch ops, and bench.py's layer check then read the tuner's stale tensor behind a fused Bottleneck)."""
    det, w = make_detector(pkg, wdir, "s", 320, batch=4, autotune=True, chains=2)
    assert det.model.chains == 2
    frame = pkg.synth.frames(1, 320, 320, seed=1234)[0]
    _, _, _, worst = layer_check(pkg, det, w, frame, "s", 320)
    print("worst layer", worst)


def test_forward_layers_yolov8s_640(pkg, wdir):
    det, w = make_detector(pkg, wdir, "s", 640)
    frame = pkg.synth.frames(1, 640, 640, seed=1234)[0]
    d, heads, pred, worst = layer_check(pkg, det, w, frame, "s", 640)
    print("worst layer", worst)
    # decode on identical logits
    A = det.model.n_anchors
    maps, off = [], 0
    for s in (80, 40, 20):
        maps.append(heads[off:off + s * s * 144].reshape(s, s, 144).astype(np.float32)); off += s * s * 144
    ref_pred = Y.decode(maps)
    assert ref_pred.shape == pred.shape == (84, A)
    np.testing.assert_allclose(pred, ref_pred, rtol=2e-4, atol=2e-4)
    # NMS + rescale on the engine's own pre-NMS tensor: integer-exact
    dets, anchors = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 100)
    ref_xyxy = Y.scale_boxes(dets[:, :4], 640, 640, 640, 640)
    assert len(d) == len(dets) and len(d) > 10
    assert np.array_equal(d.xyxy.view(np.int32), ref_xyxy.view(np.int32))
    assert np.array_equal(d.confidence.view(np.int32), dets[:, 4].view(np.int32))
    assert d.class_id.tolist() == dets[:, 5].astype(np.int32).tolist()
    assert d.class_names[0] == pkg.yolo_spec.COCO_NAMES[int(d.class_id[0])]
    det.close()


@pytest.mark.parametrize("scale,size", [("n", 320), ("m", 320), ("s", 320), ("n", 640)])
def test_forward_layers_other_scales(pkg, wdir, scale, size):
    """n and m have channel counts that are not multiples of 32 (16, 48): the kernel's
    general K-chunk path.  n at 640 x 640, one frame: BASELINE config 1's workload (synthetic weights: the real checkpoint needs Ultralytics)."""
    det, w = make_detector(pkg, wdir, scale, size)
    frame = pkg.synth.frames(1, size, size, seed=77)[0]
    d, heads, pred, worst = layer_check(pkg, det, w, frame, scale, size)
    dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 100)
    assert len(d) == len(dets)
    assert np.array_equal(d.xyxy.view(np.int32), Y.scale_boxes(dets[:, :4], size, size, size, size).view(np.int32)) if len(dets) else True
    det.close()


@pytest.mark.parametrize("tile", [T["ROWS_128x32"], T["ROWS_K64_64x64"], T["ROWS_256x64_W8"]])      # 4 waves (32- / 64-deep), 8 waves
def test_tap_reuse_conv_tiles(pkg, wdir, monkeypatch, tile):
    """conv3x3_rows (the 3x3/s1 tap-reuse kernel) in each of its tile shapes, forced onto every
    layer where it is legal; all layers are then checked one by one against the oracle."""
    monkeypatch.setenv("RTMODT_TILE_3X3S1", str(tile))
    det, w = make_detector(pkg, wdir, "s", 320, autotune=False, batch=2)
    frames = list(pkg.synth.frames(2, 320, 320, seed=31))
    det.detect_batch(frames)
    names = [c.name for c in pkg.weights.spec("s")]
    for img in (0, 1):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        gpu = fetch_layers(pkg, det, names, img)
        taps = {}
        Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
        for n in gpu:
            tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"tile {tile} img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}"
    det.close()


@pytest.mark.parametrize("scale,size,batch", [("s", 320, 2), ("s", 640, 1), ("n", 320, 3), ("s", 288, 1)])
def test_fused_bottleneck_kernel(pkg, wdir, monkeypatch, scale, size, batch):
    """bottleneck_fused (conv3x3 -> conv3x3 [+x] with the intermediate in LDS) forced on for every
    Bottleneck it supports (c in 32/64/128); sizes that are not multiples of the 16x16 / 8x16
    tiles exercise the partial-tile and zero-padding paths (288 -> 72/36/18-pixel maps)."""
    monkeypatch.setenv("RTMODT_BNECK", "1")
    monkeypatch.setenv("RTMODT_TILE_3X3S1", "0")
    det, w = make_detector(pkg, wdir, scale, size, autotune=False, batch=batch)
    frames = list(pkg.synth.frames(batch, size, size, seed=41))
    det.detect_batch(frames)
    names = [c.name for c in pkg.weights.spec(scale)]
    n_fused = 0
    for img in range(batch):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        gpu = fetch_layers(pkg, det, names, img)
        n_fused = len(names) - len(gpu)
        taps = {}
        Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
        for n in gpu:
            # the pair's intermediate is fp16 in LDS on the engine, fp32 in the oracle: one extra rounding
            tol = (4e-3 if ".m." in n and n.endswith(".cv2") else 2e-3) * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}"
    assert n_fused >= (4 if scale == "s" else 2), n_fused
    det.close()


def iou_1to1(a, b):
    x1 = np.maximum(a[:, 0], b[:, 0]); y1 = np.maximum(a[:, 1], b[:, 1])
    x2 = np.minimum(a[:, 2], b[:, 2]); y2 = np.minimum(a[:, 3], b[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    return inter / ((a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-9)


def match_report(rx, rc, rk, d, thr):
    """For every oracle detection with score > thr: best-IoU engine detection of the same class."""
    rows = []
    for i in np.nonzero(rc > thr)[0]:
        same = np.nonzero(d.class_id == rk[i])[0]
        if len(same) == 0:
            rows.append((int(i), float(rc[i]), 0.0, -1))
            continue
        ious = iou_1to1(np.repeat(rx[i:i + 1], len(same), 0), d.xyxy[same])
        j = int(np.argmax(ious))
        rows.append((int(i), float(rc[i]), float(ious[j]), int(same[j])))
    return rows


_E2E = {}


def e2e_detector(pkg, wdir, autotune):
    """One detector per tuning mode for all the free-running cases (module lifetime; closed by the last case's fixture teardown)."""
    if autotune not in _E2E:
        det, wts = make_detector(pkg, wdir, "s", 640, classes=[0, 1, 2, 3, 5, 7, 17, 18], max_det=300, autotune=autotune)
        _E2E[autotune] = (det, wts, [n for n, _, _ in det.profile(1)])
    return _E2E[autotune]


@pytest.fixture(scope="module", autouse=True)
def _close_e2e_detectors():
    yield
    for det, _, _ in _E2E.values():
        det.close()
    _E2E.clear()


@pytest.mark.parametrize("autotune", [False, True], ids=["default-tiles", "autotuned"])
@pytest.mark.parametrize("src_hw,seed", [((1080, 1920), 5), ((640, 640), 1234), ((640, 640), 1235), ((640, 640), 1237), ((640, 640), 1239), ((640, 640), -77)])
def test_end_to_end_vs_fp32_oracle(pkg, wdir, src_hw, seed, autotune):
    """Whole pipeline, FREE-RUNNING fp16 engine vs all-fp32 oracle (no teacher forcing: the drift of 63 fp16 convs is in).
    north_star: IoU >= 0.99 per box, identical NMS survivors.  Two engine configurations: the default tiles (`autotune=False`:
    the same launches on every box -- with `weights.synthetic`'s quantised calibration the whole case is box-independent)
    and the autotuned one (what the bench runs; its picks are printed on failure).  Asserted, and printed with -s:
      (1) head logits: max and p99 |engine - oracle| over the three Detect maps (box and class logits separately);
      (2) pre-NMS tensor: boxes within 1 % of their extent + 0.5 px, scores within 0.008;
      (3) NMS on the engine's own pre-NMS tensor: bit-exact (identical survivors given identical inputs);
      (4) EVERY decision NMS takes -- candidate membership, best class, pairwise IoU > 0.45, walk order -- is compared between
          the two tensors (tests/nms_audit.py).  A decision may differ only where the ORACLE's margin is inside the drift
          tolerance (score 0.005, IoU 0.008; measured drift 0.004 / 0.005): then the engine's survivors are exactly NMS of the
          oracle's tensor with near-ties flipped, however far one flip cascades (round 2's one-hop explainer could not follow
          a chain A kept -> B suppressed -> C kept and failed on the driver's box).  One decision outside the tolerance fails;
      (5) per-box statistics: oracle detections reproduced with IoU >= 0.99 and the same class (>= 90 %)."""
    import nms_audit as NA
    h, w = src_hw
    det, wts, launches = e2e_detector(pkg, wdir, autotune)
    classes = det.classes
    # (a negative seed: a STRUCTURED frame -- smooth blobs on a gradient, the input on which the synthetic head saturates into dense clusters of candidates,
    #  VERDICT r04 weak 2 -- instead of uniform noise)
    frame = np.ascontiguousarray(pkg.synth.frames(1, h, w, seed=seed)[0] if seed >= 0 else pkg.synth.structured_frames(1, h, w, seed=-seed)[0])
    d = det.detect(frame)
    (rx, rc, rk), im = Y.detect(frame, wts, "s", (640, 640), 0.35, 0.45, classes, 300, return_intermediate=True)
    _, heads, pred = det.debug_fetch(0, want_input=False)
    cfg = f"weights {pkg.weights.digest(wts)}; launches: " + " | ".join(launches)
    # (1) free-running head logits
    off, box_err, cls_err = 0, [], []
    for lvl, s_ in enumerate((80, 40, 20)):
        g = heads[off:off + s_ * s_ * 144].reshape(s_, s_, 144).astype(np.float32); off += s_ * s_ * 144
        e = np.abs(g - im["heads"][lvl])
        box_err.append(e[..., :64].ravel()); cls_err.append(e[..., 64:].ravel())
    box_err, cls_err = np.concatenate(box_err), np.concatenate(cls_err)
    stats = dict(box_max=float(box_err.max()), box_p99=float(np.percentile(box_err, 99)), cls_max=float(cls_err.max()), cls_p99=float(np.percentile(cls_err, 99)))
    print(f"free-running head logits {src_hw}: box max {stats['box_max']:.4f} p99 {stats['box_p99']:.4f}; cls max {stats['cls_max']:.4f} p99 {stats['cls_p99']:.4f}")
    assert stats["box_p99"] < 0.05 and stats["cls_p99"] < 0.05 and stats["box_max"] < 0.5 and stats["cls_max"] < 0.5, (stats, cfg)
    # (2) pre-NMS tensors agree to fp16 tolerance
    cand = im["pred"][4:].max(0) > 0.2
    extent = np.maximum(im["pred"][2, cand], im["pred"][3, cand])             # box size in pixels (P5 boxes are ~480 px wide)
    assert np.all(np.abs(pred[:4, cand] - im["pred"][:4, cand]) <= 0.01 * extent + 0.5), cfg      # fp16 net vs fp32 net: 1 % of the box + half a pixel
    score_drift = float(np.abs(pred[4:] - im["pred"][4:]).max())
    assert score_drift < 0.008, (score_drift, cfg)
    # (3) NMS on the engine's tensor is exact
    dets_e, anch_e = Y.non_max_suppression(pred, 0.35, 0.45, classes, False, 300)
    assert np.array_equal(d.xyxy.view(np.int32), Y.scale_boxes(dets_e[:, :4], 640, 640, h, w).view(np.int32)), cfg
    assert np.array_equal(d.confidence.view(np.int32), dets_e[:, 4].view(np.int32)) and d.class_id.tolist() == dets_e[:, 5].astype(np.int32).tolist(), cfg
    # (4) every NMS decision, engine tensor vs oracle tensor
    res = NA.audit(pred, im["pred"], 0.35, 0.45, classes, score_tol=0.005, iou_tol=0.008)
    only_e, only_o = NA.survivors_diff(anch_e, im["anchors"])
    print(f"end to end {src_hw} seed {seed}: {len(rc)} oracle / {len(d)} engine detections; survivors only in the engine {only_e}, only in the oracle {only_o}")
    print(NA.describe(res))
    assert len(d) < 300 and len(rc) < 300                                   # (max_det never truncates here: survivor sets are order-free)
    assert not res["hard"], f"NMS decisions differ beyond the fp16 drift tolerance -- the engine's pre-NMS tensor is wrong:\n{NA.describe(res, 40)}\n{cfg}"
    assert res["iou_drift"] < 0.012, (res["iou_drift"], cfg)
    if only_e or only_o:
        assert res["near"], f"survivors differ although no NMS decision does\n{cfg}"
    # (5) per-box reproduction: EVERY oracle detection is accounted for -- reproduced at IoU >= 0.99 with the same class; or its anchor is not
    # an engine survivor: then it is one of `only_o`, which exist only through the audited near-tie flips asserted in (4); or its anchor survives in the engine too
    # and the two boxes differ by the coordinate drift bounded in (2) (1 % of the extent + half a pixel per coordinate): then the IoU of the SAME anchor's two boxes
    # is at least what that bound leaves of a box of its size -- small boxes, printed.  Nothing else is accepted.
    rows = match_report(rx, rc, rk, d, 0.35)
    hit = sum(r[2] >= 0.99 for r in rows)
    frac_all = hit / len(rows) if rows else 1.0
    eng_anchor = {int(a): j for j, a in enumerate(anch_e)}
    flipped, drifted, unexplained = [], [], []
    for i, sc, iou, j in rows:
        if iou >= 0.99:
            continue
        a = int(im["anchors"][i])
        if a not in eng_anchor:
            (flipped if (a in only_o and res["near"]) else unexplained).append((a, round(sc, 4), round(iou, 4), "oracle-only survivor"))
            continue
        bo, be = im["pred"][:4, a].astype(np.float64), pred[:4, a].astype(np.float64)      # (cx, cy, w, h) in network pixels, same anchor
        e_tol = 0.01 * max(bo[2], bo[3]) + 0.5
        bound = max(bo[2] - 2 * e_tol, 0.0) * max(bo[3] - 2 * e_tol, 0.0) / (bo[2] * bo[3])      # every side moved inwards by a centre shift + half a size change
        to_xyxy = lambda v: np.array([[v[0] - v[2] / 2, v[1] - v[3] / 2, v[0] + v[2] / 2, v[1] + v[3] / 2]], np.float32)
        same = float(iou_1to1(to_xyxy(bo), to_xyxy(be))[0])
        ok = same >= bound - 1e-3 and same >= 0.9 and int(d.class_id[eng_anchor[a]]) == int(rk[i])
        (drifted if ok else unexplained).append((a, round(sc, 4), round(same, 4), f"same anchor, box {bo[2]:.1f} x {bo[3]:.1f} px, bound {bound:.3f}"))
    print(f"  IoU >= 0.99 + same class: {hit}/{len(rows)} = {frac_all:.4f}; near-tie cascades: {len(flipped)}; same anchor below 0.99 by coordinate drift: {drifted}; score drift {score_drift:.4f}")
    assert len(rows) >= 5
    assert not unexplained, (unexplained, cfg)
    assert frac_all >= 0.9, (frac_all, cfg)                                   # (sanity floor only; the bar is the accounting above.  Measured 0.977 - 1.0: profiles/r03/diag_e2e)
    assert np.all(d.xyxy[:, [0, 2]] >= 0) and np.all(d.xyxy[:, [0, 2]] <= w) and np.all(d.xyxy[:, [1, 3]] <= h)


def test_batch_equals_single_and_graph_equals_eager(pkg, wdir, monkeypatch):
    """Same tile configuration => same accumulation order => bit-identical detections whatever
    the batch size, sub-batch chaining or graph replay (the autotuner would otherwise pick
    different tiles for different GEMM shapes)."""
    monkeypatch.setenv("RTMODT_TILE", "2")
    det1, _ = make_detector(pkg, wdir, "s", 320, autotune=False)
    det4, _ = make_detector(pkg, wdir, "s", 320, batch=4, autotune=False, chains=2)
    det4e, _ = make_detector(pkg, wdir, "s", 320, batch=4, use_graph=False, autotune=False)
    frames = list(pkg.synth.frames(4, 320, 320, seed=9))
    single = [det1.detect(f) for f in frames]
    for det in (det4, det4e):
        got = det.detect_batch(frames)
        for a, b in zip(single, got):
            assert len(a) == len(b)
            assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32))
            assert a.class_id.tolist() == b.class_id.tolist()
    part = det4.detect_batch(frames[:2])                                       # n < batch
    assert len(part) == 2 and np.array_equal(part[1].xyxy, single[1].xyxy)
    with pytest.raises(ValueError):
        det4.detect_batch(frames + frames)
    for d in (det1, det4, det4e):
        d.close()


def test_device_resident_frames_and_tracker_chain(pkg, wdir):
    """Throughput path: frames already in HBM -> detect -> tracker consumes the detections on
    the device (no host hop) == host path fed the same detections."""
    from oracle import tracker_oracle as T
    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    B = 4
    det, _ = make_detector(pkg, wdir, "s", 320, batch=B)
    frames = pkg.synth.frames(6 * B, 320, 320, seed=21).reshape(6, B, 320, 320, 3)
    buf = pkg._ffi.DeviceBuffer(frames.nbytes)
    buf.upload(frames)
    per = 320 * 320 * 3
    core = core_cls(n_streams=B, max_dets=128, max_tracks=512)
    oracles = [T.TrackerOracle() for _ in range(B)]
    for t in range(6):
        det.enqueue([buf.ptr + (t * B + i) * per for i in range(B)], height=320, width=320)
        core.update_from_detector(det)
        dets = det.fetch()
        host = det.detect_batch(list(frames[t]))
        for i in range(B):
            assert np.array_equal(dets[i].xyxy.view(np.int32), host[i].xyxy.view(np.int32))
            oracles[i].update(dets[i].xyxy, dets[i].confidence, dets[i].class_id)
        for i in range(B):
            assert np.array_equal(T.state_digest(core.snapshot(i)), T.state_digest(oracles[i].snapshot())), (t, i)
    assert sum(len(core.snapshot(i)["ids"]) for i in range(B)) > 0
    total, fwd = det.last_timing()
    assert 0 < fwd <= total
    buf.free()
    det.close()


def test_config5_yolov8m_1280_dense_scene(pkg, wdir):
    """BASELINE config 5: YOLOv8m at 1280x1280 (33 600 anchors, 83 fused convs, channel counts
    48/96/192/384/576) + the 500-box tracker stress sequence on the same handles."""
    from oracle import tracker_oracle as T
    path = os.path.join(str(wdir), "yolov8m_1280.rtw")
    if not os.path.exists(path):
        pkg.weights.save(path, pkg.weights.synthetic("m", input_size=640), "m")      # calibrated at 640 (cheaper), run at 1280
    w, _, _, _ = pkg.weights.load(path)
    det = pkg.Detector(path, input_size=(1280, 1280), max_det=300, warmup=False)
    assert det.model.n_anchors == 33600 and det.model.n_convs == 80           # 83 fused convs, the 3 Detect first-conv pairs share a launch
    assert det.model.conv_flops_per_frame == Y.conv_flops("m", 1280, 1280) == 315742617600
    frame = pkg.synth.frames(1, 1280, 1280, seed=55)[0]
    d = det.detect(frame)
    inp, heads, pred = det.debug_fetch(0)
    # a few layers in isolation (full m/1280 oracle forward is ~300 GFLOP of NumPy: check the stem-side and head-side ends)
    names = ["0", "1", "2.cv1", "2.m.0.cv1", "2.m.0.cv2", "2.m.1.cv2", "2.cv2"]
    gpu = {n: det.debug_layer(n).astype(np.float32) for n in names}
    x = inp.astype(np.float32)
    ref0 = Y.conv2d_nhwc(x, *w["0"], stride=2)
    assert np.abs(ref0 - gpu["0"]).max() <= 2e-3 * np.abs(ref0).max() + 2e-3
    ref1 = Y.conv2d_nhwc(gpu["0"], *w["1"], stride=2)
    assert np.abs(ref1 - gpu["1"]).max() <= 2e-3 * np.abs(ref1).max() + 2e-3
    c = gpu["2.cv1"].shape[2] // 2
    t = Y.conv2d_nhwc(gpu["2.cv1"][..., c:], *w["2.m.0.cv1"])                        # cin = 48: general K-chunk path
    assert np.abs(t - gpu["2.m.0.cv1"]).max() <= 2e-3 * np.abs(t).max() + 2e-3
    t2 = Y.conv2d_nhwc(gpu["2.m.0.cv1"], *w["2.m.0.cv2"]) + gpu["2.cv1"][..., c:]
    assert np.abs(t2 - gpu["2.m.0.cv2"]).max() <= 2e-3 * np.abs(t2).max() + 2e-3
    # ... and, teacher-forced with the engine's own tensors, the layers whose tile paths depend on M at this size: the rest of C2f 2, one P4 and one
    # P5 C2f (persistent-tile ownership, ping-pong tiles at 80 x 80 / 40 x 40 with cin 192 / 288 / 576), SPPF's 9.cv2, and the three Detect stacks
    # (grouped launches over 160 / 80 / 40 pixel levels) -- about 100 GFLOP of im2col + sgemm instead of the net's 316
    all_names = [cv.name for cv in pkg.weights.spec("m")]
    stored = fetch_layers(pkg, det, all_names)
    want = {"2.m.1.cv1", "2.m.1.cv2", "2.cv2", "6.cv1", "6.m.0.cv1", "6.m.0.cv2", "6.m.3.cv2", "6.cv2", "8.cv1", "8.m.0.cv1", "8.m.0.cv2", "8.m.1.cv2", "8.cv2", "9.cv2"}
    want |= {f"22.cv{br}.{lvl}.{k}" for br in (2, 3) for lvl in range(3) for k in (0, 1)}
    # the launches whose tile ownership depends on M at this size -- all of layer 4 (P3 C2f, n = 4 for m, 160 x 160), the strided convs 3, 5, 16, 19,
    # and one whole neck C2f at 160 x 160 (15: its cv1 reads the upsampled half of the concat from the half-resolution tensor) and at 80 x 80 (18)
    want |= {"3", "5", "16", "19", "4.cv1", "4.cv2", "15.cv1", "15.cv2", "18.cv1", "18.cv2"}
    want |= {f"4.m.{j}.cv{k}" for j in range(4) for k in (1, 2)} | {f"{m}.m.{j}.cv{k}" for m in (15, 18) for j in range(2) for k in (1, 2)}
    want &= set(stored)

    def producers(n):          # the convs whose stored outputs feed conv n (a layer the engine did not store cannot be teacher-forced from)
        parts = n.split(".")
        if parts[0] == "22":
            return [f"22.{parts[1]}.{parts[2]}.0"] if parts[3] == "1" else [{"0": "15.cv2", "1": "18.cv2", "2": "21.cv2"}[parts[2]]]
        if len(parts) == 4:    # X.m.j.cvK
            j = int(parts[2])
            return [f"{parts[0]}.m.{j}.cv1"] if parts[3] == "cv2" else ([f"{parts[0]}.cv1"] if j == 0 else [f"{parts[0]}.m.{j - 1}.cv2"])
        if len(parts) == 1:    # a plain Conv: fed by the module in front of it
            return [{"3": "2.cv2", "5": "4.cv2", "16": "15.cv2", "19": "18.cv2"}[n]]
        if parts[1] == "cv2":
            return [f"{parts[0]}.cv1"] + [m for m in all_names if m.startswith(parts[0] + ".m.") and m.endswith(".cv2")] if parts[0] != "9" else ["9.cv1"]
        if parts[0] in ("15", "18"):      # the neck's C2f.cv1 reads a concat: (upsampled 12 | 4) and (16 | 12)
            return {"15": ["12.cv2", "4.cv2"], "18": ["16", "12.cv2"]}[parts[0]]
        return [str(int(parts[0]) - 1)]      # X.cv1 of the backbone C2f / SPPF modules asked for here: fed by conv X - 1
    want = {n for n in want if all(q in stored for q in producers(n))}
    assert {"2.cv2", "6.cv2", "8.cv2", "9.cv2", "3", "5", "16", "19", "4.cv1", "4.cv2", "15.cv1", "15.cv2", "18.cv1", "18.cv2"} <= want and sum(n.startswith("22.") for n in want) >= 6 and \
        sum(n.startswith("4.m.") for n in want) >= 4, sorted(want)
    taps = {}
    Y.forward(x, w, "m", taps=taps, force=stored, only=want)
    worst = ("", 0.0)
    for n in sorted(want):
        tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
        err = float(np.abs(taps[n] - stored[n]).max())
        assert err <= tol, f"m @ 1280 layer {n}: max err {err:.4g} > tol {tol:.4g}; launches: {launch_list(det)}"
        if err / tol > worst[1]:
            worst = (n, err / tol)
    print(f"m @ 1280: {len(want)} layers teacher-forced, worst err / tol {worst[1]:.3f} ({worst[0]})")
    # decode + NMS on the engine's own tensors: exact
    maps, off = [], 0
    for s_ in (160, 80, 40):
        maps.append(heads[off:off + s_ * s_ * 144].reshape(s_, s_, 144).astype(np.float32)); off += s_ * s_ * 144
    np.testing.assert_allclose(pred, Y.decode(maps), rtol=2e-4, atol=2e-4)
    dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 300)
    assert len(d) == len(dets)
    assert np.array_equal(d.xyxy.view(np.int32), Y.scale_boxes(dets[:, :4], 1280, 1280, 1280, 1280).view(np.int32))
    det.close()
    # tracker: 500 boxes per frame on a 1280 canvas (fixture G6 covers bit-exactness; here: capacity + speed path)
    trk = pkg.MultiObjectTracker("bytetrack")
    orc = T.TrackerOracle()
    xy, cf, cl = pkg.synth.box_sequence(500, 1280, 12, seed=9)
    for f in range(12):
        trk.update(pkg.Detections(xy[f], cf, cl)); orc.update(xy[f], cf, cl)
    assert np.array_equal(T.state_digest(trk._core.snapshot()), T.state_digest(orc.snapshot()))


@pytest.mark.parametrize("scale,size", [("l", 960), ("x", 320)])
def test_yolov8l_960_tuner_skips_tiles_that_do_not_fit(pkg, wdir, scale, size):
    """(YOLOv8x @ 320: the widest scale -- 80 / 160 / 320 / 640 channels, cin % 32 != 0 on layer 2: the general K-chunk path -- created and checked the same way;
    before round 5 neither l nor x could be created at all: a reference into d->tensors dangled in the Detect head builder once the tensor list grew.)
    ADVICE r04 (medium): the ping-pong kernels index their output with 24-bit multiplies (padded H x W x C of ONE image < 2^24).  YOLOv8l's layer-2 concat
    tensor has 5 x 64 = 320 channels: at 960 x 960 it is 242 x 242 x 320 = 18.7 M elements -- the tuner's candidate filter (and every cache hit) must skip those
    tiles for the convs that write into it instead of failing `rtmodt_detector_create` (round 4: the launch check's E_INVALID aborted autotune).  Create with the
    tuner ON, one frame, then: the tuner's launch list holds no ping-pong tile on layer 2's Bottleneck outputs, NMS is bit-exact on the engine's own tensor, and
    layer 2 (every conv of it) is within tolerance of the fp32 oracle, teacher-forced."""
    path = os.path.join(str(wdir), f"yolov8{scale}_{size}_cal160.rtw")
    if not os.path.exists(path):
        pkg.weights.save(path, pkg.weights.synthetic(scale, input_size=160), scale)      # calibrated at 160 (cheap), run at `size`
    w, _, _, _ = pkg.weights.load(path)
    det = pkg.Detector(path, input_size=(size, size), max_det=300, warmup=False)         # autotune on (the default)
    frame = pkg.synth.frames(1, size, size, seed=99)[0]
    d = det.detect(frame)
    prof = [n for n, _, _ in det.profile(1)]
    l2 = [n for n in prof if n.startswith("2.m.")]
    assert l2, prof[:12]
    for n in l2:                                                                       # 2.m.j (cv1+cv2): its cv2 writes into the 320-channel tensor
        if "two launches" in n:
            tiles = n.split("two launches:")[1].strip(" ]").split(",")
            assert "pp" not in tiles[1], (n, "a ping-pong tile was chosen for a tensor it cannot index")
    inp, heads, pred = det.debug_fetch(0)
    dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 300)
    assert len(d) == len(dets) and np.array_equal(d.xyxy.view(np.int32), Y.scale_boxes(dets[:, :4], size, size, size, size).view(np.int32))
    names = [cv.name for cv in pkg.weights.spec(scale)]
    stored = fetch_layers(pkg, det, names)
    # layer 2's convs are all COMPUTED by the oracle (also the Bottlenecks' first convs, which a fused launch keeps in LDS), each from the engine's stored inputs;
    # 2.cv1 only when its input (layer 1) was stored, i.e. when it did not run as layer 1's tail
    only = {n for n in names if n.startswith("2.")} - (set() if "1" in stored else {"2.cv1"})
    want = only & set(stored)
    assert "2.cv2" in want and "2.m.2.cv2" in want and len(want) >= 4, sorted(want)
    taps = {}
    Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=stored, only=only)
    for n in sorted(want):
        tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
        assert float(np.abs(taps[n] - stored[n]).max()) <= tol, f"{scale} @ {size} layer {n}; launches: {prof[:12]}"
    det.close()


@pytest.mark.parametrize("scale,size,batch,chains", [("n", 224, 1, 1), ("n", 416, 3, 1), ("s", 416, 3, -1), ("s", 544, 2, -2), ("s", 352, 5, 2), ("m", 224, 2, 1), ("m", 416, 1, 1), ("s", 1024, 1, 1)])
def test_odd_configurations(pkg, wdir, scale, size, batch, chains):
    """Shapes no other test visits (round 5 found YOLOv8l / x broken only because nothing created them): input sizes that are not multiples of 64 (the fused front end
    and the persistent C2f kernel must step aside: 416, 352, 544, 224), odd batches through chains and stages, the largest s engine (1024).  Tuner on.  Per image:
    letterbox bit-exact, NMS survivors bit-equal to the oracle on the engine's own pre-NMS tensor, the first stored layers within tolerance (teacher-forced)."""
    path = os.path.join(str(wdir), f"yolov8{scale}_{size}_cal160.rtw")
    if not os.path.exists(path):
        pkg.weights.save(path, pkg.weights.synthetic(scale, input_size=160), scale)
    w, _, _, _ = pkg.weights.load(path)
    det = pkg.Detector(path, input_size=(size, size), max_det=300, warmup=False, batch=batch, chains=chains)
    frames = list(pkg.synth.frames(batch, size - 17, size, seed=size + batch))       # 17 rows short: letterbox pads top / bottom, no resize
    got = det.detect_batch(frames)
    names = [cv.name for cv in pkg.weights.spec(scale)]
    for i in range(batch):
        inp, _, pred = det.debug_fetch(i, want_heads=False)
        assert np.array_equal(inp.astype(np.float32), Y.preprocess(frames[i], size, size).astype(np.float16).astype(np.float32)), i
        dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 300)
        ref = Y.scale_boxes(dets[:, :4], size, size, size - 17, size) if len(dets) else np.empty((0, 4), np.float32)
        assert len(got[i]) == len(dets) and np.array_equal(got[i].xyxy.view(np.int32), ref.view(np.int32)), (i, launch_list(det)[:6])
    inp, _, _ = det.debug_fetch(batch - 1, want_heads=False, want_pred=False)
    stored = fetch_layers(pkg, det, names, batch - 1)
    only = {n for n in names if n.split(".")[0] in ("0", "1", "2", "3")}      # the oracle computes all of layers 0-3, each conv from the engine's tensor where one is stored
    want = only & set(stored)                                                    # (behind a fused launch: from its own fp32 value of the LDS-resident intermediate)
    assert len(want) >= 2 and "2.cv2" in want, sorted(want)
    taps = {}
    Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=stored, only=only)
    for n in sorted(want):
        tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
        assert float(np.abs(taps[n] - stored[n]).max()) <= tol, f"{scale} @ {size} batch {batch} layer {n}; launches: {launch_list(det)[:8]}"
    det.close()


def test_reference_constructor_behaviour(pkg, wdir, tmp_path):
    path = os.path.join(str(wdir), "yolov8n_160.rtw")
    pkg.weights.save(path, pkg.weights.synthetic("n", input_size=160), "n")
    det = pkg.Detector(str(tmp_path / "missing.engine"), fallback_model=path, input_size=(160, 160))   # fallback + 10x warm-up
    assert det.half is True and det.device == "cuda:0" and det.max_det == 100 and det.model.scale == "n"
    out = det.detect(np.zeros((160, 160, 3), np.uint8))
    assert len(out) == 0 and out.xyxy.shape == (0, 4) and out.xyxy.dtype == np.float32 and out.class_id.dtype == np.int32
    prof = det.profile(2)
    assert len(prof) > 40 and all(ms >= 0 for _, ms, _ in prof)
    det.close()
    bad = tmp_path / "bad.rtw"
    bad.write_bytes(b"not a weight file")
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        pkg.Detector(str(bad), input_size=(160, 160))
    assert e.value.code == pkg._ffi.E_IO


@pytest.mark.parametrize("handoff", [True, False])
def test_pipeline_loop_and_stage_profiler(pkg, wdir, handoff):
    """the reference's per-frame loop with the sync-bracketed stage profiler,
    including the sub-stages (preprocess / nms) the reference names but never measures.  With the device hand-off
    (default) the tracker consumes the detections on the device; without it the reference's literal data flow
    (host arrays into tracker.update) -- the tracker state must be the oracle's either way."""
    from oracle import tracker_oracle as T
    det, _ = make_detector(pkg, wdir, "s", 320)
    trk = pkg.MultiObjectTracker("bytetrack")
    frames = pkg.synth.frames(6, 320, 320, seed=12)
    prof = pkg.profiling.LatencyProfiler(gpu_sync=True, warmup_frames=10, log_interval=20)
    assert prof.gpu_sync is True
    out = pkg.pipeline.run(pkg.pipeline.SyntheticSource(frames), det, trk, prof, max_frames=50, device_handoff=handoff)
    for st in ("decode", "preprocess", "inference", "nms", "tracking", "total"):
        assert out[f"{st}_mean_ms"] > 0 and out[f"{st}_p50_ms"] > 0 and out[f"{st}_p99_ms"] >= out[f"{st}_p50_ms"], st
    parts = sum(out[f"{s}_mean_ms"] for s in ("decode", "preprocess", "inference", "nms", "tracking"))
    assert abs(parts - out["total_mean_ms"]) < 1e-6 * max(1.0, parts)
    assert out["fps_mean"] > 0 and out["last_tracks"] == 0               # facade returns [] like the reference
    # the loop fed the tracker exactly the detector's detections: replay them through the oracle
    orc = T.TrackerOracle()
    for i in range(50):
        d = det.detect(frames[i % 6])
        orc.update(d.xyxy, d.confidence, d.class_id)
    assert np.array_equal(T.state_digest(trk._core.snapshot()), T.state_digest(orc.snapshot()))
    det.close()


def test_pipeline_fed_by_frame_reader_through_pinned_ring(pkg, wdir):
    """the ingest reader (reference interface: src/ingestion/rtsp_reader.py) decodes straight into a
    page-locked ring; the loop of rank 1 runs on it unchanged, and a frame handed out without a copy (`read(copy=False)`)
    is uploaded from where the reader put it -- same detections as from an ordinary NumPy copy of it."""
    det, _ = make_detector(pkg, wdir, "s", 320)
    trk = pkg.MultiObjectTracker("bytetrack")
    frames = pkg.synth.frames(6, 320, 320, seed=12)
    ring = pkg.pipeline.PinnedFrameRing(3, 320, 320)
    with pkg.ingestion.FrameReader("synthetic", backend="synthetic", resolution=(320, 320), frames=frames, fps=500.0, ring=ring) as reader:
        import time
        t0 = time.perf_counter()
        while not reader.read()[0] and time.perf_counter() - t0 < 3.0:
            time.sleep(0.002)
        prof = pkg.profiling.LatencyProfiler(gpu_sync=True, warmup_frames=5, log_interval=20)
        out = pkg.pipeline.run(reader, det, trk, prof, max_frames=30)
        assert out["total_p50_ms"] > 0 and out["decode_p50_ms"] > 0
        ok, f, fid = reader.read(copy=False)
        assert ok and fid >= 1 and np.array_equal(f, frames[(fid - 1) % 6])
        a = det.detect(f)                                    # page-locked slot: asynchronous DMA path
        b = det.detect(frames[(fid - 1) % 6].copy())          # pageable copy
        assert np.array_equal(a.xyxy, b.xyxy) and np.array_equal(a.class_id, b.class_id) and np.array_equal(a.confidence, b.confidence)
    ring.close()
    det.close()


# ------------------------------------------------------------------ rect (LetterBox auto=True) mode
@pytest.mark.parametrize("h,w,shape", [(1080, 1920, (384, 640)), (480, 640, (480, 640)), (720, 405, (640, 384)), (333, 500, (448, 640))])
def test_rect_letterbox_mode(pkg, wdir, h, w, shape):
    """`predict` on a .pt model letterboxes to the minimal stride-32 rectangle; Detector(rect=True) builds
    one engine per rectangle.  Input image bit-exact vs the oracle's auto=True letterbox, detections vs the
    oracle run on the engine's own head maps (identical NMS survivors), boxes mapped back with the
    rectangle's gain/pad."""
    det, wts = make_detector(pkg, wdir, "n", 640, rect=True, autotune=False)
    assert det.rect_shape(h, w, 640) == shape
    frame = next(iter(pkg.synth.structured_frames(1, h, w, seed=9)))
    got = det.detect(frame)
    assert det.model.input_hw == shape
    inp, heads, pred = det.debug_fetch(0)
    want = Y.preprocess(frame, 640, 640, auto=True)
    assert want.shape[:2] == shape
    assert np.array_equal(inp.astype(np.float32), want.astype(np.float16).astype(np.float32))
    dets, anchors = Y.non_max_suppression(pred, det.confidence, det.iou, None, False, det.max_det, 80)
    xyxy = Y.scale_boxes(dets[:, :4], shape[0], shape[1], h, w) if len(dets) else np.empty((0, 4), np.float32)
    assert len(got) == len(dets)
    assert np.array_equal(got.class_id, dets[:, 5].astype(np.int32))
    assert np.array_equal(got.confidence, dets[:, 4].astype(np.float32))
    assert np.abs(got.xyxy - xyxy).max(initial=0) <= 1e-3
    # a second frame size switches engines; the square engine from __init__ is still cached
    det.detect(np.zeros((640, 640, 3), np.uint8))
    assert det.model.input_hw == (640, 640) and len(det._models) == (1 if shape == (640, 640) else 2)
    det.close()


@pytest.mark.parametrize("h,w,rect", [(640, 640, False), (600, 632, False), (609, 637, False), (1080, 1920, False), (1080, 1920, True), (320, 320, False)])
def test_front_end_fused_is_bit_identical(pkg, wdir, monkeypatch, h, w, rect):
    """csrc/front.hip: [letterbox +] stem + layer 1 + 2.cv1 as ONE launch -- the stem's and layer 1's outputs never leave the CU -- against the launches it
    replaces, forced on / off with the tuner out of the way.  Same MFMA instruction, same k order, fp16 at the same two places: 2.cv1's stored tensor,
    everything behind it and the detections must be BIT-identical.  Sources: the frames' bytes (no resize: 640x640; a smaller frame with letterbox pads
    on an odd pitch; 320x320) and the letterboxed image tensor (1080p resized; rect = its minimal 384x640 rectangle).  The fused-away tensors are
    reported as such, and 2.cv1 stays within the ordinary layer tolerance (2e-3) of the fp32 oracle although TWO fp16 intermediates now stand behind it."""
    size = 320 if h == 320 else 640
    pitch = 3 * w + (7 if h == 609 else 0)
    wide = pkg.synth.structured_frames(3, h, pitch // 3 + 1, seed=h + w).reshape(3, -1)[:, :h * pitch]
    buf = pkg._ffi.DeviceBuffer(wide.nbytes + 64)
    buf.upload(np.ascontiguousarray(wide))
    ptrs = [buf.ptr + i * h * pitch for i in range(3)]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTMODT_FRONT", mode)
        path = os.path.join(str(wdir), f"yolov8s_{size}_noise.rtw")
        if not os.path.exists(path):
            pkg.weights.save(path, pkg.weights.synthetic("s", input_size=size), "s")
        wts = pkg.weights.load(path)[0]
        det = pkg.Detector(path, input_size=(size, size), warmup=False, autotune=False, batch=3, rect=rect)
        det.enqueue(ptrs, height=h, width=w, pitch=pitch)
        dets = det.fetch()
        prof = [n for n, _, _ in det.profile(1)]
        assert ("front end fused" in prof[0]) == (mode == "1"), prof[:3]
        if mode == "1":
            assert "inside the front-end launch" in prof[1] and "inside the front-end launch" in prof[2], prof[:3]
            for gone in ("0", "1"):
                with pytest.raises(pkg._ffi.RtmodtError) as ei:
                    det.debug_layer(gone, 0)
                assert ei.value.code == pkg._ffi.E_UNSUPPORTED
        layers = [{n: det.debug_layer(n, i) for n in ("2.cv1", "2.cv2", "4.cv2", "9.cv2", "21.cv2")} for i in range(3)]
        preds = [det.debug_fetch(i, want_input=True, want_heads=False) for i in range(3)]
        if mode == "1" and not rect:                               # teacher-forced against the oracle: 0 and 1 are recomputed in fp32 from the engine's own input
            inp = preds[0][0]
            taps = {}
            Y.forward(inp.astype(np.float32), wts, "s", taps=taps, force={"2.cv1": layers[0]["2.cv1"].astype(np.float32)}, only={"0", "1", "2.cv1"})
            ref = taps["2.cv1"]
            err, tol = float(np.abs(ref - layers[0]["2.cv1"].astype(np.float32)).max()), 2e-3 * float(np.abs(ref).max()) + 2e-3
            print(f"front end fused, {h}x{w}: 2.cv1 err/tol {err / tol:.3f}")
            assert err <= tol, (err, tol)
        outs[mode] = (layers, preds, dets)
        det.close()
    for i in range(3):
        for n in outs["0"][0][i]:
            assert np.array_equal(outs["0"][0][i][n].view(np.uint16), outs["1"][0][i][n].view(np.uint16)), f"layer {n}, image {i}"
        assert np.array_equal(outs["0"][1][i][0].view(np.uint16), outs["1"][1][i][0].view(np.uint16)), f"input image {i}"
        assert np.array_equal(outs["0"][1][i][2], outs["1"][1][i][2]), f"pre-NMS tensor, image {i}"
        assert np.array_equal(outs["0"][2][i].xyxy, outs["1"][2][i].xyxy) and np.array_equal(outs["0"][2][i].confidence, outs["1"][2][i].confidence)
    buf.free()


@pytest.mark.parametrize("chains", [1, 2, -1, -2], ids=["plain", "two-chains", "two-stages", "three-stages"])
def test_front_end_fused_under_every_engine_shape(pkg, wdir, monkeypatch, chains):
    """The fused front end is launched by enqueue_batch itself (its frame pointers change every batch), once per sub-batch chain or per arena copy of the
    staged engine: forced on, batches of 4 pipelined through the plain engine, two sub-batch chains, two and three stages must give the detections of the
    unfused plain engine, bit for bit (device frames that need no resize, and host frames that take the letterbox kernel + the tensor source)."""
    frames = pkg.synth.structured_frames(12, 320, 320, seed=77).reshape(3, 4, 320, 320, 3)
    small = pkg.synth.structured_frames(4, 240, 416, seed=78)                      # resized: letterbox kernel -> image tensor -> front_fused<TENSOR>
    monkeypatch.setenv("RTMODT_FRONT", "0")
    ref_det, _ = make_detector(pkg, wdir, "s", 320, autotune=False, batch=4, chains=1)
    ref = [ref_det.detect_batch(list(frames[t])) for t in range(3)] + [ref_det.detect_batch(list(small))]
    ref_det.close()
    monkeypatch.setenv("RTMODT_FRONT", "1")
    det, _ = make_detector(pkg, wdir, "s", 320, autotune=False, batch=4, chains=chains)
    assert "front end fused" in det.profile(1)[0][0]
    buf = pkg._ffi.DeviceBuffer(frames.nbytes)
    buf.upload(frames)
    per = 320 * 320 * 3
    depth = (det.model.stages + 1) if getattr(det.model, "stages", 1) > 1 else 2
    got = []
    for t in range(3):
        det.enqueue([buf.ptr + (t * 4 + i) * per for i in range(4)], height=320, width=320)
        if t >= depth - 1:
            got.append(det.fetch())
    while len(got) < 3:
        got.append(det.fetch())
    got.append(det.detect_batch(list(small)))
    for t in range(4):
        for i in range(4):
            assert np.array_equal(got[t][i].xyxy.view(np.int32), ref[t][i].xyxy.view(np.int32)) and got[t][i].class_id.tolist() == ref[t][i].class_id.tolist(), (chains, t, i)
    assert sum(len(d) for b in got for d in b) > 0
    buf.free()
    det.close()


@pytest.mark.parametrize("h,w", [(640, 640), (480, 640), (640, 512), (636, 640), (640, 634), (640, 630)])
def test_letterbox_fused_into_stem_is_bit_identical(pkg, wdir, monkeypatch, h, w):
    """Frames that need no resize skip the letterbox kernel: the stem conv builds its MFMA fragments from the BGR
    bytes (114 padding, zero canvas border, c/255 table).  Its output must equal the two-kernel path bit for bit;
    batch of 3 device-resident frames whose row pitch is wider than the image."""
    pitch = 3 * (w + 5)
    wide = pkg.synth.structured_frames(3, h, w + 5, seed=h + w)
    buf = pkg._ffi.DeviceBuffer(wide.nbytes)
    buf.upload(wide)
    ptrs = [buf.ptr + i * h * pitch for i in range(3)]
    outs = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("RTMODT_STEM_FUSE", fuse)
        det, _ = make_detector(pkg, wdir, "s", 640, autotune=False, batch=3)
        det.enqueue(ptrs, height=h, width=w, pitch=pitch)
        dets = det.fetch()
        prof = [n for n, _, _ in det.profile(1)]
        assert ("fused" in prof[0]) == (fuse == "1"), prof[0]
        outs[fuse] = ([det.debug_layer("0", i) for i in range(3)], [det.debug_fetch(i, want_heads=False, want_pred=False)[0] for i in range(3)], dets)
        det.close()
    for i in range(3):
        assert np.array_equal(outs["0"][0][i].view(np.uint16), outs["1"][0][i].view(np.uint16)), f"stem output, image {i}"
        assert np.array_equal(outs["0"][1][i].view(np.uint16), outs["1"][1][i].view(np.uint16)), f"input image, image {i}"
        assert np.array_equal(outs["0"][2][i].xyxy, outs["1"][2][i].xyxy)
    want = Y.preprocess(wide[1][:, :w], 640, 640)
    assert np.array_equal(outs["1"][1][1].astype(np.float32), want.astype(np.float16).astype(np.float32))
    buf.free()


def test_autotune_cache_replays_the_same_configuration(pkg, wdir, monkeypatch, tmp_path):
    """RTMODT_TUNE_CACHE: the first detector times every tile and records its choices; the second one (same
    shapes) reads them back -- identical launch configuration, no timing runs; a third one with another batch
    size adds its own keys to the same file."""
    import time
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("RTMODT_TUNE_CACHE", str(cache))
    t0 = time.perf_counter()
    a, _ = make_detector(pkg, wdir, "n", 320, autotune=True, batch=2)
    t1 = time.perf_counter()
    names_a = [n for n, _, _ in a.profile(1)]
    a.close()
    lines = cache.read_text().splitlines()
    assert lines[0].startswith("#rtmodt-tune tiles=")                # the tile table the ids refer to
    assert len(lines) > 30 and all(len(l.split()) == 4 for l in lines[1:])
    t2 = time.perf_counter()
    b, _ = make_detector(pkg, wdir, "n", 320, autotune=True, batch=2)
    t3 = time.perf_counter()
    assert [n for n, _, _ in b.profile(1)] == names_a
    assert (t3 - t2) < 0.5 * (t1 - t0), (t1 - t0, t3 - t2)
    frames = list(pkg.synth.frames(2, 320, 320, seed=5))
    assert all(len(d.xyxy) == len(d.confidence) for d in b.detect_batch(frames))
    b.close()
    c, _ = make_detector(pkg, wdir, "n", 320, autotune=True, batch=1)
    c.close()
    assert len(cache.read_text().splitlines()) > len(lines)


@pytest.mark.parametrize("bias", [-4.1, 9.0, 14.0])
def test_decode_fast_path_equals_score_by_score(pkg, tmp_path, bias):
    """decode_kernel takes the arg-max over class LOGITS and one sigmoid when every logit of the anchor is < 8
    (the sigmoid is strictly increasing on the fp16 grid there) and the score-by-score loop otherwise; the
    `pred` dump always takes the loop.  Class ids, scores and NMS survivors from the engine's candidates must
    equal those derived from the dumped pred -- for ordinary logits (bias -4.1), logits straddling the switch
    (9) and saturated ones where float32 sigmoids tie (14)."""
    path = str(tmp_path / "w.rtw")
    w = pkg.weights.synthetic("n", input_size=320, cls_bias=bias)
    pkg.weights.save(path, w, "n")
    det = pkg.Detector(path, input_size=(320, 320), confidence=0.25, max_det=300, warmup=False, autotune=False)
    frame = pkg.synth.frames(1, 320, 320, seed=8)[0]
    got = det.detect(frame)
    _, heads, pred = det.debug_fetch(0, want_input=False)
    cls_logits = np.concatenate([heads[o:o + n * 144].reshape(n, 144)[:, 64:] for o, n in ((0, 1600), (1600 * 144, 400), (2000 * 144, 100))]).astype(np.float32)
    if bias > 8:
        assert (cls_logits.max(axis=1) >= 8).mean() > 0.3            # the loop really is exercised
    if bias > 12:
        assert (pred[4:].max(axis=0) == 1.0).any()                  # saturated float32 sigmoids: ties by rounding
    dets, _ = Y.non_max_suppression(pred, 0.25, det.iou, None, False, 300, 80)
    assert len(got) == len(dets) > 0
    assert np.array_equal(got.class_id, dets[:, 5].astype(np.int32))
    assert np.array_equal(got.confidence.view(np.int32), dets[:, 4].astype(np.float32).view(np.int32))
    det.close()


@pytest.mark.parametrize("F", [2, 3])
def test_frame_batching_keeps_tracker_order(pkg, wdir, monkeypatch, F):
    """bench.py's default launch set holds F consecutive frames of every stream (image f * S + s); the tracker
    consumes them slice by slice on the device.  Detections and every stream's tracker state must equal the
    frame-at-a-time run (same tiles forced so the convs are bit-identical) and the oracle."""
    from oracle import tracker_oracle as T
    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_BNECK", "0")
    S, steps = 3, 4
    frames = pkg.synth.frames(S * F * steps, 320, 320, seed=33).reshape(steps, F, S, 320, 320, 3)
    buf = pkg._ffi.DeviceBuffer(frames.nbytes)
    buf.upload(frames)
    per = 320 * 320 * 3
    big, _ = make_detector(pkg, wdir, "s", 320, batch=S * F, autotune=False, confidence=0.2)
    one, _ = make_detector(pkg, wdir, "s", 320, batch=S, autotune=False, confidence=0.2)
    core_b, core_1 = core_cls(n_streams=S, max_dets=128, max_tracks=512), core_cls(n_streams=S, max_dets=128, max_tracks=512)
    core_x = core_cls(n_streams=S, max_dets=128, max_tracks=512)      # the same batch in ONE tracker launch (what bench.py issues)
    oracles = [T.TrackerOracle() for _ in range(S)]
    for t in range(steps):
        big.enqueue([buf.ptr + ((t * F + f) * S + s) * per for f in range(F) for s in range(S)], height=320, width=320)
        for f in range(F):
            core_b.update_from_detector(big, f * S, S)
        core_x.update_from_detector(big, 0, S, frames_per_stream=F)
        got = big.fetch()
        assert len(got) == S * F
        for f in range(F):
            one.enqueue([buf.ptr + ((t * F + f) * S + s) * per for s in range(S)], height=320, width=320)
            core_1.update_from_detector(one)
            ref = one.fetch()
            for s in range(S):
                d = got[f * S + s]
                assert np.array_equal(d.xyxy.view(np.int32), ref[s].xyxy.view(np.int32)) and np.array_equal(d.class_id, ref[s].class_id)
                oracles[s].update(d.xyxy, d.confidence, d.class_id)
        for s in range(S):
            assert np.array_equal(T.state_digest(core_b.snapshot(s)), T.state_digest(core_1.snapshot(s))), (t, s)
            assert np.array_equal(T.state_digest(core_b.snapshot(s)), T.state_digest(oracles[s].snapshot())), (t, s)
            assert np.array_equal(T.state_digest(core_x.snapshot(s)), T.state_digest(oracles[s].snapshot())), ("one launch", t, s)
    assert sum(len(core_b.snapshot(s)["ids"]) for s in range(S)) > 0
    with pytest.raises(pkg._ffi.RtmodtError):
        core_b.update_from_detector(big, S * F - 1, S)          # slice runs past the batch
    with pytest.raises(pkg._ffi.RtmodtError):
        core_x.update_from_detector(big, S, S, frames_per_stream=F)      # F frames from slot S on: past the batch
    buf.free(); big.close(); one.close()


def test_epilogue_variants_store_identical_values(pkg, wdir, monkeypatch):
    """The conv epilogue either stores 8 bytes per lane straight from the MFMA accumulator layout (0) or stages the
    fp16 tile in LDS and stores 16 bytes per lane (1: tile kernels -- the default; 2: also the 3x3 tap-reuse kernel,
    forced here).  bias / SiLU / residual / nearest-2x copy happen in fp32 before the single rounding in all of them,
    so every layer must be bit-identical."""
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TILE_3X3S1", str(T["ROWS_256x64_W8"]))
    frames = list(pkg.synth.frames(2, 320, 320, seed=61))
    names = [c.name for c in pkg.weights.spec("s")]
    outs = {}
    for mode in ("0", "1", "2"):
        monkeypatch.setenv("RTMODT_EPI16", mode)
        det, _ = make_detector(pkg, wdir, "s", 320, autotune=False, batch=2)
        det.detect_batch(frames)
        outs[mode] = [fetch_layers(pkg, det, names, i) for i in range(2)]
        det.close()
    for mode in ("1", "2"):
        for i in range(2):
            for n in outs["0"][i]:
                assert np.array_equal(outs["0"][i][n].view(np.uint16), outs[mode][i][n].view(np.uint16)), (mode, i, n)


@pytest.mark.parametrize("size,batch", [(320, 2), (288, 1), (640, 1)])
def test_conv_with_fused_1x1_tail(pkg, wdir, monkeypatch, size, batch):
    """Backbone Conv -> C2f.cv1 pairs ("1" -> "2.cv1", "3" -> "4.cv1" of YOLOv8s) as ONE launch: the conv's output tile
    stays in LDS, the 1x1 runs on it there, only the 1x1's output is stored.  Forced on / off with the tuner out of
    the way (every conv on the 64x64 tile, so the k order is the same everywhere): every stored layer must be bit-identical
    between the two and within tolerance of the oracle; the fused-away tensors are reported as such."""
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TILE", "2")
    frames = list(pkg.synth.frames(batch, size, size, seed=71))
    names = [c.name for c in pkg.weights.spec("s")]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTMODT_TAIL", mode)
        det, w = make_detector(pkg, wdir, "s", size, autotune=False, batch=batch)
        det.detect_batch(frames)
        prof = [n for n, _, _ in det.profile(1)]
        assert (sum("tail:" in n for n in prof) == 2) == (mode == "1"), prof[:8]
        outs[mode] = []
        for img in range(batch):
            inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
            gpu = fetch_layers(pkg, det, names, img)
            outs[mode].append(gpu)
            if mode == "1":
                assert "1" not in gpu and "3" not in gpu and "2.cv1" in gpu and "4.cv1" in gpu
                taps = {}
                Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
                for n in gpu:
                    tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
                    assert float(np.abs(taps[n] - gpu[n]).max()) <= tol, (img, n)
        det.close()
    for img in range(batch):                                         # every conv on the 64x64 tile: same k order everywhere
        for n in outs["1"][img]:
            assert np.array_equal(outs["0"][img][n], outs["1"][img][n]), (img, n)


@pytest.mark.parametrize("scale,size,batch", [("s", 320, 2), ("m", 256, 1), ("s", 288, 3)])
def test_head_final_equals_separate_launches(pkg, wdir, monkeypatch, scale, size, batch):
    """Detect's last 1x1 convs + decode as one launch (head_final: logits rounded to fp16 in LDS, decoded there) against
    the grouped 1x1 launch + decode kernel it replaces, every conv on the 64x64 tile so that the k order is the same:
    head rows, pred, detections -- all bit-identical; 288 gives anchor counts that are not multiples of the 64-anchor tile."""
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_BNECK", "0")
    frames = list(pkg.synth.frames(batch, size, size, seed=81))
    res = {}
    for mode in ("off", "on"):
        if mode == "off":
            monkeypatch.setenv("RTMODT_NO_HEAD_FINAL", "1")
        else:
            monkeypatch.delenv("RTMODT_NO_HEAD_FINAL")
        det, _ = make_detector(pkg, wdir, scale, size, autotune=False, batch=batch, confidence=0.05, max_det=300)
        dets = det.detect_batch(frames)
        prof = [n for n, _, _ in det.profile(1)]
        assert ("head_final" in prof[-1]) == (mode == "on"), prof[-2:]
        res[mode] = (dets, [det.debug_fetch(i, want_input=False) for i in range(batch)])
        det.close()
    for i in range(batch):
        a, b = res["off"][0][i], res["on"][0][i]
        assert len(a) == len(b) > 0
        assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32)) and np.array_equal(a.class_id, b.class_id)
        assert np.array_equal(a.confidence.view(np.int32), b.confidence.view(np.int32))
        assert np.array_equal(res["off"][1][i][1].view(np.uint16), res["on"][1][i][1].view(np.uint16))       # head rows
        assert np.array_equal(res["off"][1][i][2].view(np.int32), res["on"][1][i][2].view(np.int32))         # pred


@pytest.mark.parametrize("tile", [T["K64_128x128_S2_W8"], T["K64_128x128_S3_W8"], T["K64_128x64_S3_W8"], T["K64_256x64_S2_W8"]])
def test_eight_wave_tiles(pkg, wdir, monkeypatch, tile):
    """The 64-deep tile kernel with EIGHT waves per workgroup (the global->LDS path sustains ~5 B/clk per wave, so the
    big tiles issue their operands from twice as many waves): forced onto every single-launch conv with cin % 64 == 0,
    all layers against the oracle, both epilogues."""
    monkeypatch.setenv("RTMODT_TILE_K64", str(tile))
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    for epi in ("1", "0"):
        monkeypatch.setenv("RTMODT_EPI16", epi)
        det, w = make_detector(pkg, wdir, "s", 320, autotune=False, batch=2)
        used = [n for n, _, _ in det.profile(1) if "/8w" in n or "/16w" in n or "k64pf:" in n]
        assert len(used) >= 15, used
        frames = list(pkg.synth.frames(2, 320, 320, seed=91))
        det.detect_batch(frames)
        names = [c.name for c in pkg.weights.spec("s")]
        for img in (0, 1):
            inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
            gpu = fetch_layers(pkg, det, names, img)
            taps = {}
            Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
            for n in gpu:
                tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
                err = float(np.abs(taps[n] - gpu[n]).max())
                assert err <= tol, f"tile {tile} epi {epi} img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}"
        det.close()


@pytest.mark.parametrize("size,batch", [(320, 2), (288, 1), (640, 1)])
def test_bottleneck_with_c2f_cv2_tail(pkg, wdir, monkeypatch, size, batch):
    """C2f with one Bottleneck (layers 2 and 15 of YOLOv8s): the fused Bottleneck launch also runs C2f.cv2 on
    [the two earlier concat chunks from HBM | its own output tile in LDS]; the Bottleneck's output tensor is never
    stored.  Forced on / off with every other conv on the 64x64 tile: stored layers bit-identical between the two
    (same k order: concat order) and within tolerance of the oracle; partial 16x16 tiles at 288."""
    monkeypatch.setenv("RTMODT_BNECK", "1")
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    frames = list(pkg.synth.frames(batch, size, size, seed=101))
    names = [c.name for c in pkg.weights.spec("s")]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTMODT_BNECK_TAIL", mode)
        det, w = make_detector(pkg, wdir, "s", size, autotune=False, batch=batch)
        det.detect_batch(frames)
        prof = [n for n, _, _ in det.profile(1)]
        assert (sum("C2f.cv2 tail" in n for n in prof) == 2) == (mode == "1"), [n for n in prof if "bottleneck" in n]
        outs[mode] = []
        for img in range(batch):
            inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
            gpu = fetch_layers(pkg, det, names, img)
            outs[mode].append(gpu)
            if mode == "1":
                assert "2.m.0.cv2" not in gpu and "15.m.0.cv2" not in gpu and "2.cv2" in gpu and "15.cv2" in gpu and "4.m.1.cv2" in gpu
                taps = {}
                Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
                for n in gpu:
                    tol = 2e-3 * np.abs(taps[n]).max() + 2e-3      # (also behind the two LDS-resident fp16 intermediates: measured 0.2 of it)
                    assert float(np.abs(taps[n] - gpu[n]).max()) <= tol, (img, n)
        det.close()
    for img in range(batch):
        for n in outs["1"][img]:
            assert np.array_equal(outs["0"][img][n], outs["1"][img][n]), (img, n)


@pytest.mark.parametrize("size,batch", [(640, 2), (320, 3), (64, 1)])
def test_persistent_c2f32_kernel_is_bit_identical(pkg, wdir, monkeypatch, size, batch):
    """csrc/bneck32.hip: YOLOv8s' layer-2 Bottleneck (c = 32) with C2f.cv2 as its tail on persistent 256-thread workgroups (8 x 16 tiles, weights of both 3x3
    convs in registers, the next tile's patch prefetched by LDS-DMA) against bottleneck_fused<32, 16, 16, 4>: same arithmetic order, so 2.cv2 and
    everything behind it must be BIT-identical; and within the ordinary layer tolerance (2e-3) of the fp32 oracle, two fp16 intermediates in LDS notwithstanding."""
    monkeypatch.setenv("RTMODT_BNECK", "1")
    monkeypatch.setenv("RTMODT_BNECK_TAIL", "1")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    frames = list(pkg.synth.frames(batch, size, size, seed=303))
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTMODT_BNECK32", mode)
        det, w = make_detector(pkg, wdir, "s", size, autotune=False, batch=batch)
        dets = det.detect_batch(frames)
        outs[mode] = ([{n: det.debug_layer(n, i) for n in ("2.cv1", "2.cv2", "4.cv2", "21.cv2")} for i in range(batch)], dets)
        if mode == "1":
            inp, _, _ = det.debug_fetch(0, want_heads=False, want_pred=False)
            lay = outs[mode][0][0]
            taps = {}
            Y.forward(inp.astype(np.float32), w, "s", taps=taps, force={n: lay[n].astype(np.float32) for n in ("2.cv1", "2.cv2")}, only={"2.m.0.cv1", "2.m.0.cv2", "2.cv2"})
            err, tol = float(np.abs(taps["2.cv2"] - lay["2.cv2"].astype(np.float32)).max()), 2e-3 * float(np.abs(taps["2.cv2"]).max()) + 2e-3
            print(f"persistent c2f32 @ {size}: 2.cv2 err/tol {err / tol:.3f}")
            assert err <= tol, (err, tol)
        det.close()
    for i in range(batch):
        for n in outs["0"][0][i]:
            assert np.array_equal(outs["0"][0][i][n].view(np.uint16), outs["1"][0][i][n].view(np.uint16)), (n, i)
        assert np.array_equal(outs["0"][1][i].xyxy, outs["1"][1][i].xyxy)


@pytest.mark.parametrize("src_hw,host", [((320, 320), False), ((320, 320), True), ((240, 416), True)])
def test_free_running_chains_match_one_chain(pkg, wdir, monkeypatch, src_hw, host):
    """A 16-frame launch set runs as two sub-batch chains on their own streams (stem -> graph -> Detect's tail per chain,
    joined only by the post-processing stream).  With every conv forced onto one tile the detections of pipelined
    batches (two in flight) must equal those of the single-chain engine and of the per-batch-joined variant, for
    device frames, host frames through the copy stream, and frames that take the unfused letterbox (resize)."""
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_BNECK", "0")
    if host:
        monkeypatch.setenv("RTMODT_PAD_STREAMS", "1")     # chain 1's first candidate stream then shares the main stream's hardware queue: the probe must move on
    B, steps = 16, 5
    h, w = src_hw
    frames = pkg.synth.frames(B * steps, h, w, seed=77).reshape(steps, B, h, w, 3)
    buf = None
    if not host:
        buf = pkg._ffi.DeviceBuffer(frames.nbytes)
        buf.upload(frames)
    per = h * w * 3

    def run(chains, join):
        monkeypatch.setenv("RTMODT_CHAINS", str(chains))
        monkeypatch.setenv("RTMODT_CHAIN_JOIN", "1" if join else "0")
        det, _ = make_detector(pkg, wdir, "s", 320, batch=B, autotune=False, confidence=0.02)
        outs = []
        for t in range(steps):                                       # two batches in flight, as the pipeline runs them
            if host:
                det.enqueue(list(frames[t]))
            else:
                det.enqueue([buf.ptr + (t * B + i) * per for i in range(B)], height=h, width=w)
            if t:
                outs.append(det.fetch())
        outs.append(det.fetch())
        names = [n for n, _, _ in det.profile(1)]
        actual = det.model.chains                                    # fewer than asked for when the process has no more hardware queues to itself
        det.close()
        return outs, names, actual

    ref, _, one = run(1, False)
    assert one == 1 and sum(len(d) for batch in ref for d in batch) > 0
    for chains, join in ((2, False), (2, True), (4, False)):
        got, names, actual = run(chains, join)
        assert actual == chains or (chains == 4 and actual in (2, 4)), (chains, actual)
        assert any(f"M={B // actual * 80 * 80} " in n for n in names), names[:3]      # the profile lists one chain's launches
        for t in range(steps):
            for i in range(B):
                a, b = got[t][i], ref[t][i]
                assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32)) and np.array_equal(a.class_id, b.class_id) and \
                    np.array_equal(a.confidence.view(np.int32), b.confidence.view(np.int32)), (chains, join, t, i)
    if buf is not None:
        buf.free()


@pytest.mark.parametrize("src_hw,host", [((320, 320), False), ((320, 320), True), ((240, 416), True)])
def test_staged_pipeline_matches_single_stream_engine(pkg, wdir, monkeypatch, src_hw, host):
    """Staged mode (chains=-1): the whole batch per launch, backbone on the main stream and neck + Detect on a second
    one, consecutive batches in alternate arena copies.  Detections of pipelined batches equal the plain engine's on
    a common tile; after an odd number of batches the debug reads come from the second arena copy and every stored
    layer still matches the oracle."""
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_BNECK", "0")
    B, steps = 8, (23 if not host else 5)          # the long run: every arena copy and ring slot reused many times, S + 1 batches in flight throughout
    h, w = src_hw
    frames = pkg.synth.frames(B * steps, h, w, seed=91).reshape(steps, B, h, w, 3)
    buf = None
    if not host:
        buf = pkg._ffi.DeviceBuffer(frames.nbytes)
        buf.upload(frames)
    per = h * w * 3
    names = [c.name for c in pkg.weights.spec("s")]

    def run(chains):
        det, wts = make_detector(pkg, wdir, "s", 320, batch=B, autotune=False, confidence=0.02, chains=chains)
        want = 1 - chains if chains < 0 else 1
        assert det.model.chains == 1 and (det.model.stages == want or (want == 3 and det.model.stages == 1)), det.model.stages   # 3 stages need every queue of the process
        outs = []
        depth = det.model.stages + 1 if det.model.stages > 1 else 2      # batches in flight: S stages are run S + 1 deep
        for t in range(steps):
            if host:
                det.enqueue(list(frames[t]))
            else:
                det.enqueue([buf.ptr + (t * B + i) * per for i in range(B)], height=h, width=w)
            if t >= depth - 1:
                outs.append(det.fetch())
        for _ in range(depth - 1):
            outs.append(det.fetch())
        if chains < 0 and src_hw == (320, 320):                         # 5 (23) batches so far; the next runs in arena copy 5 % S or 23 % S (1 or 2)
            if host:
                det.enqueue(list(frames[0]))
            else:
                det.enqueue([buf.ptr + i * per for i in range(B)], height=h, width=w)
            det.fetch()
            inp, _, _ = det.debug_fetch(B - 1, want_heads=False, want_pred=False)
            gpu = fetch_layers(pkg, det, names, B - 1)
            taps = {}
            Y.forward(inp.astype(np.float32), wts, "s", taps=taps, force=gpu)
            for n in gpu:
                assert float(np.abs(taps[n] - gpu[n]).max()) <= 2e-3 * np.abs(taps[n]).max() + 2e-3, n
            assert np.array_equal(inp.astype(np.float32), Y.preprocess(frames[0][B - 1], 320, 320).astype(np.float16).astype(np.float32))
        det.close()
        return outs

    ref = run(1)
    assert sum(len(d) for batch in ref for d in batch) > 0
    for chains in (-1, -2):
        got = run(chains)
        for t in range(steps):
            for i in range(B):
                a, b = got[t][i], ref[t][i]
                assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32)) and np.array_equal(a.class_id, b.class_id) and \
                    np.array_equal(a.confidence.view(np.int32), b.confidence.view(np.int32)), (chains, t, i)
    if buf is not None:
        buf.free()


@pytest.mark.parametrize("size,batch", [(320, 2), (288, 1)])
def test_neck_concat_read_from_half_resolution(pkg, wdir, monkeypatch, size, batch):
    """Neck Upsample + Concat (layers 10/11, 13/14): by default C2f.cv1 of layers 12 and 15 reads the upsampled channels
    straight from the half-resolution producer (ConvLaunch::in_lo) and no upsampled copy is written; with
    RTMODT_UP_READ=0 the producer's epilogue writes the copy into the concat slice.  On a common 64-deep tile the two
    give bit-identical layers, and both match the oracle (odd map sizes at 288: 9 -> 18 -> 36)."""
    monkeypatch.setenv("RTMODT_TILE", "2")
    monkeypatch.setenv("RTMODT_TILE_K64", str(T["K64_128x64_S3_W8"]))
    monkeypatch.setenv("RTMODT_BNECK", "0")
    frames = list(pkg.synth.frames(batch, size, size, seed=57))
    names = [c.name for c in pkg.weights.spec("s")]
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("RTMODT_UP_READ", mode)
        det, w = make_detector(pkg, wdir, "s", size, autotune=False, batch=batch)
        det.detect_batch(frames)
        outs[mode] = []
        for img in range(batch):
            inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
            gpu = fetch_layers(pkg, det, names, img)
            outs[mode].append(gpu)
            taps = {}
            Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
            for n in ("12.cv1", "15.cv1", "12.cv2", "9.cv2"):
                assert float(np.abs(taps[n] - gpu[n]).max()) <= 2e-3 * np.abs(taps[n]).max() + 2e-3, (mode, img, n)
        det.close()
    for img in range(batch):
        for n in outs["1"][img]:
            assert np.array_equal(outs["0"][img][n], outs["1"][img][n]), (img, n)


def test_benchmarked_shape_parity(pkg, wdir):
    """BASELINE config 4's per-GPU shard exactly as bench.py builds it: YOLOv8s @ 640, batch 32 (8 streams x 4 consecutive
    frames), autotuned tiles, three-stage engine, S + 1 batches in flight, tracker fed on the device.
      * every stored layer of images 0 and 15 of the last batch within fp16 tolerance of the fp32 oracle (teacher-forced);
      * NMS survivors, boxes, scores, classes of all 16 images bit-equal to oracle NMS on the engine's own pre-NMS tensor;
      * pipelined enqueue/fetch results bit-equal to a synchronous detect_batch of the same frames on the same engine;
      * the 8 streams' tracker state bit-equal to TrackerOracle fed the fetched detections frame by frame."""
    from oracle import tracker_oracle as T
    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    S, F, steps, size = 8, 4, 6, 640
    B = S * F
    det, w = make_detector(pkg, wdir, "s", size, batch=B, autotune=True, chains=-2, max_det=100)
    assert det.model.stages == 3, "the bench configuration needs all four hardware queues of the process"
    frames = np.stack([pkg.synth.frames(F * steps, size, size, seed=1234 + s) for s in range(S)], 1).reshape(steps, F, S, size, size, 3)
    buf = pkg._ffi.DeviceBuffer(frames.nbytes)
    buf.upload(frames)
    per = size * size * 3
    core = core_cls(n_streams=S, max_dets=128, max_tracks=2048)
    oracles = [T.TrackerOracle() for _ in range(S)]
    depth = det.model.stages + 1
    outs = []

    def submit(t):
        det.enqueue([buf.ptr + ((t * F + f) * S + s) * per for f in range(F) for s in range(S)], height=size, width=size)
        core.update_from_detector(det, 0, S, frames_per_stream=F)          # one launch, the F frames of a stream in order inside its workgroup

    for t in range(steps):
        submit(t)
        if t >= depth - 1:
            outs.append(det.fetch())
    for _ in range(depth - 1):
        outs.append(det.fetch())
    assert len(outs) == steps
    # tracker: device hand-off == oracle on the fetched detections
    for t in range(steps):
        for f in range(F):
            for s in range(S):
                d = outs[t][f * S + s]
                oracles[s].update(d.xyxy, d.confidence, d.class_id)
    for s in range(S):
        assert np.array_equal(T.state_digest(core.snapshot(s)), T.state_digest(oracles[s].snapshot())), s
    assert sum(len(core.snapshot(s)["ids"]) for s in range(S)) > 0
    # the newest batch is still resident: pre-NMS tensor -> oracle NMS, all 16 images
    last = outs[-1]
    for i in range(B):
        _, _, pred = det.debug_fetch(i, want_input=False, want_heads=False)
        dets, _ = Y.non_max_suppression(pred, 0.35, 0.45, None, False, 100)
        ref = Y.scale_boxes(dets[:, :4], size, size, size, size) if len(dets) else np.empty((0, 4), np.float32)
        assert len(last[i]) == len(dets), i
        assert np.array_equal(last[i].xyxy.view(np.int32), ref.view(np.int32)), i
        assert np.array_equal(last[i].confidence.view(np.int32), dets[:, 4].astype(np.float32).view(np.int32)), i
        assert last[i].class_id.tolist() == dets[:, 5].astype(np.int32).tolist(), i
    assert sum(len(d) for d in last) > 16
    # layers of images 0 and 15 against the fp32 oracle
    names = [c.name for c in pkg.weights.spec("s")]
    loose = {"2.cv1", "4.cv1", "6.cv1", "8.cv1", "2.cv2", "15.cv2"}       # consumers of an LDS-resident fp16 intermediate when the tuner fused the pair
    for img in (0, B - 1):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        f, s = divmod(img, S)
        assert np.array_equal(inp.astype(np.float32), Y.preprocess(frames[steps - 1][f][s], size, size).astype(np.float16).astype(np.float32))
        gpu = fetch_layers(pkg, det, names, img)
        assert len(gpu) >= 50
        taps = {}
        Y.forward(inp.astype(np.float32), w, "s", taps=taps, force=gpu)
        worst = {"plain": ("", 0.0), "behind an LDS-resident fp16 intermediate": ("", 0.0)}
        for n in gpu:
            tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}"
            key = "behind an LDS-resident fp16 intermediate" if (n in loose or (".m." in n and n.endswith(".cv2"))) else "plain"
            if err / tol > worst[key][1]:
                worst[key] = (n, err / tol)
        # rounds 1-4 allowed 4e-3 * max|ref| behind a fused pair's LDS-resident fp16 intermediate.  Measured (round 5, printed with -s): those
        # layers sit at 0.22 - 0.23 of the 2e-3 bound, the plain ones at 0.18 - 0.19 -- the factor 2 was never needed, so ONE tolerance class is left
        print(f"benchmarked shape, image {img}: worst err / tol (2e-3 * max|ref| + 2e-3 everywhere): " + "; ".join(f"{k}: {v[1]:.3f} ({v[0]})" for k, v in worst.items()))
    # the same frames synchronously through the same engine
    sync = det.detect_batch([frames[steps - 1][f][s] for f in range(F) for s in range(S)])
    for i in range(B):
        assert np.array_equal(sync[i].xyxy.view(np.int32), last[i].xyxy.view(np.int32)) and sync[i].class_id.tolist() == last[i].class_id.tolist(), i
    buf.free()
    det.close()


@pytest.mark.parametrize("h,w", [(640, 640), (480, 640)])
def test_page_locked_frames_read_in_place(pkg, wdir, monkeypatch, h, w):
    """RTMODT_ZERO_COPY=1: host frames in page-locked memory that need no resize are not copied, the stem kernel reads them
    over PCIe in place (engine.hip: in_place).  Stem output, input image and detections must equal, bit for bit, the staged-copy path
    (RTMODT_ZERO_COPY=0: one DMA for the contiguous ring slots) and the pageable path (16 separate copies); pipelined,
    with the ring slots of later batches in flight while earlier ones are read."""
    B, steps = 4, 5
    src = pkg.synth.structured_frames(B * steps, h, w, seed=h * 3 + w)
    ring = pkg.pipeline.PinnedFrameRing(B * steps, h, w)
    for i in range(B * steps):
        ring.write(i, src[i])
    outs = {}
    for mode in ("inplace", "copy", "pageable"):
        monkeypatch.setenv("RTMODT_ZERO_COPY", "0" if mode == "copy" else "1")
        det, _ = make_detector(pkg, wdir, "s", 640, autotune=False, batch=B, chains=-1, confidence=0.05)
        res = []
        for t in range(steps):
            det.enqueue([(src[t * B + i] if mode == "pageable" else ring.frame(t * B + i)) for i in range(B)])
            if t >= 2:
                res.append(det.fetch())
        while len(res) < steps:
            res.append(det.fetch())
        stem = [det.debug_layer("0", i) for i in range(B)]
        inp = [det.debug_fetch(i, want_heads=False, want_pred=False)[0] for i in range(B)]
        outs[mode] = (res, stem, inp)
        det.close()
    want = Y.preprocess(src[(steps - 1) * B + 1], 640, 640).astype(np.float16)
    assert np.array_equal(outs["inplace"][2][1].view(np.uint16), want.view(np.uint16))
    assert sum(len(d) for batch in outs["inplace"][0] for d in batch) > 0
    for mode in ("copy", "pageable"):
        for t in range(steps):
            for i in range(B):
                a, b = outs["inplace"][0][t][i], outs[mode][0][t][i]
                assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32)) and np.array_equal(a.confidence.view(np.int32), b.confidence.view(np.int32)) and \
                    a.class_id.tolist() == b.class_id.tolist(), (mode, t, i)
        for i in range(B):
            assert np.array_equal(outs["inplace"][1][i].view(np.uint16), outs[mode][1][i].view(np.uint16)), (mode, i)
            assert np.array_equal(outs["inplace"][2][i].view(np.uint16), outs[mode][2][i].view(np.uint16)), (mode, i)
    ring.close()


def test_autotune_cache_rejects_foreign_and_illegal_entries(pkg, wdir, monkeypatch, tmp_path):
    """A cache written for another tile table (no / different header line) is ignored as a whole, and a record whose tile id
    is not legal for its launch (a tap-reuse tile on a 1x1 conv, a tail tile as a plain tile) is re-tuned instead of applied."""
    cache = tmp_path / "tune.txt"
    monkeypatch.setenv("RTMODT_TUNE_CACHE", str(cache))
    a, w = make_detector(pkg, wdir, "n", 320, autotune=True, batch=2)
    good = cache.read_text().splitlines()
    a.close()
    # (1) headerless file with every record pointing at a tap-reuse tile (illegal for 1x1 and stride-2 convs)
    rows, tail = T["ROWS_K64_64x64"], T["TAIL_K64_128x128"]
    cache.write_text("\n".join(f"{l.split()[0]}\t{rows} {rows} 1" for l in good[1:]) + "\n")
    b, _ = make_detector(pkg, wdir, "n", 320, autotune=True, batch=2)
    assert cache.read_text().splitlines()[0] == good[0]              # ignored, re-tuned, rewritten with the header
    frames = list(pkg.synth.frames(2, 320, 320, seed=5))
    ref = b.detect_batch(frames)
    b.close()
    # (2) right header, illegal ids: every hit is re-checked
    cache.write_text(good[0] + "\n" + "\n".join(f"{l.split()[0]}\t{rows} {tail} 1" for l in good[1:]) + "\n")
    c, _ = make_detector(pkg, wdir, "n", 320, autotune=True, batch=2)
    inp, _, _ = c.debug_fetch(0, want_heads=False, want_pred=False) if c.detect_batch(frames) else (None, None, None)
    names = [x.name for x in pkg.weights.spec("n")]
    gpu = fetch_layers(pkg, c, names, 0)
    taps = {}
    Y.forward(inp.astype(np.float32), w, "n", taps=taps, force=gpu)
    for n in gpu:
        assert float(np.abs(taps[n] - gpu[n]).max()) <= 4e-3 * np.abs(taps[n]).max() + 2e-3, n
    c.close()
    assert len(ref) == 2


@pytest.mark.parametrize("tile,size,batch,up_read,scale", [(T["PT_128x128_S2"], 320, 32, "0", "s"), (T["PT_128x64_S2"], 320, 32, "1", "s"),
                                                                 (T["PT_128x128_S2"], 640, 8, "1", "s"), (T["PT_128x64_S2"], 288, 3, "0", "s"),
                                                                 (T["PT_128x128_S2"], 64, 8, "1", "s"), (T["PT_128x64_S2"], 64, 24, "0", "s"),      # 64 x 64: one to twelve pixel tiles per conv -- fewer workgroups than XCDs
                                                                 (T["PT_128x128_S2"], 320, 8, "1", "m"), (T["PT_128x64_S2"], 320, 16, "1", "n")])   # other channel counts (m: 192 / 384 / 576, n: 64 / 128 / 256)
def test_persistent_tile_kernel(pkg, wdir, monkeypatch, tile, size, batch, up_read, scale):
    """conv_mfma64_pt: a persistent workgroup walks over pixel tiles of one cout slice; the stage ring keeps prefetching
    across tile boundaries and the epilogue stores straight from the accumulators.  Forced onto every conv where it is legal
    (any kernel size / stride with cin % 64 == 0, full tiles, no second destination; the Bottleneck shortcuts and -- with
    `up_read` -- the neck's half-resolution concat sources included); all layers of the first and the last image against the
    oracle.  288 x 288 x 3: legal nowhere, every conv keeps its default."""
    monkeypatch.setenv("RTMODT_TILE_K64", str(tile))
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    monkeypatch.setenv("RTMODT_TILE_3X3S1", "-1")            # (no tap-reuse tiles: the 3x3 convs take the forced tile too)
    monkeypatch.setenv("RTMODT_UP_READ", up_read)
    det, w = make_detector(pkg, wdir, scale, size, autotune=False, batch=batch)
    used = [n for n, _, _ in det.profile(1) if "pt:" in n]
    assert (len(used) >= (8 if scale == "s" else 4)) if size != 288 else (len(used) == 0), used
    frames = list(pkg.synth.frames(batch, size, size, seed=17 + tile))
    det.detect_batch(frames)
    names = [c.name for c in pkg.weights.spec(scale)]
    for img in sorted({0, batch - 1}):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        gpu = fetch_layers(pkg, det, names, img)
        taps = {}
        Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
        for n in gpu:
            tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"tile {tile} img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}"
    det.close()


def test_converted_checkpoint_through_the_engine(pkg, wdir, tmp_path):
    """SURVEY 8f rank 2 on the GPU (VERDICT r02 missing 1): an Ultralytics-shaped `.pt` (fake class paths, fp16 tensors,
    non-trivial BatchNorm statistics) -> `weights.convert_pt` (restricted unpickler, BN fold, NHWC fp16) -> RTMODTW1 -> the HIP
    engine, checked against torch running the checkpoint's own UNFOLDED Conv + BatchNorm + SiLU graph
    (reference: src/detection/detector.py:82-90 -- `YOLO(path)` loads the .pt and runs it as is):
      * every stored layer of the engine vs the oracle on the converted file, teacher-forced, usual layer tolerance;
      * free-running head logits vs the unfolded torch model: p99 < 0.05, max < 0.5 (the e2e bars);
      * decode + NMS: every decision on the engine's tensor vs the unfolded model's tensor (nms_audit), none hard."""
    import torch
    import nms_audit as NA
    from fake_ultralytics import forward_heads, make_checkpoint
    model, rtw, _ = make_checkpoint(pkg, tmp_path, "n", 320)
    w, scale, nc, _ = pkg.weights.load(rtw)
    det = pkg.Detector(rtw, input_size=(320, 320), max_det=300, warmup=False)
    assert det.model.scale == "n" and det.model.nc == 80
    frame = pkg.synth.frames(1, 320, 320, seed=77)[0]
    d, heads, pred, worst = layer_check(pkg, det, w, frame, "n", 320)
    print("worst layer vs the folded oracle", worst)
    inp, _, _ = det.debug_fetch(0, want_heads=False, want_pred=False)
    with torch.no_grad():
        ref = [h[0].permute(1, 2, 0).numpy() for h in forward_heads(model, torch.from_numpy(np.ascontiguousarray(inp.astype(np.float32).transpose(2, 0, 1)))[None])]
    off, errs = 0, []
    for lvl, s_ in enumerate((40, 20, 10)):
        g = heads[off:off + s_ * s_ * 144].reshape(s_, s_, 144).astype(np.float32); off += s_ * s_ * 144
        errs.append(np.abs(g - ref[lvl]).ravel())
    errs = np.concatenate(errs)
    print(f"free-running engine on the converted file vs the unfolded torch model: head logits max {errs.max():.4f} p99 {np.percentile(errs, 99):.4f}")
    assert np.percentile(errs, 99) < 0.05 and errs.max() < 0.5
    pr = Y.decode(ref)
    res = NA.audit(pred, pr, 0.35, 0.45, None, score_tol=0.005, iou_tol=0.008)
    print(NA.describe(res))
    assert res["n_candidates"] > 50 and not res["hard"], NA.describe(res, 40)
    dets, anch = Y.non_max_suppression(pr, 0.35, 0.45, None, False, 300)
    assert len(d) > 5 and abs(len(d) - len(dets)) <= max(3, len(dets) // 10)
    det.close()



@pytest.mark.parametrize("tile,size,batch,scale", [(T["PP_256x128"], 320, 32, "s"), (T["PP_256x64"], 320, 32, "s"), (T["PP_256x192"], 320, 32, "s"), (T["PP_512x64"], 320, 32, "s"), (T["PP_256x128"], 288, 3, "s"), (T["PP_256x64"], 640, 4, "s"), (T["PP_512x64"], 288, 3, "s"), (T["PP_512x64"], 640, 2, "s"), (T["PP_256x128"], 320, 8, "m"), (T["PP_256x64"], 320, 5, "n"), (T["PP_256x192"], 288, 3, "s"), (T["PP_256x192"], 640, 2, "s"), (T["PP_256x192"], 320, 4, "m")])
def test_ping_pong_3x3_kernel(pkg, wdir, monkeypatch, tile, size, batch, scale):
    """conv3x3_pp (TILE_PP_*, csrc/conv_pp.hip): the 3x3 / stride-1 kernel whose two wave halves run one barrier interval apart, forced onto
    every conv where it is legal -- Bottlenecks with their shortcuts (fp32 staging + 16-byte shortcut reads), the grouped Detect launches (a
    workgroup walks tiles of several problems; with the balanced schedule of pp_lpt_schedule), the 192-wide form on Detect stage 0, the 512-position
    form (pp:512x64: every wave 64 positions x all 64 couts) -- at batches
    where a workgroup runs several tiles, at 288 x 288 (partial tiles, fewer tiles than workgroups) and on the n / m widths.  Every stored layer
    of the first and the last image against the oracle fed the engine's own inputs."""
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    monkeypatch.setenv("RTMODT_TILE_3X3S1", str(tile))
    frames = list(pkg.synth.frames(batch, size, size, seed=41 + tile))
    names = [c.name for c in pkg.weights.spec(scale)]
    det, w = make_detector(pkg, wdir, scale, size, autotune=False, batch=batch, confidence=0.05)
    used = [n for n, _, _ in det.profile(1) if "pp:" in n]
    assert len(used) >= (1 if tile == T["PP_256x192"] else 6), [n for n, _, _ in det.profile(1)]
    det.detect_batch(frames)
    for img in sorted({0, batch - 1}):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        gpu = fetch_layers(pkg, det, names, img)
        taps = {}
        Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
        for n in gpu:
            tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"tile {tile} img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}; launches: {used}"
    det.close()


@pytest.mark.parametrize("tile,size,batch,up_read,scale", [(T["PPT_256x128"], 320, 32, "1", "s"), (T["PPT_256x128"], 320, 32, "0", "s"), (T["PPT_256x128"], 288, 3, "1", "s"), (T["PPT_256x128"], 640, 2, "1", "s"), (T["PPT_256x128"], 320, 8, "1", "m")])
def test_ping_pong_tile_kernel(pkg, wdir, monkeypatch, tile, size, batch, up_read, scale):
    """conv_tile_pp (TILE_PPT_*): the ping-pong schedule without tap reuse -- 1x1 convs (incl. the neck layers that read their upsampled channels
    from the half-resolution tensor, up_read = 1) and the 3x3 / stride-2 convs -- forced wherever it is legal (cin % 64 == 0, K >= 192, no shortcut);
    persistent workgroups over several tiles, partial tiles at 288 x 288.  Every stored layer of the first and the last image against the oracle."""
    monkeypatch.setenv("RTMODT_BNECK", "0")
    monkeypatch.setenv("RTMODT_TAIL", "0")
    monkeypatch.setenv("RTMODT_UP_READ", up_read)
    monkeypatch.setenv("RTMODT_TILE_K64", str(tile))
    monkeypatch.setenv("RTMODT_TILE_3X3S1", "-1")            # (no tap-reuse tiles: the 3x3 / stride-1 convs without a shortcut take the forced tile too)
    frames = list(pkg.synth.frames(batch, size, size, seed=61 + tile))
    names = [c.name for c in pkg.weights.spec(scale)]
    det, w = make_detector(pkg, wdir, scale, size, autotune=False, batch=batch, confidence=0.05)
    used = [n for n, _, _ in det.profile(1) if "ppt:" in n]
    assert len(used) >= 8, [n for n, _, _ in det.profile(1)]
    det.detect_batch(frames)
    for img in sorted({0, batch - 1}):
        inp, _, _ = det.debug_fetch(img, want_heads=False, want_pred=False)
        gpu = fetch_layers(pkg, det, names, img)
        taps = {}
        Y.forward(inp.astype(np.float32), w, scale, taps=taps, force=gpu)
        for n in gpu:
            tol = 2e-3 * np.abs(taps[n]).max() + 2e-3
            err = float(np.abs(taps[n] - gpu[n]).max())
            assert err <= tol, f"tile {tile} img {img} layer {n}: max err {err:.4g} > tol {tol:.4g}; launches: {used}"
    det.close()

def test_in_kernel_clock_sampling(pkg, wdir):
    """rtmodt_detector_clock_enable / _read: one wave behind every batch's NMS reads the shader-cycle counter against the constant
    100 MHz counter; the clock must be a plausible gfx950 shader clock, one sample per batch, and sampling must not change results."""
    det, _ = make_detector(pkg, wdir, "s", 320, batch=4, autotune=False)
    frames = list(pkg.synth.frames(4, 320, 320, seed=3))
    ref = det.detect_batch(frames)
    det.clock_sampling(True)
    for _ in range(6):
        got = det.detect_batch(frames)
    mean, lo, hi, n = det.clock_read()
    assert n == 6 and 0.3 < lo <= mean <= hi < 2.6, (mean, lo, hi, n)
    assert det.clock_read()[3] == 0                                   # read resets
    det.clock_sampling(False)
    det.detect_batch(frames)
    assert det.clock_read()[3] == 0
    for a, b in zip(ref, got):
        assert np.array_equal(a.xyxy.view(np.int32), b.xyxy.view(np.int32)) and a.class_id.tolist() == b.class_id.tolist()
    det.close()
