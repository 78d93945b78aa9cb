"""The decision-level NMS audit (tests/nms_audit.py) that the free-running fp16-vs-fp32 GPU tests rely on must itself
be right: drift-sized perturbations may only produce near-tie flips (however far the cascade runs), and a perturbation
larger than the tolerance on a decision that matters must be reported as hard."""
import numpy as np

import nms_audit as NA
from oracle import yolo_oracle as Y


def dense_pred(seed, n=2100, frac=0.5):
    rng = np.random.default_rng(seed)
    pred = np.zeros((84, n), np.float32)
    pred[0] = rng.uniform(0, 640, n); pred[1] = rng.uniform(0, 640, n)
    pred[2] = rng.uniform(40, 200, n); pred[3] = rng.uniform(40, 200, n)
    hot = np.nonzero(rng.uniform(size=n) < frac)[0]
    cls = rng.integers(0, 4, n)
    pred[4:] = rng.uniform(0.0, 0.05, (80, n)).astype(np.float32)
    pred[4 + cls[hot], hot] = rng.uniform(0.30, 0.99, len(hot)).astype(np.float32)
    return pred, rng


def test_drift_sized_noise_gives_only_near_ties_and_explains_every_cascade():
    seen_diff = 0
    for seed in range(6):
        po, rng = dense_pred(seed)
        pe = po.copy()
        pe[:4] += rng.normal(0, 0.05, pe[:4].shape).astype(np.float32)              # ~fp16 box drift, px
        pe[4:] += rng.uniform(-0.003, 0.003, pe[4:].shape).astype(np.float32)        # score drift
        res = NA.audit(pe, po, 0.35, 0.45, None, score_tol=0.004, iou_tol=0.02)
        assert not res["hard"], NA.describe(res)
        assert res["score_drift"] <= 0.0031 and res["iou_drift"] < 0.02
        _, ae = Y.non_max_suppression(pe, 0.35, 0.45, None, False, 3000)
        _, ao = Y.non_max_suppression(po, 0.35, 0.45, None, False, 3000)
        only_e, only_o = NA.survivors_diff(ae, ao)
        seen_diff += len(only_e) + len(only_o)
        if only_e or only_o:
            assert res["near"], "survivors differ but no decision does"
    assert seen_diff > 0, "the sample never exercised a flip: make it denser"


def test_identical_decisions_imply_identical_survivors():
    """No flipped decision at all => the same survivors (the property the audit's verdict rests on)."""
    for seed in range(4):
        po, rng = dense_pred(100 + seed, n=800, frac=0.3)
        pe = po.copy()
        pe[:4] += rng.normal(0, 0.002, pe[:4].shape).astype(np.float32)
        pe[4:] *= np.float32(1.0 + 1e-6)
        res = NA.audit(pe, po, 0.35, 0.45, None, score_tol=0.004, iou_tol=0.02)
        _, ae = Y.non_max_suppression(pe, 0.35, 0.45, None, False, 3000)
        _, ao = Y.non_max_suppression(po, 0.35, 0.45, None, False, 3000)
        if not res["near"] and not res["hard"]:
            assert ae.tolist() == ao.tolist()


def test_defects_are_hard():
    po, rng = dense_pred(7)
    _, ao = Y.non_max_suppression(po, 0.35, 0.45, None, False, 3000)
    # (a) a confident survivor's score knocked below the threshold
    a = int(ao[0])
    pe = po.copy(); pe[4:, a] *= 0.1
    res = NA.audit(pe, po, 0.35, 0.45, None, score_tol=0.004, iou_tol=0.02)
    assert any(k in ("member", "class") and a in an for k, an, _ in res["hard"]), NA.describe(res)
    # (b) a survivor's box moved onto another survivor of its class (IoU 0 -> 1)
    k = np.argmax(po[4:, ao], axis=0)
    same = [i for i in range(1, len(ao)) if k[i] == k[0]]
    b = int(ao[same[0]])
    pe = po.copy(); pe[:4, b] = po[:4, a]
    res = NA.audit(pe, po, 0.35, 0.45, None, score_tol=0.004, iou_tol=0.02)
    assert any(kind == "overlap" and set(an) == {a, b} for kind, an, _ in res["hard"]), NA.describe(res)
    # (c) two overlapping candidates swap their walk order by more than the drift
    cand = np.nonzero(po[4:].max(0) > 0.35)[0]
    kc = np.argmax(po[4:, cand], axis=0)
    io = NA._iou(NA._xyxy(po, cand))
    i, j = next((i, j) for i in range(len(cand)) for j in range(i + 1, len(cand)) if kc[i] == kc[j] and io[i, j] > 0.5
                and abs(po[4 + kc[i], cand[i]] - po[4 + kc[j], cand[j]]) > 0.05)
    pe = po.copy()
    pe[4 + kc[i], cand[i]], pe[4 + kc[j], cand[j]] = po[4 + kc[j], cand[j]], po[4 + kc[i], cand[i]]
    res = NA.audit(pe, po, 0.35, 0.45, None, score_tol=0.004, iou_tol=0.02)
    assert any(kind == "order" for kind, _, _ in res["hard"]), NA.describe(res)
    # (d) class filter: a candidate whose best class leaves the kept set
    pe = po.copy(); pe[4 + 70, a] = 0.999
    res = NA.audit(pe, po, 0.35, 0.45, [0, 1, 2, 3], score_tol=0.004, iou_tol=0.02)
    assert any(kind == "class" and a in an for kind, an, _ in res["hard"]), NA.describe(res)
