"""Decision-level audit of NMS on two pre-NMS tensors of the same image (test helper, not a test).

`non_max_suppression` is a chain of discrete decisions over continuous inputs:
  (1) per anchor: best class (first max), candidate iff its score > conf (strict) and its class is kept;
  (2) per pair of same-class candidates: does IoU exceed the threshold (strict);
  (3) per overlapping pair: which of the two is walked first (score descending, anchor index ascending).
Given the same answers to (1)-(3) the survivor set is the same, whatever the tensors hold.  An fp16 engine and
an fp32 oracle give slightly different tensors, so some answers may differ -- legitimately only where the
ORACLE's margin for that decision is inside the numerical drift between the nets (a "near-tie"); one flipped
near-tie may then cascade through any number of later suppressions (A kept instead of dropped -> B dropped ->
C kept ...).  The one-hop explainer this replaces could not follow such a chain and called it a defect.

`audit()` therefore compares EVERY decision, not survivors: it returns the decisions that differ, split into
near-ties (allowed, reported) and hard mismatches (a decision whose oracle margin is larger than the drift
tolerance still came out differently: the engine's tensor is wrong beyond fp16 drift -- a kernel defect).
If `hard` is empty, the engine's survivors are exactly NMS of the oracle's tensor with only near-tie
decisions flipped, which is the strongest statement an fp16-vs-fp32 comparison can make.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def _best(pred, nc):
    cls = pred[4:4 + nc]
    k = np.argmax(cls, axis=0)                                  # first max
    return k, cls[k, np.arange(pred.shape[1])].astype(F32)


def _xyxy(pred, idx):
    p = pred[:4, idx].T.astype(F32)
    half = p[:, 2:4] / F32(2)
    return np.concatenate([p[:, 0:2] - half, p[:, 0:2] + half], axis=1).astype(F32)


def _iou(b):
    """Pairwise IoU, float32, the arithmetic of torchvision.ops.nms (no eps)."""
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
    x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(inter > 0, inter / (area[:, None] + area[None, :] - inter), F32(0)).astype(F32)


def audit(pred_e, pred_o, conf=0.35, iou_thr=0.45, classes=None, agnostic=False, nc=80, score_tol=0.01, iou_tol=0.02):
    """Engine tensor `pred_e` vs oracle tensor `pred_o`, both (4+nc, A) float32.

    Returns a dict:
      near   list of (kind, anchors, oracle margin)  decisions that differ inside the tolerance
      hard   same, outside the tolerance              (must be empty)
      n_candidates, n_pairs                           how much was audited
      iou_drift, score_drift                          max |engine - oracle| over the audited candidates / overlapping pairs
    kinds: "member" (score vs conf), "class" (best class), "overlap" (IoU vs threshold), "order" (walk order).
    """
    conf, iou_thr = F32(conf), F32(iou_thr)
    ke, se = _best(pred_e, nc)
    ko, so = _best(pred_o, nc)
    keep_cls = None if classes is None else np.asarray(classes)

    def member(k, s):
        m = s > conf
        if keep_cls is not None:
            m &= np.isin(k, keep_cls)
        return m

    me, mo = member(ke, se), member(ko, so)
    near, hard = [], []
    A = pred_e.shape[1]
    an = np.arange(A)
    # (1a) best class: the engine's class must be the oracle's, or its oracle score within score_tol of the oracle's best
    for a in np.nonzero((me | mo) & (ke != ko))[0]:
        gap = float(so[a] - pred_o[4 + ke[a], a])
        (near if gap <= 2 * score_tol else hard).append(("class", (int(a),), gap))
    # (1b) membership
    for a in np.nonzero(me != mo)[0]:
        if ke[a] != ko[a]:
            continue                                             # judged as a class decision above
        gap = abs(float(so[a]) - float(conf))
        (near if gap <= score_tol else hard).append(("member", (int(a),), gap))
    # (2), (3): pairs among the union of candidates, same class as the ENGINE walks them
    cand = np.nonzero(me | mo)[0]
    n = len(cand)
    out = dict(near=near, hard=hard, n_candidates=int(n), n_pairs=0, iou_drift=0.0,
               score_drift=float(np.abs(se[cand] - so[cand]).max(initial=0.0)))
    if n < 2:
        return out
    be, bo = _xyxy(pred_e, cand), _xyxy(pred_o, cand)
    ie, io = _iou(be), _iou(bo)
    kc = ke[cand]
    same = np.ones((n, n), bool) if agnostic else (kc[:, None] == kc[None, :])      # (an anchor whose class flipped is already recorded above)
    upper = np.triu(np.ones((n, n), bool), 1) & same
    ov_e, ov_o = ie > iou_thr, io > iou_thr
    out["n_pairs"] = int(upper.sum())
    touched = upper & (ov_e | ov_o)
    out["iou_drift"] = float(np.abs(ie - io)[touched].max(initial=0.0))
    for i, j in zip(*np.nonzero(upper & (ov_e != ov_o))):
        gap = abs(float(io[i, j]) - float(iou_thr))
        (near if gap <= iou_tol else hard).append(("overlap", (int(cand[i]), int(cand[j])), gap))
    s_e, s_o = se[cand], so[cand]
    first_e = (s_e[:, None] > s_e[None, :]) | ((s_e[:, None] == s_e[None, :]) & (cand[:, None] < cand[None, :]))
    first_o = (s_o[:, None] > s_o[None, :]) | ((s_o[:, None] == s_o[None, :]) & (cand[:, None] < cand[None, :]))
    for i, j in zip(*np.nonzero(touched & (first_e != first_o))):
        gap = abs(float(s_o[i]) - float(s_o[j]))
        (near if gap <= 2 * score_tol else hard).append(("order", (int(cand[i]), int(cand[j])), gap))
    return out


def survivors_diff(anchors_e, anchors_o):
    """(only the engine keeps, only the oracle keeps) as sorted anchor lists."""
    e, o = set(int(a) for a in anchors_e), set(int(a) for a in anchors_o)
    return sorted(e - o), sorted(o - e)


def describe(res, limit=12):
    rows = [f"audited {res['n_candidates']} candidates / {res['n_pairs']} same-class pairs; score drift {res['score_drift']:.4f}, "
            f"IoU drift on overlapping pairs {res['iou_drift']:.4f}; near-tie decisions flipped: {len(res['near'])}, hard mismatches: {len(res['hard'])}"]
    for tag in ("hard", "near"):
        for kind, anchors, gap in res[tag][:limit]:
            rows.append(f"  {tag:4s} {kind:7s} anchors {anchors} oracle margin {gap:.5f}")
    return "\n".join(rows)
