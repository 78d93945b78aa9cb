"""CPU oracle of the OPT-IN Kalman motion model of the tracker -- TEST INFRASTRUCTURE, not product code.

The reference has no Kalman filter: ``/root/reference/src/tracking/tracker.py:99-104`` overwrites a matched track's
box with the detection, and ``requirements.txt:23`` lists filterpy "for DeepSORT" only (SURVEY finding 3).
BASELINE.json's ``north_star`` names "ByteTrack's batched Kalman predict/update", so the build offers it as an option
(``MultiObjectTracker(kalman=True)``, default off; parity mode is untouched).  PARITY UNPINNED by construction: there is
nothing in the reference to pin it to.  The algorithm is the published ByteTrack ``KalmanFilter`` (8-state constant
velocity over (cx, cy, a, h), ``_std_weight_position = 1/20``, ``_std_weight_velocity = 1/160``; ``initiate`` / ``predict``
/ ``project`` / ``update``), restated twice:

* :class:`KalmanFull64` -- the textbook 8x8 matrix form in float64 (what the published code evaluates);
* :func:`kf_initiate` / :func:`kf_predict` / :func:`kf_update` -- the form the HIP kernel runs: F, H, Q, R and the
  initial P are such that P stays block-diagonal (one 2x2 block [[a, b], [b, c]] per coordinate), so the 8-state filter
  is four independent (position, velocity) filters whose only coupling is the noise scale ``h``.  float32, one rounding
  per operation, in the exact order written here (the kernel is built with FMA contraction off and correctly rounded
  division, so it matches bit for bit).  ``tests/test_oracle_kalman.py`` checks this form against the matrix form.

:class:`TrackerOracleKalman` is :class:`oracle.tracker_oracle.TrackerOracle` with the motion model switched on:
every track is predicted at the start of a frame (a track that was not matched in the previous frame has its height
velocity zeroed first, as ByteTrack's ``multi_predict`` does), association runs on the PREDICTED boxes, a matched track
is corrected with the detection, a new track is initiated from it.  ``xyxy`` keeps the reference's meaning (the last
matched detection, tracker.py:99); the filter state is an extra (mean[8], cov[12]) per track.
"""
from __future__ import annotations

import numpy as np

from .tracker_oracle import F32, TrackerOracle, batch_iou

WP = F32(0.05)            # _std_weight_position = 1 / 20
WV = F32(0.00625)         # _std_weight_velocity = 1 / 160
A_INIT_P, A_INIT_V = F32(1e-2), F32(1e-5)      # aspect-ratio std at initiate
A_PRED_P, A_PRED_V = F32(1e-2), F32(1e-5)      # ... in predict
A_PROJ = F32(1e-1)                              # ... in project


def xyxy_to_xyah(b):
    """(x1, y1, x2, y2) -> (cx, cy, a, h), float32, op by op (ByteTrack tlwh_to_xyah on tlwh = (x1, y1, w, h))."""
    b = np.asarray(b, F32).reshape(-1, 4)
    w = b[:, 2] - b[:, 0]
    h = b[:, 3] - b[:, 1]
    cx = b[:, 0] + w * F32(0.5)
    cy = b[:, 1] + h * F32(0.5)
    a = w / np.maximum(h, F32(1e-6))
    return np.stack([cx, cy, a, h], 1).astype(F32)


def xyah_to_xyxy(m):
    m = np.asarray(m, F32).reshape(-1, 4)
    w = m[:, 2] * m[:, 3]
    x1 = m[:, 0] - w * F32(0.5)
    y1 = m[:, 1] - m[:, 3] * F32(0.5)
    return np.stack([x1, y1, x1 + w, y1 + m[:, 3]], 1).astype(F32)


def _sq(x):
    return (x * x).astype(F32)


def kf_initiate(z):
    """z (n,4) xyah -> mean (n,8), cov (n,12) = per coordinate (a, b, c) of [[a, b], [b, c]]."""
    z = np.asarray(z, F32).reshape(-1, 4)
    n = z.shape[0]
    mean = np.zeros((n, 8), F32)
    mean[:, :4] = z
    h = z[:, 3]
    sp = (F32(2) * WP) * h
    sv = (F32(10) * WV) * h
    cov = np.zeros((n, 4, 3), F32)
    for k in (0, 1, 3):
        cov[:, k, 0] = _sq(sp)
        cov[:, k, 2] = _sq(sv)
    cov[:, 2, 0] = A_INIT_P * A_INIT_P
    cov[:, 2, 2] = A_INIT_V * A_INIT_V
    return mean, cov.reshape(n, 12)


def kf_predict(mean, cov):
    mean = np.array(mean, F32).reshape(-1, 8)
    cov = np.array(cov, F32).reshape(-1, 4, 3)
    h = mean[:, 3].copy()
    sp, sv = WP * h, WV * h
    for k in range(4):
        qp = _sq(sp) if k != 2 else np.full_like(h, A_PRED_P * A_PRED_P)
        qv = _sq(sv) if k != 2 else np.full_like(h, A_PRED_V * A_PRED_V)
        a, b, c = cov[:, k, 0].copy(), cov[:, k, 1].copy(), cov[:, k, 2].copy()
        mean[:, k] = mean[:, k] + mean[:, 4 + k]
        cov[:, k, 0] = ((a + (b + b)) + c) + qp
        cov[:, k, 1] = b + c
        cov[:, k, 2] = c + qv
    return mean, cov.reshape(-1, 12)


def kf_update(mean, cov, z):
    mean = np.array(mean, F32).reshape(-1, 8)
    cov = np.array(cov, F32).reshape(-1, 4, 3)
    z = np.asarray(z, F32).reshape(-1, 4)
    h = mean[:, 3].copy()
    sp = WP * h
    for k in range(4):
        r = _sq(sp) if k != 2 else np.full_like(h, A_PROJ * A_PROJ)
        a, b, c = cov[:, k, 0].copy(), cov[:, k, 1].copy(), cov[:, k, 2].copy()
        s = a + r
        k0 = a / s
        k1 = b / s
        y = z[:, k] - mean[:, k]
        mean[:, k] = mean[:, k] + k0 * y
        mean[:, 4 + k] = mean[:, 4 + k] + k1 * y
        cov[:, k, 0] = a - k0 * a
        cov[:, k, 1] = b - k0 * b
        cov[:, k, 2] = c - k1 * b
    return mean, cov.reshape(-1, 12)


class KalmanFull64:
    """The published ByteTrack KalmanFilter, 8x8 matrices, float64 (one track at a time)."""

    def __init__(self):
        self.F = np.eye(8)
        for i in range(4):
            self.F[i, 4 + i] = 1.0
        self.H = np.eye(4, 8)
        self.wp, self.wv = 1.0 / 20, 1.0 / 160

    def initiate(self, z):
        mean = np.r_[np.asarray(z, np.float64), np.zeros(4)]
        h = z[3]
        std = [2 * self.wp * h, 2 * self.wp * h, 1e-2, 2 * self.wp * h, 10 * self.wv * h, 10 * self.wv * h, 1e-5, 10 * self.wv * h]
        return mean, np.diag(np.square(std))

    def predict(self, mean, cov):
        h = mean[3]
        std = [self.wp * h, self.wp * h, 1e-2, self.wp * h, self.wv * h, self.wv * h, 1e-5, self.wv * h]
        return self.F @ mean, self.F @ cov @ self.F.T + np.diag(np.square(std))

    def update(self, mean, cov, z):
        h = mean[3]
        R = np.diag(np.square([self.wp * h, self.wp * h, 1e-1, self.wp * h]))
        S = self.H @ cov @ self.H.T + R
        K = cov @ self.H.T @ np.linalg.inv(S)
        y = np.asarray(z, np.float64) - self.H @ mean
        return mean + K @ y, cov - K @ S @ K.T


class TrackerOracleKalman(TrackerOracle):
    """TrackerOracle (tracker.py:43-148) + the opt-in motion model described in the module docstring."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.mean = np.zeros((0, 8), F32)
        self.cov = np.zeros((0, 12), F32)

    def _predict_all(self):
        if self.ids.shape[0] == 0:
            return
        lost = self.tsu >= 2                                  # not matched in the previous frame (a matched track leaves with tsu == 1)
        self.mean[lost, 7] = F32(0)
        self.mean, self.cov = kf_predict(self.mean, self.cov)

    def update(self, xyxy, conf, cls):
        xyxy = np.asarray(xyxy, dtype=F32).reshape(-1, 4)
        conf = np.asarray(conf, dtype=F32).reshape(-1)
        cls = np.asarray(cls, dtype=np.int32).reshape(-1)
        self._predict_all()
        if conf.shape[0] == 0:
            self.tsu = self.tsu + 1
            return np.nonzero(self.tsu == 0)[0]
        pred = xyah_to_xyxy(self.mean[:, :4])                  # association sees the predicted boxes
        hi = conf >= F32(self.track_thresh)
        hb, hc, hk = xyxy[hi], conf[hi], cls[hi]
        lb, lc, lk = xyxy[~hi], conf[~hi], cls[~hi]
        m = self.ids.shape[0]
        um_t, um_d = list(range(m)), list(range(hb.shape[0]))
        if m > 0 and hb.shape[0] > 0:
            mt, md, um_t, um_d = self._assign(batch_iou(pred, hb), self.match_thresh)
            self._commit(np.asarray(mt, np.int64), hb, hc, hk, md)
        if len(um_t) > 0 and lb.shape[0] > 0:
            rem = np.asarray(um_t, np.int64)
            mt2, md2, _, _ = self._assign(batch_iou(pred[rem], lb), self.match_thresh)
            self._commit(rem[np.asarray(mt2, np.int64)], lb, lc, lk, md2)
        k = len(um_d)
        if k:
            d = np.asarray(um_d, np.int64)
            self.ids = np.concatenate([self.ids, self.next_id + np.arange(k, dtype=np.int64)])
            self.next_id += k
            self.xyxy = np.concatenate([self.xyxy, hb[d]], 0)
            self.conf = np.concatenate([self.conf, hc[d]])
            self.cls = np.concatenate([self.cls, hk[d]])
            self.age = np.concatenate([self.age, np.ones(k, np.int32)])
            self.tsu = np.concatenate([self.tsu, np.zeros(k, np.int32)])
            m0, c0 = kf_initiate(xyxy_to_xyah(hb[d]))
            self.mean = np.concatenate([self.mean, m0], 0)
            self.cov = np.concatenate([self.cov, c0], 0)
        self.tsu = self.tsu + 1
        keep = self.tsu <= self.track_buffer
        self.ids, self.xyxy, self.conf = self.ids[keep], self.xyxy[keep], self.conf[keep]
        self.cls, self.age, self.tsu = self.cls[keep], self.age[keep], self.tsu[keep]
        self.mean, self.cov = self.mean[keep], self.cov[keep]
        return np.nonzero(self.tsu == 0)[0]

    def _commit(self, rows, boxes, confs, clss, det_idx):
        if len(det_idx) == 0:
            return
        d = np.asarray(det_idx, np.int64)
        self._overwrite(rows, boxes, confs, clss, det_idx)
        self.mean[rows], self.cov[rows] = kf_update(self.mean[rows], self.cov[rows], xyxy_to_xyah(boxes[d]))

    def snapshot(self):
        s = super().snapshot()
        s["mean"], s["cov"] = self.mean.copy(), self.cov.copy()
        return s
