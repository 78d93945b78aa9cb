"""CPU: the opt-in Kalman motion model's oracle (oracle/kalman_oracle.py).  The reference has no Kalman filter
(tracker.py:99-104 overwrites the box), so there is no fixture to pin it to: the decoupled float32 form the kernel runs is
checked against the textbook 8x8 float64 matrix form of the published ByteTrack filter, and the tracker oracle built on
it against constant-velocity ground truth."""
import numpy as np

from oracle import kalman_oracle as K
from oracle import tracker_oracle as T


def test_decoupled_f32_form_equals_matrix_form():
    rng = np.random.default_rng(3)
    full = K.KalmanFull64()
    for trial in range(20):
        z0 = np.array([rng.uniform(50, 600), rng.uniform(50, 600), rng.uniform(0.3, 2.0), rng.uniform(20, 200)])
        m64, p64 = full.initiate(z0)
        m32, c32 = K.kf_initiate(z0[None].astype(np.float32))
        v = rng.uniform(-3, 3, 4) * [1, 1, 0.002, 0.3]
        for step in range(25):
            m64, p64 = full.predict(m64, p64)
            m32, c32 = K.kf_predict(m32, c32)
            if step % 4 != 3:                                # every fourth frame is a miss: predict only
                z = z0 + v * (step + 1) + rng.normal(0, 0.5, 4) * [1, 1, 0.005, 1]
                m64, p64 = full.update(m64, p64, z)
                m32, c32 = K.kf_update(m32, c32, z[None].astype(np.float32))
            np.testing.assert_allclose(m32[0], m64, rtol=2e-4, atol=2e-3)
            blocks = c32.reshape(4, 3)
            for k in range(4):
                want = [p64[k, k], p64[k, 4 + k], p64[4 + k, 4 + k]]
                np.testing.assert_allclose(blocks[k], want, rtol=2e-3, atol=1e-9)
            off = p64.copy()                                 # the 8x8 covariance really is block-diagonal
            for k in range(4):
                off[k, k] = off[k, 4 + k] = off[4 + k, k] = off[4 + k, 4 + k] = 0
            assert np.abs(off).max() < 1e-9


def test_box_conversions_round_trip():
    b = np.array([[10, 20, 110, 220], [0.5, 0.25, 3.5, 9.25], [300, 300, 300, 340]], np.float32)
    np.testing.assert_allclose(K.xyah_to_xyxy(K.xyxy_to_xyah(b)), b, rtol=1e-6, atol=1e-4)


def accelerating_scene(n=12, frames=120, seed=11):
    """n boxes that start at rest and speed up by 0.4 % of their width per frame, up to 0.25 widths per frame."""
    rng = np.random.default_rng(seed)
    wh = rng.uniform(40, 80, (n, 2)).astype(np.float32)
    c = np.stack([np.linspace(100, 3000, n), rng.uniform(200, 900, n)], 1).astype(np.float64)
    sign = rng.choice([-1.0, 1.0], (n, 2)) * np.array([1.0, 0.2])
    out = []
    for f in range(frames):
        c = c + np.minimum(0.25, 0.004 * f) * wh * sign
        out.append(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32))
    return out, np.full(n, 0.9, np.float32), (np.arange(n) % 5).astype(np.int32)


def test_tracker_with_motion_model_follows_fast_objects():
    """The reference matches on IoU >= 0.8 between a track's LAST box and the detection (tracker.py:92-98): an object that
    moves more than ~0.11 box-widths per frame is lost every frame and respawned under a new id.  With the motion model the
    association sees the predicted box and the ids persist (the filter learns the velocity while the object is slow)."""
    boxes, conf, cls = accelerating_scene()
    n = len(conf)
    plain, kal = T.TrackerOracle(), K.TrackerOracleKalman()
    for f, b in enumerate(boxes):
        if f == 90:                                          # an empty frame: predict only
            plain.update(np.zeros((0, 4), np.float32), [], [])
            kal.update(np.zeros((0, 4), np.float32), [], [])
            continue
        plain.update(b, conf, cls)
        kal.update(b, conf, cls)
    assert plain.next_id - 1 > 10 * n                        # the parity tracker respawns everything once the boxes are fast
    assert kal.next_id - 1 == n, kal.next_id                 # ... the motion model never loses one
    live = kal.tsu == 1
    assert live.sum() == n and np.all(kal.age[live] == len(boxes) - 1)    # (the empty frame ages nothing: tracker.py:70-73)
    s = kal.snapshot()
    assert s["mean"].shape == (n, 8) and s["cov"].shape == (n, 12)
    true_v = (boxes[-1] - boxes[-2])[:, :2]
    np.testing.assert_allclose(s["mean"][:, 4:6], true_v, rtol=0.1, atol=0.5)      # learnt velocities


def test_static_scene_matches_the_parity_tracker():
    """With nothing moving the predicted boxes equal the last detections to within rounding, so ids, ages and tsu agree
    with the parity tracker (boxes are the detections themselves in both)."""
    rng = np.random.default_rng(5)
    boxes = np.concatenate([rng.uniform(0, 500, (20, 2)), rng.uniform(520, 640, (20, 2))], 1).astype(np.float32)
    conf = rng.uniform(0.3, 0.95, 20).astype(np.float32)
    cls = rng.integers(0, 3, 20).astype(np.int32)
    a, b = T.TrackerOracle(), K.TrackerOracleKalman()
    for f in range(12):
        a.update(boxes, conf, cls)
        b.update(boxes, conf, cls)
    sa, sb = a.snapshot(), b.snapshot()
    for k in ("ids", "age", "tsu", "cls"):
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(sa["xyxy"], sb["xyxy"])
