"""CPU: the tracker oracle is pinned bit-for-bit to fixtures produced by the
reference's own tracker.py (oracle/gen_golden_tracker.py)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import tracker_oracle as T
from conftest import GOLDEN, unpack_frames

KEYS = ("ids", "xyxy", "conf", "cls", "age", "tsu")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def check_sequence(z, frames, **params):
    o = T.TrackerOracle(**params)
    for f, (b, c, k) in enumerate(frames):
        ret = o.update(b, c, k)
        assert len(ret) == z["n_returned"][f] == 0      # SURVEY finding 4: facade always returns []
        s = o.snapshot()
        assert len(s["ids"]) == z["n_tracks"][f], f"frame {f}"
        assert s["next_id"] == z["next_id"][f], f"frame {f}"
        assert np.array_equal(T.state_digest(s), z["digest"][f]), f"frame {f}"
        if f"f{f:04d}_ids" in z:
            for key in KEYS:
                assert np.array_equal(bits(s[key]), bits(z[f"f{f:04d}_{key}"])), (f, key)
    return o


def test_g1_batch_iou_bit_exact():
    z = load("tracker_g1_iou.npz")
    got = T.batch_iou(z["a"], z["b"])
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.int32), z["iou"].view(np.int32))


@pytest.mark.parametrize("fn", [T.assign_greedy, T.assign_greedy_parallel])
def test_g2_assignment(fn):
    z = load("tracker_g2_assign.npz")
    for name in z["names"]:
        mr, mc, ur, uc = fn(z[f"{name}_cost"], 0.8)
        assert mr == z[f"{name}_mr"].tolist(), name
        assert mc == z[f"{name}_mc"].tolist(), name
        assert ur == z[f"{name}_ur"].tolist(), name
        assert uc == z[f"{name}_uc"].tolist(), name


def test_g2_known_answers():
    z = load("tracker_g2_assign.npz")
    # contested column: row 1 loses column 0 and does NOT fall back to .81
    assert z["contested_mr"].tolist() == [0, 2] and z["contested_mc"].tolist() == [0, 2]
    assert z["contested_ur"].tolist() == [1] and z["contested_uc"].tolist() == [1]
    assert z["alltie_mr"].tolist() == [0] and z["alltie_ur"].tolist() == [1]
    # float32(0.8) accepted, next float32 below rejected
    assert z["thr_edge_mr"].tolist() == [0] and z["thr_edge_ur"].tolist() == [1]


def test_g3_lifecycle():
    z = load("tracker_g3_lifecycle.npz")
    fr = unpack_frames(z)
    check_sequence(z, fr)
    # known answers straight from the reference run
    assert z["f0000_ids"].tolist() == [1, 2]                       # low-conf box 1 never spawns
    assert z["f0002_cls"].tolist() == [7, 2]                       # low-conf rematch overwrites class
    assert abs(float(z["f0002_conf"][0]) - 0.4) < 1e-6
    assert z["next_id"][3] == 3                                    # unmatched low det did not spawn
    assert z["n_tracks"][103] == 2 and z["f0103_tsu"].tolist() == [102, 102]  # 100 empty frames age, never expire
    assert z["n_tracks"][-2] == 0                                  # expired on non-empty unmatched frames
    assert z["f%04d_ids" % (len(fr) - 1)].tolist() == [3]


def test_g3c_expiry_on_30th_unmatched_nonempty_frame():
    z = load("tracker_g3c_expiry.npz")
    check_sequence(z, unpack_frames(z))
    assert z["n_tracks"][29] == 1 and z["f0029_tsu"].tolist() == [30]   # survives 29 unmatched frames
    assert z["n_tracks"][30] == 0                                       # removed on the 30th
    assert z["f0033_ids"].tolist() == [2]


def test_g3b_nondefault_params():
    z = load("tracker_g3b_params.npz")
    check_sequence(z, unpack_frames(z), track_thresh=0.6, track_buffer=5, match_thresh=0.7)


@pytest.mark.parametrize("name", ["tracker_g5_seq200.npz", "tracker_g6_seq500.npz"])
def test_g5_g6_sequences(name, pkg):
    z = load(name)
    xy, cf, cl = pkg.synth.box_sequence(int(z["seq_n"]), int(z["seq_canvas"]), int(z["seq_frames"]), int(z["seq_seed"]))
    sha = hashlib.sha256(xy.tobytes() + cf.tobytes() + cl.tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, dtype=np.uint8), z["in_sha"]), "synthetic generator drifted"
    check_sequence(z, [(xy[f], cf, cl) for f in range(xy.shape[0])])


def test_g7_ragged():
    z = load("tracker_g7_ragged.npz")
    check_sequence(z, unpack_frames(z))


def test_g8_adversarial():
    """Ties and degenerate inputs run through the reference itself (oracle/gen_golden_tracker.py:g8_adversarial): duplicate detections,
    confidences exactly at / one ulp below track_thresh, zero-area and inverted boxes, huge coordinates, low-confidence-only and
    empty frames, crowds whose tracks share a best column."""
    z = load("tracker_g8_adversarial.npz")
    check_sequence(z, unpack_frames(z))


def test_parallel_greedy_equals_sequential_on_sequences(pkg):
    xy, cf, cl = pkg.synth.box_sequence(150, 640, 40, seed=3)
    a, b = T.TrackerOracle(), T.TrackerOracle(assign="greedy_parallel")
    for f in range(40):
        a.update(xy[f], cf, cl)
        b.update(xy[f], cf, cl)
        assert np.array_equal(T.state_digest(a.snapshot()), T.state_digest(b.snapshot()))


def test_lapjv_restatement_is_optimal_and_respects_limit():
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 1, size=(12, 9)).astype(np.float32)
    mr, mc, ur, uc = T.assign_lapjv(c, 0.8)
    assert len(set(mc)) == len(mc)
    assert all(c[i, j] > np.float32(0.8) - 1e-6 for i, j in zip(mr, mc))
    assert sorted(mr + ur) == list(range(12))


def _best_matching_bruteforce(iou, thresh):
    """max over all partial matchings of sum(cost_limit - cost) -- the definition lap's extended matrix encodes"""
    m, n = iou.shape
    lim = 1 - thresh
    g = lim - (np.float32(1) - iou).astype(np.float64)
    best = [0.0, ()]

    def rec(i, used, tot, pairs):
        if i == m:
            if tot > best[0] + 1e-15:
                best[0], best[1] = tot, pairs
            return
        rec(i + 1, used, tot, pairs)
        for j in range(n):
            if not used & (1 << j) and g[i, j] > 0:
                rec(i + 1, used | (1 << j), tot + g[i, j], pairs + ((i, j),))

    rec(0, 0, 0.0, ())
    return best


def test_lapjv_restatement_equals_bruteforce_optimum():
    rng = np.random.default_rng(3)
    for _ in range(40):
        m, n = rng.integers(1, 6, size=2)
        iou = rng.uniform(0.6, 1.0, size=(m, n)).astype(np.float32)
        mr, mc, ur, uc = T.assign_lapjv(iou, 0.8)
        tot, pairs = _best_matching_bruteforce(iou, 0.8)
        assert tuple(zip(mr, mc)) == pairs
        assert sorted(mr + ur) == list(range(m)) and sorted(mc + uc) == list(range(n))
