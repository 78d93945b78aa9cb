"""GPU: the single-launch HIP tracker against the golden fixtures (produced by the
reference's tracker.py) and the pinned oracle.  Everything goes through the C ABI."""
import hashlib
import os

import numpy as np
import pytest

from oracle import tracker_oracle as T
from conftest import GOLDEN, unpack_frames

pytestmark = pytest.mark.gpu
KEYS = ("ids", "xyxy", "conf", "cls", "age", "tsu")


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def test_iou_matrix_bit_exact_g1(pkg):
    z = np.load(os.path.join(GOLDEN, "tracker_g1_iou.npz"))
    got = pkg._ffi.iou_matrix(z["a"], z["b"])
    assert np.array_equal(got.view(np.int32), z["iou"].view(np.int32))


def test_greedy_assignment_g2(pkg):
    z = np.load(os.path.join(GOLDEN, "tracker_g2_assign.npz"))
    for name in z["names"]:
        mr, mc, ur, uc = pkg._ffi.assign_greedy(z[f"{name}_cost"], 0.8)
        assert mr == z[f"{name}_mr"].tolist(), name
        assert mc == z[f"{name}_mc"].tolist(), name
        assert ur == z[f"{name}_ur"].tolist(), name
        assert uc == z[f"{name}_uc"].tolist(), name


def run_sequence(pkg, z, frames, **params):
    trk = pkg.MultiObjectTracker("bytetrack", **params)
    for f, (b, c, k) in enumerate(frames):
        out = trk.update(pkg.Detections(b, c, k))
        assert out == []                                     # reference facade returns [] every frame
        s = trk._core.snapshot()
        assert len(s["ids"]) == z["n_tracks"][f], f"frame {f}"
        assert s["next_id"] == z["next_id"][f], f"frame {f}"
        assert np.array_equal(T.state_digest(s), z["digest"][f]), f"frame {f}"
        if f"f{f:04d}_ids" in z:
            for key in KEYS:
                assert np.array_equal(bits(s[key]), bits(z[f"f{f:04d}_{key}"])), (f, key)
    return trk


@pytest.mark.parametrize("name,params", [
    ("tracker_g3_lifecycle.npz", {}),
    ("tracker_g3c_expiry.npz", {}),
    ("tracker_g3b_params.npz", {"bytetrack": {"track_thresh": 0.6, "track_buffer": 5, "match_thresh": 0.7, "mot20": False}}),
    ("tracker_g7_ragged.npz", {}),
    ("tracker_g8_adversarial.npz", {}),      # ties, threshold-equal confidences, zero-area / inverted boxes (round 3)
])
def test_fixture_sequences(pkg, name, params):
    z = np.load(os.path.join(GOLDEN, name))
    trk = run_sequence(pkg, z, unpack_frames(z), **params)
    # reference attribute names work
    assert trk._core._next_id == z["next_id"][-1]
    assert [t["track_id"] for t in trk._core._tracks] == z["f%04d_ids" % (len(z["n_tracks"]) - 1)].tolist()


@pytest.mark.parametrize("name", ["tracker_g5_seq200.npz", "tracker_g6_seq500.npz"])
def test_baseline_sequences(pkg, name):
    """BASELINE configs 3 and 5: 200 boxes @640^2 x120 frames, 500 boxes @1280^2 x60 frames."""
    z = np.load(os.path.join(GOLDEN, name))
    xy, cf, cl = pkg.synth.box_sequence(int(z["seq_n"]), int(z["seq_canvas"]), int(z["seq_frames"]), int(z["seq_seed"]))
    sha = hashlib.sha256(xy.tobytes() + cf.tobytes() + cl.tobytes()).digest()
    assert np.array_equal(np.frombuffer(sha, dtype=np.uint8), z["in_sha"])
    run_sequence(pkg, z, [(xy[f], cf, cl) for f in range(xy.shape[0])])


def test_multi_stream_batch_equals_oracle(pkg):
    """8 independent streams advanced by ONE launch each frame (BASELINE config 4 shape)."""
    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    S, N = 8, 256
    core = core_cls(n_streams=S, max_dets=N, max_tracks=1024)
    oracles = [T.TrackerOracle() for _ in range(S)]
    seqs = [pkg.synth.box_sequence(40 + 25 * s, 640, 30, seed=100 + s) for s in range(S)]
    for f in range(30):
        xy = np.zeros((S, N, 4), np.float32); cf = np.zeros((S, N), np.float32); cl = np.zeros((S, N), np.int32)
        cnt = np.zeros(S, np.int32)
        for s, (b, c, k) in enumerate(seqs):
            n = 0 if (f + s) % 11 == 0 else b.shape[1]        # some empty frames
            cnt[s] = n
            xy[s, :n], cf[s, :n], cl[s, :n] = b[f][:n], c[:n], k[:n]
            oracles[s].update(b[f][:n], c[:n], k[:n])
        act = core.update_batch(xy, cf, cl, cnt)
        assert act.tolist() == [0] * S
        for s in range(S):
            assert np.array_equal(T.state_digest(core.snapshot(s)), T.state_digest(oracles[s].snapshot())), (f, s)


def test_matched_report_and_trails(pkg):
    trk = pkg.MultiObjectTracker("bytetrack")
    trk.report = "matched"
    xy, cf, cl = pkg.synth.box_sequence(10, 320, 40, seed=2)
    cf[:] = 0.9
    for f in range(40):
        out = trk.update(pkg.Detections(xy[f], cf, cl))
    assert len(out) > 0 and all(t.time_since_update == 1 for t in out)
    assert max(len(t.trail) for t in out) == 30                # trail capped at 30 points (tracker.py:219,247)
    t = out[0]
    assert t.trail[-1] == (int((t.xyxy[0] + t.xyxy[2]) / 2), int((t.xyxy[1] + t.xyxy[3]) / 2))


def test_capacity_errors(pkg):
    from importlib import import_module
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    core = core_cls(max_tracks=8, max_dets=16)
    xy, cf, cl = pkg.synth.box_sequence(12, 320, 1, seed=1)
    cf[:] = 0.9
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        core.update(xy[0], cf, cl)
    assert e.value.code == pkg._ffi.E_CAPACITY
    core2 = core_cls(max_tracks=64, max_dets=4)
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        core2.update(xy[0], cf, cl)
    assert e.value.code == pkg._ffi.E_CAPACITY
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        core_cls(assign_mode=7)
    assert e.value.code == pkg._ffi.E_UNSUPPORTED


# ------------------------------------------------------------------ lap.lapjv branch (tracker.py:168-181), PARITY UNPINNED
def contested_boxes(rng, n, size=640.0, dup=0.35):
    """n track boxes and a detection list in which a share of the tracks has 2-3 near-identical
    detections and some tracks are near-duplicates of each other: IoU > 0.8 candidate pairs that share
    rows and columns, i.e. the components the exact solver (not the isolated-edge shortcut) must handle."""
    wh = rng.uniform(40, 120, size=(n, 2))
    c = rng.uniform(60, size - 60, size=(n, 2))
    t = np.concatenate([c - wh / 2, c + wh / 2], axis=1).astype(np.float32)
    k = int(n * dup)
    t[n - k:] = t[:k] + rng.normal(0, 1.0, size=(k, 4)).astype(np.float32)          # near-duplicate tracks
    d = [t + rng.normal(0, 0.8, size=t.shape).astype(np.float32)]
    for _ in range(2):
        sel = rng.random(n) < dup
        d.append(t[sel] + rng.normal(0, 1.2, size=(int(sel.sum()), 4)).astype(np.float32))
    d = np.concatenate(d).astype(np.float32)
    return t, d[rng.permutation(len(d))]


def gain(iou, mr, mc, thresh):
    lim = 1 - thresh
    return sum(lim - float(np.float32(1) - iou[i, j]) for i, j in zip(mr, mc))


@pytest.mark.parametrize("case", ["uniform12x9", "contested40", "contested150", "sparse200", "single", "ties"])
def test_lapjv_assignment_vs_oracle(pkg, case):
    rng = np.random.default_rng(11)
    if case == "uniform12x9":
        mats = [rng.uniform(0.6, 1.0, size=(12, 9)).astype(np.float32) for _ in range(6)]
    elif case == "single":
        mats = [np.array([[0.9]], np.float32), np.array([[0.5]], np.float32), np.array([[0.81, 0.95, 0.1]], np.float32),
                np.array([[0.81], [0.95], [0.1]], np.float32)]
    elif case == "ties":                                   # float32(0.8) itself: cost 0.19999999 < 1 - 0.8 (double) -> a candidate
        mats = [np.array([[np.float32(0.8), 0.0], [0.0, np.nextafter(np.float32(0.8), np.float32(0))]], np.float32)]
    else:
        n = {"contested40": 40, "contested150": 150, "sparse200": 200}[case]
        mats = []
        for _ in range(3):
            t, d = contested_boxes(rng, n, dup=0.0 if case == "sparse200" else 0.35)
            mats.append(T.batch_iou(t, d))
    for iou in mats:
        ref = T.assign_lapjv(iou, 0.8)
        got = pkg._ffi.assign_lapjv(iou, 0.8)
        assert abs(gain(iou, got[0], got[1], 0.8) - gain(iou, ref[0], ref[1], 0.8)) < 1e-12       # both optimal
        assert got == ref, (case, got[:2], ref[:2])


def test_lapjv_tracker_sequence_vs_oracle(pkg):
    """_ByteTrackCore with the lapjv branch over a crowded sequence (duplicated detections every frame)
    against the oracle running scipy's exact solver on lap's extended matrix."""
    rng = np.random.default_rng(5)
    core = pkg.tracking.tracker._ByteTrackCore(assign_mode=pkg._ffi.ASSIGN_LAPJV)
    ora = T.TrackerOracle(assign="lapjv")
    xy, cf, cl = pkg.synth.box_sequence(120, 640, 60, seed=77)
    for f in range(60):
        dup = rng.random(120) < 0.3
        b = np.concatenate([xy[f], xy[f][dup] + rng.normal(0, 1.0, size=(int(dup.sum()), 4)).astype(np.float32)]).astype(np.float32)
        c = np.concatenate([cf, rng.uniform(0.2, 0.95, size=int(dup.sum())).astype(np.float32)])
        k = np.concatenate([cl, cl[dup]]).astype(np.int32)
        core.update(b, c, k)
        ora.update(b, c, k)
        assert np.array_equal(T.state_digest(core.snapshot()), T.state_digest(ora.snapshot())), f"frame {f}"
    core.close()


def test_lapjv_too_dense_is_an_error(pkg):
    iou = np.full((300, 300), 0.95, np.float32)            # one 300 x 300 component: beyond the LDS budget
    with pytest.raises(pkg._ffi.RtmodtError) as e:
        pkg._ffi.assign_lapjv(iou, 0.8)
    assert e.value.code == pkg._ffi.E_CAPACITY


# ------------------------------------------------------------------ opt-in Kalman motion model (no reference counterpart)
def kalman_equal(core, orc, stream=0, tag=""):
    s, k, o = core.snapshot(stream), core.kalman_snapshot(stream), orc.snapshot()
    for key in KEYS:
        assert np.array_equal(bits(s[key]), bits(o[key])), (tag, key)
    assert s["next_id"] == o["next_id"], tag
    assert np.array_equal(k["mean"].view(np.int32), o["mean"].view(np.int32)), (tag, "mean")
    assert np.array_equal(k["cov"].view(np.int32), o["cov"].view(np.int32)), (tag, "cov")


def test_kalman_tracker_bit_exact_vs_oracle(pkg):
    """MultiObjectTracker(kalman=True): predict / associate on predicted boxes / update / initiate inside the one launch.
    Ids, ages, tsu, boxes AND the filter state (mean[8], covariance blocks) equal the NumPy oracle bit for bit, on the
    accelerating scene that the parity tracker cannot follow, on the jittery 200-box sequence of BASELINE config 3 with
    empty and sparse frames mixed in, and after the tracks have been compacted (expiry)."""
    from oracle import kalman_oracle as K
    from test_oracle_kalman import accelerating_scene
    boxes, conf, cls = accelerating_scene()
    trk = pkg.MultiObjectTracker("bytetrack", kalman=True)
    assert trk._core.kalman is True
    orc = K.TrackerOracleKalman()
    for f, b in enumerate(boxes):
        if f == 90:
            b, c, k = np.zeros((0, 4), np.float32), np.zeros(0, np.float32), np.zeros(0, np.int32)
        else:
            c, k = conf, cls
        assert trk.update(pkg.Detections(b, c, k)) == []
        orc.update(b, c, k)
        kalman_equal(trk._core, orc, tag=f"accelerating frame {f}")
    assert orc.next_id - 1 == len(conf)                      # no track was ever lost
    # config 3's sequence: 200 boxes, jitter, low-confidence second pass, expiry after track_buffer frames
    xy, cf, cl = pkg.synth.box_sequence(200, 640, 70, seed=1234)
    trk2 = pkg.MultiObjectTracker("bytetrack", kalman=True, track_buffer=8)
    orc2 = K.TrackerOracleKalman(track_buffer=8)
    rng = np.random.default_rng(0)
    for f in range(70):
        keep = rng.uniform(size=200) < (0.0 if f in (20, 21) else 0.5 if 30 <= f < 45 else 1.0)
        b, c, k = xy[f][keep], cf[keep], cl[keep]
        trk2.update(pkg.Detections(b, c, k))
        orc2.update(b, c, k)
        kalman_equal(trk2._core, orc2, tag=f"config-3 frame {f}")
    assert orc2.next_id - 1 > 120 and len(orc2.ids) < orc2.next_id - 1      # tracks were lost, respawned and expired along the way


def test_kalman_multi_stream_and_enable_rules(pkg):
    from importlib import import_module
    from oracle import kalman_oracle as K
    core_cls = import_module(pkg.__name__ + ".tracking.tracker")._ByteTrackCore
    S, N = 3, 64
    core = core_cls(n_streams=S, max_dets=N, max_tracks=256, kalman=True)
    orcs = [K.TrackerOracleKalman() for _ in range(S)]
    seqs = [pkg.synth.box_sequence(40 + 8 * s, 640, 25, seed=70 + s) for s in range(S)]
    for f in range(25):
        xyxy, conf, cls, cnt = np.zeros((S, N, 4), np.float32), np.zeros((S, N), np.float32), np.zeros((S, N), np.int32), np.zeros(S, np.int32)
        for s in range(S):
            n = 0 if (f + s) % 9 == 8 else len(seqs[s][1])
            xyxy[s, :n], conf[s, :n], cls[s, :n], cnt[s] = seqs[s][0][f][:n], seqs[s][1][:n], seqs[s][2][:n], n
            orcs[s].update(xyxy[s, :n], conf[s, :n], cls[s, :n])
        core.update_batch(xyxy, conf, cls, cnt)
        for s in range(S):
            kalman_equal(core, orcs[s], stream=s, tag=f"stream {s} frame {f}")
    core.close()
    plain = core_cls(n_streams=1, max_dets=N, max_tracks=256)
    with pytest.raises(pkg._ffi.RtmodtError):
        plain.kalman_snapshot()                              # not enabled
    plain.update(seqs[0][0][0], seqs[0][1], seqs[0][2])
    with pytest.raises(pkg._ffi.RtmodtError):
        pkg._ffi.check(pkg._ffi.lib().rtmodt_tracker_enable_kalman(plain._h))      # tracks already exist
    plain.close()
