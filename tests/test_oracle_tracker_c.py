"""CPU: the plain-C tracker restatement agrees bit-for-bit with the golden fixtures."""
import os

import numpy as np
import pytest

from oracle import tracker_oracle as T
from oracle import tracker_oracle_c as TC
from conftest import GOLDEN, unpack_frames


def test_c_iou_bit_exact():
    z = np.load(os.path.join(GOLDEN, "tracker_g1_iou.npz"))
    assert np.array_equal(TC.batch_iou(z["a"], z["b"]).view(np.int32), z["iou"].view(np.int32))


@pytest.mark.parametrize("name,params", [
    ("tracker_g3_lifecycle.npz", {}),
    ("tracker_g3c_expiry.npz", {}),
    ("tracker_g3b_params.npz", dict(track_thresh=0.6, track_buffer=5, match_thresh=0.7)),
    ("tracker_g7_ragged.npz", {}),
    ("tracker_g8_adversarial.npz", {}),
])
def test_c_sequences(name, params):
    z = np.load(os.path.join(GOLDEN, name))
    o = TC.TrackerOracleC(**params)
    for f, (b, c, k) in enumerate(unpack_frames(z)):
        assert o.update(b, c, k) == 0
        assert np.array_equal(T.state_digest(o.snapshot()), z["digest"][f]), f


@pytest.mark.parametrize("name", ["tracker_g5_seq200.npz", "tracker_g6_seq500.npz"])
def test_c_baseline_sequences(name, pkg):
    z = np.load(os.path.join(GOLDEN, name))
    xy, cf, cl = pkg.synth.box_sequence(int(z["seq_n"]), int(z["seq_canvas"]), int(z["seq_frames"]), int(z["seq_seed"]))
    o = TC.TrackerOracleC()
    for f in range(xy.shape[0]):
        o.update(xy[f], cf, cl)
        assert np.array_equal(T.state_digest(o.snapshot()), z["digest"][f]), f
