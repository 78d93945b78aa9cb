"""CPU: the zone-engine restatement against the fixture produced by the reference's own
src/events/zone_engine.py (oracle/gen_golden_zones.py), and the restated cv::pointPolygonTest against an
independent exact test."""
import gzip
import json
import os
from fractions import Fraction

import numpy as np
import pytest

from oracle import zone_oracle as Z
from conftest import GOLDEN


def load_cases():
    with gzip.open(os.path.join(GOLDEN, "zones_g1.json.gz"), "rt") as f:
        return json.load(f)


@pytest.mark.parametrize("case", ["epoch_clock", "small_clock"])
def test_zone_state_machine_matches_reference_fixture(case):
    g = load_cases()
    eng = Z.ZoneOracle(g["zones"])
    c = g["cases"][case]
    n_events = 0
    for fr, want in zip(c["frames"], c["expect"]):
        tracks = [(i, np.asarray(b, np.float32), k) for i, b, k in zip(fr["ids"], fr["xyxy"], fr["cls"])]
        got = eng.process(tracks, fr["frame_id"], fr["now"])
        ref = [{k: v for k, v in e.items() if k != "class_name"} for e in want["events"]]
        assert got == ref, f"frame {fr['frame_id']}"
        snap = eng.snapshot()
        assert snap["occupancy"] == want["occupancy"] and snap["cooldown"] == want["cooldown"], f"frame {fr['frame_id']}"
        n_events += len(got)
    assert n_events > 30


def exact_pip(poly, x, y):
    """Independent statement: 0 if the point lies on a closed edge segment (exact integer test), else the
    parity of edges crossed by the ray to +x, half-open in y, intersection abscissa as an exact Fraction."""
    n = len(poly)
    for i in range(n):
        (ax, ay), (bx, by) = poly[i - 1], poly[i]
        cross = (bx - ax) * (y - ay) - (by - ay) * (x - ax)
        if cross == 0 and min(ax, bx) <= x <= max(ax, bx) and min(ay, by) <= y <= max(ay, by):
            return 0
    inside = False
    for i in range(n):
        (ax, ay), (bx, by) = poly[i - 1], poly[i]
        if (ay > y) != (by > y):
            xi = Fraction(ax) + Fraction((y - ay) * (bx - ax), (by - ay))
            if xi > x:
                inside = not inside
    return 1 if inside else -1


def test_point_polygon_test_restatement_vs_exact():
    rng = np.random.default_rng(4)
    polys = [np.array(p, np.int32) for p in (
        [[100, 200], [400, 200], [400, 600], [100, 600]],
        [[450, 100], [750, 100], [750, 400], [600, 250], [450, 400]],
        [[0, 0], [10, 0], [10, 10], [5, 10], [5, 5], [0, 5]],
        [[0, 0], [8, 8], [16, 0], [16, 16], [0, 16]],
        [[3, 1], [9, 1], [9, 1], [12, 7], [6, 12], [0, 7]],              # repeated vertex
    )]
    for _ in range(20):                                                   # random simple-ish polygons (star-shaped)
        k = int(rng.integers(3, 12))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        rad = rng.uniform(5, 40, k)
        polys.append(np.stack([50 + rad * np.cos(ang), 50 + rad * np.sin(ang)], 1).round().astype(np.int32))
    n = 0
    for poly in polys:
        pl = [(int(a), int(b)) for a, b in poly]
        lo, hi = poly.min(0) - 2, poly.max(0) + 3
        pts = [(int(x), int(y)) for x in range(lo[0], hi[0], max(1, (hi[0] - lo[0]) // 40)) for y in range(lo[1], hi[1], max(1, (hi[1] - lo[1]) // 40))]
        pts += pl                                                         # every vertex
        for x, y in pts:
            assert Z.point_polygon_test(poly, x, y) == exact_pip(pl, x, y), (pl, x, y)
            n += 1
    assert n > 5000
    assert Z.point_polygon_test(np.zeros((0, 2), np.int32), 1, 1) == -1


def test_centroid_truncates_toward_zero_in_float32():
    assert Z.centroid(np.array([10.6, 20.2, 11.5, 21.9], np.float32)) == (11, 21)
    assert Z.centroid(np.array([-3.5, -1.0, 0.2, 0.1], np.float32)) == (-1, 0)
