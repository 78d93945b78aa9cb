"""CPU restatement of the reference's zone event engine -- TEST INFRASTRUCTURE ONLY.

Follows ``/root/reference/src/events/zone_engine.py:82-132`` (``ZoneEventEngine.process``)
statement by statement, with the wall clock made an argument (the reference calls
``time.time()`` once per ``process``, ``:84``).

Pinning status
* occupancy / dwell / cooldown state machine -- PINNED: ``oracle/gen_golden_zones.py`` runs the
  reference's own ``zone_engine.py`` under a scripted clock and writes
  ``tests/golden/zones_g1.json.gz``; ``tests/test_oracle_zones.py`` replays it through this file.
* ``cv2.pointPolygonTest`` (``zone_engine.py:95``) -- PARITY UNPINNED: OpenCV
  (``opencv-python-headless>=4.8.0``, requirements.txt:26) is not installed anywhere this runs and
  the reference holds no fixture for it.  ``point_polygon_test`` restates the published algorithm
  of ``cv::pointPolygonTest`` (imgproc/geometry.cpp) for an int32 contour, an integer point and
  ``measureDist=False`` -- its "purely integer" branch -- and is self-checked against an
  independent exact rational ray-casting test.  The golden generator gives the reference THIS
  function as its ``cv2.pointPolygonTest``, so the fixture pins everything around it, not it.
"""
from __future__ import annotations

import numpy as np


def point_polygon_test(poly: np.ndarray, x: int, y: int) -> int:
    """``cv2.pointPolygonTest(poly, (x, y), False)``: +1 inside, -1 outside, 0 on an edge or vertex.
    Integer branch of cv::pointPolygonTest: for every edge v0 -> v either skip it (both ends on one
    side of the horizontal through the point, or both left of it; a point lying on a horizontal edge
    or on the vertex v is reported as 0 first) or take the sign of the int64 cross product."""
    poly = np.asarray(poly).reshape(-1, 2)
    total = len(poly)
    if total == 0:
        return -1
    x, y = int(x), int(y)
    counter = 0
    vx, vy = int(poly[total - 1][0]), int(poly[total - 1][1])
    for i in range(total):
        v0x, v0y = vx, vy
        vx, vy = int(poly[i][0]), int(poly[i][1])
        if (v0y <= y and vy <= y) or (v0y > y and vy > y) or (v0x < x and vx < x):
            if y == vy and (x == vx or (y == v0y and ((v0x <= x <= vx) or (vx <= x <= v0x)))):
                return 0
            continue
        dist = (y - v0y) * (vx - v0x) - (x - v0x) * (vy - v0y)
        if dist == 0:
            return 0
        if vy < v0y:
            dist = -dist
        counter += dist > 0
    return -1 if counter % 2 == 0 else 1


def centroid(xyxy) -> tuple:
    """zone_engine.py:91-92: ``int((x1 + x2) / 2)`` on float32 array elements (float32 add, float32
    halving, truncation toward zero)."""
    b = np.asarray(xyxy, dtype=np.float32)
    return int((b[0] + b[2]) / 2), int((b[1] + b[3]) / 2)


class ZoneOracle:
    """State machine of ``ZoneEventEngine`` (zone_engine.py:64-132).  ``zones``: list of dicts with the
    reference's keys (``name, polygon, trigger, dwell_time_sec, cooldown_sec``; defaults :146-149)."""

    def __init__(self, zone_configs):
        self.zones = [dict(name=z["name"], polygon=np.array(z["polygon"], dtype=np.int32),
                           trigger=z.get("trigger", "intrusion"), dwell_time_sec=z.get("dwell_time_sec", 2.0),
                           cooldown_sec=z.get("cooldown_sec", 10.0)) for z in zone_configs]
        self.occupancy = {}        # track_id -> {zone_name: first_seen}         (:74)
        self.cooldown = {}         # (track_id, zone_name) -> last_alert         (:76)

    def process(self, tracks, frame_id: int, now: float):
        """``tracks``: iterable of (track_id, xyxy float32[4], class_id).  Returns the events of this frame
        as dicts (zone_engine.py:104-115 minus the wall-clock timestamp string)."""
        events = []
        active = set()
        for tid, xyxy, cls in tracks:
            tid = int(tid)
            active.add(tid)
            cx, cy = centroid(xyxy)
            for z in self.zones:
                inside = point_polygon_test(z["polygon"], cx, cy) >= 0          # :95
                if inside:
                    occ = self.occupancy.setdefault(tid, {})
                    if z["name"] not in occ:
                        occ[z["name"]] = now
                    dwell = now - occ[z["name"]]
                    if dwell >= z["dwell_time_sec"]:
                        key = (tid, z["name"])
                        if now - self.cooldown.get(key, 0.0) >= z["cooldown_sec"]:
                            events.append(dict(event_type=z["trigger"], zone_name=z["name"], track_id=tid, class_id=int(cls),
                                               dwell_time_sec=round(dwell, 2), bbox_xyxy=[float(v) for v in np.asarray(xyxy, np.float32)],
                                               centroid=[cx, cy], frame_id=int(frame_id)))
                            self.cooldown[key] = now
                elif tid in self.occupancy:
                    self.occupancy[tid].pop(z["name"], None)                     # :122-123
        for sid in set(self.occupancy) - active:                                 # :126-128
            del self.occupancy[sid]
        return events

    def snapshot(self):
        """Canonical, JSON-able view of the two ledgers (empty per-track dicts dropped: unobservable)."""
        occ = sorted([int(t), str(n), float(v)] for t, d in self.occupancy.items() for n, v in d.items())
        cd = sorted([int(t), str(n), float(v)] for (t, n), v in self.cooldown.items())
        return {"occupancy": occ, "cooldown": cd}
