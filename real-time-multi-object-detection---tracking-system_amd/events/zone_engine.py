"""Polygon-zone intrusion / dwell events on the native engine.

Host-side mirror of the reference's ``src/events/zone_engine.py`` (``ZoneEvent`` :29-45, ``Zone`` :50-58,
``ZoneEventEngine`` :64-157): same constructor, ``process(tracks, frame_id)``, ``get_zone_polygons()`` and
the same JSON-lines alert log.  The per-frame work -- centroid, point-in-polygon against every zone,
occupancy / dwell / cooldown ledgers keyed by (track id, zone name) -- runs in ``csrc/zones.hip``; there is
no CPU implementation here.

Two ways in:

* ``process(tracks, frame_id)``: the reference's call (``tools/run_pipeline.py:145``), duck-typed on
  ``track_id / xyxy / class_id / class_name``.  With the reference's tracker that list is always
  empty (SURVEY finding 4), so no event can ever fire there.
* ``process_tracker(tracker, frame_id)``: straight on the device-resident state of a
  ``MultiObjectTracker`` / ``_ByteTrackCore`` right after its update -- the tracks never visit the host.
  ``tracker.report`` decides which tracks count as passed (``"matched"``: matched or spawned this
  frame; ``"reference"``: none).
"""
from __future__ import annotations

import ctypes as C
import json
import logging
import time
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Optional, Sequence

import numpy as np

from .. import _ffi

log = logging.getLogger("rtmodt.events")


@dataclass
class ZoneEvent:
    """Event record written to the alert log (reference: zone_engine.py:29-45)."""
    timestamp_utc: str
    event_type: str
    zone_name: str
    track_id: int
    class_id: int
    class_name: str
    dwell_time_sec: float
    bbox_xyxy: list
    centroid: list
    frame_id: int
    metadata: dict = field(default_factory=dict)

    def to_json(self) -> str:
        return json.dumps(asdict(self), default=str)


@dataclass
class Zone:
    """Zone definition (reference: zone_engine.py:50-58)."""
    name: str
    polygon: np.ndarray
    trigger: str
    dwell_time_sec: float = 2.0
    cooldown_sec: float = 10.0
    direction: Optional[str] = None


class ZoneEventEngine:
    """Evaluate tracks against polygon zones and emit events (reference: zone_engine.py:64-157)."""

    def __init__(self, zone_configs: list, log_path: str = "logs/events.jsonl", *, device=0, n_streams: int = 1,
                 max_tracks: int = 2048, max_events: int = 256, max_idle_frames: Optional[int] = None) -> None:
        self.zones = [self._parse_zone(z) for z in zone_configs]
        self.log_path = Path(log_path)
        self.log_path.parent.mkdir(parents=True, exist_ok=True)
        self.n_streams, self.max_tracks, self.max_events = int(n_streams), int(max_tracks), int(max_events)
        self._device = _ffi.device_ordinal(device)
        first = {}
        for i, z in enumerate(self.zones):                 # the reference's dicts are keyed by zone NAME (:98-100, :105)
            first.setdefault(z.name, i)
        self._polys = [np.ascontiguousarray(z.polygon, dtype=np.int32).reshape(-1, 2) for z in self.zones]
        cfgs = (_ffi.ZoneCfg * max(len(self.zones), 1))()
        for i, z in enumerate(self.zones):
            cfgs[i] = _ffi.ZoneCfg(self._polys[i].ctypes.data_as(C.POINTER(C.c_int32)), len(self._polys[i]), float(z.dwell_time_sec),
                                   float(z.cooldown_sec), first[z.name])
        h = C.c_void_p()
        idle = (1 << 62) if max_idle_frames is None else int(max_idle_frames)      # None: never, like the reference's _cooldown (:76)
        _ffi.check(_ffi.lib().rtmodt_zones_create(self._device, cfgs, len(self.zones), self.n_streams, self.max_tracks, self.max_events,
                                                  idle, C.byref(h)))
        self._h = h
        E = self.max_events
        self._ev_track = np.empty(E, np.int32)
        self._ev_zone = np.empty(E, np.int32)
        self._ev_dwell = np.empty(E, np.float64)
        self._ev_c = np.empty((E, 2), np.int32)
        log.info("ZoneEventEngine loaded %d zones.", len(self.zones))

    # ------------------------------------------------------------------ reference API
    def process(self, tracks: Sequence, frame_id: int, *, stream: int = 0, now: Optional[float] = None) -> list:
        """Check all tracks against all zones; returns the new events (zone_engine.py:82-132)."""
        now = time.time() if now is None else float(now)                      # :84
        n = len(tracks)
        ids = np.fromiter((int(t.track_id) for t in tracks), np.int64, n)
        xyxy = np.ascontiguousarray([np.asarray(t.xyxy, np.float32) for t in tracks], np.float32).reshape(n, 4)
        cls = np.fromiter((int(t.class_id) for t in tracks), np.int32, n)
        ne = C.c_int32(0)
        _ffi.check(_ffi.lib().rtmodt_zones_process(self._h, int(stream), _ffi.ptr(ids), _ffi.ptr(xyxy), _ffi.ptr(cls), n, now, int(frame_id),
                                                   _ffi.ptr(self._ev_track), _ffi.ptr(self._ev_zone), _ffi.ptr(self._ev_dwell),
                                                   _ffi.ptr(self._ev_c), C.byref(ne)))
        events = []
        for e in range(ne.value):
            t = tracks[int(self._ev_track[e])]
            events.append(self._emit(self.zones[int(self._ev_zone[e])], int(t.track_id), int(t.class_id), getattr(t, "class_name", ""),
                                     float(self._ev_dwell[e]), [float(v) for v in t.xyxy], self._ev_c[e].tolist(), frame_id))
        return events

    def process_tracker(self, tracker, frame_id: int, *, now: Optional[float] = None, class_names=None) -> list:
        """All streams of ``tracker`` at once, on its device-resident state.  Returns one event list per stream."""
        core = getattr(tracker, "_core", tracker)
        report = getattr(tracker, "report", "matched")
        now = time.time() if now is None else float(now)
        S, E = core.n_streams, self.max_events
        if not hasattr(self, "_t_id") or self._t_id.shape[0] != S:
            self._t_id = np.empty((S, E), np.int64); self._t_zone = np.empty((S, E), np.int32); self._t_dwell = np.empty((S, E), np.float64)
            self._t_box = np.empty((S, E, 4), np.float32); self._t_c = np.empty((S, E, 2), np.int32); self._t_cls = np.empty((S, E), np.int32)
            self._t_n = np.zeros(S, np.int32)
        _ffi.check(_ffi.lib().rtmodt_zones_process_tracker(self._h, core._h, now, int(frame_id), 1 if report == "matched" else 0,
                                                           _ffi.ptr(self._t_id), _ffi.ptr(self._t_zone), _ffi.ptr(self._t_dwell),
                                                           _ffi.ptr(self._t_box), _ffi.ptr(self._t_c), _ffi.ptr(self._t_cls), _ffi.ptr(self._t_n)))
        out = []
        for s in range(S):
            evs = []
            for e in range(int(self._t_n[s])):
                k = int(self._t_cls[s, e])
                name = class_names.get(k, str(k)) if isinstance(class_names, dict) else ""
                evs.append(self._emit(self.zones[int(self._t_zone[s, e])], int(self._t_id[s, e]), k, name, float(self._t_dwell[s, e]),
                                      [float(v) for v in self._t_box[s, e]], self._t_c[s, e].tolist(), frame_id))
            out.append(evs)
        return out

    def get_zone_polygons(self) -> list:
        """For visualization overlay (zone_engine.py:134-136)."""
        return [(z.name, z.polygon) for z in self.zones]

    def snapshot(self, stream: int = 0) -> dict:
        """The two ledgers in the oracle's canonical form (parity surface: ``_occupancy`` / ``_cooldown``)."""
        cap, Z = 2 * self.max_tracks, len(self.zones)
        ids = np.empty(cap, np.int64); mask = np.empty(cap, np.uint32)
        first = np.empty((cap, max(Z, 1)), np.float64); alert = np.empty((cap, max(Z, 1)), np.float64)
        n = C.c_int32(0)
        _ffi.check(_ffi.lib().rtmodt_zones_state(self._h, int(stream), _ffi.ptr(ids), _ffi.ptr(mask), _ffi.ptr(first), _ffi.ptr(alert), C.byref(n)))
        first, alert = first[:n.value, :Z].reshape(n.value, Z) if Z else first[:0], alert[:n.value, :Z].reshape(n.value, Z) if Z else alert[:0]
        keys = {}
        for i, z in enumerate(self.zones):
            keys.setdefault(z.name, i)
        occ, cd = [], []
        for r in range(n.value):
            for name, k in keys.items():
                if mask[r] >> k & 1:
                    occ.append([int(ids[r]), name, float(first[r, k])])
                if alert[r, k] != 0.0:
                    cd.append([int(ids[r]), name, float(alert[r, k])])
        return {"occupancy": sorted(occ), "cooldown": sorted(cd)}

    def close(self) -> None:
        if getattr(self, "_h", None):
            _ffi.lib().rtmodt_zones_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ internals
    def _emit(self, zone: Zone, track_id, class_id, class_name, dwell, bbox, centroid, frame_id) -> ZoneEvent:
        evt = ZoneEvent(timestamp_utc=time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), event_type=zone.trigger, zone_name=zone.name,
                        track_id=track_id, class_id=class_id, class_name=class_name, dwell_time_sec=round(dwell, 2), bbox_xyxy=bbox,
                        centroid=centroid, frame_id=frame_id)                    # zone_engine.py:104-115
        self._write(evt)
        return evt

    @staticmethod
    def _parse_zone(cfg: dict) -> Zone:
        return Zone(name=cfg["name"], polygon=np.array(cfg["polygon"], dtype=np.int32), trigger=cfg.get("trigger", "intrusion"),
                    dwell_time_sec=cfg.get("dwell_time_sec", 2.0), cooldown_sec=cfg.get("cooldown_sec", 10.0),
                    direction=cfg.get("direction"))                             # zone_engine.py:141-151

    def _write(self, evt: ZoneEvent) -> None:
        with open(self.log_path, "a") as f:
            f.write(evt.to_json() + "\n")
        log.info("EVENT | %s | zone=%s track=%s dwell=%.1fs", evt.event_type, evt.zone_name, evt.track_id, evt.dwell_time_sec)
