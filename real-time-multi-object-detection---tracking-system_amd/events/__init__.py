from .zone_engine import Zone, ZoneEvent, ZoneEventEngine

__all__ = ["ZoneEventEngine", "ZoneEvent", "Zone"]
