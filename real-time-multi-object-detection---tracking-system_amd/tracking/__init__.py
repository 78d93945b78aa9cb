from .tracker import MultiObjectTracker, Track

__all__ = ["MultiObjectTracker", "Track"]
