from .detector import Detections, Detector

__all__ = ["Detector", "Detections"]
