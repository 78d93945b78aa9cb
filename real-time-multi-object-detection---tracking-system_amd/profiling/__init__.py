from .latency_profiler import LatencyProfiler

__all__ = ["LatencyProfiler"]
