"""Multi-GPU layout of the hot path: independent video streams sharded across ranks.

The reference is a single-process, single-GPU, batch-1 loop (tools/run_pipeline.py:121-166,
config/default.yaml:35); there is nothing to translate.  On an MI355X node the natural unit
is the stream: every stream owns its frames and its tracker state, nothing is exchanged
between streams, so one process per GPU takes ``stream_id % n_gpus == rank`` (SURVEY.md
section 8e) and the data path has NO collective.  RCCL over xGMI carries only

* the start/stop barriers and the max-over-ranks wall time of a measurement, and
* an optional all-reduce(SUM) of a tiny int64 stats vector (frames, detections, live tracks)
  per report interval -- ~100 bytes, latency-bound, so ring/tree choice and the per-link
  xGMI bandwidth are irrelevant.

``backend="nccl"`` is RCCL on ROCm; ``"gloo"`` runs the same code on CPUs (tests).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence


def shard(n_streams_total: int, world: int, rank: int) -> list:
    """Global stream ids owned by ``rank``: ``gpu = stream_id mod n_gpu``."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_streams_total, world))


def owner(stream_id: int, world: int) -> int:
    return stream_id % world


class NodeSync:
    """Barrier / max-time / stats-sum over the ranks of one node.  With ``world == 1`` no
    process group is created and every call is local."""

    def __init__(self, world: Optional[int] = None, rank: Optional[int] = None, local_rank: Optional[int] = None,
                 backend: str = "nccl", init_method: Optional[str] = None):
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank))) if local_rank is None else local_rank
        self.backend = backend
        self._dist = None
        self._torch = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            self._torch, self._dist = torch, dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            if init_method:
                kw["init_method"] = init_method
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)

    @property
    def device(self):
        """Where the tiny reduce tensors live: the rank's GPU for RCCL, host memory for gloo."""
        return f"cuda:{self.local_rank}" if self.backend == "nccl" else "cpu"

    def barrier(self) -> None:
        if self._dist is not None:
            self._dist.barrier()

    def device_synchronize(self) -> None:
        if self._torch is not None and self.backend == "nccl":
            self._torch.cuda.synchronize()

    def max_time(self, seconds: float) -> float:
        if self._dist is None:
            return seconds
        t = self._torch.tensor([seconds], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, value: float) -> list:
        """Every rank's value, in rank order, on every rank (8 bytes per rank: an all-gather written as an all-reduce(SUM) of a
        one-hot vector, so that it runs on RCCL and gloo alike)."""
        if self._dist is None:
            return [float(value)]
        t = self._torch.zeros(self.world, dtype=self._torch.float64, device=self.device)
        t[self.rank] = float(value)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def sum_stats(self, values: Sequence[int]) -> list:
        if self._dist is None:
            return [int(v) for v in values]
        t = self._torch.tensor(list(values), dtype=self._torch.int64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def close(self) -> None:
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None
