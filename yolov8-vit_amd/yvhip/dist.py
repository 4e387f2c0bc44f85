"""Multi-GPU plumbing of the inference path: one process per GPU, image batches shard
embarrassingly, NO data-path collective.  The only exchanges are control-plane scalars (timing max,
optional result gather) over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in CPU tests)."""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch


def env_rank() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n items for `rank`; sizes differ by at most one, earlier ranks larger."""
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a host scalar (the bench's elapsed time)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_objects(local: list) -> List[list]:
    """All ranks' per-image result lists, in rank order (image order is preserved by shard_range)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local]
    out: List[list] = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    return out


class BucketReducer:
    """Bucketed SUM all-reduce of one flat gradient buffer whose entries become final from the END towards
    the start (backward visits parameters in reverse order).  `ready(low)` launches, asynchronously, every
    bucket that lies entirely at or above offset `low`; `finish()` waits for all of them.  Buckets are
    fixed slices counted from the end, so every rank issues the same collectives in the same order."""

    def __init__(self, flat: torch.Tensor, bucket_elems: int, force: bool = False):
        """force (or YV_DP_FORCE_COLLECTIVE=1): issue the collectives even in a one-rank group - a one-GPU box can then
        exercise the real transport (RCCL communicator set-up, the async all-reduce on the wgrad side stream, the handle
        waits against the trainer's two streams); a SUM over one rank leaves the gradients bit for bit unchanged."""
        self.flat, self.bucket = flat, max(int(bucket_elems), 1)
        self.force = bool(force) or os.environ.get("YV_DP_FORCE_COLLECTIVE", "0") == "1"
        self.reset()

    def reset(self):
        self.next_hi = self.flat.numel()
        self.pending = []
        self.launched = []                      # (lo, hi) in launch order (for tests / tracing)

    def ready(self, low: int):
        import torch.distributed as dist
        active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or self.force)
        while self.next_hi > 0:
            lo = max(self.next_hi - self.bucket, 0)
            if lo < low:
                break
            if active:
                self.pending.append(dist.all_reduce(self.flat[lo:self.next_hi], op=dist.ReduceOp.SUM, async_op=True))
            self.launched.append((lo, self.next_hi))
            self.next_hi = lo

    def finish(self):
        self.ready(0)
        for h in self.pending:
            h.wait()
        self.pending = []


def union_length(intervals) -> float:
    """Total length of the union of [start, end] intervals (bench.py: time during which at least one launch of the
    dominant kernel is executing, from launch-attached event timestamps of concurrent streams)."""
    busy, cur_s, cur_e = 0.0, None, None
    for a, b in sorted(intervals):
        if cur_e is None or a > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = a, b
        else:
            cur_e = max(cur_e, b)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy
