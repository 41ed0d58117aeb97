"""Multi-GPU layout of a batch: one process per GPU, instances sharded, no collective on the data path.

Instances of a leaf are independent (no cross-instance data on the audio path for leaves without
gmem/msg coupling), so rank r of W owns the contiguous range instance_range(N, r, W) and runs its own engine.
The only communication is control-plane: a barrier around the timed region and a tiny all-reduce of run statistics
(max elapsed, total frames, worst null-test residual) -- RCCL on GPUs ("nccl" backend), gloo in CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple


def instance_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the instances rank owns: contiguous, sizes differ by at most one, earlier ranks get the extras."""
    if world <= 0 or not 0 <= rank < world or n_total < 0:
        raise ValueError(f"bad shard request n={n_total} rank={rank} world={world}")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(instance: int, n_total: int, world: int) -> int:
    base, extra = divmod(n_total, world)
    edge = extra * (base + 1)
    if instance < edge:
        return instance // (base + 1)
    return extra + (instance - edge) // base if base else world - 1


@dataclass
class Shard:
    lo: int                   # first global instance of this rank
    hi: int                   # one past its last
    n_total: int              # instances of the whole job
    scaling: str              # "strong": n_total fixed, split over the ranks; "weak": a fixed count on every rank

    @property
    def count(self) -> int:
        return self.hi - self.lo


def plan(rank: int, world: int, instances_total: int = 0, instances_per_rank: int = 0) -> Shard:
    """The shard of one rank, as bench.py and the C group API (zab_group_create) lay a job out: `instances_per_rank` > 0 gives
    every rank that many (weak scaling); otherwise `instances_total` is split with instance_range (strong scaling). A rank
    that would own nothing is an error -- the job is mis-sized, not something to run silently on fewer GPUs."""
    if instances_per_rank > 0:
        sh = Shard(rank * instances_per_rank, (rank + 1) * instances_per_rank, world * instances_per_rank, "weak")
    else:
        lo, hi = instance_range(instances_total, rank, world)
        sh = Shard(lo, hi, instances_total, "strong")
    if sh.count <= 0:
        raise ValueError(f"rank {rank} of {world} owns no instances ({sh.n_total} in total)")
    return sh


@dataclass
class RunStats:
    elapsed_s: float          # wall time of the timed region on this rank
    units: float              # samples (instances x channels x frames x steps) this rank processed
    max_abs_err: float = 0.0  # worst |out - oracle| this rank saw (0 when not checked)


def reduce_stats(local: RunStats, dist=None, device=None) -> RunStats:
    """Whole-job statistics: MAX of elapsed (the job is as slow as its slowest rank), SUM of units, MAX of error."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    t = torch.tensor([local.elapsed_s, local.max_abs_err], dtype=torch.float64, device=device)
    u = torch.tensor([local.units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return RunStats(float(t[0]), float(u[0]), float(t[1]))
