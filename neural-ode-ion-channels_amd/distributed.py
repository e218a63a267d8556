"""Trajectory sharding across the GPUs of one node and the path's only collective.

Trajectories are independent (each `odeint` call of the reference is its own solve, e.g. train-s1.py:566-580), so
the batch index is cut into contiguous shards, one process per GPU, and the forward solve needs no data-path
collective.  An objective evaluation (mean |i_pred - i_ref| as train-s1.py:329, or a sum of squares as
train-d0.py:508-540) reduces two fp64 scalars with ONE all-reduce -- RCCL over xGMI when the process group is
`nccl`, gloo on CPU in the tests.
"""
import os

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_bounds(n, rank, world):
    """Contiguous [lo, hi) of `n` items for `rank`; the first n % world ranks get one extra item."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_bounds_by_cost(cost, world):
    """Contiguous shards of equal predicted COST instead of equal count: [(lo, hi)] * world, covering range(len(cost)).

    Solves differ 2-3x in step count, so equal counts leave the ranks with the easy candidates idle at the
    all-reduce.  cost: per-item prediction (the previous evaluation's RHS-evaluation counts, schedule.pilot_cost(), ...);
    the cut after rank r is placed where the running sum first reaches (r + 1) / world of the total.  Every rank computes
    the same cuts from the same `cost`, so no communication is needed; zero or non-finite costs fall back to equal counts."""
    import numpy as np
    c = np.asarray(cost.detach().cpu() if isinstance(cost, torch.Tensor) else cost, dtype=np.float64).reshape(-1)
    n = c.size
    if world < 1:
        raise ValueError("world must be positive")
    if n == 0 or not np.isfinite(c).all() or (c < 0).any() or c.sum() <= 0:
        return [shard_bounds(n, r, world) for r in range(world)]
    cum = np.cumsum(c)
    cuts = [0]
    for r in range(1, world):
        k = int(np.searchsorted(cum, cum[-1] * r / world, side="left")) + 1   # first prefix reaching r/world of the total
        cuts.append(min(max(k, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def init_process_group(backend=None, device=None):
    """One process per GPU; nccl (= RCCL) when a HIP device is given, gloo otherwise.  No-op for world size 1."""
    import torch.distributed as dist
    rank, world, local = env_rank_world()
    if world == 1 or dist.is_initialized():
        return dist if dist.is_initialized() else None
    if backend is None:
        backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
    kw = {"device_id": torch.device(device)} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def allreduce_sum_count(total, count, group=None):
    """Sum a (sum, count) pair over all ranks with one all-reduce; returns (global_sum, global_count) tensors."""
    import torch.distributed as dist
    pair = torch.stack([torch.as_tensor(total, dtype=torch.float64).reshape(()),
                        torch.as_tensor(float(count), dtype=torch.float64, device=torch.as_tensor(total).device).reshape(())])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
    return pair[0], pair[1]


def sharded_mean_abs_loss(solve_shard, i_ref, n_traj, group=None):
    """Global mean |i_pred - i_ref| over `n_traj` trajectories sharded over the ranks.

    solve_shard(lo, hi) -> [hi - lo, Nt] current traces of this rank's trajectories (a torch tensor on the rank's
    device); i_ref: the matching [hi - lo, Nt] reference slice or a callable (lo, hi) -> tensor.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n_traj, rank, world)
    pred = solve_shard(lo, hi)
    ref = i_ref(lo, hi) if callable(i_ref) else i_ref
    ref = torch.as_tensor(ref, dtype=pred.dtype, device=pred.device)
    s, c = allreduce_sum_count((pred - ref).abs().sum(), pred.numel(), group=group)
    return s / c
