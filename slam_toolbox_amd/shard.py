"""Multi-GPU sharding of the matcher path.

Scan matches are independent units: each (query scan, base chain) pair owns its correlation grid, lookup
table and response volume, so N GPUs = N processes (one per GPU, `torch.distributed`, backend "nccl" = RCCL
on the GPU box, "gloo" in the CPU tests) that each match their share of the pairs.  There is NO data-path
collective: the only communication is the gather of the 13 result doubles per pair (response, mean[3],
cov[9]) and, in bench.py, the barrier + max-over-ranks of the wall time.

Loop-closure semantics (Mapper.cpp:1487-1560, TryCloseLoop): candidates are matched speculatively for the
current poses and consumed in order until the first acceptance (`first_accepted`); everything after an
accepted closure is stale because CorrectPoses() moves the nodes.
"""
from __future__ import annotations

import numpy as np


def shard_units(n_units: int, rank: int, world: int) -> np.ndarray:
    """Round-robin assignment of unit indices to `rank` (pairs of similar cost are neighbours in the
    candidate list, so round-robin balances better than contiguous blocks)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return np.arange(rank, n_units, world, dtype=np.int64)


def gather_results(local_ids, local_results, n_units: int, width: int = 13, group=None, device="cpu"):
    """All ranks end up with the (n_units, width) result table in unit order.  One fixed-size all-gather of
    the padded per-rank tables (ids carried along), no object pickling."""
    import torch
    import torch.distributed as dist
    local_ids = np.asarray(local_ids, dtype=np.int64)
    local_results = np.asarray(local_results, dtype=np.float64).reshape(len(local_ids), width)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out = np.full((n_units, width), np.nan)
        out[local_ids] = local_results
        return out
    world = dist.get_world_size(group)
    cap = (n_units + world - 1) // world
    buf = torch.full((cap, width + 1), -1.0, dtype=torch.float64, device=device)
    if len(local_ids):
        buf[: len(local_ids), 0] = torch.from_numpy(local_ids.astype(np.float64)).to(device)
        buf[: len(local_ids), 1:] = torch.from_numpy(local_results).to(device)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = np.full((n_units, width), np.nan)
    for p in parts:
        p = p.cpu().numpy()
        ok = p[:, 0] >= 0
        out[p[ok, 0].astype(np.int64)] = p[ok, 1:]
    return out


def max_over_ranks(seconds: float, group=None, device="cpu") -> float:
    """bench.py contract: the timed region of a multi-rank run is the slowest rank's."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def match_candidates_sharded(match_fn, n_units: int, rank: int, world: int, batch: int = 32, group=None,
                             device="cpu"):
    """Matches unit indices shard_units(n_units, rank, world) in batches of `batch` with
    match_fn(list_of_unit_indices) -> (responses (k,), means (k,3), covs (k,3,3)) and returns the full
    (n_units, 13) table on every rank."""
    ids = shard_units(n_units, rank, world)
    rows = []
    for b in range(0, len(ids), batch):
        chunk = ids[b: b + batch]
        resp, means, covs = match_fn([int(i) for i in chunk])
        rows.append(np.concatenate([np.asarray(resp).reshape(-1, 1), np.asarray(means).reshape(-1, 3),
                                    np.asarray(covs).reshape(-1, 9)], axis=1))
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, 13))
    return gather_results(ids, local, n_units, 13, group=group, device=device)


def first_accepted(table, min_response: float, max_variance: float):
    """TryCloseLoop's acceptance rule on a gathered table, in candidate order: coarse response above
    m_pLoopMatchMinimumResponseCoarse and both positional variances below m_pLoopMatchMaximumVarianceCoarse
    (Mapper.cpp:1515-1518).  Returns the index of the first accepted candidate or -1."""
    table = np.asarray(table)
    for i in range(table.shape[0]):
        if table[i, 0] > min_response and table[i, 4] < max_variance and table[i, 8] < max_variance:
            return i
    return -1
