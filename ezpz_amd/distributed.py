"""Batch sharding of independent constraint systems across the GPUs of one node (one process per GPU).

The LM path has no exchange step between systems, so the data path needs no collective at all: every rank
solves its own contiguous shard.  When the whole batch originates on one rank, `solve_batch_sharded` moves
the guesses out and the results back with exactly one scatter and one gather (RCCL over xGMI with the
"nccl" backend; each of the root's 7 links carries one peer's slice).  The reference has no counterpart
(it solves one system per call, single-threaded); SURVEY.md section 8(e).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from ._lib import STATUS_DTYPE


def shard_bounds(batch: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ceil(batch/world) systems per rank; trailing ranks may be short or empty."""
    per = -(-batch // world) if world > 0 else batch
    return [(min(r * per, batch), min((r + 1) * per, batch)) for r in range(world)]


def _default_local_solve(system, config):
    def run(x0: torch.Tensor):
        n = x0.shape[1]
        x_out = torch.empty_like(x0)
        status = torch.zeros((x0.shape[0], STATUS_DTYPE.itemsize), dtype=torch.uint8, device=x0.device)
        stream = torch.cuda.current_stream(x0.device).cuda_stream
        system.solve_batch_device(x0.data_ptr(), x0.shape[0], x_out.data_ptr(), status.data_ptr(), 0, stream, config)
        return x_out, status

    return run


def solve_batch_sharded(system, x0_root: Optional[torch.Tensor], n_vars: int, root: int = 0, config=None,
                        group=None, local_solve: Optional[Callable] = None, device=None):
    """Scatter guesses from `root`, solve every shard locally, gather x* and status on `root`.

    x0_root: [batch, n_vars] float64 on `root` (None elsewhere).  Returns (x [batch, n_vars], status [batch, 32] uint8)
    on `root` and (None, None) on the other ranks.  `local_solve(x0_shard) -> (x_shard, status_shard)` defaults to
    the HIP path of `system` (an ezpz_amd.System on this rank's device).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    solve = local_solve or _default_local_solve(system, config)
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        meta[0] = x0_root.shape[0]
    dist.broadcast(meta, src=root, group=group)
    batch = int(meta[0])
    bounds = shard_bounds(batch, world)
    per = max(b - a for a, b in bounds) if batch else 0
    mine = torch.empty((per, n_vars), dtype=torch.float64, device=device)
    if rank == root:
        chunks = []
        for a, b in bounds:
            c = torch.empty((per, n_vars), dtype=torch.float64, device=device)
            if b > a:
                c[: b - a] = x0_root[a:b]
                c[b - a:] = x0_root[b - 1]  # padding rows: a valid system, dropped after the gather
            elif batch:
                c[:] = x0_root[0]
            chunks.append(c)
        dist.scatter(mine, chunks, src=root, group=group)
    else:
        dist.scatter(mine, None, src=root, group=group)
    if per:
        x_mine, st_mine = solve(mine)
    else:
        x_mine = mine
        st_mine = torch.zeros((0, STATUS_DTYPE.itemsize), dtype=torch.uint8, device=device)
    if rank == root:
        xs = [torch.empty_like(x_mine) for _ in range(world)]
        sts = [torch.empty_like(st_mine) for _ in range(world)]
        dist.gather(x_mine, xs, dst=root, group=group)
        dist.gather(st_mine, sts, dst=root, group=group)
        x = torch.cat([xs[r][: b - a] for r, (a, b) in enumerate(bounds)], dim=0)
        st = torch.cat([sts[r][: b - a] for r, (a, b) in enumerate(bounds)], dim=0)
        return x, st
    dist.gather(x_mine, None, dst=root, group=group)
    dist.gather(st_mine, None, dst=root, group=group)
    return None, None
