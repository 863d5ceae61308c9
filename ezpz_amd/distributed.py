"""Batch sharding of independent constraint systems across the GPUs of one node (one process per GPU).

The LM path has no exchange step between systems, so the data path needs no collective at all: every rank
solves its own contiguous shard.  When the whole batch originates on one rank, `solve_batch_sharded` moves
the guesses out and the results back as one scatter and one gather made of point-to-point transfers (RCCL
send / recv over xGMI with the "nccl" backend -- what a scatter is on the wire; each of the root's 7 links carries
one peer's slice): the root sends VIEWS of its batch (shards are contiguous row ranges, of uneven length when the
batch does not divide), solves its own shard in place of the batch tensors and receives straight into the result
tensors -- no padded chunks, no concatenation.  The reference has no counterpart (it solves one system per call,
single-threaded); SURVEY.md section 8(e).  Callers that are not one-process-per-GPU use the C ABI's
`ezpz_multi_solve_batch` (one host thread per device) instead.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist

from ._lib import STATUS_DTYPE

TAG_X, TAG_STATUS = 1, 2


def shard_bounds(batch: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous ceil(batch/world) systems per rank; trailing ranks may be short or empty."""
    per = -(-batch // world) if world > 0 else batch
    return [(min(r * per, batch), min((r + 1) * per, batch)) for r in range(world)]


def _default_local_solve(system, config):
    """The HIP path of `system` (an ezpz_amd.System on this rank's device), writing into the tensors it is given."""

    def run(x0: torch.Tensor, x_out: torch.Tensor, status: torch.Tensor):
        stream = torch.cuda.current_stream(x0.device).cuda_stream
        system.solve_batch_device(x0.data_ptr(), x0.shape[0], x_out.data_ptr(), status.data_ptr(), 0, stream, config)

    return run


def _wait(reqs):
    for r in reqs:
        r.wait()


def solve_batch_sharded(system, x0_root: Optional[torch.Tensor], n_vars: int, root: int = 0, config=None,
                        group=None, local_solve: Optional[Callable] = None, device=None):
    """Scatter guesses from `root`, solve every shard locally, gather x* and status on `root`.

    x0_root: [batch, n_vars] float64 on `root` (None elsewhere).  Returns (x [batch, n_vars], status [batch, 32] uint8)
    on `root` and (None, None) on the other ranks.  `local_solve(x0_shard) -> (x_shard, status_shard)` replaces the HIP
    path of `system` (tests on CPU).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    peer = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)
    width = STATUS_DTYPE.itemsize
    meta = torch.zeros(1, dtype=torch.int64, device=device)
    if rank == root:
        x0_root = x0_root.contiguous()
        meta[0] = x0_root.shape[0]
    dist.broadcast(meta, src=peer(root), group=group)
    batch = int(meta[0])
    bounds = shard_bounds(batch, world)
    a, b = bounds[rank]

    def solve_into(x0, x_out, status):
        if local_solve is not None:
            xs, sts = local_solve(x0)
            x_out.copy_(xs)
            status.copy_(sts)
        else:
            _default_local_solve(system, config)(x0, x_out, status)

    if rank == root:
        x_all = torch.empty((batch, n_vars), dtype=torch.float64, device=device)
        st_all = torch.zeros((batch, width), dtype=torch.uint8, device=device)
        others = [(r, pa, pb) for r, (pa, pb) in enumerate(bounds) if r != root and pb > pa]
        sends = [dist.P2POp(dist.isend, x0_root[pa:pb], peer(r), group, TAG_X) for r, pa, pb in others]
        reqs = dist.batch_isend_irecv(sends) if sends else []
        if b > a:  # the root's own shard: solved from a view of the batch into views of the results
            solve_into(x0_root[a:b], x_all[a:b], st_all[a:b])
        _wait(reqs)
        recvs = []
        for r, pa, pb in others:
            recvs.append(dist.P2POp(dist.irecv, x_all[pa:pb], peer(r), group, TAG_X))
            recvs.append(dist.P2POp(dist.irecv, st_all[pa:pb], peer(r), group, TAG_STATUS))
        if recvs:
            _wait(dist.batch_isend_irecv(recvs))
        return x_all, st_all
    if b > a:
        mine = torch.empty((b - a, n_vars), dtype=torch.float64, device=device)
        _wait(dist.batch_isend_irecv([dist.P2POp(dist.irecv, mine, peer(root), group, TAG_X)]))
        x_mine = torch.empty_like(mine)
        st_mine = torch.zeros((b - a, width), dtype=torch.uint8, device=device)
        solve_into(mine, x_mine, st_mine)
        _wait(dist.batch_isend_irecv([dist.P2POp(dist.isend, x_mine, peer(root), group, TAG_X),
                                      dist.P2POp(dist.isend, st_mine, peer(root), group, TAG_STATUS)]))
    return None, None
