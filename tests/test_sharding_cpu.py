"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the batch scatter / local solve / gather plumbing.
The local solve is the CPU oracle here (this container has no GPU); on a GPU box the same function drives the
HIP path (`ezpz_amd.distributed._default_local_solve`)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, read_case


def _worker(rank, world, port, batch, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import gen
        from ezpz_amd import STATUS_DTYPE
        from ezpz_amd.distributed import shard_bounds, solve_batch_sharded
        from oracle import oracle as O
        from oracle import textual as T

        ref = T.load(read_case("square"))
        n = ref.num_vars

        def local_solve(x0):
            rc, x, it, conv, nun = O.solve_batch(ref.constraints, x0.numpy())
            st = np.zeros(len(x), dtype=STATUS_DTYPE)
            st["iterations"], st["converged"], st["n_unsatisfied"] = it, conv, nun
            return torch.from_numpy(x), torch.from_numpy(st.view(np.uint8).reshape(len(x), -1).copy())

        x0 = None
        if rank == 0:
            x0 = torch.from_numpy(ref.guesses[None, :] + gen.keyed_uniform(3, batch, n, -0.1, 0.1))
        x, st = solve_batch_sharded(None, x0, n, root=0, local_solve=local_solve)
        if rank == 0:
            rc, xo, it, conv, nun = O.solve_batch(ref.constraints, x0.numpy())
            stv = st.numpy().view(STATUS_DTYPE).reshape(-1)
            ok = (x.shape == (batch, n) and np.array_equal(x.numpy(), xo) and np.array_equal(stv["iterations"], it)
                  and np.array_equal(stv["converged"], conv))
            q.put(("ok" if ok else "mismatch", shard_bounds(batch, world)))
        else:
            assert x is None and st is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 64), (2, 7), (3, 10), (2, 1)])
def test_scatter_solve_gather_matches_single_process(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + batch
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    status, bounds = q.get(timeout=5)
    assert status == "ok"
    assert bounds[0][0] == 0 and bounds[-1][1] == batch
    assert all(b0[1] == b1[0] for b0, b1 in zip(bounds, bounds[1:]))


def test_shard_bounds():
    from ezpz_amd.distributed import shard_bounds

    assert shard_bounds(1000000, 8) == [(i * 125000, (i + 1) * 125000) for i in range(8)]
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]
