"""Diagnostic: the graph families of tests/gen.py:graph_sketch (random two-predecessor graph "tree", wide band, hub, comb), the seeds of
tests/test_gpu_fuzz.py: which take the frontal plan (and with which elimination order), one solve on it against the record walk."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
import gen, front_ref as FR

dev = torch.device('cuda', 0)
cfg = E.Config(max_iterations=50)


def one_solve(s, g):
    x0 = torch.from_numpy(np.asarray(g)[None, :].copy()).to(dev)
    xo = torch.empty_like(x0); st = torch.zeros((1, 32), dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2): s.solve_batch_device(x0.data_ptr(), 1, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): s.solve_batch_device(x0.data_ptr(), 1, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / 10, int(st.cpu().numpy().view(E.STATUS_DTYPE)["iterations"][0])


seeds = [int(a) for a in sys.argv[1:]] or (list(range(48)) + [51, 153, 189])
taken = 0
for seed in seeds:
    rng = np.random.default_rng(7000 + seed)
    family = ["tree", "band", "hub", "comb"][seed % 4]
    npts = int(rng.integers(20, 500))
    recs, true = gen.graph_sketch(family, npts, rng)
    n = len(true)
    if n < 48: continue
    x0 = true + rng.uniform(-0.01, 0.01, n)
    a = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY); ia = a.info()
    r = E.System(recs, n, team_size=E.TEAM_LATENCY_RECORDS); ir = r.info()
    (ta, it), (tr, _) = one_solve(a, x0), one_solve(r, x0)
    taken += ia['team_mode'] == 5
    plan = FR.Plan(recs, n, wgs=0, max_wgs=64)
    model = f"fronts model {plan.model_cycles:7d} cycles, largest front {plan.max_rows:2d} rows" if plan.ok else "no frontal plan"
    print(f"{family:5s} seed {seed:3d} n={n:4d} {it:2d} iterations: automatic team_mode {ia['team_mode']} ({ia['grid_workgroups']:2d} workgroups) {ta*1e6:9.1f} us = {ta*2.1e9/max(it,1):8.0f} cycles/iteration | "
          f"record walk (mode {ir['team_mode']}, levels {ir['n_levels']:3d}, nnzL {ir['nnz_l']:6d}) {tr*1e6:9.1f} us = {tr*2.1e9/max(it,1):8.0f} | x{tr/ta:5.2f} | {model}")
print(f"{taken} of these systems take the frontal plan")
