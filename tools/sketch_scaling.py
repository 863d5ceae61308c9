"""Diagnostic: one connected sketch of mixed kinds (tests/gen.py:connected_sketch) at growing sizes -- launch shape,
device time of one solve, batch throughput with the guesses resident in HBM, and the CPU oracle beside them."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ezpz_amd as E
from oracle import oracle as O
import gen

dev = torch.device('cuda', 0)
sizes = [int(a) for a in sys.argv[1:]] or [8, 25, 75, 150, 400, 1000, 2500]
for npts in sizes:
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    t = time.perf_counter(); s = E.System(recs, len(g), team_size=int(__import__("os").environ.get("TEAM", "0"), 0)); t_sym = time.perf_counter() - t
    info = s.info()
    cfg = E.Config(max_iterations=60)
    def run(B, reps):
        x0 = torch.from_numpy(np.tile(g, (B, 1))).to(dev)
        xo = torch.empty_like(x0); st = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(2): s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): s.solve_batch_device(x0.data_ptr(), B, xo.data_ptr(), st.data_ptr(), 0, stream, cfg)
        torch.cuda.synchronize(); return (time.perf_counter() - t) / reps, xo, st
    t1, xo, st = run(1, 20)
    B = int(__import__("os").environ.get("BATCH", 0)) or max(256, min(65536, (1 << 26) // (8 * len(g))))
    tB, _, _ = run(B, 5)
    t = time.perf_counter(); want = O.solve(recs, g, O.Config(max_iterations=60), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16); t_cpu = time.perf_counter() - t
    x = xo[0].cpu().numpy()
    err = float(np.max(np.abs(x - want.final_values) / np.maximum(1.0, np.abs(want.final_values))))
    print(f"npts={npts:5d} n={len(g):5d} mode={info['team_mode']} team={info['team_size']:4d} parts={info['n_partitions']} wgs={info['grid_workgroups']} levels={info['n_levels']:3d} "
          f"nnzL/nnzA={info['nnz_l']/max(1,info['nnz_a']):.2f} ws_lds={info['workspace_in_lds']} | symbolic {t_sym*1e3:7.2f} ms | one solve {t1*1e6:8.1f} us | "
          f"batch {B:6d}: {B/tB:12.0f} solves/s | oracle {t_cpu*1e6:9.1f} us ({1/t_cpu:8.0f}/s), iters {want.iterations}, max rel diff {err:.1e}")
