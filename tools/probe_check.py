import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, ezpz_amd as E, gen
from oracle import oracle as O
for npts, drop in ((40, 0), (40, 3), (150, 0), (150, 3), (150, 1), (400, 2), (1000, 0), (1000, 3)):
    recs, g = gen.connected_sketch(npts, 4242)
    if drop: recs = recs[:-drop]
    n = len(g)
    for team in (0, E.TEAM_AUTO_LATENCY):
        s = E.System(recs, n, team_size=team)
        x, st, _ = s.solve_batch(g[None, :], E.Config(max_iterations=60))
        t = time.perf_counter(); mask, part = s.freedom_batch(x); t1 = time.perf_counter() - t
        t = time.perf_counter()
        for _ in range(3): mask, part = s.freedom_batch(x)
        dt = (time.perf_counter() - t) / 3
        _, J, _ = s.eval_batch(x)
        under, want = O.freedom_analysis_dense(J[0])
        got = sorted(np.nonzero(mask[0])[0].tolist())
        print(f"npts {npts} drop {drop} team {team:#x}: first {t1*1e3:7.2f} ms, then {dt*1e3:7.2f} ms | under {len(got)} (oracle {len(under)}) equal {got == sorted(under)} | max |part - oracle| {np.max(np.abs(part[0] - want)):.2e}")
