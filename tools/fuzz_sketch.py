"""Diagnostic: random connected sketches (tests/gen.py:connected_sketch) at three perturbation sizes on the three launch
shapes (one wavefront, latency workgroup, 128-lane workgroup) against the oracle: iteration count, convergence flag and
coordinates (1e-6 relative)."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import ezpz_amd as E
from oracle import oracle as O
import gen
bad = 0; tot = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    npts = [20, 33, 48, 70, 100, 140, 200, 320][seed % 8]
    recs, g = gen.connected_sketch(npts, 9000 + seed)
    for team in (0, E.TEAM_AUTO_LATENCY, 128):
        s = E.System(recs, len(g), team_size=team)
        rng = np.random.default_rng(seed)
        x0 = np.stack([g, g + rng.uniform(-0.1, 0.1, len(g)), g + rng.uniform(-0.3, 0.3, len(g))])
        cfg = dict(max_iterations=60)
        x, st, _ = s.solve_batch(x0, E.Config(**cfg))
        for b in range(3):
            w = O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16)
            tot += 1
            same_it = int(st["iterations"][b]) == w.iterations and bool(st["converged"][b]) == w.converged
            err = float(np.max(np.abs(x[b] - w.final_values) / np.maximum(1.0, np.abs(w.final_values))))
            if not same_it or (w.converged and err > 1e-6):
                bad += 1
                print("MISMATCH seed", seed, "npts", npts, "team", team, "b", b, "iters", int(st["iterations"][b]), w.iterations, "conv", bool(st["converged"][b]), w.converged, "err %.2e" % err, "mode", s.info()["team_mode"], s.info()["team_size"])
print("checked", tot, "mismatches", bad)
