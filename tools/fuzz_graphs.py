"""Diagnostic: consistent, fully determined sketches of several graph families (random tree with chords, wide band, hub,
comb) on the launch shapes (automatic, one-solve, lanes across the batch, plain 512-lane walk) against the oracle:
convergence flag, iteration count (+-1 when long) and coordinates (1e-6 relative)."""
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ezpz_amd as E
from oracle import oracle as O

def sketch(family, npts, rng):
    pt = lambda i: (2 * i, 2 * i + 1)
    true = np.zeros((npts, 2))
    cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
    for i in range(1, npts):
        if family == "tree":
            a, b = int(rng.integers(0, i)), int(rng.integers(0, i))
        elif family == "band":
            a, b = i - 1, max(0, i - int(rng.integers(2, 13)))
        elif family == "hub":
            a, b = 0, max(0, i - 1)
        else:  # comb: a spine with teeth
            a, b = (i - 1, max(0, i - 2)) if i % 5 else (max(0, i - 5), max(0, i - 10))
        true[i] = true[a] + rng.uniform(0.6, 2.0, 2) * rng.choice([-1.0, 1.0], 2)
        cons.append(O.distance(pt(i), pt(a), float(np.hypot(*(true[i] - true[a])))))
        if b != a:
            cons.append(O.distance(pt(i), pt(b), float(np.hypot(*(true[i] - true[b])))) if rng.random() < 0.5
                        else O.horizontal_distance(pt(i), pt(b), float(true[i][0] - true[b][0])))
        else:
            cons.append(O.vertical_distance(pt(i), pt(a), float(true[i][1] - true[a][1])))
    return O.stack(cons), true.reshape(-1)

bad = tot = 0
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 64)
for seed in range(lo, hi):
    rng = np.random.default_rng(7000 + seed)
    family = ["tree", "band", "hub", "comb"][seed % 4]
    npts = int(rng.integers(20, 500))
    recs, true = sketch(family, npts, rng)
    n = len(true)
    x0 = np.stack([true + rng.uniform(-0.01, 0.01, n), true + rng.uniform(-0.03, 0.03, n)])
    cfg = dict(max_iterations=50)
    wants = [O.solve(recs, x0[b], O.Config(**cfg), linsolve=O.LINSOLVE_SPARSE, warn_cap=1 << 16) for b in range(2)]
    for team in (0, E.TEAM_AUTO_LATENCY, E.TEAM_BATCH_LANES, 512):
        s = E.System(recs, n, team_size=team)
        x, st, _ = s.solve_batch(x0, E.Config(**cfg))
        for b in range(2):
            w = wants[b]
            tot += 1
            ok_it = abs(int(st["iterations"][b]) - w.iterations) <= (0 if w.iterations <= 10 else 2) and bool(st["converged"][b]) == w.converged
            err = float(np.max(np.abs(x[b] - w.final_values) / np.maximum(1.0, np.abs(w.final_values))))
            if not ok_it or (w.converged and err > 1e-6):
                bad += 1
                i = s.info()
                print("MISMATCH", family, "seed", seed, "npts", npts, "team", team, "b", b, "iters", int(st["iterations"][b]), w.iterations,
                      "conv", bool(st["converged"][b]), w.converged, "err %.2e" % err, "mode", i["team_mode"], i["team_size"], "levels", i["n_levels"])
print("checked", tot, "mismatches", bad)
