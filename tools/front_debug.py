"""Diagnostic: the first LM step of the frontal shape against the CPU port's, variable by variable, with the front each variable is
eliminated in (tests/front_ref.py reads the plan).  usage (GPU box): python tools/front_debug.py <points> [workgroups] [iterations]"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import ezpz_amd as E
from oracle import oracle as O
import gen, front_ref as FR
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 25
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1
recs, g = gen.connected_sketch(npts, 1000 + npts)
n = len(g)
os.environ["EZPZ_FRONT_WGS"] = str(wgs)
s = E.System(recs, n, team_size=E.TEAM_FRONTS)
print(s.info())
x0 = g[None, :] + gen.keyed_uniform(npts, 6, n, -0.02, 0.02)
x, st, _ = s.solve_batch(x0, E.Config(max_iterations=iters))
rc, xo, it, conv, nun = O.solve_batch(recs, x0, O.Config(max_iterations=iters), linsolve=O.LINSOLVE_SPARSE)
p = FR.Plan(recs, n, wgs=wgs)
owner = {}
for gi in range(p.n_wgs):
    W = p.wgs[gi]
    descs, level_ptr, children, rows, exports, maps = p.wg_tables(gi)
    vg = p.arr("<u4", W["o_var_glob"], int(W["n_loc"]))
    for k, d in enumerate(descs):
        lv = int(np.searchsorted(level_ptr, k, side="right")) - 1
        for q in range(int(d["K"])):
            owner[int(vg[int(rows[int(d["rows"]) + q])])] = (gi, k, lv, int(d["K"]), int(d["S"]), q)
for b in range(len(x0)):
    err = np.abs(x[b] - xo[b]) / np.maximum(1.0, np.abs(xo[b]))
    print(f"system {b}: iterations {int(st['iterations'][b])} / {int(it[b])}, max err {err.max():.3e}")
    if err.max() > 1e-9:
        badv = np.nonzero(err > 1e-9)[0]
        fronts = sorted({owner[int(v)][:5] for v in badv})
        print("   wrong variables:", len(badv), "of", n, "in fronts (wg, front, level, K, S):", fronts[:40])
        print("   wrong variables (id, (wg, front, level, K, S, pivot), got, want):", [(int(v), owner[int(v)], float(x[b][v]), float(xo[b][v])) for v in badv[:12]])
        if iters == 1:
            d, bad = FR.linear_step(p, x0[b], 1e-9)
            print("   numpy executor of the same plan: max err of x0 + d against the CPU port", float(np.max(np.abs(x0[b] + d - xo[b]))))
        good = sorted({owner[int(v)][:5] for v in range(n)} - set(fronts))
        print("   fronts with every variable right:", good[:40])
        break
