"""Diagnostic: ms per ezpz_system_freedom_batch call through each route of the pivoted QR of systems too large for one workgroup
(freedom.hip: freedom_device), by systems per call: the chain of launches per step (EZPZ_FREEDOM_CHAIN=1), without the resident route
(=2: one system -> the cooperative launch streaming the matrix, several -> the chain), and the default (resident rounds, or the
chain).  usage (GPU box): python tools/qr_routes.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import ezpz_amd as E
import test_gpu_freedom_fuzz as T
os.environ["EZPZ_FREEDOM_PROBES"] = "0"
for npts in (400, 700, 1000):
    rng = np.random.default_rng(5)
    recs, true = T.variant(rng, "tree", npts, 1.0, 6, 1, 1, 1e-5)
    n = len(true)
    s = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
    for nb in (1, 2, 3, 6, 12):
        X = np.stack([true + 0.01 * rng.uniform(-1, 1, n) for _ in range(nb)])
        out = []
        for chain in ("1", "2", ""):
            os.environ["EZPZ_FREEDOM_CHAIN"] = chain
            s.freedom_batch(X)
            t0 = time.time()
            for _ in range(3): m, p = s.freedom_batch(X)
            out.append("%s: %.1f ms" % ({"1": "chain", "2": "cooperative", "": "default"}[chain], (time.time() - t0) / 3 * 1e3))
        print(f"n {n}, {nb} systems per call: " + ", ".join(out), flush=True)
