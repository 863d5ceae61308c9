"""FreedomAnalysis of ONE large connected component (the WIDE layout: pivoted QR over the whole device), time and a digest of
the participation values -- run as it is (one cooperative launch, the matrix resident in the workgroups' registers), with
EZPZ_FREEDOM_CHAIN=2 (one cooperative launch streaming the trailing matrix per step) and with EZPZ_FREEDOM_CHAIN=1 (round 3's
chain of one launch pair per step): the underconstrained sets must agree (the sums run in three fixed orders).
usage: python tools/freedom_wide.py [points ...]   (variables = 2 x points)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ezpz_amd as E
import gen
ROUTES = {'1': 'chain of launch pairs (EZPZ_FREEDOM_CHAIN=1)', '2': 'one cooperative launch, trailing matrix streamed per step (EZPZ_FREEDOM_CHAIN=2)'}
for npts in [int(a) for a in sys.argv[1:]] or [150, 400, 1000]:
    recs, g = gen.connected_sketch(npts, 4242)
    recs = recs[:-3]
    n = len(g)
    s = E.System(recs, n)
    x, st, _ = s.solve_batch(g[None, :], E.Config(max_iterations=60))
    mask, part = s.freedom_batch(x)
    t = time.perf_counter()
    reps = 3
    for _ in range(reps):
        mask, part = s.freedom_batch(x)
    dt = (time.perf_counter() - t) / reps
    print(f"{n} variables: {dt * 1e3:8.2f} ms per analysis | underconstrained {int(mask.sum())} | digest {hashlib.sha256(part.tobytes()).hexdigest()[:16]}"
          f" | {ROUTES.get(os.environ.get('EZPZ_FREEDOM_CHAIN'), 'one cooperative launch, matrix resident in registers')}")
