"""FreedomAnalysis of ONE large connected component, time and a digest of the participation values (rounded to 1e-9) -- run as it is
(round 5: null-space probes on the frontal factorisation, a fully constrained sketch and one that lost three constraints), with
EZPZ_FREEDOM_PROBES=0 (the WIDE layout: pivoted QR over the whole device, one cooperative launch, the matrix resident in the
workgroups' registers), and on top of that EZPZ_FREEDOM_CHAIN=2 (one cooperative launch streaming the trailing matrix per step) /
EZPZ_FREEDOM_CHAIN=1 (round 3's chain of one launch pair per step): the underconstrained sets must agree.
usage: python tools/freedom_wide.py [points ...]   (variables = 2 x points)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ezpz_amd as E
import gen
ROUTES = {'1': 'chain of launch pairs (EZPZ_FREEDOM_CHAIN=1)', '2': 'one cooperative launch, trailing matrix streamed per step (EZPZ_FREEDOM_CHAIN=2)'}
probes = os.environ.get("EZPZ_FREEDOM_PROBES", "1") != "0"
for npts in [int(a) for a in sys.argv[1:]] or [150, 400, 1000]:
    for drop in ((0, 3) if probes else (3,)):
        recs, g = gen.connected_sketch(npts, 4242)
        if drop: recs = recs[:-drop]
        n = len(g)
        s = E.System(recs, n)
        x, st, _ = s.solve_batch(g[None, :], E.Config(max_iterations=60))
        mask, part = s.freedom_batch(x)
        t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            mask, part = s.freedom_batch(x)
        dt = (time.perf_counter() - t) / reps
        route = ("null-space probes on the frontal factorisation (no QR)" if probes
                 else ROUTES.get(os.environ.get('EZPZ_FREEDOM_CHAIN'), 'pivoted QR: one cooperative launch, matrix resident in registers'))
        print(f"{n} variables, the last {drop} constraints dropped: {dt * 1e3:8.2f} ms per analysis | underconstrained {int(mask.sum())} | digest "
              f"{hashlib.sha256(np.round(part, 9).tobytes()).hexdigest()[:16]} | {route}")
