"""Diagnostic (CPU only): the symbolic phase and launch-shape code on thousands of random connected systems of four graph
families (band, hub, grid, random tree with chords), best on a bounds-checked build:
python -c "import ezpz_amd.build as b; b.build(extra_flags=['-D_GLIBCXX_ASSERTIONS'], lib_path=b.LIB.replace('.so', '_assert.so'))" """
import os, sys
os.environ.setdefault("EZPZ_AMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ezpz_amd", "libezpz_amd_assert.so"))
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, gen, ezpz_amd as E
from oracle import oracle as O
rng = np.random.default_rng(5)
n_ok = 0
for trial in range(3000):
    kind = trial % 4
    npts = int(rng.integers(10, 900))
    if kind == 0:
        recs, g = gen.connected_sketch(npts, 100 + trial)
    elif kind == 1:  # hub
        cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
        for k in range(1, npts):
            cons += [O.distance((2*k, 2*k+1), (0, 1), 1.0 + k), O.horizontal_distance((2*k, 2*k+1), (0, 1), 0.5 * k)]
        recs, g = O.stack(cons), rng.uniform(-5, 5, 2 * npts)
    elif kind == 2:  # grid-like: each point tied to left and upper neighbour
        w = int(np.sqrt(npts)) + 1; npts = w * w
        cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
        pt = lambda i: (2*i, 2*i+1)
        for i in range(1, npts):
            r, c = divmod(i, w)
            if c > 0: cons.append(O.distance(pt(i), pt(i - 1), 1.0))
            if r > 0: cons.append(O.distance(pt(i), pt(i - w), 1.0))
            if c == 0 and r > 0: cons.append(O.horizontal_distance(pt(i), pt(i - w), 0.0))
            if r == 0: cons.append(O.vertical_distance(pt(i), pt(i - 1), 0.0))
        recs, g = O.stack(cons), rng.uniform(-5, 5, 2 * npts)
    else:  # random tree + extra chords
        cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
        pt = lambda i: (2*i, 2*i+1)
        for i in range(1, npts):
            a = int(rng.integers(0, i)); b = int(rng.integers(0, i))
            cons += [O.distance(pt(i), pt(a), 1.0), O.vertical_distance(pt(i), pt(b), 0.3) if a != b else O.horizontal_distance(pt(i), pt(a), 0.2)]
        recs, g = O.stack(cons), rng.uniform(-5, 5, 2 * npts)
    try:
        i = E.analyze(recs, len(g))
    except Exception as e:
        print("FAIL trial", trial, "kind", kind, "npts", npts, e); continue
    for team in (0, E.TEAM_AUTO_LATENCY, E.TEAM_LATENCY_PHASES, E.TEAM_AUTO_LISTS, E.TEAM_BATCH_LANES):  # every automatic shape: without a device the analysis runs, then -100
        try:
            E.System(recs, len(g), team_size=team)
        except E.NonLinearSystemError as e:
            if e.code != -100: print("FAIL trial", trial, "kind", kind, "npts", npts, "team", team, e)
    n_ok += 1
print("analysed", n_ok, "systems without a bounds assertion")
