"""The symbolic phase of the FRONTAL launch shape (csrc/fronts.cpp) on the CPU: the plan blob is executed in numpy
(tests/front_ref.py, operation for operation what front_kernel.hip.hpp does) and its step must equal the dense solve of
(JtJ + lambda I) d = -Jt r built from the oracle's Jacobian rows (reference: ezpz/src/solver/newton.rs:73-102).  No GPU needed:
this is host logic.  The GPU tests of the kernel itself are tests/test_gpu_fronts.py."""
import numpy as np
import pytest

import gen
import front_ref as FR
from oracle import oracle as O


def check(recs, g, wgs, lam=1e-3, lds_bytes=160 * 1024, rel=1e-9):
    n = len(g)
    p = FR.Plan(recs, n, wgs=wgs, lds_bytes=lds_bytes)
    assert p.ok
    d, bad = FR.linear_step(p, g, lam)
    want = FR.dense_step(recs, n, g, lam)
    assert not bad and not np.any(np.isnan(d))
    assert np.max(np.abs(d - want)) <= rel * max(1.0, np.max(np.abs(want))), float(np.max(np.abs(d - want)))
    return p


@pytest.mark.parametrize("npts,wgs", [(3, 1), (8, 1), (25, 1), (75, 1), (75, 2), (150, 1), (150, 3), (150, 8), (400, 0)])
def test_connected_sketch_plan_equals_the_dense_solve(npts, wgs):
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    p = check(recs, g + 0.01, wgs)
    assert p.max_rows <= 63 and p.max_pivots <= 16
    if wgs > 1:
        assert 1 < p.n_wgs <= wgs and p.n_chunks > 0
    if wgs == 0:
        assert p.n_wgs > 1  # 800 variables: several workgroups by the planner's own choice


@pytest.mark.parametrize("family", ["tree", "band", "hub", "comb"])
@pytest.mark.parametrize("wgs", [1, 4])
def test_graph_families_plan_equals_the_dense_solve(family, wgs):
    rng = np.random.default_rng(11)
    recs, true = gen.graph_sketch(family, 60, rng)
    check(recs, true + rng.uniform(-0.02, 0.02, len(true)), wgs)


def test_all_kinds_weights_duplicate_columns_and_free_variables():
    """Random block of all 25 kinds (duplicate ids inside a row, two-row kinds whose rows touch different variables), weights,
    and variables no constraint touches: the plan keeps one front per loose variable and the step is the dense solve's."""
    rng = np.random.default_rng(5)
    n = 40
    cons = []
    for kind in range(25):
        c = gen.arb_constraint(rng, kind, hi=32)
        c["weight"] = float(rng.uniform(0.5, 2.0))
        cons.append(c)
    recs = O.stack(cons)
    x = rng.uniform(-5.0, 5.0, n)
    p = check(recs, x, 1, lam=0.5, rel=1e-8)
    assert p.n_fronts >= 8  # variables 32..39 are touched by nothing: one front each
    check(recs, x, 3, lam=0.5, rel=1e-8)


def test_a_front_too_large_for_a_wavefront_is_refused():
    """A clique of 40 points (every pair at a distance): one front of 80 rows -- more than a wavefront's 63: the shape does not
    apply and the other shapes serve."""
    pts = 40
    cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
    for i in range(pts):
        for j in range(i):
            cons.append(O.distance((2 * i, 2 * i + 1), (2 * j, 2 * j + 1), 1.0))
    p = FR.Plan(O.stack(cons), 2 * pts)
    assert not p.ok


def test_workgroup_shares_fit_the_lds_they_are_given():
    recs, g = gen.connected_sketch(400, 1400)
    p = FR.Plan(recs, len(g), wgs=0, lds_bytes=64 * 1024)
    assert p.ok and p.lds_bytes <= 64 * 1024 and p.n_wgs >= 4
    assert not FR.Plan(recs, len(g), wgs=1, lds_bytes=64 * 1024).ok  # 800 variables do not fit 64 KB on one workgroup


@pytest.mark.parametrize("npts,wgs", [(25, 1), (150, 1), (400, 0), (1000, 0)])
def test_wavefront_schedules_cover_every_front_once_and_cannot_deadlock(npts, wgs):
    """FrontWg::t_sched: eight forward and eight backward lists per workgroup.  Every front of the workgroup is on exactly one list of
    each kind; parents and child counts of the descriptors agree; and the wavefronts, each running its list IN ORDER and waiting for
    what its next front needs (forward: the children of this workgroup, backward: the parent), all get through -- whatever the
    timing: stepped here with the slowest possible interleaving (one front of one wavefront at a time, lowest wavefront first)."""
    recs, g = gen.connected_sketch(npts, 1000 + npts)
    plan = FR.Plan(recs, len(g), wgs=wgs)
    waves = 8
    for gi in range(plan.n_wgs):
        W = plan.wgs[gi]
        descs = plan.wg_tables(gi)[0]
        nf = len(descs)
        t0 = int(W["o_tables"]) + int(W["t_sched"])
        head = plan.arr("<u2", t0, 2 * (waves + 1))
        total = int(head[2 * waves + 1])
        words = plan.arr("<u2", t0, total)
        lists = [[[int(k) for k in words[int(head[p * (waves + 1) + w]):int(head[p * (waves + 1) + w + 1])]] for w in range(waves)] for p in range(2)]
        parent = [int(d["parent_local"]) for d in descs]
        kids = [0] * nf
        for k in range(nf):
            if parent[k] != 0xFFFFFFFF:
                assert parent[k] < nf and parent[k] != k
                kids[parent[k]] += 1
        assert kids == [int(d["n_kids_local"]) for d in descs]
        for p in range(2):
            assert sorted(k for l in lists[p] for k in l) == list(range(nf)), (gi, p)
            done = [False] * nf
            signed = [0] * nf
            at = [0] * waves
            progressed = True
            while progressed:
                progressed = False
                for w in range(waves):
                    if at[w] == len(lists[p][w]):
                        continue
                    k = lists[p][w][at[w]]
                    ready = signed[k] == kids[k] if p == 0 else (parent[k] == 0xFFFFFFFF or done[parent[k]])
                    if ready:
                        done[k] = True
                        if p == 0 and parent[k] != 0xFFFFFFFF:
                            signed[parent[k]] += 1
                        at[w] += 1
                        progressed = True
                        break
            assert all(done), (gi, p, [k for k in range(nf) if not done[k]][:8])


def test_minimum_degree_serves_graphs_whose_level_separators_are_too_wide():
    """The planner orders by nested dissection and, where that makes a front of more than a wavefront's 63 rows, by minimum degree
    (FrontPlan::ordering 1): among the graph families of the GPU fuzz (tests/gen.py:graph_sketch, the seeds of test_gpu_fuzz.py) some
    take the second -- their plans' steps equal the dense solve like any other -- and some have no plan at all under either."""
    used = {0: 0, 1: 0, None: 0}
    checked = 0
    for seed in range(48):
        rng = np.random.default_rng(7000 + seed)
        family = ["tree", "band", "hub", "comb"][seed % 4]
        npts = int(rng.integers(20, 500))
        recs, true = gen.graph_sketch(family, npts, rng)
        n = len(true)
        if n < 48 or n > 300:
            continue  # (the numpy executor is slow: the small ones)
        p = FR.Plan(recs, n, wgs=0)
        used[p.ordering if p.ok else None] += 1
        if p.ok and p.ordering == 1:
            assert p.max_rows <= 63 and p.n_components == 1
            d, bad = FR.linear_step(p, true, 1e-3)
            want = FR.dense_step(recs, n, true, 1e-3)
            assert not bad and np.max(np.abs(d - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))
            checked += 1
    assert used[0] >= 5 and checked >= 1, used
