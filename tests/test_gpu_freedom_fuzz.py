"""FreedomAnalysis by null-space probes (csrc/freedom.hip: freedom_by_probes) is another algorithm than the reference's -- the
projector onto null(J) applied to random vectors by the frontal factorisation instead of a column-pivoted QR with the rank rule
|R_ii| > 1e-8 max |R_ii| (find_dof.rs:36-49) -- behind heuristic thresholds (Ritz values, a second opinion at lambda / 1000).  Round
5's review: pinned by one generator seed.  This fuzz sweeps what could make the two disagree: four graph families x 24 ... 1000
points x 0 ... 6 constraints dropped x constraints DUPLICATED or NEARLY DEPENDENT (a copy of a constraint with its parameter and
the point it ties moved by 1e-4 ... 1e-11: a singular value of J swept through the rank decision) x coordinates scaled by 1e-3 ... 1e3.
Every system: the underconstrained id list equal to the oracle's dense pivoted QR of the same Jacobian (freedom_analysis_dense) --
whether the probes decided or handed the system to the device's own QR; the exits taken are counted and logged
(gpurun_out/freedom_probes_fuzz.txt when that directory exists: profiles/r06_freedom_probes_fuzz.txt is that file from the round's
GPU run)."""
import ctypes as C
import os

import numpy as np
import pytest

import gen
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FAMILIES = ["tree", "band", "hub", "comb"]


@pytest.fixture(scope="module")
def E():
    import ezpz_amd

    if ezpz_amd.device_count() < 1:
        pytest.fail("GPU tests need a HIP device: the product path has no CPU fallback")
    return ezpz_amd


def exits(E):
    out = (C.c_ulonglong * 8)()
    E.lib().ezpz_debug_freedom_exits(out)
    return np.array(list(out), dtype=np.int64)


def variant(rng, family, npts, scale, drop, weak, dup, eps):
    """records, true values: a graph sketch scaled by `scale`; its last `drop` constraints gone, `weak` of them back with weight
    `eps` -- a degree of freedom held by a row of norm ~eps: a singular value of J swept through the reference's rank decision --
    and `dup` constraints repeated with weight 1 + eps (dependent rows: the rank must not notice)."""
    recs, true = gen.graph_sketch(family, npts, rng)
    recs = recs.copy()
    recs["param"] *= scale  # (every parameter of these sketches is a length or a coordinate)
    true = true * scale
    kept, gone = (recs[:-drop], recs[-drop:]) if drop else (recs, recs[:0])
    extra = []
    for c in gone[:weak]:
        d = c.copy()
        d["weight"] = eps
        extra.append(d)
    for _ in range(dup):
        d = kept[int(rng.integers(2, len(kept)))].copy()
        d["weight"] = 1.0 + eps
        extra.append(d)
    if extra:
        kept = np.concatenate([kept, np.array(extra, dtype=recs.dtype)])
    return kept, true


@pytest.mark.parametrize("family", FAMILIES)
def test_probes_equal_the_pivoted_qr_over_the_fuzz(E, family, monkeypatch):
    """Per system, by what the reference's rule itself can tell (numpy's singular values of the same Jacobian, relative to its
    largest column norm -- the reference's first pivot):
    * CLEAR (no singular value within 1e-10 ... 1e-6 of the scale: every pivot is orders of magnitude on one side of the rule's
      1e-8): the id list equals the oracle's dense pivoted QR, by probes or by the device's QR;
    * otherwise (a weak row holds a direction by about the threshold: the reference's own answer flips with the last bits of
      J): the probes must not have decided on their own -- the answer is the device's own pivoted QR's (EZPZ_FREEDOM_PROBES=0) bit
      for bit, or the oracle's.
    Variables whose participation lies within a factor 4 of the reference's cut (1e-3 of the largest, squared: find_dof.rs:90-103)
    are left out of the comparison of id lists: the cut is sharp, their side of it is not."""
    rng = np.random.default_rng({"tree": 11, "band": 12, "hub": 13, "comb": 14}[family])
    before = exits(E)
    checked = clear = mismatch = 0
    log = []
    for npts in (24, 60, 150, 400, 700):
        reps = {24: 16, 60: 12, 150: 8, 400: 4, 700: 2}[npts]
        for rep in range(reps):
            scale = float(10.0 ** rng.uniform(-3, 3))
            drop = int(rng.integers(0, 7))
            weak = int(rng.integers(0, drop + 1))
            dup = int(rng.integers(0, 3))
            eps = float(rng.choice([1e-11, 1e-10, 1e-9, 3e-9, 1e-8, 3e-8, 1e-7, 1e-6, 1e-5, 1e-4]))
            recs, true = variant(rng, family, npts, scale, drop, weak, dup, eps)
            n = len(true)
            sysobj = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
            # at the true layout (consistent: every residual ~0) and at two jittered, re-solved starts
            X = np.stack([true, true + scale * rng.uniform(-0.01, 0.01, n), true + scale * rng.uniform(-0.01, 0.01, n)])
            x, st, _ = sysobj.solve_batch(X, E.Config(max_iterations=60))
            x[0] = true
            monkeypatch.delenv("EZPZ_FREEDOM_PROBES", raising=False)
            mask, part = sysobj.freedom_batch(x)
            monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")
            mask_qr, part_qr = sysobj.freedom_batch(x)
            monkeypatch.delenv("EZPZ_FREEDOM_PROBES", raising=False)
            _, J, _ = sysobj.eval_batch(x)
            for b in range(len(x)):
                checked += 1
                sv = np.linalg.svd(J[b], compute_uv=False)
                ref_scale = float(np.sqrt((J[b] ** 2).sum(axis=0).max()))
                rel = sv / ref_scale
                is_clear = not np.any((rel > 1e-10) & (rel < 1e-6))
                tag = f"{family} npts {npts} rep {rep} system {b}: scale {scale:.3g} drop {drop} weak {weak} dup {dup} eps {eps:g}"
                if is_clear:
                    clear += 1
                    under, want = O.freedom_analysis_dense(J[b])
                    cut = (1e-3 * np.sqrt(want.max())) ** 2 if want.max() > 0 else 0.0
                    sure = ~((want > 0.25 * cut) & (want < 4.0 * cut)) if cut > 0 else np.ones(n, bool)
                    want_mask = np.zeros(n, bool)
                    want_mask[under] = True
                    for name, got in (("probes", mask[b].astype(bool)), ("device QR", mask_qr[b].astype(bool))):
                        if not np.array_equal(got[sure], want_mask[sure]):
                            mismatch += 1
                            log.append(f"MISMATCH ({name}) {tag}: {int(got.sum())} vs oracle {len(under)} underconstrained; fronts {sysobj.info()['front_workgroups']}")
                elif not np.array_equal(mask[b], mask_qr[b]):
                    # (the two pivoted QRs may themselves part ways here -- the device's and the oracle's, the same algorithm in another
                    # summation order, on a pivot within a factor of ten of the rule's threshold: either is the reference's answer)
                    under, want = O.freedom_analysis_dense(J[b])
                    if np.nonzero(mask[b])[0].tolist() == under:
                        continue
                    mismatch += 1
                    log.append(f"MISMATCH (probes decided where the rank rule is at its threshold) {tag}: {int(mask[b].sum())} vs device QR {int(mask_qr[b].sum())} "
                               f"(the oracle's QR: {len(under)}); singular values near it: {np.sort(rel[(rel > 1e-10) & (rel < 1e-6)])[:4]}, below: {int((rel <= 1e-10).sum())}; "
                               f"participation sums: probes {part[b].sum():.3f}, device QR {part_qr[b].sum():.3f}, oracle {want.sum():.3f}")
    took = exits(E) - before
    names = ["fully constrained by 8 probes", "null vectors found", "second opinions", "QR: not finite", "QR: >= 5 candidates",
             "QR: a direction the reference's rule must decide", "QR: unsettled", "QR: probes not applicable"]
    line = f"{family}: {checked} systems ({clear} clear of the rank rule's threshold), {mismatch} mismatches; exits: " + ", ".join(f"{n} {int(t)}" for n, t in zip(names, took))
    print(line)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "freedom_probes_fuzz.txt"), "a") as f:
            f.write(line + "\n" + "".join(l + "\n" for l in log))
    assert mismatch == 0, "\n".join(log[:10])
    assert checked >= 120 and clear >= 40


def test_pivoted_qr_on_several_workgroups_is_repeatable(E, monkeypatch):
    """The device's pivoted QR of systems too large for one workgroup has three routes (freedom.hip: the matrix resident in
    registers, one cooperative launch streaming it, a chain of launches per step).  Each must give the same bits on the same
    input every time -- round 6's fuzz caught the cooperative one, three 1400-variable systems side by side, differing from call
    to call (a store still on its way when the rendezvous wrote the L2 back; a store racing with the pivot search) -- and all
    three the same id lists where no pivot sits at the rank rule's threshold."""
    rng = np.random.default_rng(5)
    recs, true = variant(rng, "tree", 700, 7.66, 6, 1, 1, 1e-5)
    n = len(true)
    sysobj = E.System(recs, n, team_size=E.TEAM_AUTO_LATENCY)
    X = np.stack([true, true + 0.05 * rng.uniform(-1, 1, n), true + 0.05 * rng.uniform(-1, 1, n)])
    x, st, _ = sysobj.solve_batch(X, E.Config(max_iterations=60))
    x[0] = true
    monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")
    masks = {}
    for chain in ("1", "2", ""):
        monkeypatch.setenv("EZPZ_FREEDOM_CHAIN", chain)
        for xs in (x, x[1:2]):  # three systems side by side; one (the resident route where it fits)
            runs = [sysobj.freedom_batch(xs) for _ in range(6)]
            for m, p in runs[1:]:
                assert np.array_equal(m, runs[0][0]) and np.array_equal(p, runs[0][1]), f"route {chain!r}, {len(xs)} systems: not repeatable"
            masks[(chain, len(xs))] = runs[0][0]
    for key, m in masks.items():
        assert np.array_equal(m, masks[("1", 3)][-len(m):] if len(m) == 3 else masks[("1", 3)][1:2]), key
    assert masks[("1", 3)].sum(axis=1).min() >= 5  # (five constraints dropped for good: the degrees of freedom are there)


def test_large_qr_beside_a_co_running_kernel(E, tmp_path, monkeypatch):
    """The resident route of the large pivoted QR is a cooperative launch over the whole device (250 workgroups of 1024 lanes at 2000
    variables): beside another stream's or process's kernel some of them never become resident, the waiting ones give up after their
    bound and flag the system.  The host entry then runs the chain of launches, which waits for nobody (round 6; before: EZPZ_ERR_HIP):
    the analysis beside tools/noise.hip's kernel equals the one taken alone."""
    import ctypes
    import subprocess

    so = str(tmp_path / "libnoise.so")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "noise.hip")],
                          stderr=subprocess.DEVNULL)
    N = ctypes.CDLL(so)
    N.noise_start.argtypes = [ctypes.c_int]
    recs, g = gen.connected_sketch(1000, 2000)
    loose = E.System(recs[:-2], len(g), team_size=E.TEAM_AUTO_LATENCY)
    x, st, _ = loose.solve_batch(g[None, :], E.Config(max_iterations=40))
    monkeypatch.setenv("EZPZ_FREEDOM_PROBES", "0")
    mask0, part0 = loose.freedom_batch(x)
    assert mask0.sum() > 0
    assert N.noise_start(160) == 0
    try:
        mask1, part1 = loose.freedom_batch(x)
    finally:
        assert N.noise_stop() == 0
    assert np.array_equal(mask1, mask0) and np.allclose(part1, part0, atol=1e-9)
    mask2, part2 = loose.freedom_batch(x)  # (and alone again: the resident route, the same bits as before)
    assert np.array_equal(mask2, mask0) and np.array_equal(part2, part0)
