"""A sensitivity-aware parity bar for solves whose LM path amplifies rounding.

The HIP kernels of connected sketches factorise in another elimination order than the oracle (and sum long lists by
groups of lanes): every operation is the reference's, the rounding is not.  On a well-conditioned solve that moves the
answer by a few ulps; on an ill-conditioned one (long chains of distance constraints, under-determined systems held by
lambda alone) the ORACLE ITSELF moves by far more than 1e-6 -- and may change its iteration count, or stop converging --
when every start coordinate is moved by one ulp.  So the bar for a system is measured on the oracle, not assumed:

    coordinates   |x - x_oracle| <= max(1e-6, 20 x the largest difference among the oracle's own answers from the
                  start and from K copies of it with every coordinate moved by +-1 ulp), relative to max(1, |x_oracle|)
    iterations    equal to the oracle's, or inside the range of counts those K + 1 oracle runs produce (the counts of a
                  chaotic path are samples -- comb 51 of the graph fuzz gives 18, 20, 22, 24, 28, 32 ... 50 over 96
                  perturbations -- so K grows 8 -> 32 -> 96 before a count is declared outside)
    converged     among the flags those runs produce

and every system of a test is checked -- none is excluded.  The K extra oracle runs are only made for the systems that
miss the plain bar (1e-6, equal iterations), which keeps the tests fast."""
import numpy as np

from oracle import oracle as O

K_PERTURBED = 8


def perturbed_starts(x0_row, k=K_PERTURBED, seed=0):
    """k copies of the start with every coordinate moved by one ulp: all up, all down, then random signs."""
    rng = np.random.default_rng(seed)
    up, down = np.nextafter(x0_row, np.inf), np.nextafter(x0_row, -np.inf)
    out = [up, down]
    while len(out) < k:
        out.append(np.where(rng.integers(0, 2, len(x0_row)).astype(bool), up, down))
    return np.stack(out[:k])


def oracle_spread(recs, x0_row, cfg=None, linsolve=O.LINSOLVE_SPARSE, k=K_PERTURBED):
    """(iteration counts, converged flags, largest relative difference of the answers) over the oracle's runs from the
    start and from its k one-ulp perturbations."""
    starts = np.concatenate([x0_row[None, :], perturbed_starts(x0_row, k)])
    rc, xo, it, conv, nun = O.solve_batch(recs, starts, cfg, linsolve=linsolve)
    assert rc == 0
    with np.errstate(invalid="ignore"):
        diff = np.abs(xo[1:] - xo[0]) / np.maximum(1.0, np.abs(xo[0]))
    spread = float(np.nanmax(diff)) if np.any(~np.isnan(diff)) else 0.0
    return set(int(v) for v in it), set(bool(v) for v in conv), spread


def assert_batch_matches_oracle(recs, x0, x, iterations, converged, cfg=None, linsolve=O.LINSOLVE_SPARSE, rel=1e-6,
                                oracle_result=None, what="", check_iterations=True):
    """Every system of a batch against the oracle with the bar above.  `oracle_result` = (xo, it, conv) when the caller
    has it already.  Returns how many systems needed the measured bar (the rest met 1e-6 / equal iterations).
    `check_iterations=False`: for solves that end on the step test at a least-squares minimum of an inconsistent system,
    where the count is decided by when the noise in |d| first drops below 1e-12 (coordinates and flags only)."""
    x0, x = np.asarray(x0), np.asarray(x)
    if oracle_result is None:
        rc, xo, it, conv, _ = O.solve_batch(recs, x0, cfg, linsolve=linsolve)
        assert rc == 0
    else:
        xo, it, conv = oracle_result
    assert np.array_equal(np.isnan(x), np.isnan(xo)), what
    with np.errstate(invalid="ignore"):
        err = np.abs(x - xo) / np.maximum(1.0, np.abs(xo))
    err = np.where(np.isnan(err), 0.0, err).max(axis=1) if x.shape[1] else np.zeros(len(x))
    plain = (err <= rel) & (np.asarray(converged).astype(bool) == np.asarray(conv).astype(bool))
    if check_iterations:
        plain &= np.asarray(iterations).astype(np.int64) == np.asarray(it).astype(np.int64)
    needed = 0
    for b in np.nonzero(~plain)[0]:
        needed += 1
        for k in (K_PERTURBED, 32, 96):
            its, convs, spread = oracle_spread(recs, x0[b], cfg, linsolve, k)
            inside = (not check_iterations or min(its) <= int(iterations[b]) <= max(its)) and bool(converged[b]) in convs and \
                err[b] <= max(rel, 20.0 * spread)
            if inside:
                break
        if check_iterations:
            assert min(its) <= int(iterations[b]) <= max(its), (what, int(b), int(iterations[b]), sorted(its))
        assert bool(converged[b]) in convs, (what, int(b), bool(converged[b]), convs)
        assert err[b] <= max(rel, 20.0 * spread), (what, int(b), float(err[b]), spread)
    return needed
