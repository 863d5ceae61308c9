"""Generates tests/golden/oracle_vectors.json from the CPU oracle (which is itself pinned to the reference's
known answers by tests/test_oracle_pins.py).  Inputs -> expected outputs for every reference fixture plus
jittered variants, consumed by the GPU parity tests.  Run: python tests/golden/make_vectors.py"""
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import gen  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import textual as T  # noqa: E402


def main():
    out = {}
    for path in sorted(glob.glob(os.path.join(HERE, "test_cases", "*", "*.md"))):
        case = os.path.relpath(path, os.path.join(HERE, "test_cases"))
        if case.startswith("massive"):
            continue
        cs = T.load(open(path).read())
        variants = [cs.guesses]
        jit = gen.keyed_uniform(0x657A707A, 3, cs.num_vars, -0.1, 0.1)
        variants += [cs.guesses + jit[k] for k in range(3)]
        recs = []
        for g in variants:
            o = O.solve(cs.constraints, g, analysis=True)
            # Conditioning of the answer itself: how far the oracle's own result moves when the inputs are
            # perturbed by one ulp.  Under-determined systems are regularised only by lambda ~ 1e-9..1e-10, so
            # rounding noise in the null space is amplified by ~1/lambda and 1e-6 agreement is not defined there.
            sens = 0.0
            for t in range(4):
                rng = np.random.default_rng(1234 + t)
                o2 = O.solve(cs.constraints, g * (1.0 + rng.uniform(-1.0, 1.0, len(g)) * 2.0 ** -52))
                if o2.iterations == o.iterations:
                    sens = max(sens, float(np.max(np.abs(o2.final_values - o.final_values)
                                                  / np.maximum(1.0, np.abs(o.final_values)))))
            recs.append({
                "ulp_sensitivity": sens,
                # FreedomAnalysis of the oracle (find_dof.rs): the variables no constraint determines.  Only these
                # may use the widened tolerance in the GPU parity test; every other coordinate is held to 1e-6.
                "underconstrained": o.underconstrained,
                "guesses": [float(v) for v in g],
                "final_values": [float(v) for v in o.final_values],
                "iterations": o.iterations,
                "converged": o.converged,
                "unsatisfied": o.unsatisfied,
                "n_warnings": len(o.warnings),
                "final_residual_inf": o.final_residual_inf,
            })
        out[case] = recs
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out), "cases")


if __name__ == "__main__":
    main()
