"""CPU-side checks of the product: the C-ABI library loads and exports every symbol include/ezpz_amd.h
declares, the C++ text front end agrees with the oracle's loader, the host symbolic phase reproduces the
sizes SURVEY.md 8(d) states, and the host mirror packs the same records as the oracle's constructors.
No compute entry point is called here (that needs a GPU; see tests/test_gpu_parity.py)."""
import ctypes as C
import glob
import os
import re
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, read_case
from oracle import oracle as O
from oracle import textual as T

import ezpz_amd as E
from ezpz_amd._lib import EXPORTS


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ezpz_amd.h")).read()
    declared = set(re.findall(r"\b(ezpz_[a-z_]+)\s*\(", header))
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    L = E.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_launch_policy_table_and_its_scaling():
    """csrc/policy.hpp: the one table of launch-shape thresholds.  The 256-CU values are the measured ones; the
    device-filling ones scale with the CU count (a CPX partition of 32 CUs, a 64-CU DPX-like mask), the per-workgroup ones
    do not; and the symbolic phase obeys them (a 20-variable system gets a lane plan, a 21-variable one does not)."""
    from ezpz_amd._lib import CLaunchPolicy

    def policy(cus):
        p = CLaunchPolicy()
        assert E.lib().ezpz_launch_policy(cus, C.byref(p)) == 0
        return p

    full = policy(256)
    assert (full.lanes_min_systems_small, full.lanes_min_systems_large, full.lanes_large_from_vars) == (65536, 32768, 601)
    assert (full.jit_lane_min_batch, full.jit_comp_min_batch, full.jit_comp_min_values, full.jit_after_launches) == (4096, 1024, 1 << 21, 256)
    assert (full.rec_min_vars_one_solve, full.rec_min_vars_batch, full.rec_one_wavefront_max_vars, full.rec_max_components) == (25, 57, 160, 127)
    assert (full.lane_max_vars, full.lane_max_constraints, full.comp_min_components) == (20, 40, 128)
    assert (full.zero_copy_max_bytes, full.h2h_piece_min_bytes, full.h2h_piece_max_bytes, full.h2h_pieces_per_call) == (1 << 20, 4 << 20, 16 << 20, 16)
    assert (full.front_min_vars_one_solve, full.front_min_vars_batch, full.front_vars_per_workgroup, full.front_max_workgroups,
            full.front_small_call_wgs_per_round) == (48, 48, 160, 64, 4)
    zero = policy(0)  # 0 = the full chip
    assert all(getattr(zero, f) == getattr(full, f) for f, _ in CLaunchPolicy._fields_)
    for cus in (32, 64, 128, 304):
        p = policy(cus)
        assert p.compute_units == cus
        for f in ("lanes_min_systems_small", "lanes_min_systems_large", "jit_lane_min_batch", "jit_comp_min_batch", "jit_comp_min_values"):
            assert getattr(p, f) * 256 == getattr(full, f) * cus, (f, cus)
        for f, _ in CLaunchPolicy._fields_:
            if not f.startswith(("lanes_min", "jit_lane_min", "jit_comp_min", "compute_units")):
                assert getattr(p, f) == getattr(full, f), (f, cus)
    assert E.lib().ezpz_launch_policy(-1, C.byref(CLaunchPolicy())) == -103
    # the symbolic phase obeys the table: one lane per system up to lane_max_vars variables
    def chain(npts):
        cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
        for i in range(1, npts):
            cons.append(O.distance((2 * i, 2 * i + 1), (2 * i - 2, 2 * i - 1), 1.0))
            cons.append(O.vertical((2 * i, 2 * i + 1), (2 * i - 2, 2 * i - 1)))
        return O.stack(cons), 2 * npts
    recs, n = chain(full.lane_max_vars // 2)
    assert "ezpz_jit_lane" in E.specialized_source(recs, n)
    recs, n = chain(full.lane_max_vars // 2 + 1)
    assert "ezpz_jit_lane" not in E.specialized_source(recs, n)


def test_struct_layouts_match_the_header():
    from ezpz_amd._lib import CConfig, COutcome, CSystemInfo, CWarning

    assert E.CONSTRAINT_DTYPE.itemsize == 56 and O.CONSTRAINT_DTYPE == E.CONSTRAINT_DTYPE
    assert E.STATUS_DTYPE.itemsize == 32
    assert C.sizeof(CConfig) == 32 and C.sizeof(CWarning) == 8 and C.sizeof(COutcome) == 80
    assert C.sizeof(CSystemInfo) == 112


def test_default_config():
    from ezpz_amd._lib import CConfig

    c = CConfig()
    E.lib().ezpz_default_config(C.byref(c))
    assert (c.max_iterations, c.residual_tolerance, c.step_tolerance, c.initial_lambda) == (35, 1e-8, 1e-12, 1e-9)
    assert E.Config() == E.Config(35, 1e-8, 1e-12, 1e-9)
    assert E.Config().with_max_iterations(200).with_convergence_tolerance(1e-10).max_iterations == 200


def test_cpp_front_end_matches_oracle_loader_on_every_fixture():
    files = sorted(glob.glob(os.path.join(GOLDEN, "test_cases", "*", "*.md")))
    assert len(files) == 29
    for f in files:
        text = open(f).read()
        a = E.textual.Problem.from_str(text).to_constraint_system()
        b = T.load(text)
        assert a.records.tobytes() == b.constraints.tobytes(), f
        assert np.array_equal(a.guesses, b.guesses), f
        assert (a.inner_points, a.inner_circles, a.inner_arcs) == (b.inner_points, b.inner_circles, b.inner_arcs)


@pytest.mark.parametrize("bad", [
    "# constraints\npoint p\n\n\n# guesses\np roughly (0,0)\n",
    "# constraints\npoint p \n\n# guesses\np roughly (0,0)\n",
    "# constraints\nfrobnicate(p)\n\n# guesses\np roughly (0,0)\n",
    "# constraints\npoint p\n\n# guesses\np about (0,0)\n",
])
def test_cpp_front_end_rejects_malformed_text(bad):
    with pytest.raises(E.textual.TextualError) as e:
        E.textual.Problem.from_str(bad)
    assert e.value.code == -110


def test_cpp_front_end_textual_errors():
    """executor.rs:673-744"""
    for text, code in [("# constraints\npoint p\n\n# guesses\nq roughly (0,0)\n", -111),
                       ("# constraints\npoint p\n\n# guesses\np roughly (0,0)\nghost roughly (1,1)\n", -112),
                       ("# constraints\npoint p\nmissing.x = 2.5\n\n# guesses\np roughly (0,0)\n", -113)]:
        with pytest.raises(E.textual.TextualError) as e:
            E.textual.Problem.from_str(text)
        assert e.value.code == code


def test_symbolic_sizes_massive_and_square():
    """massive N=500 -> C=m=n=2000, zJ=2500, zA=2500, zL=2500; square -> 10/10/8, zJ=40, A dense 8x8."""
    cs = T.load(T.gen_big_problem(500))
    i = E.analyze(cs.constraints, cs.num_vars)
    assert (i["n_constraints"], i["n_rows"], i["n_vars"], i["nnz_j"], i["nnz_a"], i["nnz_l"]) == (2000, 2000, 2000, 2500, 2500, 2500)
    assert i["n_levels"] == 2 and i["workspace_in_lds"] == 1
    sq = T.load(read_case("square"))
    j = E.analyze(sq.constraints, sq.num_vars)
    assert (j["n_constraints"], j["n_rows"], j["n_vars"], j["nnz_j"], j["nnz_a"], j["nnz_l"]) == (10, 10, 8, 40, 36, 36)


def test_symbolic_phase_of_a_connected_sketch_ends_in_a_dense_root_block():
    """Host side of the dense phases (records.cpp: make_dense_phases): on a 512-lane workgroup the top of a connected
    sketch's elimination tree -- runs of levels of one or two columns per branch -- is a few phases of the schedule;
    EZPZ_ROOT=0 keeps the plain schedule (read once per process, hence the child processes).  Block systems keep theirs, and
    so does the record walk (records.cpp: build_records), the automatic shape of such a sketch."""
    import subprocess, sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import gen, ezpz_amd as E; "
            "from oracle import textual as T; r, g = gen.connected_sketch(400, 7); cs = T.load(T.gen_big_problem(500)); "
            "print(E.analyze(r, len(g))['n_levels'], E.analyze(cs.constraints, cs.num_vars)['n_levels'])") % (ROOT, os.path.join(ROOT, "tests"))
    def levels(**env):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        return [int(v) for v in out.stdout.split()]
    # (EZPZ_REC=0: the list-walk shapes; by default a batch of one connected sketch walks records over every level)
    with_block, plain, records = levels(EZPZ_REC="0"), levels(EZPZ_REC="0", EZPZ_ROOT="0"), levels()
    assert with_block[0] + 5 <= plain[0] and with_block[1] == plain[1] == 2, (with_block, plain)
    assert records == plain, (records, plain)


def _shapes_analyse(recs, n):
    """Every automatic launch shape: where there is no device, creation analyses the system and then reports -100; where
    there is one it succeeds."""
    for team in (0, E.TEAM_AUTO_LATENCY, E.TEAM_LATENCY_PHASES, E.TEAM_BATCH_LANES):
        try:
            E.System(recs, n, team_size=team)
        except E.NonLinearSystemError as e:
            assert e.code == -100, (team, e.code)


def test_symbolic_phase_survives_random_sketches_at_the_edge_of_the_lds():
    """The launch-shape code re-packs a program after cutting its dense phases; sketches whose workspace almost fills
    the LDS (1408 variables: 156 KB) once came back from that as "too large" (the shorter level table let the level
    staging buffer in, or the shorter lists let the program move into LDS, and either took the panels' room).  The cases
    a 3000-system fuzz found, and a sample of its four graph families (band, hub, grid, random tree with chords)."""
    import gen
    rng = np.random.default_rng(5)
    for npts, seed in ((704, 288), (286, 500), (278, 688), (294, 1056)):
        recs, g = gen.connected_sketch(npts, seed)
        i = E.analyze(recs, len(g))
        assert i["n_vars"] == 2 * npts and i["n_components"] == 1
        _shapes_analyse(recs, len(g))
    # (a 100-variable comb on wavefront teams: the re-pack chose a larger workgroup than the panels had been sized for)
    for seed in (159, 83, 121, 200):
        r2 = np.random.default_rng(7000 + seed)
        recs, true = gen.graph_sketch(["tree", "band", "hub", "comb"][seed % 4], int(r2.integers(20, 500)), r2)
        _shapes_analyse(recs, len(true))
    pt = lambda i: (2 * i, 2 * i + 1)
    for trial in range(48):
        npts = int(rng.integers(10, 700))
        cons = [O.fixed(0, 0.0), O.fixed(1, 0.0)]
        if trial % 4 == 0:
            recs, g = gen.connected_sketch(npts, 100 + trial)
        else:
            if trial % 4 == 1:  # hub
                for k in range(1, npts):
                    cons += [O.distance(pt(k), pt(0), 1.0 + k), O.horizontal_distance(pt(k), pt(0), 0.5 * k)]
            elif trial % 4 == 2:  # grid: every point tied to its left and upper neighbours
                w = int(np.sqrt(npts)) + 1
                npts = w * w
                for k in range(1, npts):
                    r, c = divmod(k, w)
                    if c > 0:
                        cons.append(O.distance(pt(k), pt(k - 1), 1.0))
                    if r > 0:
                        cons.append(O.distance(pt(k), pt(k - w), 1.0))
                    cons.append(O.horizontal_distance(pt(k), pt(k - w), 0.0) if c == 0 else O.vertical_distance(pt(k), pt(k - 1), 0.0) if r == 0
                                else O.fixed(2 * k, float(c)))
            else:  # random tree with chords
                for k in range(1, npts):
                    a, b = int(rng.integers(0, k)), int(rng.integers(0, k))
                    cons += [O.distance(pt(k), pt(a), 1.0), O.vertical_distance(pt(k), pt(b), 0.3) if a != b else O.horizontal_distance(pt(k), pt(a), 0.2)]
            recs, g = O.stack(cons), rng.uniform(-5, 5, 2 * npts)
        i = E.analyze(recs, len(g))
        assert i["n_vars"] == len(g) and i["n_levels"] >= 1
        _shapes_analyse(recs, len(g))


def test_symbolic_phase_reports_missing_guess():
    """solver.rs:142-189: first id (row0 then row1, constraint order) that has no guess."""
    with pytest.raises(E.NonLinearSystemError) as e:
        E.analyze([O.fixed(0, 1.0), O.points_coincident((0, 1), (2, 7))], 4)
    assert (e.value.code, e.value.constraint_id, e.value.variable) == (-3, 1, 7)


def test_host_mirror_packs_the_same_records_as_the_oracle_constructors():
    ids = E.IdGenerator()
    p0, p1, p2, p3 = (E.DatumPoint.new(ids) for _ in range(4))
    l0, l1 = E.DatumLineSegment.new(p0, p1), E.DatumLineSegment.new(p2, p3)
    circ = E.DatumCircle(p2, E.DatumDistance.new(ids.next_id()))
    circ2 = E.DatumCircle(p3, E.DatumDistance.new(ids.next_id()))
    arc = E.DatumCircularArc(p0, p1, p2)
    t = lambda p: (p.x_id, p.y_id)
    pairs = [
        (E.Constraint.LineTangentToCircle(l0, circ, E.LineSide.Right), O.line_tangent_to_circle(t(p0), t(p1), t(p2), 8, O.LINE_RIGHT)),
        (E.Constraint.CircleTangentToCircle(circ, circ2, E.CircleSide.Interior), O.circle_tangent_to_circle(t(p2), 8, t(p3), 9, O.CIRCLE_INTERIOR)),
        (E.Constraint.Distance(p0, p1, 2.5), O.distance(t(p0), t(p1), 2.5)),
        (E.Constraint.DistanceVar(p0, p1, circ.radius), O.distance_var(t(p0), t(p1), 8)),
        (E.Constraint.VerticalDistance(p0, p1, 1.0), O.vertical_distance(t(p0), t(p1), 1.0)),
        (E.Constraint.HorizontalDistance(p0, p1, 1.0), O.horizontal_distance(t(p0), t(p1), 1.0)),
        (E.Constraint.Vertical(l0), O.vertical(t(p0), t(p1))),
        (E.Constraint.Horizontal(l0), O.horizontal(t(p0), t(p1))),
        (E.Constraint.LinesAtAngle(l0, l1, E.AngleKind.Other(E.Angle.from_degrees(30))), O.lines_at_angle(t(p0), t(p1), t(p2), t(p3), ("deg", 30.0))),
        (E.Constraint.lines_parallel([l0, l1]), O.lines_at_angle(t(p0), t(p1), t(p2), t(p3), "parallel")),
        (E.Constraint.lines_perpendicular([l0, l1]), O.lines_at_angle(t(p0), t(p1), t(p2), t(p3), "perpendicular")),
        (E.Constraint.Fixed(3, 1.5), O.fixed(3, 1.5)),
        (E.Constraint.ScalarEqual(8, 9), O.scalar_equal(8, 9)),
        (E.Constraint.PointsCoincident(p0, p1), O.points_coincident(t(p0), t(p1))),
        (E.Constraint.CircleRadius(circ, 2.0), O.circle_radius(t(p2), 8, 2.0)),
        (E.Constraint.LinesEqualLength(l0, l1), O.lines_equal_length(t(p0), t(p1), t(p2), t(p3))),
        (E.Constraint.ArcRadius(arc, 5.0), O.arc_radius(t(p0), t(p1), t(p2), 5.0)),
        (E.Constraint.Arc(arc), O.arc(t(p0), t(p1), t(p2))),
        (E.Constraint.Midpoint(l0, p2), O.midpoint(t(p0), t(p1), t(p2))),
        (E.Constraint.PointLineDistance(p2, l0, 1.0), O.point_line_distance(t(p2), t(p0), t(p1), 1.0)),
        (E.Constraint.VerticalPointLineDistance(p2, l0, 1.0), O.vertical_point_line_distance(t(p2), t(p0), t(p1), 1.0)),
        (E.Constraint.HorizontalPointLineDistance(p2, l0, 1.0), O.horizontal_point_line_distance(t(p2), t(p0), t(p1), 1.0)),
        (E.Constraint.Symmetric(l0, p2, p3), O.symmetric(t(p0), t(p1), t(p2), t(p3))),
        (E.Constraint.PointArcCoincident(arc, p3), O.point_arc_coincident(t(p0), t(p1), t(p2), t(p3))),
        (E.Constraint.ArcLength(arc, 3.0), O.arc_length(t(p0), t(p1), t(p2), 3.0)),
        (E.Constraint.ArcAngle(arc, E.Angle.from_radians(0.5)), O.arc_angle(t(p0), t(p1), t(p2), ("rad", 0.5))),
        (E.Constraint.PointsAtAngle(p0, p1, p2, E.AngleKind.Other(E.Angle.from_radians(0.25))), O.points_at_angle(t(p0), t(p1), t(p2), ("rad", 0.25))),
    ]
    assert sorted({c.kind for c, _ in pairs}) == list(range(25))
    for mine, theirs in pairs:
        assert E.ConstraintRequest.highest_priority(mine).record().tobytes() == theirs.tobytes(), mine
    req = E.ConstraintRequest.new(E.Constraint.Fixed(0, 1.0), 3).with_weight(2.0)
    assert req.record().tobytes() == O.fixed(0, 1.0, priority=3, weight=2.0).tobytes()


def test_solve_without_a_device_fails_loudly():
    """No CPU fallback: without a HIP device solve() raises NonLinearSystemError(EZPZ_ERR_NO_DEVICE)."""
    if E.device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(E.FailureOutcome) as e:
        E.solve([E.ConstraintRequest.highest_priority(E.Constraint.Fixed(0, 1.0))], [(0, 0.5)])
    assert e.value.error.code == -100


def test_product_never_imports_the_oracle():
    for path in glob.glob(os.path.join(ROOT, "ezpz_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
            src = open(path).read()
            assert "oracle" not in src.lower() or path.endswith("kinds.hpp"), path


def test_specialized_kernel_source_compiles_without_a_device():
    """The class-specialised kernel of a component plan: generated source + the embedded device headers compile for
    gfx950 with hiprtc here (no GPU needed); requests without a component plan have none."""
    import ezpz_amd as E
    from oracle import oracle as O
    from oracle import textual as T

    ref = T.load(T.gen_big_problem(200))
    src = E.specialized_source(ref.constraints, ref.num_vars, compile=True)
    assert "struct Cls0" in src and "struct Cls1" in src and "ezpz_jit_solve" in src
    # a non-linear class (distance per line) and an arc fixture replicated: the general evaluators compile too
    ref = T.load(T.gen_big_problem(150, True))
    assert "con_jacobian" in E.specialized_source(ref.constraints, ref.num_vars, compile=True)
    ref = T.load(open(os.path.join(GOLDEN, "test_cases", "arc_length", "problem.md")).read())
    recs = []
    for r in range(130):
        for c in ref.constraints:
            c = O.set_from_initial_values(c, ref.guesses).copy()
            c["ids"] = c["ids"] + r * ref.num_vars
            recs.append(c)
    assert E.specialized_source(O.stack(recs), 130 * ref.num_vars, compile=True)
    # small systems: one lane per system, parameters as literals
    for case in ("square", "two_rectangles", "circle_tangent", "arc_radius", "parallelogram", "tiny"):
        ref = T.load(open(os.path.join(GOLDEN, "test_cases", case, "problem.md")).read())
        recs = O.stack([O.set_from_initial_values(c, ref.guesses) for c in ref.constraints])
        src = E.specialized_source(recs, ref.num_vars, compile=True)
        assert "ezpz_jit_lane" in src and "lane_kernel" in src, case
        # ... and, for the latency of one solve, the same class on one wavefront per system (tables + dispatch + tail)
        wsrc = E.specialized_source(recs, ref.num_vars, compile=True, wave=True)
        assert "ezpz_jit_wave" in wsrc and "kWavePairs" in wsrc and wsrc.startswith(src[: src.index('extern "C"')]), case
    # too large for either form: a connected sketch of 60 variables
    import gen

    recs, g = gen.connected_sketch(30, 1)
    assert E.specialized_source(recs, len(g)) == ""


def test_header_compiles_and_links_as_c11(tmp_path):
    """include/ezpz_amd.h is a C header: a strict C11 program (gcc, not hipcc) builds against it, links the shared
    library and calls ezpz_analyze (host-only) and ezpz_solve (needs a device: EZPZ_ERR_NO_DEVICE here, 0 on a GPU box)."""
    import subprocess

    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "ezpz_amd.h"

int main(void) {
    /* test_cases/tiny: p = (0,0), q = (0,0) as four Fixed constraints; guesses (3,4), (5,6) */
    EzpzConstraint cs[4];
    memset(cs, 0, sizeof cs);
    for (int i = 0; i < 4; ++i) {
        cs[i].kind = EZPZ_FIXED;
        cs[i].ids[0] = (uint32_t)i;
        cs[i].param = 0.0;
        cs[i].weight = 1.0;
    }
    const uint32_t ids[4] = {0, 1, 2, 3};
    const double guesses[4] = {3.0, 4.0, 5.0, 6.0};
    EzpzSystemInfo info;
    int32_t ec = -1;
    int64_t ev = -1;
    int rc = ezpz_analyze(cs, 4, 4, &info, &ec, &ev);
    if (rc != EZPZ_OK || info.n_rows != 4 || info.n_vars != 4 || info.nnz_j != 4) return 10;
    cs[3].ids[0] = 9; /* MissingGuess: constraint 3 references variable 9 */
    rc = ezpz_analyze(cs, 4, 4, &info, &ec, &ev);
    if (rc != EZPZ_ERR_MISSING_GUESS || ec != 3 || ev != 9) return 11;
    cs[3].ids[0] = 3;
    EzpzConfig cfg;
    ezpz_default_config(&cfg);
    if (cfg.max_iterations != 35) return 12;
    double x[4];
    uint64_t unsat[4];
    EzpzWarning warn[8];
    EzpzOutcome out;
    rc = ezpz_solve(cs, 4, ids, guesses, 4, &cfg, x, unsat, warn, 8, &out);
    printf("solve rc=%d (%s) iterations=%llu\n", rc, ezpz_error_string(rc), (unsigned long long)out.iterations);
    if (rc == EZPZ_OK) {
        for (int i = 0; i < 4; ++i)
            if (x[i] > 1e-6 || x[i] < -1e-6) return 13;
        return out.n_unsatisfied == 0 ? 0 : 14;
    }
    return rc == EZPZ_ERR_NO_DEVICE && ezpz_device_count() == 0 ? 0 : 15;
}
''')
    libdir = os.path.join(ROOT, "ezpz_amd")
    E.lib()  # built
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lezpz_amd", "-Wl,-rpath," + libdir])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "solve rc=" in r.stdout


def test_code_object_cache_on_disk(tmp_path):
    """jit.cpp keeps compiled kernels under $EZPZ_JIT_CACHE_DIR: the first compilation writes one file, the second run of the
    same request reads it (no compiler), a damaged file is a miss that gets rewritten, another topology gets its own
    file, and EZPZ_JIT_CACHE=0 writes nothing.  (Fresh processes: the directory is read from the environment once.)"""
    import subprocess
    import textwrap

    script = textwrap.dedent("""
        import sys, time
        sys.path.insert(0, %r)
        import ezpz_amd as E
        lines = int(sys.argv[1])
        b = E.textual.Problem.from_str(E.textual.gen_big_problem(lines)).to_constraint_system()
        t = time.perf_counter()
        assert "ezpz_jit_solve" in E.specialized_source(b.records, b.num_vars, compile="cached")
        print("compilations", E.lib().ezpz_debug_jit_compilations())
    """ % ROOT)
    cache = tmp_path / "cache"

    def run(lines, **env):
        r = subprocess.run([sys.executable, "-c", script, str(lines)], env=dict(os.environ, EZPZ_JIT_CACHE_DIR=str(cache), **env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return int(r.stdout.split()[-1])  # hiprtc compilations of that process (not wall-clock: a loaded machine compiles slowly)

    assert run(200) == 1  # cold
    files = sorted(os.listdir(cache))
    assert len(files) == 1 and files[0].endswith(".co") and os.path.getsize(cache / files[0]) > 4096
    assert run(200) == 0 and sorted(os.listdir(cache)) == files  # warm: the kernel comes from the file
    # a damaged file (a flipped byte in the code) is a miss, and the kernel is compiled and stored again
    path = cache / files[0]
    blob = bytearray(path.read_bytes())
    blob[-100] ^= 0xFF
    path.write_bytes(bytes(blob))
    assert run(200) == 1 and path.read_bytes() != bytes(blob)
    path.write_bytes(bytes(blob[: len(blob) // 2]))  # and a truncated one
    assert run(200) == 1
    assert os.path.getsize(path) == len(blob)
    assert run(240) == 1  # another topology: its own file
    assert len(os.listdir(cache)) == 2
    off = tmp_path / "off"
    subprocess.run([sys.executable, "-c", script, "200"], env=dict(os.environ, EZPZ_JIT_CACHE_DIR=str(off), EZPZ_JIT_CACHE="0"), check=True,
                   capture_output=True, timeout=600)
    assert not off.exists()
