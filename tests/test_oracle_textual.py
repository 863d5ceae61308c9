"""The oracle's .md loader against the reference's fixtures (parser.rs:29-555, executor.rs:40-445)."""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, read_case
from oracle import oracle as O
from oracle import textual as T


def test_every_fixture_parses():
    files = sorted(glob.glob(os.path.join(GOLDEN, "test_cases", "*", "*.md")))
    assert len(files) == 29
    for f in files:
        cs = T.load(open(f).read())
        assert len(cs.constraints) > 0 and cs.num_vars > 0


def test_generator_reproduces_committed_massive_file():
    """The committed problem.md is gen_big_problem.py 600."""
    assert T.gen_big_problem(600) == read_case("massive_parallel_system")


def test_massive_variable_and_constraint_order():
    """p_k -> ids (2k, 2k+1); per line: Vertical, Fixed x, Fixed y, Fixed y."""
    cs = T.load(T.gen_big_problem(3))
    assert cs.num_vars == 12 and len(cs.constraints) == 12
    kinds = cs.constraints["kind"].tolist()
    assert kinds == [O.VERTICAL, O.FIXED, O.FIXED, O.FIXED] * 3
    c = cs.constraints
    assert c["ids"][4][:4].tolist() == [4, 5, 6, 7]
    assert (c["ids"][5][0], c["param"][5]) == (4, 1.0)
    assert (c["ids"][6][0], c["param"][6]) == (5, 0.0)
    assert (c["ids"][7][0], c["param"][7]) == (7, 4.0)
    assert cs.guesses.tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5]
    over = T.load(T.gen_big_problem(2, True))
    assert over.constraints["kind"].tolist() == [O.VERTICAL, O.FIXED, O.FIXED, O.FIXED, O.DISTANCE] * 2


def test_id_layout_points_circles_arcs():
    """geometry_variables.rs:56-177"""
    cs = T.load(read_case("circle_tangent"))
    # points p,q -> 0..3 ; circle a -> center (4,5), radius 6
    tang = cs.constraints[cs.constraints["kind"] == O.LINE_TANGENT_TO_CIRCLE][0]
    assert tang["ids"][:7].tolist() == [0, 1, 2, 3, 4, 5, 6] and tang["tag"] == O.SIDE_UNDEFINED
    arc = T.load(read_case("arc_radius"))
    # point p -> 0,1 ; arc a: a=(2,3) b=(4,5) center=(6,7); record order center,start,end
    ar = arc.constraints[arc.constraints["kind"] == O.ARC_RADIUS][0]
    assert ar["ids"][:6].tolist() == [6, 7, 2, 3, 4, 5]
    assert arc.guesses.tolist() == [4, 3, 0, 4, 4, 0, 0.1, 0.2]
    fx = arc.constraints[arc.constraints["kind"] == O.FIXED]
    assert fx["ids"][:, 0].tolist() == [6, 7]


def test_sqrt_and_angle_units():
    cs = T.load(read_case("angle_parallel_manual"))
    la = cs.constraints[cs.constraints["kind"] == O.LINES_AT_ANGLE][0]
    assert la["tag"] == O.ANGLE_OTHER_DEG and la["param"] == 720.0
    d = cs.constraints[cs.constraints["kind"] == O.DISTANCE][0]
    assert d["param"] == np.sqrt(32.0)


@pytest.mark.parametrize("bad", [
    "# constraints\npoint p\n\n\n# guesses\np roughly (0,0)\n",      # two blank lines
    "# constraints\npoint p \n\n# guesses\np roughly (0,0)\n",       # trailing space
    "# constraints\nfrobnicate(p)\n\n# guesses\np roughly (0,0)\n",  # unknown instruction
    "# constraints\npoint p\n\n# guesses\np about (0,0)\n",
    "point p\n\n# guesses\np roughly (0,0)\n",
])
def test_malformed_text_is_rejected(bad):
    with pytest.raises(T.ParseError):
        T.parse_problem(bad)


def test_textual_errors():
    """executor.rs:673-744"""
    with pytest.raises(T.TextualError) as e:
        T.load("# constraints\npoint p\n\n# guesses\nq roughly (0,0)\n")
    assert e.value.kind == "MissingGuess"
    with pytest.raises(T.TextualError) as e:
        T.load("# constraints\npoint p\n\n# guesses\np roughly (0,0)\nghost roughly (1,1)\n")
    assert e.value.kind == "UnusedGuesses"
    with pytest.raises(T.TextualError) as e:
        T.load("# constraints\npoint p\nmissing.x = 2.5\n\n# guesses\np roughly (0,0)\n")
    assert e.value.kind == "UndefinedPoint"


def test_arc_center_assignment_is_silently_dropped():
    """executor.rs:273-283: `X.center = (..)` only resolves circles."""
    cs = T.load("# constraints\narc a\na.center = (0, 0)\nis_arc(a)\n\n# guesses\na.center roughly (0, 0)\n"
                "a.a roughly (0, 5)\na.b roughly (5, 0)\n")
    assert cs.constraints["kind"].tolist() == [O.ARC]
