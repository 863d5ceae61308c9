"""Python restatement of the ezpz `.md` problem-file front end -- TEST INFRASTRUCTURE ONLY.

Follows (all paths relative to /root/reference/):
  ezpz/src/textual/parser.rs:29-555          grammar (winnow combinators, tried in `alt` order)
  ezpz/src/textual/executor.rs:40-445        label -> variable ids, instruction -> Constraint
  ezpz/src/textual/geometry_variables.rs:56-177   id layout (points x2, circles x3, arcs a,b,center x6)

Reference quirks that are kept on purpose:
  * `arc_ids` offsets arcs by 2*num_points only, ignoring circles (geometry_variables.rs:92).
  * `X.center = (..)` for an *arc* X is silently dropped (executor.rs:273-283).
  * no whitespace is accepted after the last argument of tuple-style calls or at line ends.
The product has its own C++ parser (ezpz_amd/csrc); tests cross-check the two.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import oracle as O


class ParseError(ValueError):
    pass


class TextualError(ValueError):
    """error.rs:10-33"""

    def __init__(self, kind: str, label):
        super().__init__(f"{kind}: {label}")
        self.kind = kind
        self.label = label


_FLOAT_RE = re.compile(
    r"[+-]?(?:(?:\d+(?:\.\d*)?|\.\d+)(?:[eE][+-]?\d+)?|[iI][nN][fF](?:[iI][nN][iI][tT][yY])?|[nN][aA][nN])"
)
_LABEL_RE = re.compile(r"[A-Za-z0-9]+")
_WS_RE = re.compile(r"[ \t]*")


class _Cur:
    def __init__(self, s: str):
        self.s = s
        self.i = 0

    def rest(self) -> str:
        return self.s[self.i:]

    def ws(self):
        self.i = _WS_RE.match(self.s, self.i).end()

    def lit(self, t: str):
        if not self.s.startswith(t, self.i):
            raise ParseError(f"expected {t!r} at {self.i}: {self.s[self.i:self.i + 30]!r}")
        self.i += len(t)

    def try_lit(self, t: str) -> bool:
        if self.s.startswith(t, self.i):
            self.i += len(t)
            return True
        return False

    def label(self) -> str:
        m = _LABEL_RE.match(self.s, self.i)
        if not m:
            raise ParseError(f"expected label at {self.i}: {self.s[self.i:self.i + 30]!r}")
        self.i = m.end()
        return m.group(0)

    def label_opt_suffix(self) -> str:
        lab = self.label()
        save = self.i
        if self.try_lit("."):
            try:
                lab = lab + "." + self.label()
            except ParseError:
                self.i = save
        return lab

    def number(self) -> float:
        m = _FLOAT_RE.match(self.s, self.i)
        if not m:
            raise ParseError(f"expected number at {self.i}: {self.s[self.i:self.i + 30]!r}")
        self.i = m.end()
        return float(m.group(0))

    def number_expr(self) -> float:  # parser.rs:549-555
        save = self.i
        try:
            return self.number()
        except ParseError:
            self.i = save
        self.lit("sqrt(")
        v = self.number_expr()
        self.lit(")")
        return math.sqrt(v)

    def commasep(self):  # parser.rs:223-228
        self.ws()
        self.lit(",")
        self.ws()

    def point(self) -> Tuple[float, float]:  # parser.rs:511-516
        self.lit("(")
        self.ws()
        x = self.number()
        self.lit(",")
        self.ws()
        y = self.number()
        self.lit(")")
        return (x, y)

    def labels(self, k: int) -> List[str]:  # two_points / three_points / four_points
        out = [self.label()]
        for _ in range(k - 1):
            self.commasep()
            out.append(self.label())
        self.ws()
        return out

    def open(self):  # inside_brackets, parser.rs:330-339
        self.lit("(")
        self.ws()

    def angle(self):  # parser.rs:243-251
        v = self.number()
        if self.try_lit("deg"):
            return ("deg", v)
        self.lit("rad")
        return ("rad", v)


def _instr_declare(kw):
    def f(c: _Cur):
        c.lit(kw)
        c.ws()
        return [(kw, c.label())]

    return f


def _fix_point_component(c: _Cur):  # parser.rs:477-493
    lab = c.label()
    c.lit(".")
    if c.try_lit("x"):
        comp = "x"
    else:
        c.lit("y")
        comp = "y"
    c.ws()
    c.lit("=")
    c.ws()
    return [("fix", lab, comp, c.number())]


def _fix_center_component(c: _Cur):  # parser.rs:518-534
    lab = c.label()
    c.lit(".center.")
    if c.try_lit("x"):
        comp = "x"
    else:
        c.lit("y")
        comp = "y"
    c.ws()
    c.lit("=")
    c.ws()
    return [("fixcenter", lab, comp, c.number())]


def _assign_point(c: _Cur):  # parser.rs:452-471
    lab = c.label_opt_suffix()
    c.ws()
    c.lit("=")
    c.ws()
    x, y = c.point()
    return [("fix", lab, "x", x), ("fix", lab, "y", y)]


def _call_labels(name, k, tag):
    def f(c: _Cur):
        c.lit(name)
        c.ws()
        c.open()
        labs = c.labels(k)
        c.lit(")")
        return [(tag, *labs)]

    return f


def _distance(c: _Cur):  # parser.rs:213-221
    c.lit("distance")
    c.ws()
    c.open()
    p0, p1 = c.labels(2)
    c.commasep()
    d = c.number_expr()
    c.lit(")")
    return [("distance", p0, p1, d)]


def _angle_line(c: _Cur):  # parser.rs:230-241
    c.lit("lines_at_angle")
    c.ws()
    c.open()
    labs = c.labels(4)
    c.commasep()
    a = c.angle()
    c.lit(")")
    return [("lines_at_angle", *labs, a)]


def _label_num(name, tag, expr):
    def f(c: _Cur):
        c.lit(name)
        c.ws()
        c.open()
        lab = c.label()
        c.commasep()
        v = c.number_expr() if expr else c.number()
        c.lit(")")
        return [(tag, lab, v)]

    return f


def _tangent(c: _Cur):  # parser.rs:269-281
    c.lit("tangent")
    c.ws()
    c.open()
    p0 = c.label()
    c.commasep()
    p1 = c.label()
    c.commasep()
    circ = c.label()
    c.lit(")")
    return [("tangent", p0, p1, circ)]


def _is_arc(c: _Cur):
    c.lit("is_arc")
    c.ws()
    c.open()
    lab = c.label()
    c.lit(")")
    return [("is_arc", lab)]


def _point_line_distance(c: _Cur):  # parser.rs:183-193, :371-381
    c.lit("point_line_distance")
    c.ws()
    c.open()
    p = c.label()
    c.commasep()
    l0 = c.label()
    c.commasep()
    l1 = c.label()
    c.commasep()
    d = c.number()
    c.ws()
    c.lit(")")
    return [("point_line_distance", p, l0, l1, d)]


def _line(c: _Cur):  # parser.rs:304-309
    c.lit("line")
    c.ws()
    c.open()
    p0 = c.label()
    c.commasep()
    p1 = c.label()
    c.lit(")")
    return [("line", p0, p1)]


# `alt` order of parser.rs:388-442
_ALTS = [
    _instr_declare("point"),
    _instr_declare("circle"),
    _instr_declare("arc"),
    _fix_point_component,
    _fix_center_component,
    _assign_point,
    _call_labels("horizontal", 2, "horizontal"),
    _call_labels("coincident", 2, "coincident"),
    _call_labels("point_arc_coincident", 2, "point_arc_coincident"),
    _call_labels("midpoint", 3, "midpoint"),
    _call_labels("symmetric", 4, "symmetric"),
    _call_labels("vertical", 2, "vertical"),
    _distance,
    _call_labels("parallel", 4, "parallel"),
    _call_labels("perpendicular", 4, "perpendicular"),
    _angle_line,
    _label_num("radius", "radius", True),
    _tangent,
    _label_num("arc_radius", "arc_radius", False),
    _label_num("arc_length", "arc_length", False),
    _is_arc,
    _point_line_distance,
    _line,
    _call_labels("lines_equal_length", 4, "lines_equal_length"),
]


def _parse_instruction(c: _Cur):
    c.ws()
    start = c.i
    for alt in _ALTS:
        c.i = start
        try:
            return alt(c)
        except ParseError:
            continue
    c.i = start
    raise ParseError(f"no instruction matches at {start}: {c.s[start:start + 40]!r}")


def _parse_guess(c: _Cur):  # parser.rs:84-128
    c.ws()
    lab = c.label_opt_suffix()
    c.ws()
    c.lit("roughly")
    c.ws()
    save = c.i
    try:
        return ("point", lab, c.point())
    except ParseError:
        c.i = save
    return ("scalar", lab, c.number())


@dataclass
class Problem:
    """textual.rs:30-41"""

    instructions: list
    inner_points: List[str]
    inner_circles: List[str]
    inner_arcs: List[str]
    inner_lines: List[Tuple[str, str]]
    point_guesses: List[Tuple[str, Tuple[float, float]]]
    scalar_guesses: List[Tuple[str, float]]


def parse_problem(text: str) -> Problem:  # parser.rs:29-76
    c = _Cur(text)
    c.lit("#")
    c.ws()
    c.lit("constraints")
    c.lit("\n")
    instructions = []
    instructions.extend(_parse_instruction(c))
    while True:  # separated(1.., parse_instruction, newline)
        save = c.i
        if not c.try_lit("\n"):
            break
        try:
            instructions.extend(_parse_instruction(c))
        except ParseError:
            c.i = save
            break
    c.lit("\n")
    c.lit("\n")
    c.ws()
    c.lit("#")
    c.ws()
    c.lit("guesses")
    c.lit("\n")
    guesses = [_parse_guess(c)]
    while True:
        save = c.i
        if not c.try_lit("\n"):
            break
        try:
            guesses.append(_parse_guess(c))
        except ParseError:
            c.i = save
            break
    c.try_lit("\n")
    c.ws()
    if c.i != len(c.s):
        raise ParseError(f"trailing input at {c.i}: {c.s[c.i:c.i + 40]!r}")
    return Problem(
        instructions=instructions,
        inner_points=[i[1] for i in instructions if i[0] == "point"],
        inner_circles=[i[1] for i in instructions if i[0] == "circle"],
        inner_arcs=[i[1] for i in instructions if i[0] == "arc"],
        inner_lines=[(i[1], i[2]) for i in instructions if i[0] == "line"],
        point_guesses=[(g[1], g[2]) for g in guesses if g[0] == "point"],
        scalar_guesses=[(g[1], g[2]) for g in guesses if g[0] == "scalar"],
    )


@dataclass
class ConstraintSystem:
    """executor.rs:448-459.  constraints: structured array (oracle.CONSTRAINT_DTYPE); guesses: values by id."""

    constraints: np.ndarray
    guesses: np.ndarray
    inner_points: List[str]
    inner_circles: List[str]
    inner_arcs: List[str]
    inner_lines: List[Tuple[str, str]] = field(default_factory=list)

    @property
    def num_vars(self) -> int:
        return len(self.guesses)

    def variables(self) -> List[Tuple[int, float]]:
        return [(i, float(v)) for i, v in enumerate(self.guesses)]

    # label -> value helpers following executor.rs:521-566
    def point(self, values, label: str) -> Tuple[float, float]:
        i = self.inner_points.index(label)
        return (values[2 * i], values[2 * i + 1])

    def circle(self, values, label: str):
        i = self.inner_circles.index(label)
        s = 2 * len(self.inner_points) + 3 * i
        return {"center": (values[s], values[s + 1]), "radius": values[s + 2]}

    def arc(self, values, label: str):
        i = self.inner_arcs.index(label)
        s = 2 * len(self.inner_points) + 3 * len(self.inner_circles) + 6 * i
        return {"a": (values[s], values[s + 1]), "b": (values[s + 2], values[s + 3]),
                "center": (values[s + 4], values[s + 5])}


def to_constraint_system(p: Problem) -> ConstraintSystem:  # executor.rs:40-445
    variables: List[float] = []
    guessmap_points = {}
    for lab, g in p.point_guesses:
        guessmap_points[lab] = g
    for lab in p.inner_points:
        if lab not in guessmap_points:
            raise TextualError("MissingGuess", lab)
        g = guessmap_points.pop(lab)
        variables.extend([g[0], g[1]])
    guessmap_scalars = {}
    for lab, g in p.scalar_guesses:
        guessmap_scalars[lab] = g
    for circ in p.inner_circles:
        cl = f"{circ}.center"
        if cl not in guessmap_points:
            raise TextualError("MissingGuess", cl)
        cg = guessmap_points.pop(cl)
        rl = f"{circ}.radius"
        if rl not in guessmap_scalars:
            raise TextualError("MissingGuess", rl)
        rg = guessmap_scalars.pop(rl)
        variables.extend([cg[0], cg[1], rg])
    for arc in p.inner_arcs:
        cl, al, bl = f"{arc}.center", f"{arc}.a", f"{arc}.b"
        if cl not in guessmap_points:
            raise TextualError("MissingGuess", cl)
        cg = guessmap_points.pop(cl)
        if al not in guessmap_points:
            raise TextualError("MissingGuess", al)
        ag = guessmap_points.pop(al)
        if bl not in guessmap_points:
            raise TextualError("MissingGuess", bl)
        bg = guessmap_points.pop(bl)
        variables.extend([ag[0], ag[1], bg[0], bg[1], cg[0], cg[1]])
    if guessmap_points:
        raise TextualError("UnusedGuesses", sorted(guessmap_points))
    if guessmap_scalars:
        raise TextualError("UnusedGuesses", sorted(guessmap_scalars))

    n_points = len(p.inner_points)
    # first position of each label (list.index semantics) in O(1)
    pos_points, pos_circles, pos_arcs = {}, {}, {}
    for table, labels in ((pos_points, p.inner_points), (pos_circles, p.inner_circles), (pos_arcs, p.inner_arcs)):
        for i, lab in enumerate(labels):
            table.setdefault(lab, i)

    def point_ids(i):
        return (2 * i, 2 * i + 1)

    def circle_ids(i):
        s = 2 * n_points + 3 * i
        return (s, s + 1), s + 2

    def arc_ids(i):  # geometry_variables.rs:91-104 (circles not counted)
        s = 2 * n_points + 6 * i
        return {"start": (s, s + 1), "end": (s + 2, s + 3), "center": (s + 4, s + 5)}

    def datum_point(label: str):  # executor.rs:121-174
        if label in pos_points:
            return point_ids(pos_points[label])
        if label.endswith(".center"):
            base = label[: -len(".center")]
            if base in pos_circles:
                return circle_ids(pos_circles[base])[0]
            if base in pos_arcs:
                return arc_ids(pos_arcs[base])["center"]
        if label.endswith(".a") and label[:-2] in pos_arcs:
            return arc_ids(pos_arcs[label[:-2]])["start"]
        if label.endswith(".b") and label[:-2] in pos_arcs:
            return arc_ids(pos_arcs[label[:-2]])["end"]
        raise TextualError("UndefinedPoint", label)

    def datum_distance(label: str):  # executor.rs:175-187
        if label.endswith(".radius") and label[: -len(".radius")] in pos_circles:
            return circle_ids(pos_circles[label[: -len(".radius")]])[1]
        raise TextualError("UndefinedPoint", label)

    def datum_arc(label: str):
        return (datum_point(f"{label}.center"), datum_point(f"{label}.a"), datum_point(f"{label}.b"))

    cs = []
    for ins in p.instructions:
        k = ins[0]
        if k in ("point", "circle", "arc", "line"):
            continue
        if k == "radius":
            cs.append(O.circle_radius(datum_point(f"{ins[1]}.center"), datum_distance(f"{ins[1]}.radius"), ins[2]))
        elif k == "arc_radius":
            cs.append(O.arc_radius(*datum_arc(ins[1]), ins[2]))
        elif k == "is_arc":
            cs.append(O.arc(*datum_arc(ins[1])))
        elif k == "point_line_distance":
            l0, l1 = datum_point(ins[2]), datum_point(ins[3])
            pt = datum_point(ins[1])
            cs.append(O.point_line_distance(pt, l0, l1, ins[4]))
        elif k == "tangent":
            center = datum_point(f"{ins[3]}.center")
            rad = datum_distance(f"{ins[3]}.radius")
            l0, l1 = datum_point(ins[1]), datum_point(ins[2])
            cs.append(O.line_tangent_to_circle(l0, l1, center, rad, O.SIDE_UNDEFINED))
        elif k == "fix":  # executor.rs:259-289
            _, label, comp, value = ins
            sel = 0 if comp == "x" else 1
            if label in pos_points:
                cs.append(O.fixed(point_ids(pos_points[label])[sel], value))
            elif label.endswith(".center"):
                circ = label[: -len(".center")]
                if circ in pos_circles:
                    cs.append(O.fixed(circle_ids(pos_circles[circ])[0][sel], value))
            else:
                raise TextualError("UndefinedPoint", label)
        elif k == "fixcenter":  # executor.rs:290-320
            _, obj, comp, value = ins
            sel = 0 if comp == "x" else 1
            if obj in pos_circles:
                cs.append(O.fixed(circle_ids(pos_circles[obj])[0][sel], value))
            elif obj in pos_arcs:
                cs.append(O.fixed(arc_ids(pos_arcs[obj])["center"][sel], value))
            else:
                raise TextualError("UndefinedPoint", obj)
        elif k == "vertical":
            cs.append(O.vertical(datum_point(ins[1]), datum_point(ins[2])))
        elif k == "horizontal":
            cs.append(O.horizontal(datum_point(ins[1]), datum_point(ins[2])))
        elif k == "coincident":
            cs.append(O.points_coincident(datum_point(ins[1]), datum_point(ins[2])))
        elif k == "point_arc_coincident":  # [point, arc]
            pt = datum_point(ins[1])
            cs.append(O.point_arc_coincident(*datum_arc(ins[2]), pt))
        elif k == "midpoint":
            cs.append(O.midpoint(datum_point(ins[1]), datum_point(ins[2]), datum_point(ins[3])))
        elif k == "symmetric":  # [line_p, line_q, a, b]; executor resolves a, b first, then the line
            a, b = datum_point(ins[3]), datum_point(ins[4])
            lp, lq = datum_point(ins[1]), datum_point(ins[2])
            cs.append(O.symmetric(lp, lq, a, b))
        elif k == "distance":
            cs.append(O.distance(datum_point(ins[1]), datum_point(ins[2]), ins[3]))
        elif k in ("parallel", "perpendicular"):
            pts = [datum_point(l) for l in ins[1:5]]
            cs.append(O.lines_at_angle(*pts, k))
        elif k == "lines_equal_length":
            pts = [datum_point(l) for l in ins[1:5]]
            cs.append(O.lines_equal_length(*pts))
        elif k == "lines_at_angle":
            pts = [datum_point(l) for l in ins[1:5]]
            cs.append(O.lines_at_angle(*pts, ins[5]))
        elif k == "arc_length":
            cs.append(O.arc_length(*datum_arc(ins[1]), ins[2]))
        else:  # pragma: no cover
            raise AssertionError(k)
    return ConstraintSystem(
        constraints=O.stack(cs),
        guesses=np.asarray(variables, dtype=np.float64),
        inner_points=list(p.inner_points),
        inner_circles=list(p.inner_circles),
        inner_arcs=list(p.inner_arcs),
        inner_lines=list(p.inner_lines),
    )


def load(text: str) -> ConstraintSystem:
    return to_constraint_system(parse_problem(text))


def gen_big_problem(total_lines: int, overconstrain: bool = False) -> str:
    """Same text as test_cases/massive_parallel_system/gen_big_problem.py:16-35 prints."""
    out = ["# constraints"]
    for line in range(total_lines):
        a, b = line * 2, line * 2 + 1
        out += [f"point p{a}", f"point p{b}", f"vertical(p{a}, p{b})", f"p{a}.x={line}", f"p{a}.y=0", f"p{b}.y=4"]
        if overconstrain:
            out.append(f"distance(p{a}, p{b}, 4)")
    out.append("")
    out.append("# guesses")
    for line in range(total_lines):
        a, b = line * 2, line * 2 + 1
        out += [f"p{a} roughly ({a},{a})", f"p{b} roughly ({b},{b})"]
    return "\n".join(out) + "\n"
