// The ezpz `.md` problem-file front end, host side (C++17).
//
// Mirrors, for the one purpose of feeding the solver with the reference's own fixture / CLI format:
//   grammar            reference ezpz/src/textual/parser.rs:29-555  (combinators tried in `alt` order)
//   lowering           reference ezpz/src/textual/executor.rs:40-445
//   variable layout    reference ezpz/src/textual/geometry_variables.rs:56-177
// Reference behaviours kept on purpose: arcs are offset by 2*num_points only (geometry_variables.rs:92),
// `X.center = (..)` for an arc X is dropped silently (executor.rs:273-283), no whitespace is accepted
// after the last argument of tuple-style calls or at line ends, exactly one blank line before `# guesses`.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/ezpz_amd.h"

namespace {

struct ParseFail {
    size_t at;
    std::string what;
};

struct Instr {
    std::string op;
    std::vector<std::string> labels;
    double value = 0.0;
    char comp = 0;      // 'x' / 'y' for fix / fixcenter
    uint8_t angle_tag = 0;
};

struct Cur {
    const std::string& s;
    size_t i = 0;
    explicit Cur(const std::string& str) : s(str) {}
    [[noreturn]] void fail(const std::string& what) const { throw ParseFail{i, what}; }
    void ws() {
        while (i < s.size() && (s[i] == ' ' || s[i] == '\t')) ++i;
    }
    bool try_lit(const char* t) {
        size_t n = std::strlen(t);
        if (s.compare(i, n, t) == 0) {
            i += n;
            return true;
        }
        return false;
    }
    void lit(const char* t) {
        if (!try_lit(t)) fail(std::string("expected '") + t + "'");
    }
    static bool alnum(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
    std::string label() {  // parser.rs:495-499
        size_t b = i;
        while (i < s.size() && alnum(s[i])) ++i;
        if (i == b) fail("expected label");
        return s.substr(b, i - b);
    }
    std::string label_opt_suffix() {  // parser.rs:501-509
        std::string lab = label();
        size_t save = i;
        if (try_lit(".")) {
            size_t b = i;
            while (i < s.size() && alnum(s[i])) ++i;
            if (i == b)
                i = save;
            else
                lab += "." + s.substr(b, i - b);
        }
        return lab;
    }
    // winnow::ascii::float: [+-]? (digits [. digits?] | . digits) ([eE] [+-]? digits)? | inf | infinity | nan
    double number() {
        size_t b = i, j = i;
        if (j < s.size() && (s[j] == '+' || s[j] == '-')) ++j;
        auto ci = [&](size_t at, const char* w) {
            size_t n = std::strlen(w);
            if (at + n > s.size()) return false;
            for (size_t k = 0; k < n; ++k)
                if (std::tolower((unsigned char)s[at + k]) != w[k]) return false;
            return true;
        };
        if (ci(j, "infinity")) {
            j += 8;
        } else if (ci(j, "inf")) {
            j += 3;
        } else if (ci(j, "nan")) {
            j += 3;
        } else {
            size_t d0 = j;
            while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
            bool had_int = j > d0;
            if (had_int) {
                if (j < s.size() && s[j] == '.') {
                    ++j;
                    while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
                }
            } else {
                if (j < s.size() && s[j] == '.') {
                    size_t f0 = ++j;
                    while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
                    if (j == f0) fail("expected number");
                } else {
                    fail("expected number");
                }
            }
            if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
                size_t k = j + 1;
                if (k < s.size() && (s[k] == '+' || s[k] == '-')) ++k;
                size_t e0 = k;
                while (k < s.size() && s[k] >= '0' && s[k] <= '9') ++k;
                if (k > e0) j = k;
            }
        }
        std::string tok = s.substr(b, j - b);
        i = j;
        return std::strtod(tok.c_str(), nullptr);
    }
    double number_expr() {  // parser.rs:549-555
        size_t save = i;
        try {
            return number();
        } catch (const ParseFail&) {
            i = save;
        }
        lit("sqrt(");
        double v = number_expr();
        lit(")");
        return std::sqrt(v);
    }
    void commasep() {  // parser.rs:223-228
        ws();
        lit(",");
        ws();
    }
    void point(double& x, double& y) {  // parser.rs:511-516
        lit("(");
        ws();
        x = number();
        lit(",");
        ws();
        y = number();
        lit(")");
    }
    void labels(int k, std::vector<std::string>& out) {  // two_points / three_points / four_points
        out.push_back(label());
        for (int t = 1; t < k; ++t) {
            commasep();
            out.push_back(label());
        }
        ws();
    }
    void open() {  // inside_brackets, parser.rs:330-339
        lit("(");
        ws();
    }
};

using InstrList = std::vector<Instr>;
typedef void (*AltFn)(Cur&, InstrList&);

void alt_declare(Cur& c, InstrList& out, const char* kw) {
    c.lit(kw);
    c.ws();
    Instr in;
    in.op = kw;
    in.labels.push_back(c.label());
    out.push_back(in);
}
char component(Cur& c) {
    if (c.try_lit("x")) return 'x';
    c.lit("y");
    return 'y';
}
void alt_fix_component(Cur& c, InstrList& out) {  // parser.rs:477-493
    Instr in;
    in.op = "fix";
    in.labels.push_back(c.label());
    c.lit(".");
    in.comp = component(c);
    c.ws();
    c.lit("=");
    c.ws();
    in.value = c.number();
    out.push_back(in);
}
void alt_fix_center(Cur& c, InstrList& out) {  // parser.rs:518-534
    Instr in;
    in.op = "fixcenter";
    in.labels.push_back(c.label());
    c.lit(".center.");
    in.comp = component(c);
    c.ws();
    c.lit("=");
    c.ws();
    in.value = c.number();
    out.push_back(in);
}
void alt_assign_point(Cur& c, InstrList& out) {  // parser.rs:452-471
    std::string lab = c.label_opt_suffix();
    c.ws();
    c.lit("=");
    c.ws();
    double x, y;
    c.point(x, y);
    Instr a;
    a.op = "fix";
    a.labels.push_back(lab);
    a.comp = 'x';
    a.value = x;
    Instr b = a;
    b.comp = 'y';
    b.value = y;
    out.push_back(a);
    out.push_back(b);
}
void call_labels(Cur& c, InstrList& out, const char* name, int k) {
    c.lit(name);
    c.ws();
    c.open();
    Instr in;
    in.op = name;
    c.labels(k, in.labels);
    c.lit(")");
    out.push_back(in);
}
void alt_distance(Cur& c, InstrList& out) {  // parser.rs:213-221
    c.lit("distance");
    c.ws();
    c.open();
    Instr in;
    in.op = "distance";
    c.labels(2, in.labels);
    c.commasep();
    in.value = c.number_expr();
    c.lit(")");
    out.push_back(in);
}
void alt_angle_line(Cur& c, InstrList& out) {  // parser.rs:230-251
    c.lit("lines_at_angle");
    c.ws();
    c.open();
    Instr in;
    in.op = "lines_at_angle";
    c.labels(4, in.labels);
    c.commasep();
    in.value = c.number();
    if (c.try_lit("deg"))
        in.angle_tag = EZPZ_ANGLE_OTHER_DEG;
    else {
        c.lit("rad");
        in.angle_tag = EZPZ_ANGLE_OTHER_RAD;
    }
    c.lit(")");
    out.push_back(in);
}
void label_num(Cur& c, InstrList& out, const char* name, bool expr) {
    c.lit(name);
    c.ws();
    c.open();
    Instr in;
    in.op = name;
    in.labels.push_back(c.label());
    c.commasep();
    in.value = expr ? c.number_expr() : c.number();
    c.lit(")");
    out.push_back(in);
}
void alt_tangent(Cur& c, InstrList& out) {  // parser.rs:269-281
    c.lit("tangent");
    c.ws();
    c.open();
    Instr in;
    in.op = "tangent";
    in.labels.push_back(c.label());
    c.commasep();
    in.labels.push_back(c.label());
    c.commasep();
    in.labels.push_back(c.label());
    c.lit(")");
    out.push_back(in);
}
void alt_is_arc(Cur& c, InstrList& out) {
    c.lit("is_arc");
    c.ws();
    c.open();
    Instr in;
    in.op = "is_arc";
    in.labels.push_back(c.label());
    c.lit(")");
    out.push_back(in);
}
void alt_point_line_distance(Cur& c, InstrList& out) {  // parser.rs:183-193, :371-381
    c.lit("point_line_distance");
    c.ws();
    c.open();
    Instr in;
    in.op = "point_line_distance";
    in.labels.push_back(c.label());
    c.commasep();
    in.labels.push_back(c.label());
    c.commasep();
    in.labels.push_back(c.label());
    c.commasep();
    in.value = c.number();
    c.ws();
    c.lit(")");
    out.push_back(in);
}
void alt_line(Cur& c, InstrList& out) {  // parser.rs:304-309
    c.lit("line");
    c.ws();
    c.open();
    Instr in;
    in.op = "line";
    in.labels.push_back(c.label());
    c.commasep();
    in.labels.push_back(c.label());
    c.lit(")");
    out.push_back(in);
}

// `alt` order of parser.rs:388-442
const AltFn kAlts[] = {
    [](Cur& c, InstrList& o) { alt_declare(c, o, "point"); },
    [](Cur& c, InstrList& o) { alt_declare(c, o, "circle"); },
    [](Cur& c, InstrList& o) { alt_declare(c, o, "arc"); },
    alt_fix_component,
    alt_fix_center,
    alt_assign_point,
    [](Cur& c, InstrList& o) { call_labels(c, o, "horizontal", 2); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "coincident", 2); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "point_arc_coincident", 2); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "midpoint", 3); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "symmetric", 4); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "vertical", 2); },
    alt_distance,
    [](Cur& c, InstrList& o) { call_labels(c, o, "parallel", 4); },
    [](Cur& c, InstrList& o) { call_labels(c, o, "perpendicular", 4); },
    alt_angle_line,
    [](Cur& c, InstrList& o) { label_num(c, o, "radius", true); },
    alt_tangent,
    [](Cur& c, InstrList& o) { label_num(c, o, "arc_radius", false); },
    [](Cur& c, InstrList& o) { label_num(c, o, "arc_length", false); },
    alt_is_arc,
    alt_point_line_distance,
    alt_line,
    [](Cur& c, InstrList& o) { call_labels(c, o, "lines_equal_length", 4); },
};

void parse_instruction(Cur& c, InstrList& out) {
    c.ws();
    const size_t start = c.i;
    for (AltFn alt : kAlts) {
        c.i = start;
        InstrList tmp;
        try {
            alt(c, tmp);
        } catch (const ParseFail&) {
            continue;
        }
        out.insert(out.end(), tmp.begin(), tmp.end());
        return;
    }
    c.i = start;
    c.fail("no instruction matches");
}

struct Guess {
    bool is_point;
    std::string label;
    double x, y;
};
Guess parse_guess(Cur& c) {  // parser.rs:84-128
    c.ws();
    Guess g{};
    g.label = c.label_opt_suffix();
    c.ws();
    c.lit("roughly");
    c.ws();
    size_t save = c.i;
    try {
        c.point(g.x, g.y);
        g.is_point = true;
        return g;
    } catch (const ParseFail&) {
        c.i = save;
    }
    g.x = c.number();
    g.is_point = false;
    return g;
}

struct Pt {
    uint32_t x, y;
};

}  // namespace

struct EzpzProblem {
    std::vector<EzpzConstraint> constraints;
    std::vector<double> guesses;
    std::vector<std::string> labels[3];  // points, circles, arcs
};

namespace {

struct TextErr {
    int code;
    std::string label;
};

EzpzConstraint mk(uint16_t kind, std::initializer_list<uint32_t> ids, double param = 0.0, uint8_t tag = 0) {
    EzpzConstraint c;
    std::memset(&c, 0, sizeof(c));
    c.kind = kind;
    c.tag = tag;
    c.priority = 0;  // executor.rs:429-435
    c.weight = 1.0;
    c.param = param;
    int k = 0;
    for (uint32_t v : ids) c.ids[k++] = v;
    return c;
}

void lower(const InstrList& instrs, const std::vector<Guess>& guesses, EzpzProblem& P) {
    std::vector<std::string>&pts = P.labels[0], &circles = P.labels[1], &arcs = P.labels[2];
    for (const Instr& in : instrs) {
        if (in.op == "point") pts.push_back(in.labels[0]);
        if (in.op == "circle") circles.push_back(in.labels[0]);
        if (in.op == "arc") arcs.push_back(in.labels[0]);
    }
    // executor.rs:44-116: guesses -> variables (later guesses for one label overwrite earlier ones)
    std::map<std::string, std::pair<double, double>> gp;
    std::map<std::string, double> gs;
    for (const Guess& g : guesses) {
        if (g.is_point)
            gp[g.label] = {g.x, g.y};
        else
            gs[g.label] = g.x;
    }
    auto take_point = [&](const std::string& lab) {
        auto it = gp.find(lab);
        if (it == gp.end()) throw TextErr{EZPZ_ERR_TEXT_MISSING_GUESS, lab};
        auto v = it->second;
        gp.erase(it);
        return v;
    };
    for (const std::string& p : pts) {
        auto g = take_point(p);
        P.guesses.push_back(g.first);
        P.guesses.push_back(g.second);
    }
    for (const std::string& c : circles) {
        auto cg = take_point(c + ".center");
        auto it = gs.find(c + ".radius");
        if (it == gs.end()) throw TextErr{EZPZ_ERR_TEXT_MISSING_GUESS, c + ".radius"};
        double r = it->second;
        gs.erase(it);
        P.guesses.push_back(cg.first);
        P.guesses.push_back(cg.second);
        P.guesses.push_back(r);
    }
    for (const std::string& a : arcs) {
        auto cg = take_point(a + ".center");
        auto ag = take_point(a + ".a");
        auto bg = take_point(a + ".b");
        P.guesses.insert(P.guesses.end(), {ag.first, ag.second, bg.first, bg.second, cg.first, cg.second});
    }
    if (!gp.empty()) throw TextErr{EZPZ_ERR_TEXT_UNUSED_GUESSES, gp.begin()->first};
    if (!gs.empty()) throw TextErr{EZPZ_ERR_TEXT_UNUSED_GUESSES, gs.begin()->first};

    const uint32_t np = (uint32_t)pts.size();
    // first position of a label (`.position()` in the reference), through a hash map so that large generated
    // problems (gen_big_problem.py 50000 declares 100 000 points) lower in linear time
    std::unordered_map<std::string, int> first_pos[3];
    for (int t = 0; t < 3; ++t)
        for (size_t i = 0; i < P.labels[t].size(); ++i) first_pos[t].emplace(P.labels[t][i], (int)i);
    auto index_of = [&](const std::vector<std::string>& v, const std::string& s) -> int {
        const int t = (&v == &pts) ? 0 : ((&v == &circles) ? 1 : 2);
        auto it = first_pos[t].find(s);
        return it == first_pos[t].end() ? -1 : it->second;
    };
    auto point_ids = [&](int i) { return Pt{(uint32_t)(2 * i), (uint32_t)(2 * i + 1)}; };
    auto circle_center = [&](int i) { return Pt{2 * np + 3 * (uint32_t)i, 2 * np + 3 * (uint32_t)i + 1}; };
    auto circle_radius = [&](int i) { return 2 * np + 3 * (uint32_t)i + 2; };
    // geometry_variables.rs:91-104: arcs offset by the points only
    auto arc_start = [&](int i) { return Pt{2 * np + 6 * (uint32_t)i, 2 * np + 6 * (uint32_t)i + 1}; };
    auto arc_end = [&](int i) { return Pt{2 * np + 6 * (uint32_t)i + 2, 2 * np + 6 * (uint32_t)i + 3}; };
    auto arc_center = [&](int i) { return Pt{2 * np + 6 * (uint32_t)i + 4, 2 * np + 6 * (uint32_t)i + 5}; };
    auto ends_with = [](const std::string& s, const std::string& suf) {
        return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0;
    };
    auto datum_point = [&](const std::string& label) -> Pt {  // executor.rs:121-174
        int i = index_of(pts, label);
        if (i >= 0) return point_ids(i);
        if (ends_with(label, ".center")) {
            std::string base = label.substr(0, label.size() - 7);
            if ((i = index_of(circles, base)) >= 0) return circle_center(i);
            if ((i = index_of(arcs, base)) >= 0) return arc_center(i);
        }
        if (ends_with(label, ".a")) {
            if ((i = index_of(arcs, label.substr(0, label.size() - 2))) >= 0) return arc_start(i);
        }
        if (ends_with(label, ".b")) {
            if ((i = index_of(arcs, label.substr(0, label.size() - 2))) >= 0) return arc_end(i);
        }
        throw TextErr{EZPZ_ERR_TEXT_UNDEFINED_POINT, label};
    };
    auto datum_distance = [&](const std::string& label) -> uint32_t {  // executor.rs:175-187
        if (ends_with(label, ".radius")) {
            int i = index_of(circles, label.substr(0, label.size() - 7));
            if (i >= 0) return circle_radius(i);
        }
        throw TextErr{EZPZ_ERR_TEXT_UNDEFINED_POINT, label};
    };
    auto& cs = P.constraints;
    for (const Instr& in : instrs) {
        const std::string& op = in.op;
        const auto& L = in.labels;
        if (op == "point" || op == "circle" || op == "arc" || op == "line") continue;
        if (op == "radius") {
            Pt c = datum_point(L[0] + ".center");
            uint32_t r = datum_distance(L[0] + ".radius");
            cs.push_back(mk(EZPZ_CIRCLE_RADIUS, {c.x, c.y, r}, in.value));
        } else if (op == "arc_radius" || op == "is_arc" || op == "arc_length") {
            Pt c = datum_point(L[0] + ".center"), s = datum_point(L[0] + ".a"), e = datum_point(L[0] + ".b");
            uint16_t kind = op == "arc_radius" ? EZPZ_ARC_RADIUS : (op == "is_arc" ? EZPZ_ARC : EZPZ_ARC_LENGTH);
            cs.push_back(mk(kind, {c.x, c.y, s.x, s.y, e.x, e.y}, op == "is_arc" ? 0.0 : in.value));
        } else if (op == "point_line_distance") {
            Pt l0 = datum_point(L[1]), l1 = datum_point(L[2]);
            Pt p = datum_point(L[0]);
            cs.push_back(mk(EZPZ_POINT_LINE_DISTANCE, {p.x, p.y, l0.x, l0.y, l1.x, l1.y}, in.value));
        } else if (op == "tangent") {
            Pt c = datum_point(L[2] + ".center");
            uint32_t r = datum_distance(L[2] + ".radius");
            Pt l0 = datum_point(L[0]), l1 = datum_point(L[1]);
            cs.push_back(mk(EZPZ_LINE_TANGENT_TO_CIRCLE, {l0.x, l0.y, l1.x, l1.y, c.x, c.y, r}, 0.0, EZPZ_SIDE_UNDEFINED));
        } else if (op == "fix") {  // executor.rs:259-289
            int i = index_of(pts, L[0]);
            if (i >= 0) {
                Pt p = point_ids(i);
                cs.push_back(mk(EZPZ_FIXED, {in.comp == 'x' ? p.x : p.y}, in.value));
            } else if (ends_with(L[0], ".center")) {
                int ci = index_of(circles, L[0].substr(0, L[0].size() - 7));
                if (ci >= 0) {
                    Pt p = circle_center(ci);
                    cs.push_back(mk(EZPZ_FIXED, {in.comp == 'x' ? p.x : p.y}, in.value));
                }
            } else {
                throw TextErr{EZPZ_ERR_TEXT_UNDEFINED_POINT, L[0]};
            }
        } else if (op == "fixcenter") {  // executor.rs:290-320
            int ci = index_of(circles, L[0]);
            int ai = index_of(arcs, L[0]);
            if (ci >= 0) {
                Pt p = circle_center(ci);
                cs.push_back(mk(EZPZ_FIXED, {in.comp == 'x' ? p.x : p.y}, in.value));
            } else if (ai >= 0) {
                Pt p = arc_center(ai);
                cs.push_back(mk(EZPZ_FIXED, {in.comp == 'x' ? p.x : p.y}, in.value));
            } else {
                throw TextErr{EZPZ_ERR_TEXT_UNDEFINED_POINT, L[0]};
            }
        } else if (op == "vertical" || op == "horizontal" || op == "coincident") {
            Pt a = datum_point(L[0]), b = datum_point(L[1]);
            uint16_t kind = op == "vertical" ? EZPZ_VERTICAL : (op == "horizontal" ? EZPZ_HORIZONTAL : EZPZ_POINTS_COINCIDENT);
            cs.push_back(mk(kind, {a.x, a.y, b.x, b.y}));
        } else if (op == "point_arc_coincident") {  // [point, arc]
            Pt p = datum_point(L[0]);
            Pt c = datum_point(L[1] + ".center"), s = datum_point(L[1] + ".a"), e = datum_point(L[1] + ".b");
            cs.push_back(mk(EZPZ_POINT_ARC_COINCIDENT, {c.x, c.y, s.x, s.y, e.x, e.y, p.x, p.y}));
        } else if (op == "midpoint") {
            Pt a = datum_point(L[0]), b = datum_point(L[1]), m = datum_point(L[2]);
            cs.push_back(mk(EZPZ_MIDPOINT, {a.x, a.y, b.x, b.y, m.x, m.y}));
        } else if (op == "symmetric") {  // [line_p, line_q, a, b]
            Pt a = datum_point(L[2]), b = datum_point(L[3]);
            Pt lp = datum_point(L[0]), lq = datum_point(L[1]);
            cs.push_back(mk(EZPZ_SYMMETRIC, {lp.x, lp.y, lq.x, lq.y, a.x, a.y, b.x, b.y}));
        } else if (op == "distance") {
            Pt a = datum_point(L[0]), b = datum_point(L[1]);
            cs.push_back(mk(EZPZ_DISTANCE, {a.x, a.y, b.x, b.y}, in.value));
        } else if (op == "parallel" || op == "perpendicular" || op == "lines_at_angle" || op == "lines_equal_length") {
            Pt p0 = datum_point(L[0]), p1 = datum_point(L[1]), p2 = datum_point(L[2]), p3 = datum_point(L[3]);
            if (op == "lines_equal_length")
                cs.push_back(mk(EZPZ_LINES_EQUAL_LENGTH, {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y}));
            else {
                uint8_t tag = op == "parallel" ? EZPZ_ANGLE_PARALLEL : (op == "perpendicular" ? EZPZ_ANGLE_PERPENDICULAR : in.angle_tag);
                cs.push_back(mk(EZPZ_LINES_AT_ANGLE, {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y},
                                op == "lines_at_angle" ? in.value : 0.0, tag));
            }
        }
    }
}

}  // namespace

extern "C" {

int ezpz_problem_parse(const char* text, size_t len, EzpzProblem** out, char* errbuf, size_t errcap) {
    if (!out || !text) return EZPZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    auto report = [&](const std::string& msg) {
        if (errbuf && errcap) {
            std::snprintf(errbuf, errcap, "%s", msg.c_str());
        }
    };
    std::string s(text, len);
    InstrList instrs;
    std::vector<Guess> guesses;
    try {  // parser.rs:29-76
        Cur c(s);
        c.lit("#");
        c.ws();
        c.lit("constraints");
        c.lit("\n");
        parse_instruction(c, instrs);
        for (;;) {  // separated(1.., parse_instruction, newline)
            size_t save = c.i;
            if (!c.try_lit("\n")) break;
            try {
                parse_instruction(c, instrs);
            } catch (const ParseFail&) {
                c.i = save;
                break;
            }
        }
        c.lit("\n");
        c.lit("\n");
        c.ws();
        c.lit("#");
        c.ws();
        c.lit("guesses");
        c.lit("\n");
        guesses.push_back(parse_guess(c));
        for (;;) {
            size_t save = c.i;
            if (!c.try_lit("\n")) break;
            try {
                guesses.push_back(parse_guess(c));
            } catch (const ParseFail&) {
                c.i = save;
                break;
            }
        }
        c.try_lit("\n");
        c.ws();
        if (c.i != s.size()) c.fail("trailing input");
    } catch (const ParseFail& f) {
        report("parse error at offset " + std::to_string(f.at) + ": " + f.what);
        return EZPZ_ERR_PARSE;
    }
    EzpzProblem* P = new EzpzProblem();
    try {
        lower(instrs, guesses, *P);
    } catch (const TextErr& e) {
        report(std::string(ezpz_error_string(e.code)) + ": " + e.label);
        delete P;
        return e.code;
    }
    *out = P;
    return EZPZ_OK;
}

void ezpz_problem_destroy(EzpzProblem* p) { delete p; }
size_t ezpz_problem_num_constraints(const EzpzProblem* p) { return p ? p->constraints.size() : 0; }
size_t ezpz_problem_num_vars(const EzpzProblem* p) { return p ? p->guesses.size() : 0; }
const EzpzConstraint* ezpz_problem_constraints(const EzpzProblem* p) { return p ? p->constraints.data() : nullptr; }
const double* ezpz_problem_guesses(const EzpzProblem* p) { return p ? p->guesses.data() : nullptr; }
size_t ezpz_problem_num_labels(const EzpzProblem* p, int kind) {
    return (p && kind >= 0 && kind < 3) ? p->labels[kind].size() : 0;
}
const char* ezpz_problem_label(const EzpzProblem* p, int kind, size_t index) {
    if (!p || kind < 0 || kind > 2 || index >= p->labels[kind].size()) return nullptr;
    return p->labels[kind][index].c_str();
}

}  // extern "C"
