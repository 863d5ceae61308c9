// Component plan (see comp_program.hpp).  Plain C++17, no device code.
#include "comp_program.hpp"
#include "policy.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <string>

#include "kinds.hpp"
#include "program.hpp"

namespace ezpz {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
// What is worth a component plan, and what its 16-bit fields hold.
// (policy.hpp: the one table of launch-shape thresholds)
const EzpzLaunchPolicy kPolicy = launch_policy_for(256);
const size_t kMinInstances = kPolicy.comp_min_components;  // fewer components than two wavefronts' lanes: the other launch shapes serve
const size_t kMaxClasses = kPolicy.comp_max_classes;
const size_t kMaxClassVars = kPolicy.comp_max_component_vars, kMaxClassCons = kPolicy.comp_max_component_constraints;  // (the warning mask of a chunk is one 64-bit word per lane)

// (views into two arrays shared by all components: a request of 1500 components is analysed on the path of a solve() call)
struct Span {
    const uint32_t* b = nullptr;
    uint32_t n = 0;
    size_t size() const { return n; }
    const uint32_t* begin() const { return b; }
    const uint32_t* end() const { return b + n; }
    uint32_t operator[](size_t i) const { return b[i]; }
};
struct Component {
    Span verts;  // caller's variable ids, ascending
    Span cons;   // caller's constraint positions, ascending
};

struct ClassLayout {  // what every chunk of a class shares (copied into its CompChunk)
    uint32_t nv = 0, m = 0, zj = 0, ncons = 0, n_ops = 0, ops_off = 0, cons_off = 0;
    uint32_t rows_p = 0, o_d = 0, o_r0 = 0, o_r1 = 0, o_j = 0, o_wm = 0, s_l = 0;
    uint32_t ninst_pad = 0, ids_off = 0, par_off = 0, pos_off = 0;
};

struct Class {
    ClassLayout H;
    uint32_t rep = 0;                 // representative component
    std::vector<uint32_t> instances;  // components of this class, in order of their first variable
    Program Q;                        // symbolic phase of the representative, class-internal numbering
    bool linear = true;
    bool consts_f32 = true;
    std::vector<double> jconst;       // linear classes: the (constant) Jacobian value of every slot
    uint32_t cost = 1;
};

// Partial derivatives of the nine linear kinds in emission order (constraint_eval.hip.hpp con_jacobian; reference
// ezpz/src/constraints.rs:1252-1357,:1456-1512,:1599-1642).
int linear_partials(uint32_t kind, double pd[8]) {
    switch (kind) {
    case EZPZ_VERTICAL_DISTANCE:
    case EZPZ_HORIZONTAL_DISTANCE:
    case EZPZ_VERTICAL:
    case EZPZ_HORIZONTAL:
    case EZPZ_SCALAR_EQUAL:
        pd[0] = 1.0, pd[1] = -1.0;
        return 2;
    case EZPZ_FIXED:
    case EZPZ_CIRCLE_RADIUS:
        pd[0] = 1.0;
        return 1;
    case EZPZ_POINTS_COINCIDENT:
        pd[0] = 1.0, pd[1] = -1.0, pd[2] = 1.0, pd[3] = -1.0;
        return 4;
    case EZPZ_MIDPOINT:
        pd[0] = 1.0, pd[1] = -0.5, pd[2] = -0.5, pd[3] = 1.0, pd[4] = -0.5, pd[5] = -0.5;
        return 6;
    default:
        return 0;
    }
}

bool f32_exact(double v) { return (double)(float)v == v; }

uint32_t f32_bits(double v) {
    const float f = (float)v;
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

void push_double(std::vector<uint32_t>& w, double v) {
    uint64_t u;
    std::memcpy(&u, &v, 8);
    w.push_back((uint32_t)u);
    w.push_back((uint32_t)(u >> 32));
}

void align4(std::vector<uint32_t>& w) {
    while (w.size() % 4) w.push_back(0);
}

// One operation = one or more 8-word records.  `items` holds `words_per_item` words per item; `head` (0 or 2 words) goes
// into w2:w3 of the first record (the linear build's constants), shrinking that build's item room to w4..w7.
void emit_op(std::vector<uint32_t>& out, uint32_t opcode, uint32_t a, uint32_t b, const std::vector<uint32_t>& items,
             uint32_t words_per_item, uint32_t items_per_rec, uint32_t item_word0, const uint32_t* head,
             uint32_t first_flags = 0, uint32_t last_flags = 0) {
    const size_t n = items.size() / words_per_item;
    size_t done = 0;
    bool first = true;
    do {
        const size_t take = std::min<size_t>(items_per_rec, n - done);
        const bool last = done + take == n;
        uint32_t rec[kCompRecWords] = {0, 0, 0, 0, 0, 0, 0, 0};
        rec[0] = opcode | ((uint32_t)take << 8) | (first ? kCompFirst | first_flags : 0u) | (last ? kCompLast | last_flags : 0u);
        rec[1] = a | (b << 16);
        if (head) rec[2] = head[0], rec[3] = head[1];
        for (size_t k = 0; k < take * words_per_item; ++k) rec[item_word0 + k] = items[done * words_per_item + k];
        out.insert(out.end(), rec, rec + kCompRecWords);
        done += take;
        first = false;
    } while (done < n);
}

// The class program in the blob: the operation stream of the linear solve in execution order (+ one read-ahead pad
// record) and the constraint records (+ pad).  `params`: requests whose parameters go into the records as literals
// (w13:w14; a class that is a whole system shared by every lane) -- null: parameters come from per-instance tables.
// `fuse` (general records only): the assembly of an entry of JtJ sits right before the elimination of its column / slot
// and hands its value over in the accumulator (kCompKeep / kCompCont) instead of through the entry's row of state.
void emit_class_program(std::vector<uint32_t>& blob, Class& cl, bool LIN, const EzpzConstraint* params, bool request_order = false,
                        bool fuse = false) {
    const Program& Q = cl.Q;
    ClassLayout& H = cl.H;
    const uint32_t nv = Q.c.n_vars, zlo = Q.c.zlo, ncons = Q.c.n_cons;
    // ---- operation stream of the linear solve, in execution order ----
    std::vector<uint32_t> ops, items;
    fuse = fuse && !LIN;
    auto emit_diag = [&](uint32_t v) {
        items.clear();
        for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q)
            items.push_back(Q.colj_items[2 * q] | (Q.colj_items[2 * q + 1] << 16));
        emit_op(ops, COMP_DIAG, v, 0, items, 1, kCompItemsGen, 2, nullptr, 0, fuse ? kCompKeep : 0);
    };
    auto emit_off = [&](uint32_t s) {
        items.clear();
        for (uint32_t q = Q.apair_ptr[s]; q < Q.apair_ptr[s + 1]; ++q)
            items.push_back(Q.apairs[2 * q] | (Q.apairs[2 * q + 1] << 16));
        emit_op(ops, COMP_OFF, s, 0, items, 1, kCompItemsGen, 2, nullptr, 0, fuse ? kCompKeep : 0);
    };
    for (uint32_t v = 0; v < nv && !fuse; ++v) {
        items.clear();
        if (LIN) {
            double acc = 0.0;
            for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q) {
                const double jv = cl.jconst[Q.colj_items[2 * q]];
                acc += jv * jv;
                items.push_back(Q.colj_items[2 * q + 1]);
                items.push_back(f32_bits(jv));
            }
            std::vector<uint32_t> head;
            push_double(head, acc);
            emit_op(ops, COMP_DIAG, v, 0, items, 2, kCompItemsLin, 4, head.data());
        } else {
            emit_diag(v);
        }
    }
    for (uint32_t s = 0; s < zlo && !fuse; ++s) {
        items.clear();
        if (LIN) {
            double acc = 0.0;
            for (uint32_t q = Q.apair_ptr[s]; q < Q.apair_ptr[s + 1]; ++q)
                acc += cl.jconst[Q.apairs[2 * q]] * cl.jconst[Q.apairs[2 * q + 1]];
            std::vector<uint32_t> head;
            push_double(head, acc);
            emit_op(ops, COMP_OFF, s, 0, items, 1, kCompItemsGen, 4, head.data());
        } else {
            emit_off(s);
        }
    }
    const PartDesc part = Q.parts.empty() ? PartDesc{0, 0, 0, 0} : Q.parts[0];
    if (fuse) {
        // column by column in elimination order (the numbering is level-major, so every entry a column reads is done): its
        // diagonal, then the slots below it while d_j is still in a register
        // (an entry's assembly and its elimination share ONE record when their items fit it: COMP_DIAGCOL / COMP_SLOTA)
        auto emit_merged = [&](uint32_t opcode, uint32_t a, uint32_t n0) {
            uint32_t rec[kCompRecWords] = {0, 0, 0, 0, 0, 0, 0, 0};
            rec[0] = opcode | ((uint32_t)items.size() << 8) | kCompFirst | kCompLast | (n0 << 24);
            rec[1] = a;
            for (size_t k = 0; k < items.size(); ++k) rec[2 + k] = items[k];
            ops.insert(ops.end(), rec, rec + kCompRecWords);
        };
        for (uint32_t v = 0; v < nv; ++v) {
            const uint32_t nd = Q.colj_ptr[v + 1] - Q.colj_ptr[v], nc = Q.fwd_ptr[v + 1] - Q.fwd_ptr[v];
            if (nd + nc <= kCompItemsGen) {
                items.clear();
                for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q)
                    items.push_back(Q.colj_items[2 * q] | (Q.colj_items[2 * q + 1] << 16));
                for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q)
                    items.push_back(Q.fwd_items[2 * q] | (Q.fwd_items[2 * q + 1] << 16));
                emit_merged(COMP_DIAGCOL, v, nd);
            } else {
                emit_diag(v);
                items.clear();
                for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q)
                    items.push_back(Q.fwd_items[2 * q] | (Q.fwd_items[2 * q + 1] << 16));
                emit_op(ops, COMP_COL, v, 0, items, 1, kCompItemsGen, 2, nullptr, kCompCont, 0);
            }
            for (uint32_t qs = Q.bwd_ptr[v]; qs < Q.bwd_ptr[v + 1]; ++qs) {
                const uint32_t s = Q.bwd_items[2 * qs];
                const uint32_t na = Q.apair_ptr[s + 1] - Q.apair_ptr[s], nl = Q.lpair_ptr[s + 1] - Q.lpair_ptr[s];
                if (na + nl <= kCompItemsGen) {
                    items.clear();
                    for (uint32_t q = Q.apair_ptr[s]; q < Q.apair_ptr[s + 1]; ++q)
                        items.push_back(Q.apairs[2 * q] | (Q.apairs[2 * q + 1] << 16));
                    for (uint32_t q = Q.lpair_ptr[s]; q < Q.lpair_ptr[s + 1]; ++q)
                        items.push_back(Q.lpairs[2 * q] | (Q.lpairs[2 * q + 1] << 16));
                    emit_merged(COMP_SLOTA, s, na);
                    continue;
                }
                emit_off(s);
                items.clear();
                for (uint32_t q = Q.lpair_ptr[s]; q < Q.lpair_ptr[s + 1]; ++q)
                    items.push_back(Q.lpairs[2 * q] | (Q.lpairs[2 * q + 1] << 16));
                emit_op(ops, COMP_SLOT, s, Q.l_col[s], items, 1, kCompItemsGen, 2, nullptr, kCompCont, kCompDivReg);
            }
        }
    }
    for (uint32_t lv = 0; lv < part.nlev && !fuse; ++lv) {
        const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
        const uint32_t s0 = Q.lvl_sptr[part.lvl0 + lv], s1 = Q.lvl_sptr[part.lvl0 + lv + 1];
        for (uint32_t v = c0; v < c1; ++v) {
            if (fuse) emit_diag(v);
            items.clear();
            for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q)
                items.push_back(Q.fwd_items[2 * q] | (Q.fwd_items[2 * q + 1] << 16));
            emit_op(ops, COMP_COL, v, 0, items, 1, kCompItemsGen, 2, nullptr, fuse ? kCompCont : 0, 0);
        }
        for (uint32_t s = s0; s < s1; ++s) {
            if (fuse) emit_off(s);
            items.clear();
            for (uint32_t q = Q.lpair_ptr[s]; q < Q.lpair_ptr[s + 1]; ++q)
                items.push_back(Q.lpairs[2 * q] | (Q.lpairs[2 * q + 1] << 16));
            emit_op(ops, COMP_SLOT, s, Q.l_col[s], items, 1, kCompItemsGen, 2, nullptr, fuse ? kCompCont : 0, 0);
        }
    }
    for (uint32_t lv = part.nlev; lv-- > 0;) {
        const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
        for (uint32_t v = c0; v < c1; ++v) {
            items.clear();
            for (uint32_t q = Q.bwd_ptr[v]; q < Q.bwd_ptr[v + 1]; ++q)
                items.push_back(Q.bwd_items[2 * q] | (Q.bwd_items[2 * q + 1] << 16));
            emit_op(ops, COMP_BWD, v, 0, items, 1, kCompItemsGen, 2, nullptr);
        }
    }
    H.n_ops = (uint32_t)(ops.size() / kCompRecWords);

    // ---- lay the class out in the blob: operation stream (+ one pad record), constraint records, tables ----
    align4(blob);
    H.ops_off = (uint32_t)blob.size();
    blob.insert(blob.end(), ops.begin(), ops.end());
    blob.resize(blob.size() + kCompRecWords, 0);  // the interpreter reads one record ahead
    H.cons_off = (uint32_t)blob.size();
    std::vector<uint32_t> rec_order(ncons);
    std::iota(rec_order.begin(), rec_order.end(), 0u);
    if (request_order)  // (one lane per system: sums of squares formed like the reference's sequential sum over the rows)
        std::sort(rec_order.begin(), rec_order.end(), [&](uint32_t x, uint32_t y) { return Q.cons[x].pos < Q.cons[y].pos; });
    for (uint32_t ci : rec_order) {
        const DevCon& d = Q.cons[ci];
        uint32_t rec[kCompConWords] = {};
        rec[0] = d.kind | ((uint32_t)d.tag << 8) | ((uint32_t)d.nrows << 16) | ((uint32_t)d.nslots << 24);
        rec[1] = d.row0 | (d.jbase << 16);
        for (int e = 0; e < 4; ++e) rec[2 + e] = d.ids[2 * e] | (d.ids[2 * e + 1] << 16);
        std::memcpy(&rec[6], d.jloc, 16);
        std::memcpy(&rec[10], &d.weight, 8);
        rec[12] = ci;
        if (params) std::memcpy(&rec[13], &params[d.pos].param, 8);
        rec[15] = d.pos;
        blob.insert(blob.end(), rec, rec + kCompConWords);
    }
    blob.resize(blob.size() + kCompConWords, 0);  // read-ahead pad
}

// ---- source text of the class-specialised kernel (see jit_kernel.hip.hpp) ---------------------------------------------------------
// Every statement below is one operation of the class program, in the interpreter's order (comp_kernel.hip.hpp), with
// literal indices; doubles are written as hexadecimal floating literals (exact).
std::string hexf(double v) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%a", v);
    return std::string("(") + buf + ")";
}

// The elimination of one class as straight-line statements on the scalars D<v> (diagonal), V<v> (right-hand side, then y,
// then the step) and L<s> (strict lower entries), which the caller has assembled: level by level the left-looking Cholesky
// with the forward substitution riding along, then the backward substitution into d[] and dmax -- the same statements in
// the lane kernels' `solve` and the wavefront kernel's `tail`.  `fast`: divisions by a column's diagonal through its
// refined reciprocal (jit_kernel.hip.hpp: recip_of / div_by).
void emit_elimination(std::string& o, const Class& cl, bool fast, bool fastdiv) {
    const Program& Q = cl.Q;
    auto S = [](uint32_t v) { return std::to_string(v); };
    auto div = [&](const std::string& num, uint32_t col) {
        return fast && fastdiv ? "ezpz::jit::div_by(" + num + ", D" + S(col) + ", Y" + S(col) + ", ok)" : num + " / D" + S(col);
    };
    const PartDesc part = Q.parts.empty() ? PartDesc{0, 0, 0, 0} : Q.parts[0];
    for (uint32_t lv = 0; lv < part.nlev; ++lv) {
        const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
        const uint32_t s0 = Q.lvl_sptr[part.lvl0 + lv], s1 = Q.lvl_sptr[part.lvl0 + lv + 1];
        for (uint32_t v = c0; v < c1; ++v) {
            for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q) {
                const std::string l = "L" + S(Q.fwd_items[2 * q]);
                o += "        D" + S(v) + " -= " + l + " * " + l + "; V" + S(v) + " -= " + l + " * V" + S(Q.fwd_items[2 * q + 1]) + ";\n";
            }
            o += "        if (!(D" + S(v) + " > 0.0)) bad = true;\n";
            o += "        D" + S(v) + " = sqrt(D" + S(v) + ");";
            if (fast && fastdiv) o += " const double Y" + S(v) + " = ezpz::jit::recip_of(D" + S(v) + ", ok);";
            o += " V" + S(v) + " = " + div("V" + S(v), v) + ";\n";
        }
        for (uint32_t s2 = s0; s2 < s1; ++s2) {
            for (uint32_t q = Q.lpair_ptr[s2]; q < Q.lpair_ptr[s2 + 1]; ++q)
                o += "        L" + S(s2) + " -= L" + S(Q.lpairs[2 * q]) + " * L" + S(Q.lpairs[2 * q + 1]) + ";\n";
            o += "        L" + S(s2) + " = " + div("L" + S(s2), Q.l_col[s2]) + ";\n";
        }
    }
    for (uint32_t lv = part.nlev; lv-- > 0;) {
        const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
        for (uint32_t v = c0; v < c1; ++v) {
            for (uint32_t q = Q.bwd_ptr[v]; q < Q.bwd_ptr[v + 1]; ++q)
                o += "        V" + S(v) + " -= L" + S(Q.bwd_items[2 * q]) + " * V" + S(Q.bwd_items[2 * q + 1]) + ";\n";
            o += "        V" + S(v) + " = " + div("V" + S(v), v) + "; d[" + S(v) + "] = V" + S(v) + "; dmax = ezpz::dev::fmax_abs(dmax, V" + S(v) + ");\n";
        }
    }
}

// `lane`: the class is a whole system solved by one lane (jit_kernel.hip.hpp, lane_kernel): constraint parameters and
// the caller's constraint positions are literals, and load / store / pos_of map the caller's numbering.
// EZPZ_JIT_FASTDIV (measurements): bit 0 linear classes, bit 1 non-linear classes of the component kernels, bit 2 the
// lane-per-system kernels get the shared-reciprocal division (jit_kernel.hip.hpp: recip_of / div_by); the others divide plainly.
unsigned fastdiv_mask() {
    static const unsigned mask = [] {
        const char* e = std::getenv("EZPZ_JIT_FASTDIV");
        return e ? (unsigned)std::atoi(e) : 3u;  // (lane kernels: per-lane reciprocals cost `square` its second wavefront per SIMD, 1148 -> 1008 M solves/s)
    }();
    return mask;
}

void emit_class(std::string& o, size_t k, const Class& cl, bool lane = false, const EzpzConstraint* cs = nullptr) {
    const bool fastdiv = (fastdiv_mask() >> (lane ? 2 : cl.linear ? 0 : 1)) & 1u;
    const Program& Q = cl.Q;
    const uint32_t nv = Q.c.n_vars, m = Q.c.n_rows, zj = Q.c.zj, zlo = Q.c.zlo, nc = Q.c.n_cons;
    const bool lin = cl.linear;
    auto S = [](uint32_t v) { return std::to_string(v); };
    auto con_expr = [&](const DevCon& d, uint32_t ci) {
        std::string e = "ezpz::jit::mkcon(" + S(d.kind) + ", " + S(d.tag) + ", " + S(d.nrows);
        for (int i = 0; i < 8; ++i) e += ", " + S(d.ids[i]);
        e += ", " + hexf(d.weight) + ", " + (lane ? hexf(cs[d.pos].param) : "par[" + S(ci) + "]") + ")";
        return e;
    };
    o += "struct Cls" + S((uint32_t)k) + " {\n";
    o += "    static constexpr int NV = " + S(nv) + ", M = " + S(m) + ", NC = " + S(nc) + ", ZJS = " + S(lin ? 0 : zj) +
         ", STRIDE = " + S(cl.H.ninst_pad) + ";\n";
    o += std::string("    static constexpr bool LINEAR = ") + (lin ? "true" : "false") + ";\n";
    const std::string xs = "const double (&x)[" + S(nv) + "], const double (&par)[" + S(std::max(nc, 1u)) + "]";
    // residual sweep (solver.rs:318-356): r = weight * residual, sum of squares, maximum, degenerate mask
    o += "    static __device__ __forceinline__ void residuals(" + xs + ", double (&r)[" + S(std::max(m, 1u)) +
         "], bool active, double& sq, double& mx, unsigned long long& wm) {\n";
    // (one lane per system: in request order, so that the sum of squares is formed exactly like the reference's
    // sequential sum over the rows -- the accept test `sum < previous` of a stalled solve hinges on its last bit)
    std::vector<uint32_t> sweep(nc);
    std::iota(sweep.begin(), sweep.end(), 0u);
    if (lane) std::sort(sweep.begin(), sweep.end(), [&](uint32_t a, uint32_t b) { return Q.cons[a].pos < Q.cons[b].pos; });
    for (uint32_t ci : sweep) {
        const DevCon& d = Q.cons[ci];
        o += "        { const DevCon c = " + con_expr(d, ci) + "; double r0, r1; const bool deg = ezpz::dev::con_residual<LINEAR>(c, x, r0, r1);\n";
        o += "          const double w0 = c.weight * r0; r[" + S(d.row0) + "] = w0; if (active) { sq += w0 * w0; mx = ezpz::dev::fmax_abs(mx, w0); }\n";
        if (d.nrows > 1)
            o += "          const double w1 = c.weight * r1; r[" + S(d.row0 + 1) + "] = w1; if (active) { sq += w1 * w1; mx = ezpz::dev::fmax_abs(mx, w1); }\n";
        o += "          if (!LINEAR && deg) wm |= 1ull << " + S(ci) + "; (void)r1; }\n";
    }
    o += "        (void)x; (void)par; (void)r; (void)active; (void)sq; (void)mx; (void)wm;\n    }\n";
    // Jacobian sweep (solver.rs:359-440)
    o += "    static __device__ __forceinline__ void jacobian(" + xs + ", double (&J)[" + S(lin ? 1u : std::max(zj, 1u)) +
         "], unsigned long long& wm) {\n";
    if (!lin)
        for (uint32_t ci = 0; ci < nc; ++ci) {
            const DevCon& d = Q.cons[ci];
            uint32_t loc[4];
            std::memcpy(loc, d.jloc, 16);
            o += "        { const DevCon c = " + con_expr(d, ci) + "; ezpz::dev::JacWriter<double*> w; w.jv = J; w.jbase = " + S(d.jbase) +
                 "; w.loc[0] = " + S(loc[0]) + "u; w.loc[1] = " + S(loc[1]) + "u; w.loc[2] = " + S(loc[2]) + "u; w.loc[3] = " + S(loc[3]) +
                 "u; w.weight = c.weight;\n";
            o += "          if (ezpz::dev::con_jacobian<false>(c, x, w)) wm |= 1ull << " + S(ci) + "; }\n";
        }
    o += "        (void)x; (void)par; (void)J; (void)wm;\n    }\n";
    // the linear solve (newton.rs:73-102), twice: `solve` divides by a column's diagonal through the column's refined
    // reciprocal (jit_kernel.hip.hpp: recip_of / div_by -- the tail of the compiler's own division sequence, exact while
    // no operand needs scaling; `ok` says whether every operand was in that range), `solve_exact` is the same
    // statements with plain divisions and serves the lanes whose `ok` came back false
    for (int fast = 1; fast >= 0; --fast) {
        o += std::string("    static __device__ __forceinline__ bool ") + (fast ? "solve" : "solve_exact") + "(const double (&J)[" +
             S(lin ? 1u : std::max(zj, 1u)) + "], const double (&r)[" + S(std::max(m, 1u)) + "], double lambda, double (&d)[" + S(nv) +
             "], double& dmax" + (fast ? ", bool& ok" : "") + ") {\n        bool bad = false;\n";
        auto div = [&](const std::string& num, uint32_t col) {
            return fast && fastdiv ? "ezpz::jit::div_by(" + num + ", D" + S(col) + ", Y" + S(col) + ", ok)" : num + " / D" + S(col);
        };
        auto jv = [&](uint32_t slot) { return lin ? hexf(cl.jconst[slot]) : "J[" + S(slot) + "]"; };
        for (uint32_t v = 0; v < nv; ++v) {
            o += "        double D" + S(v) + " = 0.0, V" + S(v) + " = 0.0;\n";
            for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q) {
                const std::string j = jv(Q.colj_items[2 * q]);
                o += "        D" + S(v) + " += " + j + " * " + j + "; V" + S(v) + " += " + j + " * -r[" + S(Q.colj_items[2 * q + 1]) + "];\n";
            }
            o += "        D" + S(v) + " = D" + S(v) + " + lambda;\n";
        }
        for (uint32_t s2 = 0; s2 < zlo; ++s2) {
            o += "        double L" + S(s2) + " = 0.0;\n";
            for (uint32_t q = Q.apair_ptr[s2]; q < Q.apair_ptr[s2 + 1]; ++q)
                o += "        L" + S(s2) + " += " + jv(Q.apairs[2 * q]) + " * " + jv(Q.apairs[2 * q + 1]) + ";\n";
        }
        emit_elimination(o, cl, fast != 0, fastdiv);
        o += std::string("        (void)J; (void)r;") + (fast ? " (void)ok;" : "") + "\n        return bad;\n    }\n";
    }
    // A linear class's matrix J^T J + lambda I is the same for every instance and every system: `factor` is the part of `solve` that
    // depends on lambda alone (diagonals, their refined reciprocals, the entries of L), `solve_f` the rest (right-hand side, the
    // two substitutions) on a factorisation the caller keeps -- in scalar registers, once per lambda and launch
    // (jit_kernel.hip.hpp: solve_kernel_grid_fast).  Every quantity receives exactly `solve`'s operations in `solve`'s order.
    // F: D<v> at 2 v, Y<v> at 2 v + 1, L<s> at 2 NV + s.
    o += "    static constexpr int NF = " + S(lin && !lane ? 2 * nv + zlo : 1) + ";\n";
    if (lin && !lane) {
        auto div = [&](const std::string& num, uint32_t col) {
            return fastdiv ? "ezpz::jit::div_by(" + num + ", D" + S(col) + ", Y" + S(col) + ", ok)" : num + " / D" + S(col);
        };
        auto jv = [&](uint32_t slot) { return hexf(cl.jconst[slot]); };
        const PartDesc part = Q.parts.empty() ? PartDesc{0, 0, 0, 0} : Q.parts[0];
        o += "    static __device__ __forceinline__ bool factor(double lambda, double (&F)[NF], bool& ok) {\n        bool bad = false;\n";
        for (uint32_t v = 0; v < nv; ++v) {
            o += "        double D" + S(v) + " = 0.0;\n";
            for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q) {
                const std::string j = jv(Q.colj_items[2 * q]);
                o += "        D" + S(v) + " += " + j + " * " + j + ";\n";
            }
            o += "        D" + S(v) + " = D" + S(v) + " + lambda;\n";
        }
        for (uint32_t s2 = 0; s2 < zlo; ++s2) {
            o += "        double L" + S(s2) + " = 0.0;\n";
            for (uint32_t q = Q.apair_ptr[s2]; q < Q.apair_ptr[s2 + 1]; ++q)
                o += "        L" + S(s2) + " += " + jv(Q.apairs[2 * q]) + " * " + jv(Q.apairs[2 * q + 1]) + ";\n";
        }
        for (uint32_t lv = 0; lv < part.nlev; ++lv) {
            const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
            const uint32_t s0 = Q.lvl_sptr[part.lvl0 + lv], s1 = Q.lvl_sptr[part.lvl0 + lv + 1];
            for (uint32_t v = c0; v < c1; ++v) {
                for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q) {
                    const std::string l = "L" + S(Q.fwd_items[2 * q]);
                    o += "        D" + S(v) + " -= " + l + " * " + l + ";\n";
                }
                o += "        if (!(D" + S(v) + " > 0.0)) bad = true;\n";
                o += "        D" + S(v) + " = sqrt(D" + S(v) + ");";
                o += fastdiv ? " const double Y" + S(v) + " = ezpz::jit::recip_of(D" + S(v) + ", ok);" : " const double Y" + S(v) + " = 0.0;";
                o += " F[" + S(2 * v) + "] = D" + S(v) + "; F[" + S(2 * v + 1) + "] = Y" + S(v) + ";\n";
            }
            for (uint32_t s2 = s0; s2 < s1; ++s2) {
                for (uint32_t q = Q.lpair_ptr[s2]; q < Q.lpair_ptr[s2 + 1]; ++q)
                    o += "        L" + S(s2) + " -= L" + S(Q.lpairs[2 * q]) + " * L" + S(Q.lpairs[2 * q + 1]) + ";\n";
                o += "        L" + S(s2) + " = " + div("L" + S(s2), Q.l_col[s2]) + "; F[" + S(2 * nv + s2) + "] = L" + S(s2) + ";\n";
            }
        }
        o += "        (void)ok;\n        return bad;\n    }\n";
        o += "    static __device__ __forceinline__ void solve_f(const double (&F)[NF], const double (&r)[" + S(std::max(m, 1u)) + "], double (&d)[" + S(nv) +
             "], double& dmax, bool& ok) {\n";
        for (uint32_t v = 0; v < nv; ++v) o += "        const double D" + S(v) + " = F[" + S(2 * v) + "], Y" + S(v) + " = F[" + S(2 * v + 1) + "]; (void)Y" + S(v) + ";\n";
        for (uint32_t s2 = 0; s2 < zlo; ++s2) o += "        const double L" + S(s2) + " = F[" + S(2 * nv + s2) + "];\n";
        for (uint32_t v = 0; v < nv; ++v) {
            o += "        double V" + S(v) + " = 0.0;\n";
            for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q)
                o += "        V" + S(v) + " += " + jv(Q.colj_items[2 * q]) + " * -r[" + S(Q.colj_items[2 * q + 1]) + "];\n";
        }
        for (uint32_t lv = 0; lv < part.nlev; ++lv) {
            const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
            for (uint32_t v = c0; v < c1; ++v) {
                for (uint32_t q = Q.fwd_ptr[v]; q < Q.fwd_ptr[v + 1]; ++q)
                    o += "        V" + S(v) + " -= L" + S(Q.fwd_items[2 * q]) + " * V" + S(Q.fwd_items[2 * q + 1]) + ";\n";
                o += "        V" + S(v) + " = " + div("V" + S(v), v) + ";\n";
            }
        }
        for (uint32_t lv = part.nlev; lv-- > 0;) {
            const uint32_t c0 = Q.lvl_cptr[part.lvl0 + lv], c1 = Q.lvl_cptr[part.lvl0 + lv + 1];
            for (uint32_t v = c0; v < c1; ++v) {
                for (uint32_t q = Q.bwd_ptr[v]; q < Q.bwd_ptr[v + 1]; ++q)
                    o += "        V" + S(v) + " -= L" + S(Q.bwd_items[2 * q]) + " * V" + S(Q.bwd_items[2 * q + 1]) + ";\n";
                o += "        V" + S(v) + " = " + div("V" + S(v), v) + "; d[" + S(v) + "] = V" + S(v) + "; dmax = ezpz::dev::fmax_abs(dmax, V" + S(v) + ");\n";
            }
        }
        o += "        (void)r; (void)ok;\n    }\n";
    }
    // unsatisfied check (lib.rs:305-327, :358-370)
    o += "    static __device__ __forceinline__ void unsatisfied(" + xs + ", bool active, double& unsat, uint8_t* mask, const uint32_t* pos) {\n";
    for (uint32_t ci = 0; ci < nc; ++ci) {
        const DevCon& d = Q.cons[ci];
        o += "        { const DevCon c = " + con_expr(d, ci) + "; double r0, r1; ezpz::dev::con_residual<LINEAR>(c, x, r0, r1);\n";
        o += std::string("          bool sat = fabs(r0) < ezpz::dev::EPS;") + (d.nrows > 1 ? " sat = sat && (fabs(r1) < ezpz::dev::EPS);" : "") + "\n";
        o += "          if (active) { if (!sat) unsat += 1.0; if (mask) mask[" + (lane ? S(d.pos) : "pos[(size_t)" + S(ci) + " * STRIDE]") + "] = sat ? 0 : 1; } (void)r1; }\n";
    }
    o += "        (void)x; (void)par; (void)active; (void)unsat; (void)mask; (void)pos;\n    }\n";
    o += "    static __device__ __forceinline__ void unsatisfied_from_r(const double (&r)[" + S(std::max(m, 1u)) +
         "], bool active, double& unsat, uint8_t* mask, const uint32_t* pos) {\n";
    for (uint32_t ci = 0; ci < nc; ++ci) {
        const DevCon& d = Q.cons[ci];
        o += "        { bool sat = fabs(r[" + S(d.row0) + "]) < ezpz::dev::EPS;" +
             (d.nrows > 1 ? " sat = sat && (fabs(r[" + S(d.row0 + 1) + "]) < ezpz::dev::EPS);" : "") + "\n";
        o += "          if (active) { if (!sat) unsat += 1.0; if (mask) mask[" + (lane ? S(d.pos) : "pos[(size_t)" + S(ci) + " * STRIDE]") + "] = sat ? 0 : 1; } }\n";
    }
    o += "        (void)r; (void)active; (void)unsat; (void)mask; (void)pos;\n    }\n";
    if (lane) {  // the caller's numbering: variables (row of x0 / x_out) and constraint positions
        o += "    static __device__ __forceinline__ void load(const double* row, double (&x)[" + S(nv) + "]) {\n       ";
        for (uint32_t v = 0; v < nv; ++v) o += " x[" + S(v) + "] = row[" + S(Q.var_of[v]) + "];";
        o += "\n    }\n    static __device__ __forceinline__ void store(double* row, const double (&x)[" + S(nv) + "]) {\n       ";
        for (uint32_t v = 0; v < nv; ++v) o += " row[" + S(Q.var_of[v]) + "] = x[" + S(v) + "];";
        o += "\n    }\n    static __device__ __forceinline__ uint32_t pos_of(int ci) {\n        switch (ci) {\n";
        for (uint32_t ci = 0; ci < nc; ++ci) o += "        case " + S(ci) + ": return " + S(Q.cons[ci].pos) + ";\n";
        o += "        default: return 0;\n        }\n    }\n";
    }
    o += "};\n\n";
}

}  // namespace

// EZPZ_DEBUG=comp in the environment: say on stderr why a system did not get a component plan.
#define COMP_REJECT(...)                                                            \
    do {                                                                            \
        static const bool dbg_ = debug_topic("comp");        \
        if (dbg_) std::fprintf(stderr, "[ezpz comp] no plan: " __VA_ARGS__), std::fputc('\n', stderr); \
        return false;                                                               \
    } while (0)

bool comp_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, const CompLimits& lim, CompPlan& plan) {
    plan = CompPlan();
    if (n_cs == 0 || n_vars < kMinInstances || n_cs > 0x3FFFFFFFu || n_vars > 0x3FFFFFFFu) return false;
    const uint32_t n = (uint32_t)n_vars, C = (uint32_t)n_cs;

    // ---- connected components: every variable of every row of a constraint belongs to one component ---------------------
    std::vector<uint32_t> parent(n);
    std::iota(parent.begin(), parent.end(), 0u);
    auto find = [&](uint32_t a) {
        while (parent[a] != a) a = parent[a] = parent[parent[a]];
        return a;
    };
    uint64_t m_total = 0;
    for (uint32_t i = 0; i < C; ++i) {
        if (cs[i].kind >= EZPZ_NUM_KINDS) return false;
        const KindInfo& K = kKinds[cs[i].kind];
        uint32_t first = NONE;
        for (int r = 0; r < K.n_rows; ++r)
            for (int e = 0; e < K.n_nz[r]; ++e) {
                const uint32_t v = cs[i].ids[K.nz[r][e]];
                if (v >= n) return false;  // MissingGuess: reported by the ordinary symbolic phase
                if (first == NONE)
                    first = v;
                else {
                    const uint32_t ra = find(v), rb = find(first);
                    if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
                }
            }
        m_total += K.n_rows;
    }
    std::vector<uint32_t> comp_of(n, NONE);
    uint32_t n_comps = 0;
    for (uint32_t v = 0; v < n; ++v) {  // roots are the smallest member: components come out in order of first variable
        const uint32_t r = find(v);
        if (comp_of[r] == NONE) comp_of[r] = n_comps++;
        comp_of[v] = comp_of[r];
    }
    if (n_comps < kMinInstances) COMP_REJECT("%u components", n_comps);
    // members by counting sort: variables and constraints of a component in ascending order
    std::vector<uint32_t> vert_ptr(n_comps + 1, 0), con_ptr(n_comps + 1, 0), vert_items(n), con_items(C), con_comp(C);
    for (uint32_t v = 0; v < n; ++v) ++vert_ptr[comp_of[v] + 1];
    for (uint32_t i = 0; i < C; ++i) {
        con_comp[i] = comp_of[cs[i].ids[kKinds[cs[i].kind].nz[0][0]]];
        ++con_ptr[con_comp[i] + 1];
    }
    for (uint32_t c = 0; c < n_comps; ++c) vert_ptr[c + 1] += vert_ptr[c], con_ptr[c + 1] += con_ptr[c];
    {
        std::vector<uint32_t> vfill(vert_ptr.begin(), vert_ptr.end() - 1), cfill(con_ptr.begin(), con_ptr.end() - 1);
        for (uint32_t v = 0; v < n; ++v) vert_items[vfill[comp_of[v]]++] = v;
        for (uint32_t i = 0; i < C; ++i) con_items[cfill[con_comp[i]]++] = i;
    }
    std::vector<Component> comps(n_comps);
    for (uint32_t c = 0; c < n_comps; ++c) {
        comps[c].verts = Span{vert_items.data() + vert_ptr[c], vert_ptr[c + 1] - vert_ptr[c]};
        comps[c].cons = Span{con_items.data() + con_ptr[c], con_ptr[c + 1] - con_ptr[c]};
    }
    for (const Component& c : comps)
        if (c.verts.size() > kMaxClassVars || c.cons.size() > kMaxClassCons)
            COMP_REJECT("a component of %zu variables, %zu constraints", c.verts.size(), c.cons.size());

    // ---- classes: components with the same kinds, tags, weights and local variable pattern ---------------------------------
    plan.unit_weights = true;
    for (uint32_t i = 0; i < C; ++i)
        if (cs[i].weight != 1.0) plan.unit_weights = false;
    std::vector<std::vector<uint32_t>> class_sig;  // (at most kMaxClasses: compared one by one)
    std::vector<Class> classes;
    std::vector<uint32_t> sig;
    auto local_con = [&](const Component& comp, uint32_t ci, EzpzConstraint& out) {
        out = cs[ci];
        const KindInfo& K = kKinds[out.kind];
        bool used[8] = {false, false, false, false, false, false, false, false};
        for (int r = 0; r < K.n_rows; ++r)
            for (int e = 0; e < K.n_nz[r]; ++e) used[K.nz[r][e]] = true;
        for (int k = 0; k < 8; ++k) {
            if (k < K.n_ids && used[k])
                out.ids[k] = (uint32_t)(std::lower_bound(comp.verts.begin(), comp.verts.end(), out.ids[k]) - comp.verts.begin());
            else
                out.ids[k] = 0;  // ids a kind never dereferences
        }
        out.priority = 0;
        out.flags = 0;
    };
    for (uint32_t ic = 0; ic < comps.size(); ++ic) {
        const Component& comp = comps[ic];
        sig.clear();
        sig.push_back((uint32_t)comp.verts.size());
        sig.push_back((uint32_t)comp.cons.size());
        for (uint32_t ci : comp.cons) {
            EzpzConstraint lc;
            local_con(comp, ci, lc);
            uint64_t wbits;
            std::memcpy(&wbits, &lc.weight, 8);
            sig.push_back(lc.kind | ((uint32_t)lc.tag << 16));
            sig.push_back((uint32_t)wbits);
            sig.push_back((uint32_t)(wbits >> 32));
            for (int k = 0; k < 8; ++k) sig.push_back(lc.ids[k]);
        }
        size_t k = 0;
        while (k < class_sig.size() && class_sig[k] != sig) ++k;
        if (k == class_sig.size()) {
            if (classes.size() >= kMaxClasses) COMP_REJECT("more than %zu classes", kMaxClasses);
            class_sig.push_back(sig);
            classes.emplace_back();
            classes.back().rep = ic;
        }
        classes[k].instances.push_back(ic);
    }

    // ---- per class: the symbolic phase of the representative, then its records -----------------------------------------------
    bool all_linear = true;
    for (Class& cl : classes) {
        const Component& rep = comps[cl.rep];
        std::vector<EzpzConstraint> local(rep.cons.size());
        for (size_t k = 0; k < rep.cons.size(); ++k) local_con(rep, rep.cons[k], local[k]);
        BuildError be;
        if (!build_program(local.data(), local.size(), rep.verts.size(), cl.Q, be, 1, false)) return false;
        const Program& Q = cl.Q;
        if (Q.c.n_parts != 1 || Q.c.zj >= 0xFFFF || Q.c.zlo >= 0xFFFF || Q.c.n_rows >= 0xFFFF) return false;
        for (const DevCon& d : Q.cons) cl.linear = cl.linear && kind_is_linear(d.kind);
        if (cl.linear) {  // constant Jacobian: slot values exactly as con_jacobian's writer stores them
            cl.jconst.assign(Q.c.zj, 0.0);
            for (const DevCon& d : Q.cons) {
                double pd[8];
                const int np = linear_partials(d.kind, pd);
                for (int e = 0; e < np; ++e) {
                    const uint32_t code = d.jloc[e];
                    const uint32_t slot = d.jbase + (code & 0x7Fu);
                    const double w = d.weight * pd[e];
                    if (code & 0x80u)
                        cl.jconst[slot] = cl.jconst[slot] + w;
                    else
                        cl.jconst[slot] = w;
                }
            }
            for (double v : cl.jconst) cl.consts_f32 = cl.consts_f32 && f32_exact(v);
        }
        all_linear = all_linear && cl.linear && cl.consts_f32;
        uint32_t cost = 8;
        for (const DevCon& d : Q.cons) cost += kind_is_linear(d.kind) ? 12 : 60;
        cost += 40 * Q.c.n_vars + 14 * Q.c.zlo + 4 * (uint32_t)(Q.colj_items.size() / 2 + Q.apairs.size() / 2 + Q.lpairs.size() / 2 +
                                                                Q.fwd_items.size() / 2 + Q.bwd_items.size() / 2);
        cl.cost = cost;
    }
    plan.linear = all_linear;
    const bool LIN = plan.linear;

    std::vector<uint32_t>& blob = plan.blob;
    blob.clear();
    uint32_t scratch_rows = 1;
    for (size_t k = 0; k < classes.size(); ++k) {
        Class& cl = classes[k];
        const Program& Q = cl.Q;
        const uint32_t nv = Q.c.n_vars, m = Q.c.n_rows, zj = Q.c.zj, zlo = Q.c.zlo, ncons = Q.c.n_cons;
        ClassLayout& H = cl.H;
        H.nv = nv, H.m = m, H.zj = zj, H.ncons = ncons;
        H.o_d = nv;
        H.o_r0 = 2 * nv;
        H.o_r1 = 2 * nv + m;
        H.o_j = 2 * nv + 2 * m;
        H.o_wm = H.o_j + (LIN ? 0 : zj);
        H.rows_p = H.o_wm + (LIN ? 0 : 1);
        H.s_l = nv;
        scratch_rows = std::max(scratch_rows, nv + zlo);
        const uint32_t ninst = (uint32_t)cl.instances.size();
        H.ninst_pad = (ninst + 63) & ~63u;

        emit_class_program(blob, cl, LIN, nullptr);
        align4(blob);
        H.ids_off = (uint32_t)blob.size();
        blob.resize(blob.size() + (size_t)nv * H.ninst_pad, 0);
        for (uint32_t kv = 0; kv < nv; ++kv)
            for (uint32_t i = 0; i < ninst; ++i)
                blob[H.ids_off + (size_t)kv * H.ninst_pad + i] = comps[cl.instances[i]].verts[Q.var_of[kv]];
        align4(blob);
        H.par_off = (uint32_t)blob.size();
        blob.resize(blob.size() + (size_t)2 * ncons * H.ninst_pad, 0);
        for (uint32_t ci = 0; ci < ncons; ++ci)
            for (uint32_t i = 0; i < ninst; ++i) {
                const double p = cs[comps[cl.instances[i]].cons[Q.cons[ci].pos]].param;
                std::memcpy(&blob[H.par_off + 2 * ((size_t)ci * H.ninst_pad + i)], &p, 8);
            }
        align4(blob);
        H.pos_off = (uint32_t)blob.size();
        blob.resize(blob.size() + (size_t)ncons * H.ninst_pad, 0);
        for (uint32_t ci = 0; ci < ncons; ++ci)
            for (uint32_t i = 0; i < ninst; ++i)
                blob[H.pos_off + (size_t)ci * H.ninst_pad + i] = comps[cl.instances[i]].cons[Q.cons[ci].pos];
        if (blob.size() > (64u << 20)) return false;

        plan.zj += (uint64_t)zj * ninst;
        plan.za += (uint64_t)Q.c.za * ninst;
        plan.zl += (uint64_t)(zlo + nv) * ninst;
        plan.max_levels = std::max(plan.max_levels, Q.c.n_levels);
    }

    // ---- chunks of <= 64 instances, dealt to the wavefronts (longest processing time first) --------------------------------------
    struct ChunkTmp {
        uint32_t cls, inst0, count, cost, rows;
    };
    std::vector<ChunkTmp> chunks;
    for (size_t k = 0; k < classes.size(); ++k) {
        const uint32_t ninst = (uint32_t)classes[k].instances.size();
        for (uint32_t i0 = 0; i0 < ninst; i0 += 64)
            chunks.push_back(ChunkTmp{(uint32_t)k, i0, std::min<uint32_t>(64, ninst - i0), classes[k].cost, classes[k].H.rows_p});
    }
    uint64_t rows_persistent = 0;
    for (const ChunkTmp& c : chunks) rows_persistent += c.rows;
    // Wavefronts per workgroup (= per system): more of them shorten a solve, but every one pays the reductions of
    // the LM control, and a CU holds 160 KiB of LDS: take the count that puts the most wavefronts on a CU (several
    // workgroups side by side when the state allows it), the smaller workgroup on a tie.
    const uint32_t max_waves = LIN ? lim.max_waves_linear : lim.max_waves_general;
    const size_t fixed_bytes = 3 * 4 * 16 * 8 + 64;  // reduction scratch + flags + warning counters
    auto bytes_for = [&](uint32_t w) { return (rows_persistent + (uint64_t)w * scratch_rows) * 512 + fixed_bytes; };
    uint32_t W = 0;
    uint64_t best_waves = 0;
    for (uint32_t w = 1; w <= max_waves; w <<= 1) {
        if (w > chunks.size() && w > 1) break;
        const uint64_t bytes = bytes_for(w);
        if (bytes > lim.lds_bytes) continue;
        const uint64_t per_cu = std::min<uint64_t>(lim.lds_bytes / bytes, 32 / w);
        const uint64_t waves = std::min<uint64_t>(per_cu * w, 16);  // beyond 4 per SIMD nothing is gained
        if (waves > best_waves) best_waves = waves, W = w;
    }
    // State that does not fit one CU's LDS (one large system: the 200 000-variable ladder): no interpreter, but the
    // class-specialised kernel keeps its state in registers and spreads the system over several workgroups (below).
    plan.interpretable = W != 0;
    if (!plan.interpretable) W = 1;
    std::vector<uint32_t> order(plan.interpretable ? chunks.size() : 0);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return chunks[a].cost > chunks[b].cost; });
    std::vector<uint64_t> load(W, 0);
    std::vector<std::vector<uint32_t>> of_wave(W);
    for (uint32_t c : order) {
        const uint32_t w = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
        load[w] += chunks[c].cost;
        of_wave[w].push_back(c);
    }
    align4(blob);
    plan.o_waves = (uint32_t)blob.size();
    blob.resize(blob.size() + W + 1, 0);
    while (blob.size() % 32) blob.push_back(0);
    plan.o_chunks = (uint32_t)blob.size();
    uint32_t row = 0, nch = 0;
    for (uint32_t w = 0; w < W; ++w) {
        std::sort(of_wave[w].begin(), of_wave[w].end());  // class-major, instances ascending
        blob[plan.o_waves + w] = nch;
        for (uint32_t c : of_wave[w]) {
            const ClassLayout& H = classes[chunks[c].cls].H;
            CompChunk ch{};
            ch.count = chunks[c].count;
            ch.nv = H.nv, ch.m = H.m, ch.ncons = H.ncons;
            ch.n_ops = H.n_ops, ch.ops_off = H.ops_off, ch.cons_off = H.cons_off;
            ch.row0 = row;
            ch.o_d = H.o_d, ch.o_r0 = H.o_r0, ch.o_r1 = H.o_r1, ch.o_j = H.o_j, ch.o_wm = H.o_wm, ch.s_l = H.s_l;
            ch.stride = H.ninst_pad;
            ch.ids_off = H.ids_off + chunks[c].inst0;
            ch.par_off = H.par_off + 2 * chunks[c].inst0;
            ch.pos_off = H.pos_off + chunks[c].inst0;
            const uint32_t* p = reinterpret_cast<const uint32_t*>(&ch);
            blob.insert(blob.end(), p, p + sizeof(CompChunk) / 4);
            row += chunks[c].rows;
            ++nch;
        }
    }
    blob[plan.o_waves + W] = nch;
    blob.resize(blob.size() + 16, 0);
    plan.n_waves = W;
    plan.n_chunks = nch;
    plan.n_classes = (uint32_t)classes.size();
    plan.n_instances = (uint32_t)comps.size();
    plan.rows_persistent = (uint32_t)rows_persistent;
    plan.scratch_rows = scratch_rows;
    plan.lds_bytes = plan.interpretable ? (uint32_t)bytes_for(W) : 0u;
    plan.n_vars = n;
    plan.n_cons = C;
    plan.n_rows = (uint32_t)m_total;

    // ---- the same plan as source text for the class-specialised kernel --------------------------------------------------------------
    // A wavefront gets ceil(chunks of class k / T) slots of every class k -- the same sequence for every wavefront, so
    // one body of code serves them all; a slot beyond a class's last chunk runs with no active lane.  T = the fewest
    // wavefronts per system (every one pays the LM control's reductions) whose slots still fit the register file.
    {
        std::vector<uint32_t> nchunk(classes.size());
        for (size_t k = 0; k < classes.size(); ++k) nchunk[k] = ((uint32_t)classes[k].instances.size() + 63) / 64;
        auto vgprs = [&](uint32_t T) {
            uint64_t v = 0;
            for (size_t k = 0; k < classes.size(); ++k) {
                const ClassLayout& H = classes[k].H;
                const uint64_t per = 2ull * (2 * H.nv + 2 * H.m + (classes[k].linear ? 0 : H.zj) + H.ncons) + H.nv + 4;
                v += per * ((nchunk[k] + T - 1) / T);
            }
            return v;
        };
        auto per_slot = [&](size_t k) {
            const ClassLayout& H = classes[k].H;
            return 2ull * (2 * H.nv + 2 * H.m + (classes[k].linear ? 0 : H.zj) + H.ncons) + H.nv + 4;
        };
        uint32_t T = 0, G = 1;
        std::vector<uint32_t> slots_k(classes.size(), 0);  // slots of class k per wavefront
        static const char* env_t = std::getenv("EZPZ_JIT_WAVES");
        if (env_t && std::atoi(env_t) > 0 && plan.interpretable) {
            T = (uint32_t)std::atoi(env_t);
        } else {
            // measured (2000 x 2000 / 2400 x 2400 / 800 x 800, M solves/s): the fewest wavefronts whose slots take up to
            // ~176 VGPRs win -- T = 2 / 4 / 8 on 2000 x 2000 (224 / 112 / 56 VGPRs of state): 58 / 68 / 57; T = 4 / 8 on
            // 2400 x 2400 (153 / 97): 53 / 36; T = 2 / 4 / 8 on 800 x 800 (112 / 56 / 41): 153 / 120 / 72
            for (uint32_t t = 1; t <= 8 && !T; t <<= 1)
                if (vgprs(t) <= 176) T = t;
        }
        if (T) {
            for (size_t k = 0; k < classes.size(); ++k) slots_k[k] = (nchunk[k] + T - 1) / T;
        } else {
            // One workgroup cannot hold the system: G workgroups of 4 wavefronts share it (grid reductions, see
            // jit_kernel.hip.hpp).  A wavefront takes slots of every class in proportion to the class's chunks, ~112
            // VGPRs of state in all.
            T = 4;
            uint64_t weight = 0;
            for (size_t k = 0; k < classes.size(); ++k) weight += (uint64_t)nchunk[k] * per_slot(k);
            uint32_t waves = 0;
            for (size_t k = 0; k < classes.size(); ++k) {
                slots_k[k] = std::max<uint32_t>(1, (uint32_t)((double)nchunk[k] * 112.0 / (double)weight + 0.5));
                waves = std::max(waves, (nchunk[k] + slots_k[k] - 1) / slots_k[k]);
            }
            G = (waves + T - 1) / T;
            if (G <= 1) {  // few chunks of many big classes: one workgroup after all, one slot per class and wavefront
                G = 1;
                T = 8;
                for (size_t k = 0; k < classes.size(); ++k) slots_k[k] = (nchunk[k] + T - 1) / T;
            }
        }
        uint64_t vg = 0;
        for (size_t k = 0; k < classes.size(); ++k) vg += per_slot(k) * slots_k[k];
        if (T && T <= 16 && vg <= 360 && G <= 256 && (plan.interpretable || G > 1)) {
            std::string& o = plan.jit_source;
            o = "#include \"jit_kernel.hip.hpp\"\nusing ezpz::DevCon;\n\n";
            for (size_t k = 0; k < classes.size(); ++k) emit_class(o, k, classes[k]);
            std::string seq;
            uint32_t nslots = 0;
            std::vector<std::pair<uint32_t, uint32_t>> slot_of;  // (class, index among the wavefront's slots of that class)
            for (size_t k = 0; k < classes.size(); ++k)
                for (uint32_t j = 0; j < slots_k[k]; ++j) {
                    seq += (nslots ? ", Cls" : "Cls") + std::to_string(k);
                    slot_of.emplace_back((uint32_t)k, j);
                    ++nslots;
                }
            bool any_nonlinear = false;
            for (const Class& cl : classes) any_nonlinear = any_nonlinear || !cl.linear;
            static const char* env_mw = std::getenv("EZPZ_JIT_MINWAVES");  // occupancy hint (waves per SIMD), for measurements
            // (the slots' state + ~40 working registers: with the matching occupancy as a hint the compiler spends the
            // whole register budget of that occupancy on scheduling -- 2000 x 2000: 66.9 -> 74.2 M solves/s at 3; a hint
            // the state does not fit costs spills: 2400 x 2400 at 3 instead of 2: 53 -> 43)
            const int min_waves = env_mw ? std::atoi(env_mw) : (int)std::min<uint64_t>(4, std::max<uint64_t>(1, 512 / (vg + 50)));
            const std::string bounds = std::to_string(T * 64) + (min_waves > 0 ? ", " + std::to_string(min_waves) : "");
            // eval()'s sums riding in the first iteration's rendezvous (jit_kernel.hip.hpp: FUSE) -- one rendezvous less per system.
            // Measured on one box, 65 536 systems per launch, once the kernels of one-workgroup systems carried no grid code:
            // 2000 x 2000 (four wavefronts per system, three per SIMD) 96.0 -> 101.9 M solves/s, 800 x 800 (two, three) 195.8 ->
            // 214.2, 2400 x 2400 (four, two per SIMD: eight slots per lane) 65.3 -> 60.1, the over-constrained variant (non-linear
            // classes) 24.8 -> 23.8: taken for linear systems compiled for three or more wavefronts per SIMD;
            // One workgroup per system only.
            const bool fuse = !any_nonlinear && min_waves >= 3;
            // two entries: batches, and (`_one`) one-call launches that stay resident for the caller's next request
            for (int one = 0; one < 2; ++one) {
                o += "extern \"C\" __global__ void __launch_bounds__(" + bounds + ") ezpz_jit_solve" + (one ? "_one" : "") + "(const ezpz::jit::JitArgs a) {\n";
                o += "    __shared__ double smem[ezpz::jit::kRedDoubles + 16];\n";
                if (G > 1)  // a system on several workgroups: the kernel with the reductions' grid stage (jit_kernel.hip.hpp: solve_kernel_grid)
                    o += "    ezpz::jit::solve_kernel_grid<ezpz::jit::Slots<" + seq + ">, " + std::to_string(T) + ", " + (any_nonlinear ? "true" : "false") + ", " +
                         (plan.unit_weights ? "true" : "false") + ", " + (one ? "true" : "false") + ">(a, smem);\n}\n";
                else
                    o += "    ezpz::jit::solve_kernel<ezpz::jit::Slots<" + seq + ">, " + std::to_string(T) + ", " + (any_nonlinear ? "true" : "false") + ", " +
                         (plan.unit_weights ? "true" : "false") + ", " + (one ? "true" : "false") + ", " + (fuse ? "true" : "false") + ", false>(a, smem);\n}\n";
            }
            // the caller's variables of every wavefront (of the G * T that share a system): one contiguous run?  Then the one-workgroup
            // kernel below moves a wavefront's piece of a row as full lines (jit_kernel.hip.hpp: fast_wave, IO 2) -- for systems of four
            // wavefronts or more: same box, previous build / this one, M solves/s: 2000 x 2000 134.8 -> 142.0, 2400 x 2400 113.1 ->
            // 111.5, but 800 x 800 (two wavefronts per system) 289 -> 274; and not for a system on several workgroups (the ladder, same
            // box: 256 systems per launch 0.93 -> 0.99 M, but the 64-system launches of the bench leg 0.87 -> 0.81 M: the first system's
            // piece goes through LDS before anything starts)
            std::vector<uint32_t> wave_lo(G * T, 0), wave_n(G * T, 0);
            bool contiguous = true;
            for (uint32_t w = 0; w < G * T && contiguous; ++w) {
                uint32_t lo = ~0u, hi = 0, cnt = 0;
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                    const uint32_t k = slot_of[sl].first;
                    const Class& cl = classes[k];
                    const uint32_t chunk = w * slots_k[k] + slot_of[sl].second, ninst = (uint32_t)cl.instances.size();
                    for (uint32_t i = chunk * 64; i < ninst && i < chunk * 64 + 64; ++i)
                        for (uint32_t kv = 0; kv < cl.H.nv; ++kv) {
                            const uint32_t id = comps[cl.instances[i]].verts[cl.Q.var_of[kv]];
                            lo = std::min(lo, id), hi = std::max(hi, id), ++cnt;
                        }
                }
                if (cnt == 0) continue;  // (a wavefront beyond the system's last chunk)
                // (every variable belongs to one instance: as many variables as the run is long means exactly the run)
                contiguous = hi - lo + 1 == cnt;
                wave_lo[w] = lo, wave_n[w] = cnt;
            }
            // a LINEAR system with unit weights: the kernel that does not wait for the verdicts of the LM control (jit_kernel.hip.hpp:
            // solve_kernel_fast / solve_kernel_grid_fast) and, for a system on one workgroup, the loop over the systems that kernel
            // lists (`_list`; several workgroups: the first entry reads the list itself).  The first needs far fewer registers than the
            // loop: compiled for four wavefronts per SIMD while its state fits
            if (!any_nonlinear && plan.unit_weights) {
                uint64_t fast_vg = 0;  // the next system's x, parameters (doubles), ids; one slot's x, d, r; the eight partials
                uint64_t widest = 0;
                for (size_t k = 0; k < classes.size(); ++k) {
                    const ClassLayout& H = classes[k].H;
                    fast_vg += (2ull * (H.nv + H.ncons) + H.nv) * slots_k[k];
                    widest = std::max<uint64_t>(widest, 2ull * (2 * H.nv + H.m));
                }
                fast_vg += widest + 16;
                static const char* env_fw = std::getenv("EZPZ_JIT_FAST_MINWAVES");  // occupancy hint, for measurements
                // (several workgroups per system: the variant with a wavefront per SIMD less has the registers to let the values wait in
                // LDS for stores that go out back to back; EZPZ_JIT_GRID_STAGE=0: slot by slot there too, for measurements)
                static const int grid_stage = [] {
                    const char* e = std::getenv("EZPZ_JIT_GRID_STAGE");
                    return e ? std::atoi(e) : 1;  // (2 = its piece of the row as whole lines where contiguous: measured 2-6 % slower on the ladder)
                }();
                const int fast_waves = env_fw ? std::atoi(env_fw) : (fast_vg + 40 <= 128 ? 4 : fast_vg + 40 <= 168 ? 3 : 2);
                // (twice: for the estimated occupancy and for one wavefront per SIMD less -- the loader takes the first that keeps its
                // registers out of scratch memory, jit.cpp: ensure_loaded)
                for (int variant = 0; variant < (fast_waves > 2 && !env_fw ? 2 : 1); ++variant) {
                    o += "extern \"C\" __global__ void __launch_bounds__(" + std::to_string(T * 64) + ", " + std::to_string(fast_waves - variant) +
                         ") ezpz_jit_solve_fast" + (variant ? "_b" : "") + "(const ezpz::jit::JitArgs a) {\n";
                    o += std::string("    ezpz::jit::") + (G > 1 ? "solve_kernel_grid_fast" : "solve_kernel_fast") + "<ezpz::jit::Slots<" + seq + ">, " + std::to_string(T) +
                         (G > 1 ? (variant && grid_stage ? (grid_stage >= 2 && contiguous && T >= 4 ? ", 2" : ", 1") : ", 0") : contiguous && T >= 4 ? ", true" : ", false") + ">(a);\n}\n";
                }
                if (G == 1) {
                    o += "extern \"C\" __global__ void __launch_bounds__(" + bounds + ") ezpz_jit_solve_list(const ezpz::jit::JitArgs a) {\n";
                    o += "    __shared__ double smem[ezpz::jit::kRedDoubles + 16];\n";
                    o += "    ezpz::jit::solve_kernel_grid<ezpz::jit::Slots<" + seq + ">, " + std::to_string(T) + ", false, true, false>(a, smem);\n}\n";
                }
            }
            align4(blob);
            plan.o_jit_slots = (uint32_t)blob.size();
            for (uint32_t w = 0; w < G * T; ++w)  // wavefront w of the system (workgroup w / T) owns consecutive chunks
                for (uint32_t sl = 0; sl < nslots; ++sl) {
                    const uint32_t k = slot_of[sl].first;
                    const ClassLayout& H = classes[k].H;
                    const uint32_t chunk = w * slots_k[k] + slot_of[sl].second;  // chunk index inside the class
                    const uint32_t ninst = (uint32_t)classes[k].instances.size();
                    const uint32_t inst0 = std::min(chunk * 64, H.ninst_pad ? H.ninst_pad - 64 : 0u);
                    const uint32_t count = (uint64_t)chunk * 64 < ninst ? std::min<uint32_t>(64, ninst - chunk * 64) : 0u;
                    blob.push_back(H.ids_off + inst0);
                    blob.push_back(H.par_off + 2 * inst0);
                    blob.push_back(H.pos_off + inst0);
                    blob.push_back(count);
                }
            blob.resize(blob.size() + 16, 0);
            plan.o_jit_ranges = 0;
            if (contiguous) {  // [wave] {first variable, count}
                plan.o_jit_ranges = (uint32_t)blob.size();
                for (uint32_t w = 0; w < G * T; ++w) blob.push_back(wave_lo[w]), blob.push_back(wave_n[w]);
                blob.resize(blob.size() + 16, 0);
            }
            plan.jit_waves = T;
            plan.jit_slots = nslots;
            plan.jit_wgs = G;
        }
    }
    if (!plan.interpretable && plan.jit_source.empty())
        COMP_REJECT("%llu rows of state do not fit the LDS and the system has no multi-workgroup specialised form",
                    (unsigned long long)rows_persistent);
    return true;
}

// The elimination of the wavefront kernel ACROSS its lanes (`tail_wave`): lane i holds row i of L (its entries a<k> = L(i, k),
// its diagonal and its right-hand side), columns are eliminated right-looking in the class program's order -- the pivot's
// square root and the column's division once per column instead of once per entry, the update of the later columns one
// multiply-subtract per column pair on all lanes at once, operands of other lanes through v_readlane -- and the backward
// substitution walks every column's entries in the serial order.  Every entry receives exactly the operations of `tail`
// (emit_elimination) in the same order -- the columns are taken in an order that respects every list of the class program
// (a row's updates, an entry's updates), the backward sums follow their lists;
// entries that are structurally zero hold zeros and receive multiples of zero -- so the step is the same bits
// (tests/test_gpu_lanes.py) as long as those multiples ARE zero: a right-hand side that is not finite (a NaN guess, an
// overflow) would turn 0 x y_k into NaN on rows the serial order never touches, so `fin` reports whether every y_k was
// finite and the kernel repeats the solve with the serial `tail` when not (pivots never see the right-hand side: `bad` is
// the same either way; a non-finite entry of L fails its row's pivot in both orders).  wave_row_structure checks what that rests on (lists complete and compatible with one column order, fill closed) and the serial
// `tail` stays when it does not hold.
struct WaveRows {
    bool ok = false;
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> col;  // per column k: (row i, slot) ascending in i
    std::vector<std::vector<uint32_t>> bwd;                         // per column k: its rows in the backward sum's order
    std::vector<uint32_t> order;                                    // the columns in elimination order
    std::vector<uint32_t> slot_row;
};
WaveRows wave_row_structure(const Class& cl) {
    WaveRows w;
    auto why = [&](int code) {
        static const bool dbg = debug_topic("jit");
        if (dbg) std::fprintf(stderr, "[ezpz jit] wave: elimination across lanes not used (check %d)\n", code);
        return w;
    };
    const Program& Q = cl.Q;
    const uint32_t nv = Q.c.n_vars, zlo = Q.c.zlo;
    if (nv < 2 || nv > 64) return why(1);
    w.col.assign(nv, {});
    w.slot_row.assign(zlo, ~0u);
    std::vector<std::vector<uint32_t>> slot_of(nv, std::vector<uint32_t>(nv, ~0u));
    // edges of "column a is eliminated before column b": what fixes the order in which an entry receives its updates
    std::vector<std::vector<uint32_t>> after(nv);
    std::vector<uint32_t> waiting(nv, 0);
    auto before = [&](uint32_t x, uint32_t y) {
        after[x].push_back(y);
        ++waiting[y];
    };
    for (uint32_t j = 0; j < nv; ++j) {
        uint32_t last = ~0u;
        for (uint32_t q = Q.fwd_ptr[j]; q < Q.fwd_ptr[j + 1]; ++q) {
            const uint32_t slot = Q.fwd_items[2 * q], k = Q.fwd_items[2 * q + 1];
            if (slot >= zlo || k >= j || Q.l_col[slot] != k || slot_of[j][k] != ~0u) return why(2);
            // row j's diagonal and right-hand side take their updates in the list's order, then column j is eliminated
            if (last != ~0u) before(last, k);
            before(k, j);
            last = k;
            w.slot_row[slot] = j;
            slot_of[j][k] = slot;
        }
    }
    for (uint32_t s2 = 0; s2 < zlo; ++s2)
        if (w.slot_row[s2] == ~0u) return why(3);
    w.bwd.assign(nv, {});
    for (uint32_t k = 0; k < nv; ++k) {
        for (uint32_t i = k + 1; i < nv; ++i)
            if (slot_of[i][k] != ~0u) w.col[k].push_back({i, slot_of[i][k]});
        // backward list of column k: the same entries, in the order the serial sum takes them
        if (Q.bwd_ptr[k + 1] - Q.bwd_ptr[k] != w.col[k].size()) return why(4);
        for (uint32_t q = Q.bwd_ptr[k]; q < Q.bwd_ptr[k + 1]; ++q) {
            const uint32_t slot = Q.bwd_items[2 * q], i = Q.bwd_items[2 * q + 1];
            if (i >= nv || i <= k || slot_of[i][k] != slot) return why(5);
            w.bwd[k].push_back(i);
        }
    }
    // an entry's updates: one per earlier column in which both its row and its column have entries, in the list's order
    for (uint32_t s2 = 0; s2 < zlo; ++s2) {
        const uint32_t i = w.slot_row[s2], j = Q.l_col[s2];
        size_t want = 0;
        for (uint32_t k = 0; k < j; ++k) want += slot_of[i][k] != ~0u && slot_of[j][k] != ~0u;
        if (Q.lpair_ptr[s2 + 1] - Q.lpair_ptr[s2] != want) return why(6);
        uint32_t last = ~0u;
        for (uint32_t q = Q.lpair_ptr[s2]; q < Q.lpair_ptr[s2 + 1]; ++q) {
            const uint32_t sa = Q.lpairs[2 * q], sb = Q.lpairs[2 * q + 1];
            if (sa >= zlo || sb >= zlo) return why(7);
            const uint32_t k = Q.l_col[sa];
            if (k >= j || Q.l_col[sb] != k || slot_of[i][k] != sa || slot_of[j][k] != sb) return why(7);
            if (last != ~0u) before(last, k);
            last = k;
        }
    }
    // fill closed: two entries of a column imply the entry between their rows
    for (uint32_t k = 0; k < nv; ++k)
        for (size_t x = 0; x < w.col[k].size(); ++x)
            for (size_t y = x + 1; y < w.col[k].size(); ++y)
                if (slot_of[w.col[k][y].first][w.col[k][x].first] == ~0u) return why(8);
    // one order of the columns that respects every list, taken in rounds (every column whose predecessors are all in earlier
    // rounds, ascending): the class program's own order when its lists ascend, and columns of different branches of the
    // elimination tree stay next to each other, which is what lets emit_tail_wave group their pivots
    std::vector<char> done(nv, 0);
    while (w.order.size() < nv) {
        std::vector<uint32_t> round;
        for (uint32_t v = 0; v < nv; ++v)
            if (!done[v] && waiting[v] == 0) round.push_back(v);
        if (round.empty()) return why(9);  // (lists that contradict each other: no such order)
        for (uint32_t v : round) {
            done[v] = 1;
            w.order.push_back(v);
        }
        for (uint32_t v : round)
            for (uint32_t y : after[v]) --waiting[y];
    }
    w.ok = true;
    return w;
}
void emit_wave_row_table(std::string& o, const Class& cl) {
    const WaveRows w = wave_row_structure(cl);
    if (!w.ok) return;
    const Program& Q = cl.Q;
    const uint32_t nv = Q.c.n_vars, nq = 2 * nv + Q.c.zlo;
    auto S = [](uint32_t v) { return std::to_string(v); };
    // kWaveRow[k * 64 + lane] = where lane's entry of column k sits among the assembled quantities (nq = a zero)
    o += "__device__ const uint16_t kWaveRow[" + S((nv - 1) * 64) + "] = {";
    for (uint32_t k = 0; k + 1 < nv; ++k)
        for (uint32_t l = 0; l < 64; ++l) {
            const uint32_t row = std::min(l, nv - 1);
            uint32_t at = nq;
            for (const auto& e : w.col[k])
                if (e.first == row) at = 2 * nv + e.second;
            o += ((k * 64 + l) % 32 == 0 ? "\n    " : " ") + S(at) + ",";
        }
    o += "\n};\n";
}
void emit_tail_wave(std::string& o, const Class& cl) {
    const WaveRows w = wave_row_structure(cl);
    auto S = [](uint32_t v) { return std::to_string(v); };
    if (!w.ok) {
        o += "    static constexpr bool HAS_TAIL_WAVE = false;\n    static constexpr int NROW = 1;\n";
        o += "    static __device__ __forceinline__ uint32_t row_slot(int, int) { return 0; }\n";
        o += "    static __device__ __forceinline__ bool tail_wave(const double*, double, const uint32_t (&)[1], int, double*, double&, bool&) { return false; }\n";
        return;
    }
    const uint32_t nv = cl.Q.c.n_vars;
    o += "    static constexpr bool HAS_TAIL_WAVE = true;\n    static constexpr int NROW = " + S(nv - 1) + ";\n";
    o += "    static __device__ __forceinline__ uint32_t row_slot(int k, int lane) { return kWaveRow[k * 64 + lane]; }\n";
    o += "    static __device__ __forceinline__ bool tail_wave(const double* Q, double lambda, const uint32_t (&rs)[" + S(nv - 1) +
         "], int lane, double* d, double& dmax, bool& fin) {\n        using ezpz::jit::lane_value;\n        bool bad = false;\n";
    o += "        const int me = lane < NV ? lane : NV - 1;\n        double Dm = Q[me] + lambda, Vm = Q[NV + me];\n";
    for (uint32_t k = 0; k + 1 < nv; ++k) o += "        double a" + S(k) + " = Q[rs[" + S(k) + "]];\n";
    // Updates are applied in w.order (what fixes every entry's rounding); a column's pivot, its square root and its divisions
    // only need the updates of ITS row to have been applied, so they are issued as early as that allows, together with the
    // other columns that are ready then: independent square-root / division chains next to each other instead of one after
    // the other (a sparse sketch's elimination tree is wide: 14 columns in 8 such groups for two rectangles).
    {
        const Program& Q = cl.Q;
        std::vector<char> pivoted(nv, 0), applied(nv, 0);
        auto ready = [&](uint32_t j) {
            for (uint32_t q = Q.fwd_ptr[j]; q < Q.fwd_ptr[j + 1]; ++q)
                if (!applied[Q.fwd_items[2 * q + 1]]) return false;
            return true;
        };
        size_t pos = 0;
        while (pos < w.order.size()) {
            std::vector<uint32_t> R;
            for (uint32_t j = 0; j < nv; ++j)
                if (!pivoted[j] && ready(j)) R.push_back(j);
            for (uint32_t k : R) o += "        const double P" + S(k) + " = lane_value(Dm, " + S(k) + "); if (!(P" + S(k) + " > 0.0)) bad = true;\n";
            for (uint32_t k : R) o += "        const double D" + S(k) + " = sqrt(P" + S(k) + ");\n";
            for (uint32_t k : R)
                o += "        const double Y" + S(k) + " = lane_value(Vm, " + S(k) + ") / D" + S(k) + "; fin = fin && __builtin_isfinite(Y" + S(k) + ");\n";
            for (uint32_t k : R) {
                pivoted[k] = 1;
                if (!w.col[k].empty()) o += "        a" + S(k) + " = a" + S(k) + " / D" + S(k) + ";\n";
            }
            while (pos < w.order.size() && pivoted[w.order[pos]]) {
                const uint32_t k = w.order[pos++];
                applied[k] = 1;
                if (w.col[k].empty()) continue;
                const std::string K = S(k);
                for (size_t x = 0; x + 1 < w.col[k].size(); ++x) {
                    const std::string J = S(w.col[k][x].first);
                    o += "        a" + J + " -= a" + K + " * lane_value(a" + K + ", " + J + ");\n";
                }
                o += "        Dm -= a" + K + " * a" + K + "; Vm -= a" + K + " * Y" + K + ";\n";
            }
        }
    }
    // backward substitution: a column as soon as the rows of its entries are known, the ready columns' sums and divisions
    // side by side (every sum in its list's order)
    {
        std::vector<char> known(nv, 0);
        uint32_t left = nv;
        while (left) {
            std::vector<uint32_t> G;
            for (uint32_t k = nv; k-- > 0;) {
                if (known[k]) continue;
                bool ok = true;
                for (uint32_t i : w.bwd[k]) ok = ok && known[i];
                if (ok) G.push_back(k);
            }
            for (uint32_t k : G) {
                o += "        double X" + S(k) + " = Y" + S(k) + ";\n";
                for (uint32_t i : w.bwd[k]) o += "        X" + S(k) + " -= lane_value(a" + S(k) + ", " + S(i) + ") * X" + S(i) + ";\n";
            }
            for (uint32_t k : G) {
                o += "        X" + S(k) + " = X" + S(k) + " / D" + S(k) + "; d[" + S(k) + "] = X" + S(k) + "; dmax = ezpz::dev::fmax_abs(dmax, X" + S(k) + ");\n";
                known[k] = 1;
                --left;
            }
        }
    }
    o += "        (void)Vm; (void)Dm;\n        return bad;\n    }\n";
}

// ---- one wavefront per system (jit_kernel.hip.hpp, wave_kernel): the latency shape of a small system -------------------------------
// The lane class (same statements, same order) plus what spreads the sweeps and the assembly over 64 lanes: the constraint
// records as a table (lane ci evaluates constraint ci through a dispatch over the kinds present), the operand pairs of
// every assembled quantity -- D_v and V_v per variable, L_s per strict lower entry -- padded to the longest list (a padding
// pair multiplies the zero behind the Jacobian's slots), and the elimination as `tail` on the assembled quantities.
// Empty when a list is too long for a lane's registers (a hub variable of dozens of constraints): the lane kernel serves.
void emit_wave(std::string& o, const std::string& cls, const Class& cl, const EzpzConstraint* cs, bool unit_weights) {
    o.clear();
    const Program& Q = cl.Q;
    const uint32_t nv = Q.c.n_vars, m = Q.c.n_rows, zj = Q.c.zj, zlo = Q.c.zlo, nc = Q.c.n_cons;
    if (nc > 64 || m > 64 || nv > 64 || zj + 1 + m >= 0xFFFFu) return;
    auto S = [](uint32_t v) { return std::to_string(v); };
    const uint32_t nq = 2 * nv + zlo, nqr = (nq + 63) / 64;
    // operand pairs (a | b << 16): indices into A = [J slots | 0 | -r rows]
    std::vector<std::vector<uint32_t>> pairs(nq);
    for (uint32_t v = 0; v < nv; ++v)
        for (uint32_t q = Q.colj_ptr[v]; q < Q.colj_ptr[v + 1]; ++q) {
            const uint32_t slot = Q.colj_items[2 * q], row = Q.colj_items[2 * q + 1];
            pairs[v].push_back(slot | (slot << 16));
            pairs[nv + v].push_back(slot | ((zj + 1 + row) << 16));
        }
    for (uint32_t s2 = 0; s2 < zlo; ++s2)
        for (uint32_t q = Q.apair_ptr[s2]; q < Q.apair_ptr[s2 + 1]; ++q) pairs[2 * nv + s2].push_back(Q.apairs[2 * q] | (Q.apairs[2 * q + 1] << 16));
    uint32_t pmax = 1;
    for (const auto& p : pairs) pmax = std::max<uint32_t>(pmax, (uint32_t)p.size());
    if (pmax > 16 || nqr > 3) return;
    o = cls;
    // tables
    o += "namespace {\n__device__ const ezpz::DevCon kWaveCons[" + S(nc) + "] = {\n";
    for (uint32_t ci = 0; ci < nc; ++ci) {
        const DevCon& d = Q.cons[ci];
        o += "    {{";
        for (int i = 0; i < 8; ++i) o += (i ? ", " : "") + S(d.ids[i]);
        o += "}, " + hexf(cs[d.pos].param) + ", " + hexf(d.weight) + ", " + S(d.row0) + ", " + S(d.jbase) + ", " + S(d.pos) + ", " + S(d.kind) +
             ", " + S(d.tag) + ", " + S(d.nrows) + ", " + S(d.nslots) + ", {";
        for (int i = 0; i < 16; ++i) o += (i ? ", " : "") + S(d.jloc[i]);
        o += "}},\n";
    }
    o += "};\n__device__ const uint32_t kWavePairs[" + S(nqr * pmax * 64) + "] = {";
    const uint32_t pad = zj | (zj << 16);
    for (uint32_t k = 0; k < nqr; ++k)
        for (uint32_t p = 0; p < pmax; ++p)
            for (uint32_t l = 0; l < 64; ++l) {
                const uint32_t q = k * 64 + l;
                o += (((k * pmax + p) * 64 + l) % 16 == 0 ? "\n    " : " ") + S(q < nq && p < pairs[q].size() ? pairs[q][p] : pad) + "u,";
            }
    o += "\n};\n";
    // rows in request order (constraints by position, their rows in order), the caller's id of every internal variable
    std::vector<uint32_t> order(nc);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return Q.cons[a].pos < Q.cons[b].pos; });
    o += "__device__ const uint8_t kWaveVarOf[" + S(nv) + "] = {";
    for (uint32_t v = 0; v < nv; ++v) o += S(Q.var_of[v]) + ", ";
    o += "};\n";
    if (cl.linear) {
        o += "__device__ const double kWaveJ[" + S(std::max(zj, 1u)) + "] = {";
        for (uint32_t s2 = 0; s2 < zj; ++s2) o += hexf(cl.jconst[s2]) + ", ";
        o += "};\n";
    }
    emit_wave_row_table(o, cl);
    o += "}  // namespace\n\nstruct ClsW : Cls0 {\n";
    o += "    static constexpr int ZJ = " + S(zj) + ", NQ = " + S(nq) + ", PMAX = " + S(pmax) + ";\n";
    o += "    static __device__ __forceinline__ DevCon con(int ci) { return kWaveCons[ci]; }\n";
    o += "    static __device__ __forceinline__ uint32_t pair(int k, int p, int lane) { return kWavePairs[(k * PMAX + p) * 64 + lane]; }\n";
    // the sum of squares and maximum of a residual vector in request order, straight-line (literal LDS offsets)
    o += "    static __device__ __forceinline__ void sum_rows(const double* rows, double& sq, double& mx) {\n        double w;\n";
    for (uint32_t ci : order)
        for (uint32_t rr = 0; rr < Q.cons[ci].nrows; ++rr)
            o += "        w = rows[" + S(Q.cons[ci].row0 + rr) + "]; sq += w * w; mx = ezpz::dev::fmax_abs(mx, w);\n";
    o += "        (void)w; (void)rows;\n    }\n";
    o += "    static __device__ __forceinline__ uint32_t var_of(int v) { return kWaveVarOf[v]; }\n";
    o += std::string("    static __device__ __forceinline__ double jconst(int s) { return ") + (cl.linear ? "kWaveJ[s]" : "0.0") + "; }\n";
    // the kinds present, each with its evaluator instantiated for that kind alone
    std::vector<uint32_t> kinds;
    for (const DevCon& d : Q.cons)
        if (std::find(kinds.begin(), kinds.end(), (uint32_t)d.kind) == kinds.end()) kinds.push_back(d.kind);
    o += "    template <class XP>\n    static __device__ __forceinline__ bool residual_of(const DevCon& c, XP xs, double& r0, double& r1) {\n        switch (c.kind) {\n";
    for (uint32_t k : kinds)
        o += "        case " + S(k) + ": { DevCon k = c; k.kind = " + S(k) + "; return ezpz::dev::con_residual<LINEAR>(k, xs, r0, r1); }\n";
    o += "        default: r0 = 0.0; r1 = 0.0; return false;\n        }\n    }\n";
    o += "    template <class XP, class JP>\n    static __device__ __forceinline__ bool jacobian_of(const DevCon& c, XP xs, const ezpz::dev::JacWriter<JP>& w) {\n        switch (c.kind) {\n";
    for (uint32_t k : kinds)
        o += "        case " + S(k) + ": { DevCon k = c; k.kind = " + S(k) + "; return ezpz::dev::con_jacobian<false>(k, xs, w); }\n";
    o += "        default: return false;\n        }\n    }\n";
    const bool fastdiv = (fastdiv_mask() >> 2) & 1u;
    for (int fast = 1; fast >= 0; --fast) {
        o += std::string("    static __device__ __forceinline__ bool ") + (fast ? "tail" : "tail_exact") +
             "(const double* Q, double lambda, double* d, double& dmax" + (fast ? ", bool& ok" : "") + ") {\n        bool bad = false;\n";
        for (uint32_t v = 0; v < nv; ++v)
            o += "        double D" + S(v) + " = Q[" + S(v) + "] + lambda, V" + S(v) + " = Q[" + S(nv + v) + "];\n";
        for (uint32_t s2 = 0; s2 < zlo; ++s2) o += "        double L" + S(s2) + " = Q[" + S(2 * nv + s2) + "];\n";
        emit_elimination(o, cl, fast != 0, fastdiv);
        o += std::string("        (void)Q; (void)lambda;") + (fast ? " (void)ok;" : "") + "\n        return bad;\n    }\n";
    }
    emit_tail_wave(o, cl);
    o += "};\n\nextern \"C\" __global__ void __launch_bounds__(64) ezpz_jit_wave(const ezpz::jit::LaneArgs a) {\n";
    o += std::string("    ezpz::jit::wave_kernel<ClsW, ") + (unit_weights ? "true" : "false") + ">(a);\n}\n";
}

// ---- one lane per system (jit_kernel.hip.hpp, lane_kernel) --------------------------------------------------------------------------
bool lane_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, LanePlan& plan) {
    plan = LanePlan();
    // what a lane can hold in registers: x, the accepted x, r, r_next, d, the Jacobian values of one iteration, L
    if (n_cs == 0 || n_vars == 0 || n_vars > kPolicy.lane_max_vars || n_cs > kPolicy.lane_max_constraints) return false;
    Class cl;
    BuildError be;
    if (!build_program(cs, n_cs, n_vars, cl.Q, be, 1, false)) return false;
    const Program& Q = cl.Q;
    if (Q.c.zj + Q.c.zlo > 220 || Q.c.n_rows > 48) return false;
    plan.unit_weights = true;
    for (const DevCon& d : Q.cons) {
        cl.linear = cl.linear && kind_is_linear(d.kind);
        if (d.weight != 1.0) plan.unit_weights = false;
    }
    if (cl.linear) {
        cl.jconst.assign(Q.c.zj, 0.0);
        for (const DevCon& d : Q.cons) {
            double pd[8];
            const int np = linear_partials(d.kind, pd);
            for (int e = 0; e < np; ++e) {
                const uint32_t code = d.jloc[e];
                const uint32_t slot = d.jbase + (code & 0x7Fu);
                const double w = d.weight * pd[e];
                if (code & 0x80u)
                    cl.jconst[slot] = cl.jconst[slot] + w;
                else
                    cl.jconst[slot] = w;
            }
        }
    }
    cl.H.ninst_pad = 1;
    std::string cls = "#include \"jit_kernel.hip.hpp\"\nusing ezpz::DevCon;\n\n";
    emit_class(cls, 0, cl, true, cs);
    std::string& o = plan.jit_source;
    o = cls;
    for (int one = 0; one < 2; ++one) {  // (batches; `_one`: one-call launches that stay resident, jit_kernel.hip.hpp)
        o += std::string("extern \"C\" __global__ void __launch_bounds__(") + "256" + ") ezpz_jit_lane" + (one ? "_one" : "") + "(const ezpz::jit::LaneArgs a) {\n";
        o += std::string("    ezpz::jit::lane_kernel<Cls0, ") + (plan.unit_weights ? "true" : "false") + ", " + (one ? "true" : "false") + ">(a);\n}\n";
    }
    plan.n_vars = (uint32_t)n_vars;
    plan.n_cons = (uint32_t)n_cs;
    plan.n_rows = Q.c.n_rows;
    emit_wave(plan.wave_source, cls, cl, cs, plan.unit_weights);
    return true;
}

// ---- lanes across the batch (batch_kernel.hip.hpp) ------------------------------------------------------------------------------------
bool batch_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, BatchPlan& plan) {
    plan = BatchPlan();
    if (n_cs == 0 || n_vars == 0 || n_vars > 60000 || n_cs > 60000) return false;
    Class cl;
    BuildError be;
    if (!build_program(cs, n_cs, n_vars, cl.Q, be, 1, false, true)) return false;
    const Program& Q = cl.Q;
    if (Q.c.n_parts != 1 || Q.c.zj >= 0xFFFF || Q.c.zlo >= 0xFFFF || Q.c.n_rows >= 0xFFFF) return false;
    plan.unit_weights = true;
    for (const DevCon& d : Q.cons)
        if (d.weight != 1.0) plan.unit_weights = false;
    cl.linear = false;  // the general records: Jacobian values are stored (no constant folding here)
    emit_class_program(plan.blob, cl, false, cs, true, true);
    align4(plan.blob);
    plan.var_off = (uint32_t)plan.blob.size();
    plan.blob.insert(plan.blob.end(), Q.var_of.begin(), Q.var_of.end());
    plan.blob.resize(plan.blob.size() + 16, 0);
    // ... and its inverse: the state row of the caller's variable c (the rows of a batch are moved in tiles of eight of the
    // caller's consecutive variables, batch_kernel.hip.hpp)
    plan.inv_off = (uint32_t)plan.blob.size();
    std::vector<uint32_t> inv(Q.var_of.size(), 0);
    for (uint32_t k = 0; k < (uint32_t)Q.var_of.size(); ++k) inv[Q.var_of[k]] = k;
    plan.blob.insert(plan.blob.end(), inv.begin(), inv.end());
    plan.blob.resize(plan.blob.size() + 16, 0);
    plan.nv = Q.c.n_vars, plan.m = Q.c.n_rows, plan.zj = Q.c.zj, plan.zlo = Q.c.zlo, plan.ncons = Q.c.n_cons;
    plan.n_ops = cl.H.n_ops, plan.ops_off = cl.H.ops_off, plan.cons_off = cl.H.cons_off;
    plan.o_d = plan.nv;
    plan.o_r = 2 * plan.nv;
    plan.o_rn = plan.o_r + plan.m;
    plan.o_j = plan.o_rn + plan.m;
    plan.o_dg = plan.o_j + plan.zj;
    plan.o_l = plan.o_dg + plan.nv;
    plan.rows = plan.o_l + plan.zlo;
    if (debug_topic("lanes")) {  // the operation stream by kind: records and items (an item is two loads)
        uint64_t recs[8] = {}, items[8] = {};
        for (uint32_t io = 0; io < plan.n_ops; ++io) {
            const uint32_t w = plan.blob[plan.ops_off + io * kCompRecWords];
            recs[w & 7u] += 1;
            items[w & 7u] += (w >> 8) & 0xFFu;
        }
        std::fprintf(stderr, "[ezpz lanes] nv %u m %u zj %u zlo %u rows %u ops %u |", plan.nv, plan.m, plan.zj, plan.zlo, plan.rows, plan.n_ops);
        const char* names[] = {"DIAG", "OFF", "COL", "SLOT", "BWD", "DIAGCOL", "SLOTA"};
        const uint32_t codes[] = {COMP_DIAG, COMP_OFF, COMP_COL, COMP_SLOT, COMP_BWD, COMP_DIAGCOL, COMP_SLOTA};
        for (int k = 0; k < 7; ++k)
            std::fprintf(stderr, " %s %llu recs %llu items", names[k], (unsigned long long)recs[codes[k] & 7u], (unsigned long long)items[codes[k] & 7u]);
        std::fputc('\n', stderr);
    }
    return true;
}

}  // namespace ezpz
