// Lanes across the batch: one LANE per system, for batches of one connected sketch too large for a lane's registers.
//
// The list-walk kernels (lm_kernel.hip.hpp) give a connected sketch of 50-5000 variables a wavefront or a workgroup and
// schedule its sparse Cholesky level by level: most lanes wait most of the time (a level is a few columns), which is
// the right trade for the latency of ONE solve and the wrong one for a batch.  Every system of a batch shares the
// topology, so here the 64 lanes of a wavefront are 64 systems running the SAME program -- the class program of
// comp_program.cpp built for the whole system (constraint records with their parameters, the operation stream of the
// linear solve) read through the scalar unit, every branch on it a scalar branch -- each lane strictly sequentially
// on its own system.  No lane ever waits for another: no barriers, no reductions (LDS only to transpose the rows of the
// batch on their way in and out).  The state of a wavefront's
// 64 systems lives in global memory as rows of 64 doubles (row r, lane l = word 64 r + l): every access is one
// coalesced 512-byte line, served by L2 / the Infinity Cache / HBM -- this kernel trades the on-chip residency of the
// list-walk kernels for full lanes, and is bound by that traffic.
// Every lane runs the whole LM loop of newton.rs:29-145 with its own lambda / accept / iteration count (masked
// bookkeeping; all lanes execute the same record streams); lanes whose system is done wait for a refill, which the
// wavefront does when a third of its lanes are idle (a refill costs everyone an evaluation sweep).
// Sums run in request order on one lane (the reference's own order); elimination order and operation order are those
// of the symbolic phase, as everywhere.  Replaces the same reference code as lm_kernel.hip.hpp.
#pragma once
#include <hip/hip_runtime.h>

#include "comp_kernel.hip.hpp"

namespace ezpz {

struct BatchArgs {
    const uint32_t* prog;  // the plan's blob
    uint32_t nv, m, zj, zlo, ncons, n_ops, ops_off, cons_off, var_off, inv_off;
    uint32_t o_d, o_r, o_rn, o_j, o_dg, o_l, rows;  // rows of a wavefront's workspace: x at 0, b -> y -> d, r, r_next, J, diagonal / tentative x, L
    uint32_t n_cons;                                // unsat mask row
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;  // optional
    uint64_t* warn_log;   // optional
    uint32_t warn_cap, max_iterations, unit_weights, refill_lanes;  // refill_lanes: idle lanes of a wavefront that trigger a refill
    uint64_t batch;
    double residual_tolerance, step_tolerance, initial_lambda;
    double* ws;  // workspaces: one per wavefront of the launch, `rows` x 64 doubles each
    // Stragglers (optional): once a wavefront has no system left to take and at most `strag_lanes` of its lanes are still
    // working, those lanes hand their systems over -- the system's index goes on this list, its current values to x_out, its
    // LM state (lambda, iteration and pass numbers, warnings so far) to `strag_state`, and the lane drops it; the per-system
    // teams RESUME the listed systems after this kernel (launch.hip; lm_kernel.hip.hpp: LmResume) -- a straggler is typically
    // one or two iterations from done, and solving it again from its guesses cost the teams nine.
    uint32_t* strag_list;
    uint32_t* strag_count;
    uint32_t strag_cap, strag_lanes;
    LmResume* strag_state;  // entry i: the LM state of the system listed at i (its current values go to x_out)
};

__global__ void __launch_bounds__(256, 4) batch_lane_kernel(const BatchArgs a) {
    using namespace dev;
    const int lane = threadIdx.x & 63;
    const uint64_t wave_global = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const comp_cptr prog = (comp_cptr)(uintptr_t)a.prog;
    const RowRef W{a.ws + wave_global * (uint64_t)a.rows * 64 + lane};
    const RowRef V = W + a.o_d, J = W + a.o_j, L = W + a.o_l;
    // x and the diagonal / tentative x, r and r_next: two row sets each whose roles a lane swaps when it accepts a step
    // (a pointer per lane instead of copying n + m rows through memory)
    RowRef P = W, S = W + a.o_dg, R = W + a.o_r, RN = W + a.o_rn;

    uint64_t next = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool have = false, fresh = false, r_is_at_x = true;
    uint64_t sys = 0;
    double residual_sq = 0.0, largest = 0.0, lambda = 0.0;
    uint32_t it = 0, pass = 0, pass_jac = 0, nwarn = 0;

    auto log_warning = [&](uint32_t p, uint32_t pos) {  // Warning::Degenerate, in evaluation order (solver.rs:340-346)
        if (a.warn_log && nwarn < a.warn_cap) a.warn_log[sys * a.warn_cap + nwarn] = ((uint64_t)p << 32) | pos;
        ++nwarn;
    };
    // One sweep over the constraint records (request order).  MODE 0: residuals of `xs` into rows `dst` for the lanes in
    // `store`, sum of squares / maximum per lane, degenerate evaluations logged with pass `p` for the lanes in `log`;
    // MODE 2: the unsatisfied check on the unweighted values (lib.rs:305-327) for the lanes in `store`.
    auto residual_sweep = [&](const RowRef& xs, const RowRef& dst, bool store, bool log, uint32_t p, int MODE, double& sq, double& mx,
                              double& unsat) {
        // (the next constraint's record is requested while this one is evaluated: a pad record follows the last)
        CompRec8 na = comp_load8(prog + a.cons_off), nb = comp_load8(prog + a.cons_off + 8);
        for (uint32_t ci = 0; ci < a.ncons; ++ci) {
            const CompRec8 ra = na, rb = nb;
            na = comp_load8(prog + a.cons_off + (ci + 1) * kCompConWords);
            nb = comp_load8(prog + a.cons_off + (ci + 1) * kCompConWords + 8);
            const DevCon c = comp_make_con(ra, rb, __hiloint2double((int)rb.w[6], (int)rb.w[5]));
            const uint32_t pos = rb.w[7];
            double r0, r1;
            const bool deg = con_residual<false>(c, xs, r0, r1);
            if (MODE == 2) {
                bool sat = fabs(r0) < EPS;
                if (c.nrows > 1) sat = sat && (fabs(r1) < EPS);
                if (store) {
                    if (!sat) unsat += 1.0;
                    if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + pos] = sat ? 0 : 1;
                }
                continue;
            }
            const double w0 = c.weight * r0;  // solver.rs:353
            if (store) dst[c.row0] = w0;
            sq += w0 * w0;
            mx = fmax(mx, fabs(w0));
            if (c.nrows > 1) {
                const double w1 = c.weight * r1;
                if (store) dst[c.row0 + 1] = w1;
                sq += w1 * w1;
                mx = fmax(mx, fabs(w1));
            }
            if (deg && log) log_warning(p, pos);
        }
    };
    // Jacobian sweep at the accepted values (eval() and accepted steps, newton.rs:121; solver.rs:359-440) for the lanes
    // in `mask`: their J rows, their warnings.
    auto jacobian_sweep = [&](bool mask) {
        CompRec8 na = comp_load8(prog + a.cons_off), nb = comp_load8(prog + a.cons_off + 8);
        for (uint32_t ci = 0; ci < a.ncons; ++ci) {
            const CompRec8 ra = na, rb = nb;
            na = comp_load8(prog + a.cons_off + (ci + 1) * kCompConWords);
            nb = comp_load8(prog + a.cons_off + (ci + 1) * kCompConWords + 8);
            const DevCon c = comp_make_con(ra, rb, __hiloint2double((int)rb.w[6], (int)rb.w[5]));
            if (mask) {
                JacWriter<RowRef> w;
                w.jv = J;
                w.jbase = c.jbase;
                w.loc[0] = ra.w[6];
                w.loc[1] = ra.w[7];
                w.loc[2] = rb.w[0];
                w.loc[3] = rb.w[1];
                w.weight = c.weight;
                if (con_jacobian<false>(c, P, w)) log_warning(pass_jac, rb.w[7]);
            }
        }
    };

    // Rows of the batch move between the caller's AoS layout and the lanes' rows in tiles of eight of the caller's
    // consecutive variables: a group of eight lanes moves the 64 contiguous bytes of ONE system (whole sectors -- a lane
    // reading its own system's row alone touched 8 bytes of every 64 it fetched, and the 300 reads of a 300-variable
    // system were a fifth of the batch's first round), and the wavefront transposes the tile through its own 4.6 KB of LDS
    // (row pitch 9 doubles: conflict-free both ways; wave-local, in order, no barrier).
    __shared__ double tiles[4][64 * 9];
    double* tile = tiles[threadIdx.x >> 6];
    const int sub = lane & 7, grp = lane >> 3;
    // `mine`: this lane takes part (its system's row is read / written); rows in: x0 -> P, rows out: P -> x_out
    auto move_rows = [&](bool mine, bool in) {
        uint64_t sj[8];
        bool mj[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            sj[t] = __shfl(sys, 8 * t + grp, 64);
            mj[t] = __shfl((int)mine, 8 * t + grp, 64) != 0;
        }
        for (uint32_t c0 = 0; c0 < a.nv; c0 += 8) {
            const bool col = c0 + sub < a.nv;
            if (in) {
                double v[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) v[t] = (mj[t] && col) ? a.x0[sj[t] * a.nv + c0 + sub] : 0.0;
#pragma unroll
                for (int t = 0; t < 8; ++t) tile[(8 * t + grp) * 9 + sub] = v[t];
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i)
                    if (c0 + i < a.nv && mine) P[prog[a.inv_off + c0 + i]] = tile[lane * 9 + i];
            } else {
                double v[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) v[i] = (c0 + i < a.nv && mine) ? P[prog[a.inv_off + c0 + i]] : 0.0;
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) tile[lane * 9 + i] = v[i];
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (mj[t] && col) a.x_out[sj[t] * a.nv + c0 + sub] = tile[(8 * t + grp) * 9 + sub];
            }
        }
    };

    for (;;) {
        // ---- refill: idle lanes take the next systems of the batch; eval() for them (newton.rs:45, :232-236) -------------
        const bool want = !have && next < a.batch;
        const unsigned long long wanting = __ballot(want), busy = __ballot(have);
        if (wanting && (!busy || (uint32_t)__popcll(wanting) >= a.refill_lanes)) {
            if (want) {
                sys = next;
                next += stride;
                nwarn = 0;
            }
            move_rows(want, true);
            double sq = 0.0, mx = __builtin_nan(""), none = 0.0;
            residual_sweep(P, R, want, want, 0, 0, sq, mx, none);
            if (want) {
                have = true;
                residual_sq = sq;
                largest = mx;
                lambda = a.initial_lambda;
                it = 0;
                pass = 2;
                pass_jac = 1;
                fresh = true;
                r_is_at_x = true;
            }
        }
        if (!__any(have)) break;

        // ---- one LM iteration for every lane that has a system (newton.rs:47-139) -----------------------------------------------
        bool finish = false;
        uint32_t iterations = a.max_iterations, converged = 0;
        if (have) {
            if (it >= a.max_iterations) {  // newton.rs:141-144
                finish = true;
            } else if (largest <= a.residual_tolerance) {  // newton.rs:50-60
                iterations = it;
                converged = 1;
                finish = true;
            }
        }
        const bool active = have && !finish;
        // the Jacobian at the accepted values: for the lanes that go on (newton.rs:121) and, in the same sweep, for the lanes
        // that have just converged and still owe the refresh of their last accepted step its warnings (their J rows are
        // written for nothing; a second, mostly masked sweep over the records cost every round in which a lane finished)
        if (__any(have && fresh)) jacobian_sweep(have && fresh);
        if (have) fresh = false;
        if (__any(active)) {
            // -- normal equations, Cholesky, substitutions (newton.rs:73-102): the operation stream, every lane on its own system.
            //    The operands of a record's items are all requested before the first is used: one trip to L2 / HBM per
            //    record instead of one per item; the terms are still folded in list order.  The stream runs column by column
            //    (a column, then the slots below it while d_j is still in a register).  (Requesting the operands of
            //    record i + 1 before record i is computed -- a software pipeline, with the host flagging the one hazard of
            //    the fused order -- was measured and not kept: 7.0 -> 6.7 M solves/s at 262 144 systems of 300 variables,
            //    +2 % at 16 384; the number of loads in flight is not known statically (records hold 0-6 items), so the
            //    compiler waits for all of them before the first use.)
#define LOAD_ITEMS(A, B)                                                              \
    _Pragma("unroll") for (uint32_t k = 0; k < kCompItemsGen; ++k) {                  \
        va[k] = vb[k] = 0.0;                                                          \
        if (k < ni) va[k] = A[rec.w[2 + k] & 0xFFFFu], vb[k] = B[rec.w[2 + k] >> 16]; \
    }
            double acc = 0.0, y = 0.0, dmax = __builtin_nan(""), dcur = 0.0;  // dcur: d_j of the column whose slots follow
            bool bad = false;
            CompRec8 rec = comp_load8(prog + a.ops_off);
            for (uint32_t io = 0; io < a.n_ops; ++io) {
                const CompRec8 nxt = comp_load8(prog + a.ops_off + (io + 1) * kCompRecWords);
                const uint32_t op = rec.w[0] & 0xFFu, ni = (rec.w[0] >> 8) & 0xFFu;
                const bool first = (rec.w[0] & kCompFirst) != 0, last = (rec.w[0] & kCompLast) != 0;
                // (the stream is fused: an entry of JtJ + lambda I is assembled right before its column / slot is eliminated
                // and stays in the accumulators -- no store and reload of the entry's row)
                const bool keep = (rec.w[0] & kCompKeep) != 0, cont = (rec.w[0] & kCompCont) != 0;
                const uint32_t oa = rec.w[1] & 0xFFFFu, ob = rec.w[1] >> 16;
                double va[kCompItemsGen], vb[kCompItemsGen];
                switch (op) {
                case COMP_DIAG:
                    if (first) acc = 0.0, y = 0.0;
                    LOAD_ITEMS(J, R)
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < ni) {
                            acc += va[k] * va[k];
                            y += va[k] * -vb[k];
                        }
                    if (last) {
                        acc = acc + lambda;  // newton.rs:77-84
                        if (!keep) {
                            S[oa] = acc;
                            V[oa] = y;
                        }
                    }
                    break;
                case COMP_OFF:
                    if (first) acc = 0.0;
                    LOAD_ITEMS(J, J)
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < ni) acc += va[k] * vb[k];
                    if (last && !keep) L[oa] = acc;
                    break;
                case COMP_COL:
                    if (first && !cont) {
                        acc = S[oa];
                        y = V[oa];
                    }
                    LOAD_ITEMS(L, V)
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < ni) {
                            acc -= va[k] * va[k];
                            y -= va[k] * vb[k];
                        }
                    if (last) {
                        if (!(acc > 0.0)) bad = true;  // LltError::Numeric: non-positive pivot (newton.rs:93-99)
                        const double dv = sqrt(acc);
                        dcur = dv;
                        S[oa] = dv;
                        V[oa] = y / dv;
                    }
                    break;
                case COMP_SLOT:
                    if (first && !cont) acc = L[oa];
                    LOAD_ITEMS(L, L)
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < ni) acc -= va[k] * vb[k];
                    if (last) L[oa] = acc / ((rec.w[0] & kCompDivReg) ? dcur : S[ob]);
                    break;
                case COMP_DIAGCOL: {  // a column whose assembly and elimination fit one record: one round trip instead of two
                    const uint32_t n0 = rec.w[0] >> 24;
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k) {
                        va[k] = vb[k] = 0.0;
                        if (k < ni) {
                            const uint32_t ia = rec.w[2 + k] & 0xFFFFu, ib = rec.w[2 + k] >> 16;
                            va[k] = k < n0 ? J[ia] : L[ia];
                            vb[k] = k < n0 ? R[ib] : V[ib];
                        }
                    }
                    acc = 0.0, y = 0.0;
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < n0) {
                            acc += va[k] * va[k];
                            y += va[k] * -vb[k];
                        }
                    acc = acc + lambda;  // newton.rs:77-84
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k >= n0 && k < ni) {
                            acc -= va[k] * va[k];
                            y -= va[k] * vb[k];
                        }
                    if (!(acc > 0.0)) bad = true;  // LltError::Numeric: non-positive pivot (newton.rs:93-99)
                    const double dv = sqrt(acc);
                    dcur = dv;
                    S[oa] = dv;
                    V[oa] = y / dv;
                    break;
                }
                case COMP_SLOTA: {  // likewise a slot: its entry of JtJ, the update, the division by d_j (in `dcur`)
                    const uint32_t n0 = rec.w[0] >> 24;
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k) {
                        va[k] = vb[k] = 0.0;
                        if (k < ni) {
                            const uint32_t ia = rec.w[2 + k] & 0xFFFFu, ib = rec.w[2 + k] >> 16;
                            va[k] = k < n0 ? J[ia] : L[ia];
                            vb[k] = k < n0 ? J[ib] : L[ib];
                        }
                    }
                    acc = 0.0;
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < n0) acc += va[k] * vb[k];
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k >= n0 && k < ni) acc -= va[k] * vb[k];
                    L[oa] = acc / dcur;
                    break;
                }
                case COMP_BWD:
                    if (first) acc = V[oa];
                    LOAD_ITEMS(L, V)
#pragma unroll
                    for (uint32_t k = 0; k < kCompItemsGen; ++k)
                        if (k < ni) acc -= va[k] * vb[k];
                    if (last) {
                        const double dval = acc / S[oa];
                        V[oa] = dval;
                        dmax = fmax(dmax, fabs(dval));
                        S[oa] = P[oa] + dval;  // the tentative x (newton.rs:111-114); the diagonal entry is dead now
                    }
                    break;
                default: break;
                }
                rec = nxt;
            }
#undef LOAD_ITEMS
            // -- residual at the tentative values (newton.rs:115-116); a lane whose factorisation failed skips it like the
            //    reference's `continue` (no warnings, no step)
            const bool stepping = active && !bad;
            double sq = 0.0, mx = __builtin_nan(""), none = 0.0;
            residual_sweep(S, RN, stepping, stepping, pass, 0, sq, mx, none);
            if (active && bad) {
                lambda *= LM_LAMBDA_INCR;
                ++it;
            }
            const bool accept = stepping && sq < residual_sq;  // strict, newton.rs:118
            if (stepping) {
                ++pass;
                // accept: x = x + d -- the tentative values' rows become x, r_next's rows become r; reject: x += d, x -= d like
                // the reference (newton.rs:111-114,:124-131), not a copy
                if (__any(stepping && !accept)) {
                    for (uint32_t k = 0; k < a.nv; ++k)
                        if (!accept) P[k] = S[k] - V[k];
                }
                if (accept) {
                    const RowRef tx = P, tr = R;
                    P = S, S = tx;
                    R = RN, RN = tr;
                    lambda *= LM_LAMBDA_DECR;
                    residual_sq = sq;
                    largest = mx;
                    r_is_at_x = true;
                    fresh = true;  // newton.rs:121: refresh_jacobian, evaluated at the top of the next iteration (or at the end)
                    pass_jac = pass++;
                } else {
                    lambda *= LM_LAMBDA_INCR;
                    r_is_at_x = false;
                }
                if (dmax <= a.step_tolerance) {  // newton.rs:134-139
                    iterations = it;
                    converged = 1;
                    finish = true;
                } else {
                    ++it;
                }
            }
        }

        // ---- systems that are done: the refresh their last accepted step still owes its warnings, the unsatisfied check
        //      (lib.rs:305-327, :358-370), write-back ---------------------------------------------------------------------------------
        if (__any(finish)) {
            if (__any(finish && fresh)) jacobian_sweep(finish && fresh);  // (a step-size stop right after an accepted step)
            const bool use_r = r_is_at_x && a.unit_weights != 0;
            const bool all_sat = use_r && largest < EPS && !isnan(residual_sq);
            double unsat = 0.0;
            if (__any(finish && !all_sat && !use_r)) {
                double s_ = 0.0, m_ = 0.0;
                residual_sweep(P, R, finish && !use_r, false, 0, 2, s_, m_, unsat);
            }
            if (__any(finish && (all_sat || use_r))) {
                for (uint32_t ci = 0; ci < a.ncons; ++ci) {
                    const uint32_t w0 = prog[a.cons_off + ci * kCompConWords], w1 = prog[a.cons_off + ci * kCompConWords + 1];
                    const uint32_t pos = prog[a.cons_off + ci * kCompConWords + 15];
                    const uint32_t row = w1 & 0xFFFFu;
                    if (finish && use_r) {
                        bool sat = all_sat;
                        if (!all_sat) {
                            sat = fabs(R[row]) < EPS;
                            if (((w0 >> 16) & 0xFFu) > 1) sat = sat && (fabs(R[row + 1]) < EPS);
                            if (!sat) unsat += 1.0;
                        }
                        if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + pos] = sat ? 0 : 1;
                    }
                }
            }
            move_rows(finish, false);
            if (finish) {
                EzpzStatus st;
                st.iterations = iterations;
                st.converged = converged;
                st.n_unsatisfied = (uint32_t)unsat;
                st.n_warnings = nwarn;
                st.final_residual_inf = (a.m > 0) ? largest : 0.0;
                st.final_lambda = lambda;
                a.status[sys] = st;
                have = false;
                fresh = false;
            }
        }

        // ---- stragglers: a round of LM iterations costs a wavefront the same whether 64 lanes work or one (every access is
        //      a trip to memory; ~1.5 ms per round for a 300-variable sketch on an otherwise idle device), and at one system
        //      per lane the last 0.1 % of a jittered batch took 11 rounds alone.  A wavefront that cannot refill and is down
        //      to a few lanes hands them to the per-system teams, which solve one such system in a fraction of a round.
        if (a.strag_list) {
            const unsigned long long working = __ballot(have);
            if (working && (uint32_t)__popcll(working) <= a.strag_lanes && !__any(next < a.batch)) {
                bool handing = false;
                if (have) {
                    const uint32_t idx = atomicAdd(a.strag_count, 1u);
                    if (idx < a.strag_cap) {  // (a full list: this lane simply carries on)
                        a.strag_list[idx] = (uint32_t)sys;
                        LmResume rec;
                        rec.lambda = lambda;
                        rec.it = it;
                        rec.pass = pass;
                        rec.nwarn = nwarn;
                        rec.jac_pass = fresh ? pass_jac : kNoPass;
                        a.strag_state[idx] = rec;
                        handing = true;
                    }
                }
                move_rows(handing, false);  // the current values: the teams' starting point
                if (handing) {
                    have = false;
                    fresh = false;
                }
            }
        }
    }
}

}  // namespace ezpz
