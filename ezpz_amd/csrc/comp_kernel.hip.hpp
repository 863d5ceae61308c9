// Component-resident Levenberg-Marquardt kernel: one LANE per connected component of the constraint system.
//
// For systems that are many small independent components in a few isomorphism classes (comp_program.hpp; the
// 2000 x 2000 massive_parallel_system is 1500 components in 2 classes) a workgroup owns one system and each of its
// wavefronts a few CHUNKS: up to 64 instances of one class, one per lane.  All 64 lanes of a chunk run the same class
// program, so
//   * the program (constraint records, the operation stream of the linear solve) is read through the scalar unit as
//     8/16-word records and every branch on it is a scalar branch: no per-lane index lists, no divergence;
//   * a lane's state lives in LDS rows of 64 doubles (row r, lane l = word 64 r + l: conflict-free), addressed as
//     lane base + uniform row offset; the normal equations, the Cholesky factor and the substitutions of a component
//     run start to finish on its lane with no synchronisation at all;
//   * the only cross-lane traffic is what the reference's global LM control needs (newton.rs:50-60,:96-139): ONE
//     workgroup rendezvous per LM iteration reducing {pivot failed, max |d|, sum r_next^2, max |r_next|}.  The
//     tentative residual is evaluated before the rendezvous (at x + d, from scratch rows); x itself moves only after
//     the reduction has shown that every pivot of the system was positive, so a failed factorisation leaves it
//     untouched exactly like the reference's `continue`.  Degenerate warnings of that speculative sweep are kept as a
//     per-lane bit mask and committed after the rendezvous.
// Same arithmetic as the list-walk kernels (lm_kernel.hip.hpp): the class programs come from the same symbolic phase,
// every sum runs in the same order, so per-component results are bit-identical; -ffp-contract=off.
// Replaces Model::solve_levenberg_marquardt (reference ezpz/src/solver/newton.rs:29-145), Model::residual /
// refresh_jacobian (solver.rs:318-440), the faer calls of newton.rs:73-102 and the unsatisfied check of
// lib.rs:305-327 for a batch of systems sharing the topology.
#pragma once
#include <hip/hip_runtime.h>

#include "comp_program.hpp"
#include "constraint_eval.hip.hpp"
#include "wave_ops.hip.hpp"

namespace ezpz {

struct CompArgs {
    const uint32_t* prog;  // the plan's blob
    uint32_t o_waves, o_chunks;
    uint32_t n_row;   // values per system in x0 / x_out
    uint32_t n_cons;  // constraints per system (unsat mask row)
    uint32_t n_rows_total;
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;  // optional
    uint64_t* warn_log;   // optional
    uint32_t warn_cap;
    uint64_t batch;
    uint32_t max_iterations;
    uint32_t unit_weights;
    double residual_tolerance, step_tolerance, initial_lambda;
    uint32_t scratch_row0, scratch_rows;  // LDS rows: first scratch row, rows per wavefront
    uint32_t red_row0;                    // LDS row of the reduction scratch (2 x 3 x 16 doubles), flag words, warning counters
    DoneWord done;                        // one-call launches: the completion word (dev_types.hpp)
};

namespace dev {

typedef const uint32_t __attribute__((address_space(4)))* comp_cptr;  // constant address space: scalar loads

struct CompRec8 {
    uint32_t w[8];
};
__device__ __forceinline__ CompRec8 comp_load8(comp_cptr p) {
    CompRec8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.w[i] = p[i];
    return r;
}

// A lane's rows of 64 doubles: element e of this lane is p[64 e].
struct RowRef {
    double* p;
    __device__ __forceinline__ double& operator[](uint32_t e) const { return p[(size_t)e * 64]; }
    __device__ __forceinline__ RowRef operator+(uint32_t rows) const { return RowRef{p + (size_t)rows * 64}; }
};

__device__ __forceinline__ double comp_uniform(double v) {  // a value every lane holds -> scalar registers
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)u);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// Workgroup reductions with ONE barrier: every wavefront folds its lanes into its last lane by DPP and parks the
// partials; after the barrier every wavefront folds the <= 16 partials in a fixed DPP tree, so all lanes of the
// workgroup end with the same bits, in scalar registers.  The scratch alternates between two halves (`flip`), so no
// trailing barrier is needed.  "Did any pivot fail" travels as a flag that the wavefronts OR into one of three words
// (ballot + one LDS atomic per wavefront); word k+1 is cleared while word k is in use.
struct CompRed {
    double* buf;  // 2 x 3 x 16 doubles
    int* flags;   // 3 words
    int flip;
    int turn;     // flag word of the next reduction
    __device__ __forceinline__ void sum_max(double& s0, double& m1, int lane, uint32_t wave, uint32_t nwaves) {
        s0 = reduce_wave_to_last_lane(s0, OpSum());
        m1 = reduce_wave_to_last_lane(m1, OpMax());
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        if (lane == 63) {
            b[wave] = s0;
            b[16 + wave] = m1;
        }
        __syncthreads();
        const bool in = (uint32_t)lane < nwaves;
        const int l = lane & 15;
        s0 = comp_uniform(reduce_lanes<16>(in ? b[l] : 0.0, OpSum()));
        m1 = comp_uniform(reduce_lanes<16>(in ? b[16 + l] : __builtin_nan(""), OpMax()));
    }
    // sum, max, max and the OR of `flag` over the workgroup
    __device__ __forceinline__ bool step(double& s0, double& m1, double& m2, bool flag, int lane, uint32_t wave, uint32_t nwaves) {
        s0 = reduce_wave_to_last_lane(s0, OpSum());
        m1 = reduce_wave_to_last_lane(m1, OpMax());
        m2 = reduce_wave_to_last_lane(m2, OpMax());
        double* b = buf + (flip ? 48 : 0);
        flip ^= 1;
        int* f = flags + turn;
        const int next = turn == 2 ? 0 : turn + 1;
        const bool any = __ballot(flag) != 0;
        if (lane == 63) {
            b[wave] = s0;
            b[16 + wave] = m1;
            b[32 + wave] = m2;
            if (any) atomicOr(f, 1);
            if (wave == 0) flags[next] = 0;  // last read two reductions ago, next set after this barrier
        }
        turn = next;
        __syncthreads();
        const bool in = (uint32_t)lane < nwaves;
        const int l = lane & 15;
        s0 = comp_uniform(reduce_lanes<16>(in ? b[l] : 0.0, OpSum()));
        m1 = comp_uniform(reduce_lanes<16>(in ? b[16 + l] : __builtin_nan(""), OpMax()));
        m2 = comp_uniform(reduce_lanes<16>(in ? b[32 + l] : __builtin_nan(""), OpMax()));
        return __builtin_amdgcn_readfirstlane(*f) != 0;
    }
};

// The chunk descriptor (CompChunk, 32 words) in scalar registers.
struct CompChunkRegs {
    uint32_t count, nv, m, ncons, n_ops, ops_off, cons_off, row0;
    uint32_t o_d, o_r0, o_r1, o_j, o_wm, s_l, stride, ids_off, par_off, pos_off;
};
__device__ __forceinline__ CompChunkRegs comp_load_chunk(comp_cptr p) {
    const CompRec8 h0 = comp_load8(p), h1 = comp_load8(p + 8), h2 = comp_load8(p + 16);
    CompChunkRegs c;
    c.count = h0.w[0], c.nv = h0.w[1], c.m = h0.w[2], c.ncons = h0.w[3];
    c.n_ops = h0.w[4], c.ops_off = h0.w[5], c.cons_off = h0.w[6], c.row0 = h0.w[7];
    c.o_d = h1.w[0], c.o_r0 = h1.w[1], c.o_r1 = h1.w[2], c.o_j = h1.w[3], c.o_wm = h1.w[4];
    c.s_l = h1.w[5], c.stride = h1.w[6], c.ids_off = h1.w[7];
    c.par_off = h2.w[0], c.pos_off = h2.w[1];
    return c;
}

// A class's constraint record (16 words, uniform) as the evaluators' DevCon; param is the lane's own.
__device__ __forceinline__ DevCon comp_make_con(const CompRec8& a, const CompRec8& b, double param) {
    DevCon c;
    c.kind = (uint8_t)(a.w[0] & 0xFFu);
    c.tag = (uint8_t)((a.w[0] >> 8) & 0xFFu);
    c.nrows = (uint8_t)((a.w[0] >> 16) & 0xFFu);
    c.nslots = (uint8_t)(a.w[0] >> 24);
    c.row0 = a.w[1] & 0xFFFFu;
    c.jbase = a.w[1] >> 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        c.ids[2 * e] = a.w[2 + e] & 0xFFFFu;
        c.ids[2 * e + 1] = a.w[2 + e] >> 16;
    }
    c.param = param;
    c.weight = __hiloint2double((int)b.w[3], (int)b.w[2]);
    c.pos = b.w[4];
    return c;
}

}  // namespace dev

// LIN: every class is linear with constant Jacobian (no Jacobian storage, no warnings);
// otherwise the general build (all 25 kinds, Jacobian values in LDS).  Up to 8 wavefronts per workgroup, two
// workgroups per CU when the state allows it.
template <bool LIN>
__global__ void __launch_bounds__(512) comp_solve_kernel(const CompArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using namespace dev;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane((uint32_t)tid >> 6);
    const uint32_t nwaves = (uint32_t)blockDim.x >> 6;
    const comp_cptr prog = (comp_cptr)(uintptr_t)a.prog;
    const uint32_t ch0 = prog[a.o_waves + wave], ch1 = prog[a.o_waves + wave + 1];
    CompRed red;
    red.buf = smem + (size_t)a.red_row0 * 64;
    red.flags = reinterpret_cast<int*>(red.buf + 96);
    red.flip = 0;
    red.turn = 0;
    int* nwarn2 = red.flags + 4;  // two counters, by parity of the system's turn on this workgroup
    const RowRef S{smem + (size_t)(a.scratch_row0 + wave * a.scratch_rows) * 64 + lane};  // this wavefront's scratch rows
    const bool unit_w = a.unit_weights != 0;
    if (tid < 8) red.flags[tid] = 0;
    __syncthreads();

    uint32_t parity = 0;
    // (a resident launch -- DoneWord::request, one workgroup -- serves one request after the other on the same buffers)
    // (thread 0's word to the others, behind the flag words of the reduction scratch: no static LDS beside the dynamic block)
    unsigned long long* const resident_word = reinterpret_cast<unsigned long long*>(red.buf + 104);
    const unsigned long long born = wall_clock64();
    DoneWord done = a.done;
    do {
    for (uint64_t sys = blockIdx.x; sys < a.batch; sys += gridDim.x, parity ^= 1u) {
        const double* x0 = a.x0 + sys * a.n_row;
        int* nwarn = nwarn2 + parity;
        auto log_warning = [&](uint32_t pass, uint32_t pos) {  // Warning::Degenerate, every evaluation (solver.rs:340-346)
            const int idx = atomicAdd(nwarn, 1);
            if (a.warn_log && (uint32_t)idx < a.warn_cap) a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | pos;
        };

        // One sweep of a chunk's constraints: residuals of the values `xs` into rows `dst` (weighted), their sum of
        // squares / maximum into sq / mx; MODE 0 logs degenerate evaluations at once (eval(), newton.rs:45), MODE 1
        // collects them in the chunk's warning mask (speculative sweep of a step), MODE 2 is the unsatisfied check on
        // the unweighted values (lib.rs:305-327).
        auto residual_sweep = [&](const CompChunkRegs& K, const RowRef& P, const RowRef& xs, const RowRef& dst, bool active, int MODE,
                                  uint32_t pass, double& sq, double& mx, double& unsat) {
            unsigned long long wmask = 0;
            const double* par = reinterpret_cast<const double*>(a.prog + K.par_off) + lane;
            const uint32_t* posp = a.prog + K.pos_off + lane;
            double p_next = K.ncons ? par[0] : 0.0;
            for (uint32_t ci = 0; ci < K.ncons; ++ci) {
                const CompRec8 ra = comp_load8(prog + K.cons_off + ci * kCompConWords);
                const CompRec8 rb = comp_load8(prog + K.cons_off + ci * kCompConWords + 8);
                const double param = p_next;
                if (ci + 1 < K.ncons) p_next = par[(size_t)(ci + 1) * K.stride];
                const DevCon c = comp_make_con(ra, rb, param);
                double r0, r1;
                const bool deg = con_residual<LIN>(c, xs, r0, r1);
                if (MODE == 2) {
                    bool sat = fabs(r0) < EPS;
                    if (c.nrows > 1) sat = sat && (fabs(r1) < EPS);
                    if (active) {
                        if (!sat) unsat += 1.0;
                        if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + posp[(size_t)ci * K.stride]] = sat ? 0 : 1;
                    }
                    continue;
                }
                const double w0 = c.weight * r0;  // solver.rs:353
                dst[c.row0] = w0;
                if (active) {
                    sq += w0 * w0;
                    mx = fmax(mx, fabs(w0));
                }
                if (c.nrows > 1) {
                    const double w1 = c.weight * r1;
                    dst[c.row0 + 1] = w1;
                    if (active) {
                        sq += w1 * w1;
                        mx = fmax(mx, fabs(w1));
                    }
                }
                if constexpr (!LIN) {
                    if (deg && active) {
                        if (MODE == 0)
                            log_warning(pass, posp[(size_t)ci * K.stride]);
                        else
                            wmask |= 1ull << ci;
                    }
                }
            }
            if constexpr (!LIN) {
                if (MODE == 1) P[K.o_wm] = __builtin_bit_cast(double, wmask);
            }
        };
        // Jacobian sweep of a chunk at the values `xs` (eval() and accepted steps, newton.rs:121; solver.rs:359-440).
        auto jacobian_sweep = [&](const CompChunkRegs& K, const RowRef& P, const RowRef& xs, bool active, uint32_t pass) {
            if constexpr (!LIN) {
                const double* par = reinterpret_cast<const double*>(a.prog + K.par_off) + lane;
                const uint32_t* posp = a.prog + K.pos_off + lane;
                double p_next = K.ncons ? par[0] : 0.0;
                for (uint32_t ci = 0; ci < K.ncons; ++ci) {
                    const CompRec8 ra = comp_load8(prog + K.cons_off + ci * kCompConWords);
                    const CompRec8 rb = comp_load8(prog + K.cons_off + ci * kCompConWords + 8);
                    const double param = p_next;
                    if (ci + 1 < K.ncons) p_next = par[(size_t)(ci + 1) * K.stride];
                    const DevCon c = comp_make_con(ra, rb, param);
                    JacWriter<RowRef> w;
                    w.jv = P + K.o_j;
                    w.jbase = c.jbase;
                    w.loc[0] = ra.w[6];
                    w.loc[1] = ra.w[7];
                    w.loc[2] = rb.w[0];
                    w.loc[3] = rb.w[1];
                    w.weight = c.weight;
                    const bool deg = con_jacobian<false>(c, xs, w);
                    if (deg && active) log_warning(pass, posp[(size_t)ci * K.stride]);
                }
            }
        };

        // ---- load the initial values; eval() (newton.rs:45, :232-236) ------------------------------------------------------------
        double sq = 0.0, mx = __builtin_nan(""), unsat_cnt = 0.0;
        for (uint32_t ch = ch0; ch < ch1; ++ch) {
            const CompChunkRegs K = comp_load_chunk(prog + a.o_chunks + 32 * ch);
            const RowRef P{smem + (size_t)K.row0 * 64 + lane};
            const bool active = (uint32_t)lane < K.count;
            const uint32_t* ids = a.prog + K.ids_off + lane;
            // (four values in flight at a time: the id and the value are two dependent trips to L2 / HBM)
            for (uint32_t k0 = 0; k0 < K.nv; k0 += 4) {
                uint32_t id[4];
                double v[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) id[j] = ids[(size_t)(k0 + j < K.nv ? k0 + j : k0) * K.stride];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j) v[j] = x0[id[j]];
#pragma unroll
                for (uint32_t j = 0; j < 4; ++j)
                    if (k0 + j < K.nv) P[k0 + j] = v[j];
            }
            residual_sweep(K, P, P, P + K.o_r0, active, 0, 0, sq, mx, unsat_cnt);
            jacobian_sweep(K, P, P, active, 1);
        }
        red.sum_max(sq, mx, lane, wave, nwaves);
        double residual_sq = sq, largest = mx;
        uint32_t pass = 2;
        uint32_t r_cur = 0;  // which of the two residual copies holds r
        double lambda = a.initial_lambda;
        uint32_t it = 0, iterations = a.max_iterations, converged = 0;
        bool r_is_at_x = true;

        // ---- the LM loop (newton.rs:47-139) ------------------------------------------------------------------------------------------
        for (;;) {
            if (it >= a.max_iterations) break;            // newton.rs:141-144
            if (largest <= a.residual_tolerance) {        // newton.rs:50-60
                iterations = it;
                converged = 1;
                break;
            }
            bool lane_bad = false;
            double dmax = __builtin_nan("");
            sq = 0.0;
            mx = __builtin_nan("");
            for (uint32_t ch = ch0; ch < ch1; ++ch) {
                const CompChunkRegs K = comp_load_chunk(prog + a.o_chunks + 32 * ch);
                const RowRef P{smem + (size_t)K.row0 * 64 + lane};
                const bool active = (uint32_t)lane < K.count;
                // the regions of this lane's state the operation stream addresses
                const RowRef V = P + K.o_d;                        // b -> y -> d
                const RowRef R = P + (r_cur ? K.o_r1 : K.o_r0);    // r
                const RowRef RN = P + (r_cur ? K.o_r0 : K.o_r1);   // r_next
                const RowRef J = P + K.o_j;
                const RowRef L = S + K.s_l;
                // -- normal equations, Cholesky, substitutions of this lane's component (newton.rs:73-102): the class's
                //    operation stream.  Scratch rows: diagonal (later the tentative x) at 0, L at s_l.
                double acc = 0.0, y = 0.0;
                bool chunk_bad = false;
                double chunk_dmax = __builtin_nan("");
                CompRec8 rec = comp_load8(prog + K.ops_off);
                for (uint32_t io = 0; io < K.n_ops; ++io) {
                    const CompRec8 nxt = comp_load8(prog + K.ops_off + (io + 1) * kCompRecWords);
                    const uint32_t op = rec.w[0] & 0xFFu, ni = (rec.w[0] >> 8) & 0xFFu;
                    const bool first = (rec.w[0] & kCompFirst) != 0, last = (rec.w[0] & kCompLast) != 0;
                    const uint32_t oa = rec.w[1] & 0xFFFFu, ob = rec.w[1] >> 16;
                    switch (op) {
                    case COMP_DIAG:
                        if constexpr (LIN) {
                            if (first) {
                                acc = __hiloint2double((int)rec.w[3], (int)rec.w[2]);
                                y = 0.0;
                            }
#pragma unroll
                            for (uint32_t k = 0; k < kCompItemsLin; ++k)
                                if (k < ni) {
                                    const double jv = (double)__uint_as_float(rec.w[5 + 2 * k]);
                                    y += jv * -R[rec.w[4 + 2 * k]];
                                }
                        } else {
                            if (first) acc = 0.0, y = 0.0;
#pragma unroll
                            for (uint32_t k = 0; k < kCompItemsGen; ++k)
                                if (k < ni) {
                                    const double jv = J[rec.w[2 + k] & 0xFFFFu], rv = R[rec.w[2 + k] >> 16];
                                    acc += jv * jv;
                                    y += jv * -rv;
                                }
                        }
                        if (last) {
                            S[oa] = acc + lambda;  // newton.rs:77-84
                            V[oa] = y;
                        }
                        break;
                    case COMP_OFF:
                        if constexpr (LIN) {
                            L[oa] = __hiloint2double((int)rec.w[3], (int)rec.w[2]);
                        } else {
                            if (first) acc = 0.0;
#pragma unroll
                            for (uint32_t k = 0; k < kCompItemsGen; ++k)
                                if (k < ni) acc += J[rec.w[2 + k] & 0xFFFFu] * J[rec.w[2 + k] >> 16];
                            if (last) L[oa] = acc;
                        }
                        break;
                    case COMP_COL:
                        if (first) {
                            acc = S[oa];
                            y = V[oa];
                        }
#pragma unroll
                        for (uint32_t k = 0; k < kCompItemsGen; ++k)
                            if (k < ni) {
                                const double l = L[rec.w[2 + k] & 0xFFFFu], yk = V[rec.w[2 + k] >> 16];
                                acc -= l * l;
                                y -= l * yk;
                            }
                        if (last) {
                            if (!(acc > 0.0)) chunk_bad = true;  // LltError::Numeric: non-positive pivot (newton.rs:93-99)
                            const double dv = sqrt(acc);
                            S[oa] = dv;
                            V[oa] = y / dv;
                        }
                        break;
                    case COMP_SLOT:
                        if (first) acc = L[oa];
#pragma unroll
                        for (uint32_t k = 0; k < kCompItemsGen; ++k)
                            if (k < ni) acc -= L[rec.w[2 + k] & 0xFFFFu] * L[rec.w[2 + k] >> 16];
                        if (last) L[oa] = acc / S[ob];
                        break;
                    case COMP_BWD:
                        if (first) acc = V[oa];
#pragma unroll
                        for (uint32_t k = 0; k < kCompItemsGen; ++k)
                            if (k < ni) acc -= L[rec.w[2 + k] & 0xFFFFu] * V[rec.w[2 + k] >> 16];
                        if (last) {
                            const double dval = acc / S[oa];
                            V[oa] = dval;
                            chunk_dmax = fmax(chunk_dmax, fabs(dval));
                            S[oa] = P[oa] + dval;  // the tentative x (newton.rs:111-114); the diagonal entry is dead now
                        }
                        break;
                    default: break;
                    }
                    rec = nxt;
                }
                if (active) {
                    lane_bad = lane_bad || chunk_bad;
                    dmax = fmax(dmax, chunk_dmax);
                }
                // -- residual at the tentative values (newton.rs:115-116), speculative: see the file comment
                residual_sweep(K, P, S, RN, active, 1, 0, sq, mx, unsat_cnt);
            }
            const bool bad = red.step(sq, mx, dmax, lane_bad, lane, wave, nwaves);
            if (bad) {  // numeric failure anywhere in the system => lambda *= 10, burn the iteration, x untouched
                lambda *= LM_LAMBDA_INCR;
                ++it;
                continue;
            }
            const double step_inf_norm = (a.n_row > 0) ? dmax : 0.0;
            const bool accept = sq < residual_sq;  // strict, newton.rs:118
            const uint32_t pass_res = pass++;
            const uint32_t pass_jac = pass;
            if (accept) ++pass;
            for (uint32_t ch = ch0; ch < ch1; ++ch) {
                const CompChunkRegs K = comp_load_chunk(prog + a.o_chunks + 32 * ch);
                const RowRef P{smem + (size_t)K.row0 * 64 + lane};
                const bool active = (uint32_t)lane < K.count;
                if constexpr (!LIN) {  // the warnings of the residual sweep that has now officially happened
                    unsigned long long wmask = __builtin_bit_cast(unsigned long long, P[K.o_wm]);
                    if (!active) wmask = 0;
                    while (wmask) {
                        const int ci = __builtin_ctzll(wmask);
                        wmask &= wmask - 1;
                        log_warning(pass_res, a.prog[K.pos_off + (size_t)ci * K.stride + lane]);
                    }
                }
                if (accept) {
                    for (uint32_t k = 0; k < K.nv; ++k) P[k] = P[k] + P[K.o_d + k];
                    jacobian_sweep(K, P, P, active, pass_jac);
                } else {  // reject: x += d, x -= d like the reference (newton.rs:111-114,:124-131), not a copy
                    for (uint32_t k = 0; k < K.nv; ++k) {
                        const double d = P[K.o_d + k];
                        P[k] = (P[k] + d) - d;
                    }
                }
            }
            if (accept) {
                r_cur ^= 1u;
                lambda *= LM_LAMBDA_DECR;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
            } else {
                r_is_at_x = false;  // x is now (x + d) - d, which may differ from the x of r in the last bit
                lambda *= LM_LAMBDA_INCR;
            }
            if (step_inf_norm <= a.step_tolerance) {  // newton.rs:134-139
                iterations = it;
                converged = 1;
                break;
            }
            ++it;
        }

        // ---- unsatisfied check (lib.rs:305-327, :358-370) and write-back -----------------------------------------------------------------
        // With unit weights and r evaluated at exactly this x, r already holds the unweighted residuals; when even the
        // largest |r| is below EPSILON and none is NaN (a NaN makes the sum of squares NaN) nothing can be unsatisfied.
        const bool use_r = r_is_at_x && unit_w;
        const bool all_satisfied = use_r && largest < EPS && !isnan(residual_sq);
        unsat_cnt = 0.0;
        for (uint32_t ch = ch0; ch < ch1; ++ch) {
            const CompChunkRegs K = comp_load_chunk(prog + a.o_chunks + 32 * ch);
            const RowRef P{smem + (size_t)K.row0 * 64 + lane};
            const bool active = (uint32_t)lane < K.count;
            if (all_satisfied) {
                if (a.unsat_mask && active)
                    for (uint32_t ci = 0; ci < K.ncons; ++ci)
                        a.unsat_mask[sys * a.n_cons + a.prog[K.pos_off + (size_t)ci * K.stride + lane]] = 0;
            } else if (use_r) {
                const RowRef R = P + (r_cur ? K.o_r1 : K.o_r0);
                for (uint32_t ci = 0; ci < K.ncons; ++ci) {
                    const uint32_t w0 = prog[K.cons_off + ci * kCompConWords], w1 = prog[K.cons_off + ci * kCompConWords + 1];
                    const uint32_t row = w1 & 0xFFFFu;
                    bool sat = fabs(R[row]) < EPS;
                    if (((w0 >> 16) & 0xFFu) > 1) sat = sat && (fabs(R[row + 1]) < EPS);
                    if (active) {
                        if (!sat) unsat_cnt += 1.0;
                        if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + a.prog[K.pos_off + (size_t)ci * K.stride + lane]] = sat ? 0 : 1;
                    }
                }
            } else {
                double s_ = 0.0, m_ = 0.0;
                residual_sweep(K, P, P, P, active, 2, 0, s_, m_, unsat_cnt);
            }
            if (active) {
                double* xo = a.x_out + sys * a.n_row;
                const uint32_t* ids = a.prog + K.ids_off + lane;
                for (uint32_t k = 0; k < K.nv; ++k) xo[ids[(size_t)k * K.stride]] = P[k];
            }
        }
        // the count of unsatisfied constraints; in the general build also what orders every wavefront's warnings before
        // the counter is read
        if (!all_satisfied || !LIN) {
            double none = __builtin_nan("");
            red.sum_max(unsat_cnt, none, lane, wave, nwaves);
        }
        if (tid == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = LIN ? 0u : (uint32_t)*nwarn;
            st.final_residual_inf = (a.n_rows_total > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            a.status[sys] = st;
            // this counter serves the workgroup's system after next; every wavefront passes a rendezvous of the next
            // system (which this thread joins only after the store) before it can touch it again
            if (!LIN) *nwarn = 0;
        }
    }
    publish_done(done);
    } while (resident_next(done, born, resident_word));
}

}  // namespace ezpz
