// The Levenberg-Marquardt solve kernel: one *team* of lanes owns one constraint system for the whole
// solve -- residual/Jacobian evaluation, normal equations, sparse Cholesky, triangular solves, step
// acceptance and convergence tests all run inside this one launch, with the system state resident in
// LDS.  It replaces, for a batch of systems sharing a topology,
//     Model::solve_levenberg_marquardt      reference ezpz/src/solver/newton.rs:29-145
//     Model::residual / refresh_jacobian    reference ezpz/src/solver.rs:318-440
//     faer's transpose/matmul/add/Llt/solve reference ezpz/src/solver/newton.rs:73-102
//     the unsatisfied check of solve_inner  reference ezpz/src/lib.rs:305-327
//
// Team shapes (template):
//   TEAM in {8,16,32,64}, WG=false : sub-wavefront / one-wavefront teams, several systems per 64-wide wave,
//                                    no s_barrier anywhere (lanes of a wave run in lockstep; LDS is in-order),
//                                    reductions by DPP/bpermute shuffles.
//   WG=true                        : the whole workgroup (128..1024 lanes) is one team; phases are separated
//                                    by s_barrier, reductions go wave-shuffle -> LDS -> all lanes.
//   LDSWS=false                    : state lives in a per-workgroup global-memory workspace (systems too big
//                                    for the 160 KB LDS).
// HBM traffic is only x0 in, x*/status/mask out (AoS rows, contiguous per team); the topology program is
// shared by every team and stays L2 resident.
#pragma once
#include <hip/hip_runtime.h>

#include "constraint_eval.hip.hpp"

namespace ezpz {

struct ProgramView {
    const DevCon* cons;
    const uint32_t *colj_ptr, *colj_items;
    const uint32_t *apair_ptr, *apairs;
    const uint32_t *lvl_cptr, *lvl_cols, *lvl_sptr, *l_col;
    const uint32_t *lpair_ptr, *lpairs;
    const uint32_t *fwd_ptr, *fwd_items;
    const uint32_t *bwd_ptr, *bwd_items;
    uint32_t n_cons, n_vars, n_rows, zj, zlo, n_levels;
};

struct SolveArgs {
    ProgramView p;
    const double* x0;
    double* x_out;
    EzpzStatus* status;
    uint8_t* unsat_mask;   // optional
    uint64_t* warn_log;    // optional
    double* gws;           // global workspace (LDSWS=false), ws_doubles per workgroup
    uint64_t batch;
    uint32_t warn_cap;
    uint32_t ws_doubles;   // doubles per team workspace (incl. the small int area, rounded to 2 doubles)
    uint32_t max_iterations;
    double residual_tolerance, step_tolerance, initial_lambda;
};

namespace dev {

constexpr double LM_LAMBDA_INCR = 10.0;  // newton.rs:15
constexpr double LM_LAMBDA_DECR = 0.1;   // newton.rs:16

template <int TEAM, bool WG>
struct Team {
    int lane;        // lane within the team
    int size;        // lanes in the team
    double* red;     // WG only: 2 x 16 doubles of LDS scratch
    int red_flip;

    __device__ __forceinline__ void sync() const {
        if constexpr (WG) {
            __syncthreads();
        } else {
            // Lanes of one wavefront execute in lockstep and the LDS services a wave's accesses in issue
            // order, so no hardware barrier is needed; only the compiler must not move LDS accesses across.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }

    template <class Op>
    __device__ __forceinline__ double reduce(double v, Op op) {
        if constexpr (!WG) {
#pragma unroll
            for (int off = TEAM / 2; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, TEAM));
            return __shfl(v, 0, TEAM);
        } else {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
            double* buf = red + (red_flip ? 16 : 0);
            red_flip ^= 1;
            const int wave = threadIdx.x >> 6;
            const int nwaves = (blockDim.x + 63) >> 6;
            if ((threadIdx.x & 63) == 0) buf[wave] = v;
            __syncthreads();
            double acc = buf[0];
            for (int w = 1; w < nwaves; ++w) acc = op(acc, buf[w]);
            return acc;  // the other buffer is used by the next reduction, so no trailing barrier is needed
        }
    }
    __device__ __forceinline__ double sum(double v) {
        return reduce(v, [](double a, double b) { return a + b; });
    }
    __device__ __forceinline__ double max(double v) {
        return reduce(v, [](double a, double b) { return fmax(a, b); });  // libm::fmax, newton.rs:53,:108
    }
};

// Residual sweep: r[row] = weight * residual (solver.rs:327-355).  One lane per constraint.  With
// `unweighted` the raw residuals are stored and nothing is logged (the unsatisfied check, lib.rs:305-327).
template <class T, class WP>
__device__ __forceinline__ void sweep_residual(const T& tm, const SolveArgs& a, WP ws, uint32_t o_x, uint32_t o_r,
                                               int* nwarn, uint64_t sys, uint32_t pass, bool unweighted,
                                               WP out = nullptr) {
    if (!out) out = ws;
    for (uint32_t ci = tm.lane; ci < a.p.n_cons; ci += tm.size) {
        const DevCon& c = a.p.cons[ci];
        double r0, r1;
        bool deg = con_residual(c, ws + o_x, r0, r1);
        const double wgt = unweighted ? 1.0 : c.weight;
        const uint32_t row0 = c.row0;
        out[o_r + row0] = wgt * r0;
        if (c.nrows > 1) out[o_r + row0 + 1] = wgt * r1;
        if (deg && !unweighted) {
            int idx = atomicAdd(nwarn, 1);
            if (a.warn_log && (uint32_t)idx < a.warn_cap)
                a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | c.pos;
        }
    }
}

// Jacobian sweep (solver.rs:359-440): weighted partials straight into their precomputed slots.
template <class T, class WP>
__device__ __forceinline__ void sweep_jacobian(const T& tm, const SolveArgs& a, WP ws, uint32_t o_x, uint32_t o_j,
                                               int* nwarn, uint64_t sys, uint32_t pass, WP out = nullptr) {
    if (!out) out = ws;
    for (uint32_t ci = tm.lane; ci < a.p.n_cons; ci += tm.size) {
        const DevCon& c = a.p.cons[ci];
        JacWriter<WP> w;
        w.jv = out + o_j;
        w.jbase = c.jbase;
        const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
        w.loc[0] = loc[0];
        w.loc[1] = loc[1];
        w.loc[2] = loc[2];
        w.loc[3] = loc[3];
        w.weight = c.weight;
        bool deg = con_jacobian(c, ws + o_x, w);
        if (deg) {
            int idx = atomicAdd(nwarn, 1);
            if (a.warn_log && (uint32_t)idx < a.warn_cap)
                a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | c.pos;
        }
    }
}

template <class T, class WP>
__device__ __forceinline__ double sum_squares(T& tm, WP ws, uint32_t off, uint32_t m) {
    double acc = 0.0;
    for (uint32_t i = tm.lane; i < m; i += tm.size) {
        double v = ws[off + i];
        acc += v * v;
    }
    return tm.sum(acc);
}

template <class T, class WP>
__device__ __forceinline__ double max_abs(T& tm, WP ws, uint32_t off, uint32_t m) {
    // reduce(fmax) over |.|: NaN-ignoring like libm::fmax; an all-NaN input yields NaN.  Lanes without an
    // element seed with NaN, which fmax drops, so the result only depends on real elements.
    double acc = (tm.lane < (int)m) ? fabs(ws[off + tm.lane]) : __builtin_nan("");
    for (uint32_t i = tm.lane + tm.size; i < m; i += tm.size) acc = fmax(acc, fabs(ws[off + i]));
    return tm.max(acc);
}

}  // namespace dev

template <int TEAM, bool WG, bool LDSWS>
__global__ void __launch_bounds__(WG ? 1024 : 256) lm_solve_kernel(const SolveArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using namespace dev;
    Team<TEAM, WG> tm;
    const int tid = threadIdx.x;
    tm.size = WG ? (int)blockDim.x : TEAM;
    tm.lane = WG ? tid : (tid % TEAM);
    tm.red_flip = 0;
    const uint32_t teams_per_block = WG ? 1u : (uint32_t)(blockDim.x / TEAM);
    const uint32_t team_in_block = WG ? 0u : (uint32_t)(tid / TEAM);
    const uint32_t n = a.p.n_vars, m = a.p.n_rows, zj = a.p.zj, zlo = a.p.zlo, nlev = a.p.n_levels;

    // workspace carve-up (doubles)
    double* ws;
    if constexpr (LDSWS) {
        ws = smem + (size_t)team_in_block * a.ws_doubles;
        tm.red = smem + (size_t)teams_per_block * a.ws_doubles;
    } else {
        ws = a.gws + (size_t)blockIdx.x * a.ws_doubles;
        tm.red = smem;
    }
    const uint32_t o_x = 0;
    uint32_t o_r = n;
    uint32_t o_rn = n + m;
    const uint32_t o_j = n + 2 * m;
    const uint32_t o_d = o_j + zj;   // Cholesky diagonal, by variable
    const uint32_t o_l = o_d + n;    // strictly-lower L entries, level grouped
    const uint32_t o_v = o_l + zlo;  // b, then y, then d (by variable)
    const uint32_t o_i = o_v + n;    // small int area
    int* nwarn;
    if constexpr (LDSWS) {
        nwarn = reinterpret_cast<int*>(ws + o_i);
    } else {
        nwarn = reinterpret_cast<int*>(smem + 32);  // LDS even when the bulk state is in global memory
    }

    const uint64_t n_teams = (uint64_t)gridDim.x * teams_per_block;
    for (uint64_t sys = (uint64_t)blockIdx.x * teams_per_block + team_in_block; sys < a.batch; sys += n_teams) {
        // ---- load the initial values (AoS row, coalesced) ------------------------------------------------------
        const double* x0 = a.x0 + sys * n;
        for (uint32_t i = tm.lane; i < n; i += tm.size) ws[o_x + i] = x0[i];
        if (tm.lane == 0) *nwarn = 0;
        tm.sync();

        // The LM loop of newton.rs:29-145 as a three-mode state machine, so that each of the two big
        // evaluators is instantiated exactly once (register pressure / code size):
        //   EVAL0  eval() before the loop (newton.rs:45, :232-236): residual -> r, Jacobian, sum of squares
        //   STEP   one `for this_iteration` body: test, linear solve, tentative step, residual -> r_next,
        //          accept/reject, step-size test
        //   FINAL  unweighted residuals for the unsatisfied check (lib.rs:305-327), then leave
        enum { EVAL0 = 0, STEP = 1, FINAL = 2 };
        int mode = EVAL0;
        uint32_t pass = 0;
        uint32_t it = 0;
        double residual_sq = 0.0;
        double lambda = a.initial_lambda;
        double step_inf_norm = 0.0;
        double final_inf = 0.0;
        uint32_t iterations = a.max_iterations;
        uint32_t converged = 0;
        for (;;) {
            if (mode == STEP) {
                if (it >= a.max_iterations) {  // newton.rs:141-144
                    mode = FINAL;
                } else {
                    // convergence on max |r| (newton.rs:50-60)
                    double largest = max_abs(tm, ws, o_r, m);
                    if (largest <= a.residual_tolerance) {
                        iterations = it;
                        converged = 1;
                        mode = FINAL;
                    }
                }
            }
            if (mode == STEP) {
                // ---- A = JtJ + lambda I (into L's storage) and b = Jt(-r)  (newton.rs:77-84) --------------------
                for (uint32_t v = tm.lane; v < n; v += tm.size) {
                    double acc = 0.0, b = 0.0;
                    for (uint32_t q = a.p.colj_ptr[v]; q < a.p.colj_ptr[v + 1]; ++q) {
                        double jv = ws[o_j + a.p.colj_items[2 * q]];
                        acc += jv * jv;
                        b += jv * -ws[o_r + a.p.colj_items[2 * q + 1]];
                    }
                    ws[o_d + v] = acc + lambda;
                    ws[o_v + v] = b;
                }
                for (uint32_t s = tm.lane; s < zlo; s += tm.size) {
                    double acc = 0.0;
                    for (uint32_t q = a.p.apair_ptr[s]; q < a.p.apair_ptr[s + 1]; ++q)
                        acc += ws[o_j + a.p.apairs[2 * q]] * ws[o_j + a.p.apairs[2 * q + 1]];
                    ws[o_l + s] = acc;
                }
                tm.sync();
                // ---- level-scheduled sparse Cholesky + forward substitution (newton.rs:87-102) -------------------
                double bad = 0.0;
                for (uint32_t lv = 0; lv < nlev; ++lv) {
                    const uint32_t c0 = a.p.lvl_cptr[lv], c1 = a.p.lvl_cptr[lv + 1];
                    const uint32_t s0 = a.p.lvl_sptr[lv], s1 = a.p.lvl_sptr[lv + 1];
                    for (uint32_t ci = c0 + tm.lane; ci < c1; ci += tm.size) {
                        const uint32_t v = a.p.lvl_cols[ci];
                        double acc = ws[o_d + v];
                        for (uint32_t q = a.p.fwd_ptr[v]; q < a.p.fwd_ptr[v + 1]; ++q) {
                            double l = ws[o_l + a.p.fwd_items[2 * q]];
                            acc -= l * l;
                        }
                        if (!(acc > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                        ws[o_d + v] = sqrt(acc);
                    }
                    for (uint32_t s = s0 + tm.lane; s < s1; s += tm.size) {
                        double acc = ws[o_l + s];
                        for (uint32_t q = a.p.lpair_ptr[s]; q < a.p.lpair_ptr[s + 1]; ++q)
                            acc -= ws[o_l + a.p.lpairs[2 * q]] * ws[o_l + a.p.lpairs[2 * q + 1]];
                        ws[o_l + s] = acc;
                    }
                    tm.sync();
                    for (uint32_t s = s0 + tm.lane; s < s1; s += tm.size)
                        ws[o_l + s] = ws[o_l + s] / ws[o_d + a.p.l_col[s]];
                    for (uint32_t ci = c0 + tm.lane; ci < c1; ci += tm.size) {
                        const uint32_t v = a.p.lvl_cols[ci];
                        double acc = ws[o_v + v];
                        for (uint32_t q = a.p.fwd_ptr[v]; q < a.p.fwd_ptr[v + 1]; ++q)
                            acc -= ws[o_l + a.p.fwd_items[2 * q]] * ws[o_v + a.p.fwd_items[2 * q + 1]];
                        ws[o_v + v] = acc / ws[o_d + v];
                    }
                    tm.sync();
                }
                if (tm.max(bad) > 0.0) {  // numeric failure => lambda *= 10, burn the iteration (newton.rs:96-99)
                    lambda *= LM_LAMBDA_INCR;
                    ++it;
                    continue;
                }
                // ---- backward substitution ----------------------------------------------------------------------
                for (uint32_t lv = nlev; lv-- > 0;) {
                    const uint32_t c0 = a.p.lvl_cptr[lv], c1 = a.p.lvl_cptr[lv + 1];
                    for (uint32_t ci = c0 + tm.lane; ci < c1; ci += tm.size) {
                        const uint32_t v = a.p.lvl_cols[ci];
                        double acc = ws[o_v + v];
                        for (uint32_t q = a.p.bwd_ptr[v]; q < a.p.bwd_ptr[v + 1]; ++q)
                            acc -= ws[o_l + a.p.bwd_items[2 * q]] * ws[o_v + a.p.bwd_items[2 * q + 1]];
                        ws[o_v + v] = acc / ws[o_d + v];
                    }
                    tm.sync();
                }
                // ---- tentative step (newton.rs:108-114) --------------------------------------------------------------
                step_inf_norm = (n > 0) ? max_abs(tm, ws, o_v, n) : 0.0;
                for (uint32_t i = tm.lane; i < n; i += tm.size) ws[o_x + i] = ws[o_x + i] + ws[o_v + i];
                tm.sync();
            }
            if (mode == FINAL) final_inf = (m > 0) ? max_abs(tm, ws, o_r, m) : 0.0;

            // ---- the one residual sweep: EVAL0 -> r, STEP -> r_next, FINAL -> unweighted into r_next ------------------
            const uint32_t o_dst = (mode == EVAL0) ? o_r : o_rn;
            sweep_residual(tm, a, ws, o_x, o_dst, nwarn, sys, pass, mode == FINAL);
            ++pass;
            tm.sync();
            if (mode == FINAL) break;
            const double sq = sum_squares(tm, ws, o_dst, m);
            const bool accept = (mode == EVAL0) || (sq < residual_sq);  // strict, newton.rs:118
            if (accept) {
                if (mode == STEP) {
                    uint32_t t = o_r;
                    o_r = o_rn;
                    o_rn = t;
                    lambda *= LM_LAMBDA_DECR;
                }
                // ---- the one Jacobian sweep (eval() and accepted steps, newton.rs:121) ------------------------------
                sweep_jacobian(tm, a, ws, o_x, o_j, nwarn, sys, pass);
                ++pass;
                residual_sq = sq;
            } else {  // reject: revert, raise lambda (newton.rs:124-131)
                for (uint32_t i = tm.lane; i < n; i += tm.size) ws[o_x + i] = ws[o_x + i] - ws[o_v + i];
                lambda *= LM_LAMBDA_INCR;
            }
            tm.sync();
            if (mode == STEP) {
                if (step_inf_norm <= a.step_tolerance) {  // newton.rs:134-139
                    iterations = it;
                    converged = 1;
                    mode = FINAL;
                    continue;
                }
                ++it;
            }
            if (mode == EVAL0) mode = STEP;
        }

        // ---- unsatisfied list from the unweighted residuals in r_next (lib.rs:305-327, :358-370) + write-back -------
        double unsat_cnt = 0.0;
        for (uint32_t ci = tm.lane; ci < a.p.n_cons; ci += tm.size) {
            const DevCon& c = a.p.cons[ci];
            const uint32_t row0 = c.row0;
            bool sat = fabs(ws[o_rn + row0]) < EPS;
            if (c.nrows > 1) sat = sat && (fabs(ws[o_rn + row0 + 1]) < EPS);
            if (!sat) unsat_cnt += 1.0;
            if (a.unsat_mask) a.unsat_mask[sys * a.p.n_cons + c.pos] = sat ? 0 : 1;
        }
        unsat_cnt = tm.sum(unsat_cnt);
        double* xo = a.x_out + sys * n;
        for (uint32_t i = tm.lane; i < n; i += tm.size) xo[i] = ws[o_x + i];
        if (tm.lane == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = (uint32_t)*nwarn;
            st.final_residual_inf = final_inf;
            st.final_lambda = lambda;
            a.status[sys] = st;
        }
        tm.sync();  // the workspace is reused by the next system of this team
    }
}

// Evaluation-only kernel (K1): one workgroup per value vector, state in global memory.
struct EvalArgs {
    ProgramView p;
    const double* x;
    double* r_out;
    double* jv_out;
    uint32_t* deg_out;
    uint64_t batch;
};

__global__ void __launch_bounds__(256) eval_kernel(const EvalArgs e) {
    using namespace dev;
    __shared__ int nwarn;
    Team<64, true> tm;
    tm.size = (int)blockDim.x;
    tm.lane = (int)threadIdx.x;
    tm.red = nullptr;
    tm.red_flip = 0;
    SolveArgs a{};
    a.p = e.p;
    a.warn_log = nullptr;
    a.warn_cap = 0;
    for (uint64_t sys = blockIdx.x; sys < e.batch; sys += gridDim.x) {
        if (threadIdx.x == 0) nwarn = 0;
        __syncthreads();
        const double* xs = e.x + sys * e.p.n_vars;
        double* r = e.r_out + sys * e.p.n_rows;
        double* jv = e.jv_out + sys * e.p.zj;
        // the sweeps index one base pointer by offset; give each its own base
        sweep_residual(tm, a, const_cast<double*>(xs), 0u, (uint32_t)0, &nwarn, sys, 0u, false, r);
        sweep_jacobian(tm, a, const_cast<double*>(xs), 0u, (uint32_t)0, &nwarn, sys, 1u, jv);
        __syncthreads();
        if (threadIdx.x == 0 && e.deg_out) e.deg_out[sys] = (uint32_t)nwarn;
        __syncthreads();
    }
}

}  // namespace ezpz
