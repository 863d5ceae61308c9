// FreedomAnalysis on the device: which variables of a solved system are underconstrained.
//
// Reference: ezpz/src/solver/find_dof.rs:14-103 (Model::freedom_analysis) and ezpz/src/analysis.rs:24-77.  The
// reference densifies the weighted Jacobian the LM loop left behind, takes a column-pivoted QR, reads the rank off the
// diagonal of R with tolerance 1e-8 * max|R_ii|, builds a null-space basis by back substitution, orthonormalises it and
// flags every variable whose squared row norm in that basis (its "participation") exceeds (1e-3 * max participation)^2.
//
// Here the same algorithm runs per connected component of the Jacobian's bipartite row/variable graph: J is block
// diagonal over components, so the pivoted QR of J is the interleaving of the pivoted QRs of the blocks, the rank
// tolerance needs only the largest column norm of J (= the first, largest pivot) and the participation -- the diagonal
// of the orthogonal projector onto null(J), which does not depend on the basis -- is block diagonal too.  Dense work
// drops from m*n*n to sum m_c*n_c*n_c (2000x2000: 32 MB and 1.6e10 flops -> 500 blocks of 4x4).
//
// Two layouts, one code body (freedom_component<Ctx>):
//   LANE  one lane per (system, component), private workspace interleaved in LDS, no synchronisation
//   TEAM  one workgroup per system, components in sequence, lanes over columns, workspace in LDS or global memory
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "launch_types.hpp"
#include "lm_kernel.hip.hpp"  // (the fused form evaluates the Jacobian itself: Prog, load_con / load_packed, the evaluators)

namespace ezpz {

struct FreedomArgs {
    const double* jv;  // [batch][zj] weighted Jacobian values at the final point, internal slot order
    const FreedomComp* comps;
    const uint32_t* items;
    const uint32_t* comp_vars;
    const uint32_t* col_ptr;  // per caller variable: the slots of its column
    const uint32_t* col_slots;
    double* part;       // [batch][n] out: participation
    uint8_t* mask;      // [batch][n] out: 1 = underconstrained
    uint32_t* n_under;  // [batch] out, may be null
    double* gws;        // TEAM: global workspace [gridDim][ws], null when the workspace is in LDS
    uint64_t batch;
    uint32_t n, zj, ncomp;
    uint32_t ws;     // workspace doubles of the largest component
    uint32_t group;  // LANE: systems per workgroup
    uint32_t qr_done;  // TEAM, global workspace: 1 + the component whose pivoted QR is already in the workspace (step kernels below), 0 = none
    // [systems of this launch] or null: non-zero = the resident QR (fr_qrc_kernel) gave up waiting for another workgroup's chunk.
    // The factorisation in the workspace is then not one: the system's mask becomes kFreedomPoisonedMask in every byte and its
    // count 0xFFFFFFFF (the host entry returns EZPZ_ERR_HIP), as the LM grid teams report EZPZ_ITERATIONS_TEAM_TIMEOUT.
    const uint32_t* qr_timed_out;
    // FUSED small calls (solve_analysis of one sketch: one launch instead of gather_values + eval + this kernel, round 5): the kernel
    // gathers the caller-ordered values into the program's order and evaluates the weighted Jacobian there itself
    const double* x_caller;  // non-null: fused
    double* x_int;           // [batch][n] scratch
    double* jv_out;          // = jv
    ProgramView prog;
    unsigned long long* done_flag;  // a word of mapped host memory the caller polls (one-workgroup launches), or null
    unsigned long long done_seq;
};

constexpr double kFreedomRankTol = 1e-8;  // find_dof.rs:12
constexpr uint8_t kFreedomPoisonedMask = 0xFF;
constexpr double kFreedomVarTol = 1e-3;   // find_dof.rs:98

// Gathers caller-ordered values into the program's internal variable order for eval_kernel.
__global__ void __launch_bounds__(256) gather_values_kernel(const double* x, const uint32_t* var_of, double* x_int,
                                                            uint32_t n, uint64_t total) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t b = i / n;
        const uint32_t k = (uint32_t)(i - b * n);
        x_int[i] = x[b * n + var_of[k]];
    }
}

namespace freedom {

struct Ws {
    double* p;
    uint32_t stride;
    __device__ __forceinline__ double& operator()(uint32_t e) const { return p[(size_t)e * stride]; }
};

struct LaneCtx {
    static constexpr bool kRowMajor = false;  // a private matrix: the layout does not matter
    __device__ __forceinline__ uint32_t id() const { return 0; }
    __device__ __forceinline__ uint32_t count() const { return 1; }
    __device__ __forceinline__ void sync() const {}
    __device__ __forceinline__ double sum(double v) const { return v; }
    __device__ __forceinline__ void argmax(double&, uint32_t&) const {}
};

struct BlockCtx {
    // lanes work across columns: with rows contiguous a wavefront's accesses to A(i, j..j+63) coalesce (the global
    // workspace of a large component was read 8 bytes per cache line the other way round)
    static constexpr bool kRowMajor = true;
    double* red;  // LDS scratch, 16 doubles
    __device__ __forceinline__ uint32_t id() const { return threadIdx.x; }
    __device__ __forceinline__ uint32_t count() const { return blockDim.x; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    __device__ double sum(double v) const {
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (blockDim.x > 64) {
            const uint32_t nw = blockDim.x >> 6;
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
            __syncthreads();
            v = 0.0;
            for (uint32_t w = 0; w < nw; ++w) v += red[w];
        }
        return v;
    }
    // largest value, smallest index among equals; every lane gets the result
    __device__ void argmax(double& v, uint32_t& i) const {
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(v, off);
            const uint32_t oi = (uint32_t)__shfl_xor((int)i, off);
            if (ov > v || (ov == v && oi < i)) {
                v = ov;
                i = oi;
            }
        }
        if (blockDim.x > 64) {
            const uint32_t nw = blockDim.x >> 6;
            __syncthreads();
            if ((threadIdx.x & 63) == 0) {
                red[threadIdx.x >> 6] = v;
                red[8 + (threadIdx.x >> 6)] = (double)i;
            }
            __syncthreads();
            v = red[0];
            i = (uint32_t)red[8];
            for (uint32_t w = 1; w < nw; ++w) {
                const double ov = red[w];
                const uint32_t oi = (uint32_t)red[8 + w];
                if (ov > v || (ov == v && oi < i)) {
                    v = ov;
                    i = oi;
                }
            }
        }
    }
};

__device__ __forceinline__ void atomic_max_nonneg(unsigned long long* slot, double v) {
    if (v == v) atomicMax(slot, (unsigned long long)__double_as_longlong(v));  // libm::fmax ignores NaN
}

// Workspace of one m x n component, doubles: A[m*n] | NS[n*n] | Q[n*n] | perm[n] | tau[n]
__host__ __device__ inline uint32_t component_ws(uint32_t m, uint32_t n) { return m * n + 2 * n * n + 2 * n; }

template <class C>
__device__ void freedom_component(const C& ctx, const Ws W, const FreedomComp cd, const uint32_t* __restrict__ items,
                                  const double* __restrict__ jv, const double tol, const uint32_t* __restrict__ vars,
                                  double* __restrict__ part, unsigned long long* partmax, const bool qr_done = false) {
    const uint32_t m = cd.m, n = cd.n, mn = m * n;
    const uint32_t oNS = mn, oQ = mn + n * n, oPerm = mn + 2 * n * n, oTau = oPerm + n;
    const uint32_t id = ctx.id(), cnt = ctx.count();
#define A_(i, j) W(C::kRowMajor ? (i) * n + (j) : (j) * m + (i))
#define NS_(i, j) W(oNS + (j) * n + (i))
#define Q_(i, j) W(oQ + (j) * n + (i))
#define PERM_(j) ((uint32_t)W(oPerm + (j)))
#define NORM_(j) W(oTau + (j))  // running squared column norms during the QR (tau is only needed afterwards)
    const uint32_t ndiag = m < n ? m : n;
    if (!qr_done) {
    for (uint32_t e = id; e < mn; e += cnt) W(e) = 0.0;
    for (uint32_t j = id; j < n; j += cnt) W(oPerm + j) = (double)j;
    ctx.sync();
    for (uint32_t it = cd.item0 + id; it < cd.item1; it += cnt) {
        const uint32_t e = items[2 * it + 1];  // lcol * m + lrow
        const uint32_t lcol = e / m;
        A_(e - lcol * m, lcol) = jv[items[2 * it]];
    }
    ctx.sync();
    // ---- column-pivoted Householder QR (find_dof.rs:35 ColPivQr) ------------------------------------------------------
    // The squared norm of every remaining column below the current row is what the pivot search needs; it is summed
    // (rows ascending, exactly as a fresh pass would) while the reflector is applied to the column, so a step reads
    // the trailing matrix twice instead of three times.
    for (uint32_t j = id; j < n; j += cnt) {
        double s = 0.0;
        for (uint32_t i = 0; i < m; ++i) {
            const double a = A_(i, j);
            s += a * a;
        }
        NORM_(j) = s;
    }
    ctx.sync();
    for (uint32_t k = 0; k < ndiag; ++k) {
        double bv = -1.0;
        uint32_t bj = k;
        for (uint32_t j = k + id; j < n; j += cnt) {
            const double s = NORM_(j);
            if (s > bv) {
                bv = s;
                bj = j;
            }
        }
        ctx.argmax(bv, bj);
        if (!(bv > 0.0)) break;  // nothing left (or NaN): the remaining diagonal is exactly zero
        if (bj != k) {
            for (uint32_t i = id; i < m; i += cnt) {
                const double t = A_(i, k);
                A_(i, k) = A_(i, bj);
                A_(i, bj) = t;
            }
            if (id == 0) {
                const double t = W(oPerm + k);
                W(oPerm + k) = W(oPerm + bj);
                W(oPerm + bj) = t;
                NORM_(bj) = NORM_(k);  // column k's own norm is not needed again
            }
        }
        ctx.sync();
        const double norm = sqrt(bv);
        const double alpha = A_(k, k);
        const double beta = alpha >= 0.0 ? -norm : norm;
        const double denom = alpha - beta;
        const double tau = (beta - alpha) / beta;
        ctx.sync();
        for (uint32_t i = k + 1 + id; i < m; i += cnt) A_(i, k) /= denom;
        if (id == 0) A_(k, k) = beta;
        ctx.sync();
        for (uint32_t j = k + 1 + id; j < n; j += cnt) {
            double dot = A_(k, j);
            for (uint32_t i = k + 1; i < m; ++i) dot += A_(i, k) * A_(i, j);
            dot *= tau;
            A_(k, j) -= dot;
            double s = 0.0;
            for (uint32_t i = k + 1; i < m; ++i) {
                const double a = A_(i, j) - dot * A_(i, k);
                A_(i, j) = a;
                s += a * a;
            }
            NORM_(j) = s;
        }
        ctx.sync();
    }
    }  // !qr_done
    // ---- rank and null-space basis (find_dof.rs:38-77) --------------------------------------------------------------
    uint32_t rank = 0;
    while (rank < ndiag && fabs(A_(rank, rank)) > tol) ++rank;
    const uint32_t nullity = n - rank;
    if (nullity == 0) {
        for (uint32_t j = id; j < n; j += cnt) part[vars[j]] = 0.0;
        ctx.sync();
        return;
    }
    for (uint32_t e = id; e < n * nullity; e += cnt) {
        W(oNS + e) = 0.0;
        W(oQ + e) = 0.0;
    }
    ctx.sync();
    // R11 x = -R12 e_fc, one null vector at a time, the lanes sharing each row's dot product (one lane per vector walked
    // rank^2 / 2 dependent global loads: a second per vector at 2000 variables)
    for (uint32_t fc = 0; fc < nullity; ++fc) {
        const uint32_t fv = rank + fc;
        if (id == 0) NS_(PERM_(fv), fc) = 1.0;
        ctx.sync();
        for (uint32_t i = rank; i-- > 0;) {
            double dot = 0.0;
            for (uint32_t j = i + 1 + id; j < rank; j += cnt) dot += A_(i, j) * NS_(PERM_(j), fc);
            dot = ctx.sum(dot);
            if (id == 0) NS_(PERM_(i), fc) = -(A_(i, fv) + dot) / A_(i, i);
            ctx.sync();
        }
    }
    // ---- thin Q of the basis (find_dof.rs:79) ------------------------------------------------------------------------
    for (uint32_t k = 0; k < nullity; ++k) {
        double s = 0.0;
        for (uint32_t i = k + id; i < n; i += cnt) {
            const double a = NS_(i, k);
            s += a * a;
        }
        s = ctx.sum(s);
        const double norm = sqrt(s);
        const double alpha = NS_(k, k);
        const double beta = alpha >= 0.0 ? -norm : norm;
        const double denom = alpha - beta;
        const double tau = norm > 0.0 ? (beta - alpha) / beta : 0.0;
        ctx.sync();
        if (tau != 0.0) {
            for (uint32_t i = k + 1 + id; i < n; i += cnt) NS_(i, k) /= denom;
            if (id == 0) NS_(k, k) = beta;
        }
        if (id == 0) W(oTau + k) = tau;
        ctx.sync();
        if (tau != 0.0)
            for (uint32_t j = k + 1 + id; j < nullity; j += cnt) {
                double dot = NS_(k, j);
                for (uint32_t i = k + 1; i < n; ++i) dot += NS_(i, k) * NS_(i, j);
                dot *= tau;
                NS_(k, j) -= dot;
                for (uint32_t i = k + 1; i < n; ++i) NS_(i, j) -= dot * NS_(i, k);
            }
        ctx.sync();
    }
    for (uint32_t j = id; j < nullity; j += cnt) {  // Q(:,j) = H_0 .. H_j e_j (later reflectors leave e_j alone)
        Q_(j, j) = 1.0;
        for (uint32_t k = j + 1; k-- > 0;) {
            const double tau = W(oTau + k);
            if (tau == 0.0) continue;
            double dot = Q_(k, j);
            for (uint32_t i = k + 1; i < n; ++i) dot += NS_(i, k) * Q_(i, j);
            dot *= tau;
            Q_(k, j) -= dot;
            for (uint32_t i = k + 1; i < n; ++i) Q_(i, j) -= dot * NS_(i, k);
        }
    }
    ctx.sync();
    // ---- participation (find_dof.rs:90-95) --------------------------------------------------------------------------
    double local_max = 0.0;
    for (uint32_t i = id; i < n; i += cnt) {
        double sq = 0.0;
        for (uint32_t j = 0; j < nullity; ++j) {
            const double q = Q_(i, j);
            sq += q * q;
        }
        part[vars[i]] = sq;
        if (sq > local_max) local_max = sq;
    }
    atomic_max_nonneg(partmax, local_max);
    ctx.sync();
#undef A_
#undef NORM_
#undef NS_
#undef Q_
#undef PERM_
}

}  // namespace freedom

template <bool LANE>
__global__ void __launch_bounds__(256) freedom_kernel(const FreedomArgs a) {
    using namespace freedom;
    extern __shared__ double fr_lds[];
    const uint32_t G = LANE ? a.group : 1;
    unsigned long long* sysmax = reinterpret_cast<unsigned long long*>(fr_lds);  // largest column norm per system
    unsigned long long* partmax = sysmax + G;                                    // largest participation per system
    uint32_t* under = reinterpret_cast<uint32_t*>(partmax + G);                   // [G] (rounded up to doubles)
    double* red = fr_lds + 2 * G + (G + 1) / 2;
    double* wsl = red + 16;
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    for (uint64_t base = (uint64_t)blockIdx.x * G; base < a.batch; base += (uint64_t)gridDim.x * G) {
        const uint32_t nsys = (uint32_t)((a.batch - base) < G ? (a.batch - base) : G);
        for (uint32_t g = tid; g < G; g += nthr) {
            sysmax[g] = 0;
            partmax[g] = 0;
            under[g] = 0;
        }
        __syncthreads();
        if (a.x_caller) {  // fused: this workgroup's systems gathered and evaluated here (what gather_values_kernel + eval_kernel do)
            const Prog<uint32_t> P = make_prog<uint32_t>(a.prog, a.prog.base, a.prog.base);
            for (uint32_t idx = tid; idx < nsys * a.n; idx += nthr) {
                const uint32_t g = idx / a.n, v = idx - g * a.n;
                a.x_int[(base + g) * a.n + v] = a.x_caller[(base + g) * a.n + P.var_of[v]];
            }
            __threadfence_block();
            __syncthreads();
            const uint32_t n_cons = a.prog.n_cons;
            for (uint32_t idx = tid; idx < nsys * n_cons; idx += nthr) {
                const uint32_t g = idx / n_cons, ci = idx - g * n_cons;
                DevCon c;
                if (a.prog.packed) {
                    c = load_packed(P.pcons, P.con_weight, ci, false);
                    *reinterpret_cast<uint4*>(c.jloc) = P.patterns[c.nslots];
                } else {
                    c = load_con(P.cons + ci);
                }
                dev::JacWriter<double*> w;
                w.jv = a.jv_out + (base + g) * a.zj;
                w.jbase = c.jbase;
                const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
                w.loc[0] = loc[0], w.loc[1] = loc[1], w.loc[2] = loc[2], w.loc[3] = loc[3];
                w.weight = c.weight;
                (void)dev::con_jacobian<false>(c, (const double*)(a.x_int + (base + g) * a.n), w);
            }
            __threadfence_block();
            __syncthreads();
        }
        // largest |R_ii| of the pivoted QR = largest column norm; a variable in no row is a null vector by itself
        for (uint32_t idx = tid; idx < nsys * a.n; idx += nthr) {
            const uint32_t g = idx / a.n, v = idx - g * a.n;
            const double* jvs = a.jv + (base + g) * a.zj;
            const uint32_t p0 = a.col_ptr[v], p1 = a.col_ptr[v + 1];
            if (p0 == p1) {
                a.part[(base + g) * a.n + v] = 1.0;
                atomic_max_nonneg(&partmax[g], 1.0);
            } else {
                double s = 0.0;
                for (uint32_t p = p0; p < p1; ++p) {
                    const double e = jvs[a.col_slots[p]];
                    s += e * e;
                }
                atomic_max_nonneg(&sysmax[g], sqrt(s));
            }
        }
        __syncthreads();
        if (LANE) {
            for (uint32_t item = tid; item < nsys * a.ncomp; item += nthr) {
                const uint32_t g = item / a.ncomp, c = item - g * a.ncomp;
                const double tol = kFreedomRankTol * __longlong_as_double((long long)sysmax[g]);
                const FreedomComp cd = a.comps[c];
                freedom_component(LaneCtx{}, Ws{wsl + tid, nthr}, cd, a.items, a.jv + (base + g) * a.zj, tol,
                                  a.comp_vars + cd.var0, a.part + (base + g) * a.n, &partmax[g]);
            }
        } else {
            const double tol = kFreedomRankTol * __longlong_as_double((long long)sysmax[0]);
            double* w = a.gws ? a.gws + (size_t)blockIdx.x * a.ws : wsl;
            // (the component whose QR the step kernels left in the workspace goes first: the others reuse the workspace)
            for (uint32_t q = 0; q < a.ncomp; ++q) {
                const uint32_t c = !a.qr_done ? q : q == 0 ? a.qr_done - 1 : q <= a.qr_done - 1 ? q - 1 : q;
                const FreedomComp cd = a.comps[c];
                freedom_component(BlockCtx{red}, Ws{w, 1}, cd, a.items, a.jv + base * a.zj, tol, a.comp_vars + cd.var0,
                                  a.part + base * a.n, &partmax[0], a.qr_done != 0 && q == 0);
            }
        }
        __threadfence_block();
        __syncthreads();
        for (uint32_t idx = tid; idx < nsys * a.n; idx += nthr) {  // find_dof.rs:96-103
            const uint32_t g = idx / a.n;
            const double var_tol = kFreedomVarTol * __longlong_as_double((long long)partmax[g]);
            const bool free_var = a.part[base * a.n + idx] > var_tol * var_tol;
            const bool poisoned = a.qr_timed_out && a.qr_timed_out[base + g] != 0;
            a.mask[base * a.n + idx] = poisoned ? kFreedomPoisonedMask : free_var ? 1 : 0;
            if (free_var) atomicAdd(&under[g], 1u);
        }
        __syncthreads();
        if (a.n_under)
            for (uint32_t g = tid; g < nsys; g += nthr)
                a.n_under[base + g] = a.qr_timed_out && a.qr_timed_out[base + g] != 0 ? 0xFFFFFFFFu : under[g];
        __syncthreads();
    }
    if (a.done_flag && gridDim.x == 1) {  // (every thread's stores are acknowledged; the release store publishes them to the host)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- the pivoted QR of ONE large component, spread over the device ------------------------------------------------------
// A workgroup alone streams the trailing matrix at 12 GB/s (a few wavefronts' worth of loads in flight).  For systems
// that are one big component the host therefore runs the QR as a chain of small launches per Householder step --
// fr_pivot (one workgroup per system: pivot search on the running norms, column swap, reflector) and fr_apply (64
// columns x 16 row chunks per workgroup, all systems of the chunk side by side in grid.y) -- and then the ordinary
// kernel with `qr_done` for rank, null space and participation.  Same algorithm and pivot rule; the dot products and
// norms are summed as 16 partial sums, so R differs from the one-workgroup R in the last bits (the participation it
// leads to does not depend on the basis).  Workspace layout = freedom_component's row-major one.
struct FreedomStepArgs {
    double* gws;            // [systems][ws]
    const double* jv;       // [systems][zj]
    const uint32_t* items;  // the component's (slot, lcol * m + lrow) pairs
    uint32_t* done;         // [systems]: nothing left to eliminate (all remaining columns are zero)
    double* tau;            // [systems]: tau of the current step
    uint32_t ws, zj, m, n, item0, item1, k;
};

__global__ void __launch_bounds__(256) fr_init_kernel(const FreedomStepArgs a) {
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const uint32_t mn = a.m * a.n, oPerm = mn + 2 * a.n * a.n;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < mn; e += gridDim.x * blockDim.x) W[e] = 0.0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n; j += gridDim.x * blockDim.x) W[oPerm + j] = (double)j;
    // (fr_qrc_kernel's chunk area at the start of the null-space block: sequence numbers zero)
    const uint32_t chunk_doubles = a.k;  // fr_qrc_kernel's chunk area (the host's figure; 0: not in use)
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < chunk_doubles; e += gridDim.x * blockDim.x) W[mn + e] = 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) a.done[blockIdx.y] = 0;
}
__global__ void __launch_bounds__(256) fr_scatter_kernel(const FreedomStepArgs a) {
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const double* jv = a.jv + (size_t)blockIdx.y * a.zj;
    for (uint32_t it = a.item0 + blockIdx.x * blockDim.x + threadIdx.x; it < a.item1; it += gridDim.x * blockDim.x) {
        const uint32_t e = a.items[2 * it + 1], lcol = e / a.m;
        W[(size_t)(e - lcol * a.m) * a.n + lcol] = jv[a.items[2 * it]];
    }
}
__global__ void __launch_bounds__(256) fr_norms_kernel(const FreedomStepArgs a) {
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const uint32_t oTau = a.m * a.n + 2 * a.n * a.n + a.n;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n; j += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (uint32_t i = 0; i < a.m; ++i) {
            const double v = W[(size_t)i * a.n + j];
            s += v * v;
        }
        W[oTau + j] = s;
    }
}
__global__ void __launch_bounds__(256) fr_pivot_kernel(const FreedomStepArgs a) {
    __shared__ double red[16];
    if (a.done[blockIdx.x]) return;
    double* W = a.gws + (size_t)blockIdx.x * a.ws;
    const uint32_t m = a.m, n = a.n, k = a.k, oPerm = m * n + 2 * n * n, oTau = oPerm + n;
    const freedom::BlockCtx ctx{red};
    double bv = -1.0;
    uint32_t bj = k;
    for (uint32_t j = k + threadIdx.x; j < n; j += blockDim.x) {
        const double s = W[oTau + j];
        if (s > bv) {
            bv = s;
            bj = j;
        }
    }
    ctx.argmax(bv, bj);
    if (!(bv > 0.0)) {  // nothing left (or NaN): the remaining diagonal is exactly zero
        if (threadIdx.x == 0) a.done[blockIdx.x] = 1;
        return;
    }
    if (bj != k) {
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            const double t = W[(size_t)i * n + k];
            W[(size_t)i * n + k] = W[(size_t)i * n + bj];
            W[(size_t)i * n + bj] = t;
        }
        if (threadIdx.x == 0) {
            const double t = W[oPerm + k];
            W[oPerm + k] = W[oPerm + bj];
            W[oPerm + bj] = t;
            W[oTau + bj] = W[oTau + k];
        }
    }
    __syncthreads();
    const double norm = sqrt(bv);
    const double alpha = W[(size_t)k * n + k];
    const double beta = alpha >= 0.0 ? -norm : norm;
    const double denom = alpha - beta;
    __syncthreads();
    for (uint32_t i = k + 1 + threadIdx.x; i < m; i += blockDim.x) W[(size_t)i * n + k] /= denom;
    if (threadIdx.x == 0) {
        W[(size_t)k * n + k] = beta;
        a.tau[blockIdx.x] = (beta - alpha) / beta;
    }
}
// 64 columns per workgroup, the rows below k in 16 contiguous chunks (one per wavefront).
__global__ void __launch_bounds__(1024) fr_apply_kernel(const FreedomStepArgs a) {
    __shared__ double part[16][64];
    if (a.done[blockIdx.y]) return;
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const uint32_t m = a.m, n = a.n, k = a.k, oTau = m * n + 2 * n * n + n;
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t j = k + 1 + blockIdx.x * 64 + lane;
    const bool live = j < n;
    const uint32_t rows = m - (k + 1), per = (rows + 15) / 16;
    const uint32_t i0 = k + 1 + w * per, i1 = (i0 + per < m) ? i0 + per : m;
    const double tau = a.tau[blockIdx.y];
    double dot = 0.0;
    if (live)
        for (uint32_t i = i0; i < i1; ++i) dot += W[(size_t)i * n + k] * W[(size_t)i * n + j];
    part[w][lane] = dot;
    __syncthreads();
    dot = live ? W[(size_t)k * n + j] : 0.0;
    for (uint32_t q = 0; q < 16; ++q) dot += part[q][lane];
    dot *= tau;
    __syncthreads();
    double s = 0.0;
    if (live) {
        if (w == 0) W[(size_t)k * n + j] -= dot;
        for (uint32_t i = i0; i < i1; ++i) {
            const double v = W[(size_t)i * n + j] - dot * W[(size_t)i * n + k];
            W[(size_t)i * n + j] = v;
            s += v * v;
        }
    }
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && live) {
        s = 0.0;
        for (uint32_t q = 0; q < 16; ++q) s += part[q][lane];
        W[oTau + j] = s;
    }
}

// The whole pivoted QR of the WIDE layout as ONE cooperative launch (round 4).  The chain above is one launch pair per
// Householder step -- 2000 variables: 4000 launches, ~30 us per pair, 0.12 s -- although the two kernels only ever wait
// for each other.  Here every workgroup stays for all steps:
//   * the pivot search runs redundantly in every workgroup (the running norms are a few KB: same inputs, same deterministic
//     result, nothing to exchange);
//   * the column swap and the scaling of the reflector are one pass over the rows, dealt round-robin to the whole grid;
//   * the reflector is applied by tiles of kQrCols columns x kQrChunks row chunks per workgroup (16 columns = one 128-byte
//     line per row; a 2000-variable system has 125 workgroups streaming its trailing matrix instead of the chain's 32).  The
//     reflector itself goes to LDS once per step; a lane's chunk of its column (up to kQrKeep rows) is loaded with all loads
//     in flight, stays in registers between the dot product and the update, and is written once: the trailing matrix is
//     read once and written once per step.  A chunk's partial sums fold by shuffles inside a wavefront, then over the 16
//     wavefronts through LDS.
// The sums are formed in another (fixed) order than the chain's: deterministic, equal to rounding, held against the CPU restatement's
// dense QR by the same tests.  Two rendezvous per step among the workgroups of a system (qr_rendezvous; the launch is
// cooperative, so co-residency is the runtime's guarantee).  Measured per step of a 2000-variable system (wall_clock64
// around the phases, workgroup 0): pivot search 3.5 us, swap + scale 1.9, first rendezvous 2.8, apply 17.5, second
// rendezvous 9.3 (its wait for the slowest workgroup included) = 35 us against the chain's 61: 2000 variables 122 -> 72 ms
// per analysis, 800 variables 23.5 -> 16.8 ms.  The step is bound by memory latency and by the rendezvous' cache
// maintenance (every step ends with the L2's dirty lines written back and the next begins with cold lines: ~32 MB moved
// in ~25 us); what would change that is a blocked (BLAS-3) factorisation, which the reference's column pivoting on
// running norms does not allow without changing which columns are chosen on near-ties.
// What was tried on the way: the chain's own 16 chunks and one running sum per lane as a persistent kernel (bit-identical
// to the chain): 71 us per step, no gain; cooperative groups' grid.sync(): ~20 us each; 8-column tiles x 128 chunks with 16
// kept rows (half-line tiles, 250 workgroups): apply no faster and the rendezvous twice as long.
// grid = (workgroups per system, systems side by side); 1024 lanes.
constexpr uint32_t kQrCols = 16, kQrChunks = 64, kQrKeep = 32;
// The rendezvous of one system's workgroups inside fr_qr_kernel: a counter that only grows (every arrival adds one, the k-th
// rendezvous is over when it reaches k x workgroups), zeroed by fr_init_kernel.  One lane per workgroup arrives and polls;
// its agent-scope fences publish the workgroup's stores (they sit in the XCD's L2, which the write-back covers) and drop
// stale lines before anybody reads on.  (cooperative groups' grid.sync() does the same for the whole grid and took ~20 us
// with 125 workgroups of 1024 lanes: two of them were 40 of a step's 50 us.)
// Every lane first waits for its own stores to be acknowledged: the workgroup barrier does not (a barrier orders a workgroup's
// accesses among its own wavefronts, which share an L1), and lane 0's release only waits for lane 0's wavefront -- a store of
// another wavefront still on its way to the L2 when lane 0 writes the L2 back would stay there, unseen by the other XCDs, until
// somebody's next write-back.
__device__ __forceinline__ void qr_rendezvous(unsigned int* counter, unsigned int& target, unsigned int workgroups) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        target += workgroups;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void __launch_bounds__(1024) fr_qr_kernel(const FreedomStepArgs a, const uint32_t ndiag) {
    unsigned int* const arrivals = a.done + blockIdx.y;  // (zeroed by fr_init_kernel; the chain's "done" flag is not used here)
    unsigned int arrived = 0;
    __shared__ double red[32];
    __shared__ double part[16][kQrCols];
    __shared__ double colk[kQrChunks * kQrKeep + 2 * kQrKeep];
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const uint32_t m = a.m, n = a.n, oPerm = m * n + 2 * n * n, oTau = oPerm + n;
    const uint32_t G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    const uint32_t c = tid % kQrCols, w = tid / kQrCols;  // column of the tile, row chunk
    const uint32_t tile0 = wg;
    bool done = false;  // nothing left (or NaN): the remaining diagonal is exactly zero -- the same in every workgroup
    for (uint32_t k = 0; k < ndiag; ++k) {
        double tau = 0.0, beta = 0.0;
        uint32_t bj = k;
        if (!done) {
            double bv = -1.0;
            for (uint32_t j = k + tid; j < n; j += blockDim.x) {
                const double sq = W[oTau + j];
                if (sq > bv) {
                    bv = sq;
                    bj = j;
                }
            }
            // largest value, smallest index among equals; every lane gets the result (16 wavefronts: BlockCtx::argmax is for <= 8)
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(bv, off);
                const uint32_t oi = (uint32_t)__shfl_xor((int)bj, off);
                if (ov > bv || (ov == bv && oi < bj)) {
                    bv = ov;
                    bj = oi;
                }
            }
            __syncthreads();
            if ((tid & 63u) == 0) {
                red[tid >> 6] = bv;
                red[16 + (tid >> 6)] = (double)bj;
            }
            __syncthreads();
            bv = red[0];
            bj = (uint32_t)red[16];
            for (uint32_t q = 1; q < (blockDim.x >> 6); ++q) {
                const double ov = red[q];
                const uint32_t oi = (uint32_t)red[16 + q];
                if (ov > bv || (ov == bv && oi < bj)) {
                    bv = ov;
                    bj = oi;
                }
            }
            if (!(bv > 0.0)) {
                done = true;
            } else {
                // Row k itself is left alone in this pass -- every workgroup reads alpha = W(k, bj) here, and W(k, k) is read by
                // the tile of column bj below: the swap of row k happens where its entries are consumed (the apply phase
                // writes W(k, bj), W(k, k) = beta is stored after the step's second rendezvous).
                const double norm = sqrt(bv);
                const double alpha = W[(size_t)k * n + bj];  // (W(k, k) once swapped)
                beta = alpha >= 0.0 ? -norm : norm;
                const double denom = alpha - beta;
                tau = (beta - alpha) / beta;
                // (rows dealt round-robin to the workgroups: every lane's two loads are lines of their own -- column accesses to a
                // row-major matrix -- and two workgroups taking them all took 11.6 us of a 2000-variable step)
                for (uint32_t i = tid * G + wg; i < m; i += G * blockDim.x) {
                    if (i == k) continue;
                    const double vk = W[(size_t)i * n + k];
                    double nk = vk;
                    if (bj != k) {
                        nk = W[(size_t)i * n + bj];
                        W[(size_t)i * n + bj] = vk;
                    }
                    if (i > k) nk /= denom;
                    W[(size_t)i * n + k] = nk;
                }
                // (the chain also moves column k's norm to position bj; here that store would race with the other workgroups' pivot
                // searches of this step, and nobody needs it: the apply phase writes the norm of every column right of k anew)
                if (wg == 0 && tid == 0 && bj != k) {
                    const double t = W[oPerm + k];
                    W[oPerm + k] = W[oPerm + bj];
                    W[oPerm + bj] = t;
                }
            }
        }
        qr_rendezvous(arrivals, arrived, G);  // column k is the reflector (and the norm of the column that left position k sits at its new place)
        if (!done && n - k - 1 > 0) {
            const uint32_t rows = m - (k + 1), per = (rows + kQrChunks - 1) / kQrChunks;
            const uint32_t i0 = min(m, k + 1 + w * per), i1 = min(m, i0 + per), r0 = i0 - (k + 1);
            const uint32_t tiles = (n - k - 1 + kQrCols - 1) / kQrCols;
            const bool kept = per <= kQrKeep && n >= 8;  // (the rows below the diagonal fit colk with kQrKeep to spare)
            // the reflector (column k below the diagonal) once into LDS: both passes of every lane read it from there
            if (kept) {
                for (uint32_t i = k + 1 + tid; i < m + kQrKeep; i += blockDim.x) colk[i - (k + 1)] = i < m ? W[(size_t)i * n + k] : 0.0;
                __syncthreads();
            }
            for (uint32_t t = tile0; t < tiles; t += G) {
                const uint32_t j = k + 1 + t * kQrCols + c;
                const bool live = j < n;
                // The chunk's entries of column j stay in registers between the dot pass and the update pass (up to kQrKeep rows:
                // 2048 + rows), with all their loads in flight at once: the trailing matrix is read once and written once per step
                // instead of read twice (the step is bound by HBM: 2000 variables = up to 32 MB of trailing matrix, whose lines the
                // rendezvous' fences drop from the L2 every step).
                double keep[kQrKeep];
                const uint32_t at_j = i0 * n + j;
                double d0 = 0.0, d1 = 0.0;
                if (live && kept) {
#pragma unroll
                    for (uint32_t q = 0; q < kQrKeep; ++q) {
                        // (a uniform row base and one 32-bit lane offset for all rows; rows past the chunk's end are still inside
                        // the workspace -- Q follows the matrix -- and are read as zeros)
                        const double* const row = W + (size_t)q * n;
                        const double vj = row[at_j];
                        keep[q] = i0 + q < i1 ? vj : 0.0;
                    }
#pragma unroll
                    for (uint32_t q = 0; q < kQrKeep; q += 2) {
                        d0 += colk[r0 + q] * keep[q];
                        d1 += colk[r0 + q + 1] * keep[q + 1];
                    }
                } else if (live) {
                    uint32_t i = i0;
                    for (; i + 2 <= i1; i += 2) {
                        d0 += W[(size_t)i * n + k] * W[(size_t)i * n + j];
                        d1 += W[(size_t)(i + 1) * n + k] * W[(size_t)(i + 1) * n + j];
                    }
                    for (; i < i1; ++i) d0 += W[(size_t)i * n + k] * W[(size_t)i * n + j];
                }
                // (the kQrCols columns repeat every kQrCols lanes: the chunks of a wavefront are summed by shuffles first)
                double d = d0 + d1;
                for (uint32_t off = kQrCols; off < 64; off <<= 1) d += __shfl_xor(d, off);
                if ((tid & 63u) < kQrCols) part[tid >> 6][c] = d;
                __syncthreads();
                // (row k of column j: the swapped-in value -- the old W(k, k) -- for the column the pivot came from)
                const double rowk = !live ? 0.0 : (bj != k && j == bj) ? W[(size_t)k * n + k] : W[(size_t)k * n + j];
                double dot = rowk;
                for (uint32_t q = 0; q < 16; ++q) dot += part[q][c];
                dot *= tau;
                __syncthreads();
                double s0 = 0.0, s1 = 0.0;
                if (live && w == 0) W[(size_t)k * n + j] = rowk - dot;
                if (live && kept) {
#pragma unroll
                    for (uint32_t q = 0; q < kQrKeep; ++q) {
                        const double v = keep[q] - dot * colk[r0 + q];
                        if (i0 + q < i1) {
                            (W + (size_t)q * n)[at_j] = v;
                            if (q & 1) s1 += v * v; else s0 += v * v;
                        }
                    }
                } else if (live) {
                    uint32_t i = i0;
                    for (; i + 2 <= i1; i += 2) {
                        const double v0 = W[(size_t)i * n + j] - dot * W[(size_t)i * n + k];
                        const double v1 = W[(size_t)(i + 1) * n + j] - dot * W[(size_t)(i + 1) * n + k];
                        W[(size_t)i * n + j] = v0;
                        W[(size_t)(i + 1) * n + j] = v1;
                        s0 += v0 * v0;
                        s1 += v1 * v1;
                    }
                    for (; i < i1; ++i) {
                        const double v = W[(size_t)i * n + j] - dot * W[(size_t)i * n + k];
                        W[(size_t)i * n + j] = v;
                        s0 += v * v;
                    }
                }
                double sq = s0 + s1;
                for (uint32_t off = kQrCols; off < 64; off <<= 1) sq += __shfl_xor(sq, off);
                if ((tid & 63u) < kQrCols) part[tid >> 6][c] = sq;
                __syncthreads();
                if (tid < kQrCols && live) {
                    sq = 0.0;
                    for (uint32_t q = 0; q < 16; ++q) sq += part[q][c];
                    W[oTau + j] = sq;
                }
                __syncthreads();
            }
        }
        qr_rendezvous(arrivals, arrived, G);  // the trailing matrix and its norms are those of step k + 1
        if (!done && wg == 0 && tid == 0) W[(size_t)k * n + k] = beta;
    }
}

// The pivoted QR of the WIDE layout with the matrix RESIDENT ON CHIP (round 4, second form).  fr_qr_kernel above streams the
// trailing matrix through the device once per Householder step (32 MB for 2000 variables: 17 us of a step) between two
// rendezvous whose cache maintenance covers those 32 MB.  But the device's registers hold 128 MB: here every workgroup OWNS up
// to kQcCols whole columns for the whole factorisation -- a column is 128 lanes x kQcPer rows in registers, rows interleaved
// so that the rows still below the diagonal stay spread over the lanes -- and nothing of the matrix moves until the end:
//   * a step's dot products, updates and new column norms are local to the owning workgroup (no partial sums to exchange);
//   * the pivot is found from one (norm, position) candidate per workgroup: the largest norm, the smallest POSITION among equals
//     -- positions as the swaps of the serial algorithm would leave them, tracked by every workgroup in a table of its own, so
//     ties break exactly as in the chain.  Workgroup 0 gathers the candidates and scatters the winner, one line per workgroup;
//   * every workgroup scales ITS candidate column into a reflector and publishes it (tau in the slot of row k) while the winner
//     is being decided: when the others learn who won, the winner's reflector is already on its way -- two hops through
//     memory per step on the critical path instead of three, for up to 8 MB of speculative stores per step.
// Everything that crosses workgroups travels as self-validating 16-byte chunks (value, sequence number, position) moved by
// single device-coherent 128-bit accesses, like the reductions of the LM kernels' grid teams (lm_kernel.hip.hpp): a reader
// polls until the chunk carries the step's sequence number -- no atomics, no fences, no cache maintenance.  (A first version
// with two counter rendezvous per step, release / acquire fences around them, took 27 us per step whatever the size: 800
// variables 18.4 ms, slower than fr_qr_kernel.)  The chunk area is zeroed by fr_init_kernel; sequence numbers start at 1.
// R (every column at its final position) and the permutation are written once, at the end; the reflectors are not kept (the
// rank / null-space stage reads R and the permutation only).  Sums run in another fixed order than the chain's and than
// fr_qr_kernel's: deterministic, equal to rounding, held against the CPU restatement's dense QR by the same tests.
// Limits: m <= kQcRows rows, n <= kQcCols x workgroups columns; beyond, fr_qr_kernel serves.  The launch is cooperative
// (co-residency is the runtime's guarantee); the polls are bounded all the same (a time-out poisons the result with NaN).
// Measured (workgroup 1, wall-clock stamps, 2000 variables) with the OWNER publishing after the decision: a step was 12.6 us
// -- its candidate 0.8, waiting for the winner 3.3 (two hops through memory), owner + swap 0.2, waiting for the reflector 5.3
// (the owner's divisions and stores, one hop), the two passes 3.0 -- against fr_qr_kernel's 35: 2000 variables 72 -> 25.9 ms.
// With every workgroup publishing its candidate's reflector ahead of the decision: 9.5 us, 2000 variables **19 ms** per
// analysis (122 on the launch chain), 1700: 15.5, 800: 8.2 (16.3 streaming, 23.5 chain), 300: 2.6 (4.4).  Tried on top and
// not kept: a wavefront as 8 columns x 8 rows (reads of the reflector become broadcasts, but the scaling and its stores
// spread over 16 wavefronts: 25.9 -> 37 ms); row blocks above the diagonal skipped by scalar branches (the blocks' LDS
// reads no longer overlap: 31 ms); the reflector kept in registers between the passes (32 spilled registers).
// grid = (workgroups per system, systems side by side); 1024 lanes = kQcCols column groups of 128.
constexpr uint32_t kQcRows = 2048, kQcCols = 8, kQcSeg = 128, kQcPer = kQcRows / kQcSeg, kQcMaxWgs = 256;
constexpr uint32_t kQcSmallDoubles = 2 * kQcMaxWgs + 8 * kQcMaxWgs + 8 + 2;  // candidates | results (a line each) | flags, alignment
// (then the reflector slots: 2 sets x workgroups x rows chunks; the host sizes the area and fr_init_kernel zeroes it)
typedef unsigned int qc_chunk_t __attribute__((ext_vector_type(4)));  // (value lo, value hi, sequence number, position)
__device__ __forceinline__ void qc_store(qc_chunk_t* p, double v, unsigned int seq, unsigned int pos) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    qc_chunk_t c;
    c.x = (unsigned int)u, c.y = (unsigned int)(u >> 32), c.z = seq, c.w = pos;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory");
}
__device__ __forceinline__ qc_chunk_t qc_wait(const qc_chunk_t* p, unsigned int seq, unsigned int* dead) {
    qc_chunk_t c;
    for (unsigned int spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(p) : "memory");
        if (c.z == seq) break;
        if ((spins & 1023u) == 1023u && (spins >= (1u << 22) || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long nan = __builtin_bit_cast(unsigned long long, __builtin_nan(""));
            c.x = (unsigned int)nan, c.y = (unsigned int)(nan >> 32), c.w = 0;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return c;
}
__device__ __forceinline__ double qc_value(const qc_chunk_t& c) { return __builtin_bit_cast(double, ((unsigned long long)c.y << 32) | c.x); }
__device__ __forceinline__ double qc_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__global__ void __launch_bounds__(1024) fr_qrc_kernel(const FreedomStepArgs a, const uint32_t ndiag, const uint32_t C) {
    __shared__ double v_lds[kQcRows];
    __shared__ double part[kQcCols][2];
    __shared__ double wgc_v[kQcCols];
    __shared__ uint32_t wgc_p[kQcCols];
    __shared__ double red[32];
    __shared__ unsigned short col_at[kQcRows];  // position -> column, as the serial algorithm's swaps would leave it
    __shared__ double bcast[4];
    double* W = a.gws + (size_t)blockIdx.y * a.ws;
    const uint32_t m = a.m, n = a.n, oNS = m * n, oPerm = m * n + 2 * n * n, oTau = oPerm + n;
    const uint32_t G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    // the chunk area (the null-space block is not in use yet; zeroed by fr_init_kernel)
    // (16-byte aligned: a chunk must be ONE access; m x n may be odd)
    qc_chunk_t* const area = reinterpret_cast<qc_chunk_t*>(W + oNS + ((reinterpret_cast<uintptr_t>(W + oNS) >> 3) & 1u));
    qc_chunk_t* const candc = area;                              // [workgroup]: its candidate
    qc_chunk_t* const resc = candc + kQcMaxWgs;                  // [workgroup][4]: the winner, a 64-byte line each
    // "somebody gave up waiting": the system's word of FreedomStepArgs::done (zeroed by fr_init_kernel, not otherwise used on this
    // route) -- it outlives the kernel, freedom_kernel reads it as FreedomArgs::qr_timed_out
    unsigned int* const dead = a.done + blockIdx.y;
    // [parity of the step][workgroup][row]: the reflector every workgroup forms from ITS candidate column while the winner is
    // being decided (tau at row k) -- the winner's is then already on its way when the others learn who won.  Two sets: a
    // workgroup overwrites its slot two steps later, which it only reaches after everybody has sent the next step's candidate,
    // i.e. has read this one.
    const uint32_t m_slot = (m + 3u) & ~3u;
    qc_chunk_t* const slots = resc + 4 * kQcMaxWgs + 4;
    const uint32_t cg = tid >> 7, seg = tid & 127u, half = (tid >> 6) & 1u;
    const uint32_t col = wg * C + cg;
    const bool have = cg < C && col < n;
    double areg[kQcPer];
#pragma unroll
    for (uint32_t q = 0; q < kQcPer; ++q) {
        const uint32_t i = seg + kQcSeg * q;
        areg[q] = have && i < m ? W[(size_t)i * n + col] : 0.0;
    }
    double nrm = have ? W[oTau + col] : -1.0;
    uint32_t pos = col;
    bool active = have;
    for (uint32_t i = tid; i < n; i += blockDim.x) col_at[i] = (unsigned short)i;
    __syncthreads();
    for (uint32_t k = 0; k < ndiag; ++k) {
        const unsigned int seq = k + 1;
        // this workgroup's candidate: largest squared norm, smallest position among equals
        if (seg == 0 && cg < kQcCols) {
            wgc_v[cg] = active ? nrm : -1.0;
            wgc_p[cg] = active ? pos : 0xFFFFFFFFu;
        }
        __syncthreads();
        if (tid == 0) {
            double bv = -1.0;
            uint32_t bp = 0xFFFFFFFFu, bc = 0;
            for (uint32_t c2 = 0; c2 < kQcCols; ++c2) {
                const double v = wgc_v[c2];
                const uint32_t p2 = wgc_p[c2];
                if (v > bv || (v == bv && p2 < bp)) bv = v, bp = p2, bc = c2;
            }
            qc_store(candc + wg, bv, seq, bp);
            bcast[3] = bv > 0.0 ? (double)bc : -1.0;
        }
        qc_chunk_t* const my_slot = slots + ((size_t)(k & 1u) * G + wg) * m_slot;
        // the reflector of this workgroup's candidate (find_dof.rs' Householder step, as fr_pivot_kernel forms it), published
        // before anybody knows whether it wins; workgroup 0 decides first (everybody waits for that) and publishes after
        double bv = -1.0;
        uint32_t bp = 0xFFFFFFFFu;
        // (two turns of ONE copy of the code: workgroups other than 0 publish, then wait for the decision; workgroup 0 decides
        // first -- everybody waits for that -- and publishes after)
#pragma unroll 1
        for (uint32_t turn = 0; turn < 2; ++turn) {
            if ((turn == 0) == (wg != 0)) {
                __syncthreads();
                const int bc = (int)bcast[3];
                const bool cand_group = bc >= 0 && cg == (uint32_t)bc;
                if (cand_group && seg == (k & 127u)) bcast[0] = areg[k >> 7];  // alpha = the entry of row k
                __syncthreads();
                if (cand_group) {
                    const double norm = sqrt(nrm), alpha = bcast[0];
                    const double beta = alpha >= 0.0 ? -norm : norm, denom = alpha - beta;
#pragma unroll
                    for (uint32_t q = 0; q < kQcPer; ++q) {
                        const uint32_t i = seg + kQcSeg * q;
                        if (i > k && i < m) qc_store(my_slot + i, areg[q] / denom, seq, 0);
                        if (i == k) qc_store(my_slot + k, (beta - alpha) / beta, seq, 0);  // tau rides in the slot of row k (v_k = 1)
                    }
                }
            }
            if (turn == 0) {
                if (wg == 0) {  // gather, decide, scatter
                    if (tid < G) {
                        const qc_chunk_t c = qc_wait(candc + tid, seq, dead);
                        bv = qc_value(c);
                        bp = c.w;
                        if (!(bv >= 0.0)) bv = -1.0, bp = 0xFFFFFFFFu;  // (NaN norms never win, like the chain's `s > bv`)
                    }
                    for (int off = 32; off > 0; off >>= 1) {
                        const double ov = __shfl_xor(bv, off);
                        const uint32_t op = (uint32_t)__shfl_xor((int)bp, off);
                        if (ov > bv || (ov == bv && op < bp)) bv = ov, bp = op;
                    }
                    if ((tid & 63u) == 0) {
                        red[tid >> 6] = bv;
                        red[16 + (tid >> 6)] = (double)bp;
                    }
                    __syncthreads();
                    bv = red[0];
                    bp = (uint32_t)red[16];
                    for (uint32_t q = 1; q < 16; ++q) {
                        const double ov = red[q];
                        const uint32_t op = (uint32_t)red[16 + q];
                        if (ov > bv || (ov == bv && op < bp)) bv = ov, bp = op;
                    }
                    if (tid > 0 && tid < G) qc_store(resc + 4 * tid, bv, seq, bp);
                } else {
                    if (tid == 0) {
                        const qc_chunk_t c = qc_wait(resc + 4 * wg, seq, dead);
                        bcast[1] = qc_value(c);
                        bcast[2] = (double)c.w;
                    }
                    __syncthreads();
                    bv = bcast[1];
                    bp = (uint32_t)bcast[2];
                }
            }
        }
        if (!(bv > 0.0)) break;  // nothing left (or NaN, or a time-out): the remaining diagonal is exactly zero -- the same decision everywhere
        const uint32_t pc = col_at[bp], ck = col_at[k];
        __syncthreads();
        if (tid == 0) {
            col_at[k] = (unsigned short)pc;
            col_at[bp] = (unsigned short)ck;
        }
        const bool mine = have && col == pc;  // this column group holds the pivot column
        if (have && col == ck) pos = bp;
        if (mine) pos = k;
        if (mine) {  // the winner's column: its diagonal entry of R, and out of the running (its reflector is published already)
            const double norm = sqrt(bv);
#pragma unroll
            for (uint32_t q = 0; q < kQcPer; ++q)
                if (seg + kQcSeg * q == k) areg[q] = areg[q] >= 0.0 ? -norm : norm;
            active = false;
        }
        const qc_chunk_t* const vch = slots + ((size_t)(k & 1u) * G + pc / C) * m_slot;
        {  // the reflector: a lane's (up to) two chunks requested together, one round trip instead of two
            const uint32_t i0 = k + tid, i1 = i0 + 1024u;
            if (i1 < m) {
                qc_chunk_t c0, c1;
                for (unsigned int spins = 0;; ++spins) {
                    asm volatile(
                        "global_load_dwordx4 %0, %2, off sc0 sc1\n\t"
                        "global_load_dwordx4 %1, %3, off sc0 sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(c0), "=&v"(c1)
                        : "v"(vch + i0), "v"(vch + i1)
                        : "memory");
                    if (c0.z == seq && c1.z == seq) break;
                    if ((spins & 1023u) == 1023u && (spins >= (1u << 22) || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        __hip_atomic_store(dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const unsigned long long nan = __builtin_bit_cast(unsigned long long, __builtin_nan(""));
                        c0.x = c1.x = (unsigned int)nan, c0.y = c1.y = (unsigned int)(nan >> 32);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                v_lds[i0] = qc_value(c0);
                v_lds[i1] = qc_value(c1);
            } else if (i0 < m) {
                v_lds[i0] = qc_value(qc_wait(vch + i0, seq, dead));
            }
        }
        __syncthreads();
        const double tau = v_lds[k];
        double dot = 0.0;
        if (active) {
#pragma unroll
            for (uint32_t q = 0; q < kQcPer; ++q) {
                const uint32_t i = seg + kQcSeg * q;
                if (i > k && i < m) dot += v_lds[i] * areg[q];
                if (i == k) dot += areg[q];
            }
        }
        dot = qc_wave_sum(dot);
        if ((tid & 63u) == 0) part[cg][half] = dot;
        __syncthreads();
        dot = (part[cg][0] + part[cg][1]) * tau;
        __syncthreads();
        double sq = 0.0;
        if (active) {
#pragma unroll
            for (uint32_t q = 0; q < kQcPer; ++q) {
                const uint32_t i = seg + kQcSeg * q;
                if (i > k && i < m) {
                    const double v = areg[q] - dot * v_lds[i];
                    areg[q] = v;
                    sq += v * v;
                }
                if (i == k) areg[q] -= dot;
            }
        }
        sq = qc_wave_sum(sq);
        if ((tid & 63u) == 0) part[cg][half] = sq;
        __syncthreads();
        if (active) nrm = part[cg][0] + part[cg][1];
        __syncthreads();
    }
    // R at the columns' final positions, and the permutation
    if (have) {
#pragma unroll
        for (uint32_t q = 0; q < kQcPer; ++q) {
            const uint32_t i = seg + kQcSeg * q;
            if (i < m) W[(size_t)i * n + pos] = areg[q];
        }
    }
    if (wg == 0)
        for (uint32_t p2 = tid; p2 < n; p2 += blockDim.x) W[oPerm + p2] = (double)col_at[p2];
}

}  // namespace ezpz
