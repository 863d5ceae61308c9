// The FRONTAL solve kernel: the Levenberg-Marquardt loop of one constraint system (reference ezpz/src/solver/newton.rs:29-145,
// solver.rs:318-440, lib.rs:305-327 -- the same control flow as lm_kernel.hip.hpp) with the linear solve
// (JtJ + lambda I) d = -Jt r (newton.rs:73-102: faer's sparse matmul, Llt::try_new_with_symbolic, solve) as a MULTIFRONTAL
// supernodal Cholesky factorisation on dense fronts (front_types.hpp, fronts.cpp):
//
//   * once per linear solve all lanes of the workgroup assemble every front's entries of JtJ and -Jt r from the Jacobian values of
//     the constraints whose earliest variable is eliminated there (one lane per entry, operand pairs streamed from the plan);
//   * a wavefront owns a front: it adds its children's update matrices, gathered by destination from the front's source stream
//     (extend-add; children in other workgroups through row maps, as chunks), factors the K pivot columns in registers (lane = row, pivots and
//     multipliers by v_readlane, the right-hand side as row S so that the forward substitution rides along), and leaves the
//     Schur complement of the rows below as its own update matrix;
//   * every wavefront of the workgroup runs its own list of fronts (the planner's schedule: list scheduling on the cost model) and
//     waits only for what its next front needs -- a counter in LDS that the front's children sign in on; the backward substitution
//     runs the same way top down (a front waits for its parent's stamp), a front solving its K unknowns from its panel;
//   * a system too large for one CU's LDS -- or too slow on one CU -- is cut at the top of the tree: whole subtrees go to
//     workgroups 1 .. G-1, the top to workgroup 0; update matrices and steps cross workgroups as self-validating 16-byte chunks
//     (grid_ops.hip.hpp), the LM control's sums gather at workgroup 0 and scatter.
//
// The same factorisation answers FreedomAnalysis (find_dof.rs:31-103) for the systems it serves: in PROBE mode (FrontArgs::probe_m)
// the kernel applies lambda (JtJ + lambda I)^-1 -- the projector onto null(J) as lambda -> 0 -- to vectors instead of running the
// LM loop (freedom.hip: freedom_by_probes).
//
// State per workgroup (LDS): x, d, r, r_next, Jacobian values, the fronts' panels (the factor), every front's update matrix.
// Results are those of a valid Cholesky factorisation in another elimination order than the list walks': coordinates at the
// 1e-6 bar of connected sketches, not bitwise.
#pragma once
#include <hip/hip_runtime.h>

#include "constraint_eval.hip.hpp"
#include "front_types.hpp"
#include "grid_ops.hip.hpp"
#include "wave_ops.hip.hpp"

namespace ezpz {
namespace frontal {

using namespace dev;

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// lanes of one wavefront execute in lockstep and the LDS serves a wavefront's accesses in issue order: only the compiler must
// not move accesses across
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double readlane_f64(double v, uint32_t l) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, (int)l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), (int)l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// 1 / sqrt(p) from the hardware's estimate y0 (about 26 bits) and ONE third-order step: with e = 1 - p y0^2,
// y1 = y0 (1 + e / 2 + 3 e^2 / 8) leaves an error of order e^3 -- below the last bit -- in four dependent operations where two
// coupled Newton steps are seven; the pivot chain of a front is this sequence K times.
__device__ __forceinline__ double rsqrt_newton(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double t = p * y0;
    const double e = __builtin_fma(-t, y0, 1.0);
    const double s = __builtin_fma(e, 0.375, 0.5);
    const double u = y0 * e;
    return __builtin_fma(u, s, y0);
}

// What a wavefront needs to work on a front of its workgroup.
struct Ctx {
    double* ws;                  // workspace (LDS)
    const FrontDesc* descs;      // staged tables (LDS)
    const FrontChild* children;
    const uint16_t* rows;
    const uint32_t* exports;
    const uint8_t* maps;
    const uint16_t* tri;         // entry -> (a | b << 8) of a packed lower triangle
    const uint32_t* stream;      // assembly streams (staged tables, LDS)
    gridchunk_t* chunks;         // the system's scratch chunks (null on one workgroup)
    int* dead;
    uint32_t l_jv, l_d, l_upool;
#ifdef EZPZ_STAMPS
    unsigned long long* stamps;  // diagnostic builds: (id, cycle) pairs of workgroup 0, thread 0
    int* stamp_n;
#endif
};

#ifdef EZPZ_STAMPS
#define FRONT_CX_STAMP(cx, id)                                                               \
    do {                                                                                     \
        if ((cx).stamps && blockIdx.x == 0 && threadIdx.x == 0 && *(cx).stamp_n < 2000) {     \
            (cx).stamps[2 * *(cx).stamp_n] = (id);                                           \
            (cx).stamps[2 * *(cx).stamp_n + 1] = __builtin_readcyclecounter();               \
            ++*(cx).stamp_n;                                                                 \
        }                                                                                    \
    } while (0)
#else
#define FRONT_CX_STAMP(cx, id) \
    do {                       \
    } while (0)
#endif

__device__ __forceinline__ uint32_t tri_index(uint32_t a, uint32_t b) { return a * (a + 1) / 2 + b; }

// Assembly + extend-add + partial factorisation + Schur complement of front k by the calling wavefront.  Returns true when a
// pivot was not positive (LltError::Numeric, newton.rs:93-99).
template <int KMAX>
__device__ __forceinline__ bool front_pivots(double* P, uint32_t K, uint32_t S, int lane) {
    const uint32_t S1 = S + 1;
    const uint32_t rr = (uint32_t)lane <= S ? (uint32_t)lane : S;
    double a[KMAX];
#pragma unroll
    for (int c = 0; c < KMAX; ++c) a[c] = (uint32_t)c < K ? P[c * S1 + rr] : 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
        if ((uint32_t)j < K) {
            const double piv = readlane_f64(a[j], j);
            if (!(piv > 0.0)) bad = true;
            const double rinv = rsqrt_newton(piv);
            const double l = a[j] * rinv;
            a[j] = lane == j ? rinv : l;  // (the factor's diagonal is kept as 1 / d_j)
#pragma unroll
            for (int k = j + 1; k < KMAX; ++k) a[k] = __builtin_fma(-l, readlane_f64(l, k), a[k]);
        }
    }
    if ((uint32_t)lane <= S) {
#pragma unroll
        for (int c = 0; c < KMAX; ++c)
            if ((uint32_t)c < K) P[c * S1 + lane] = a[c];
    }
    return bad;
}

// A front's descriptor from the staged tables: three 16-byte LDS reads, then every field in a scalar register.
struct DescRegs {
    uint32_t K, S, n_child, flags, panel, upd, rows, child0, src_off, src_n, src_v, up_chunk, exp0;
};
__device__ __forceinline__ DescRegs load_desc(const FrontDesc* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 q0 = q[0], q1 = q[1], q2 = q[2];
    DescRegs d;
    d.K = uni(q0.x) & 0xFFFFu;
    d.S = uni(q0.x) >> 16;
    d.n_child = uni(q0.y) & 0xFFFFu;
    d.flags = uni(q0.y) >> 16;
    d.panel = uni(q0.z);
    d.upd = uni(q0.w);
    d.rows = uni(q1.x);
    d.child0 = uni(q1.y);
    d.src_off = uni(q1.z);
    d.src_n = uni(q1.w) & 0xFFFFu;
    d.src_v = uni(q2.x);
    d.up_chunk = uni(q2.y);
    d.exp0 = uni(q2.z);
    return d;
}

// Schur complement of the rows below the pivots: U[e] -= sum_k L[a][k] L[b][k], every operand of an entry in flight at once.
template <int KMAX>
__device__ __forceinline__ void front_schur(const Ctx& cx, const DescRegs& d, int lane, unsigned int epoch) {
    const uint32_t K = d.K, S1 = d.S + 1, R = d.S - K, nU = (R + 1) * (R + 2) / 2;
    double* const ws = cx.ws;
    const uint32_t o_pk = d.panel + K, o_u = d.upd;
    const bool remote_parent = (d.flags & FRONT_REMOTE_PARENT) != 0;
    for (uint32_t e = lane; e + 1 < nU; e += 64) {  // (the last entry pairs the right-hand side's row with itself: not needed)
        const uint32_t ab = cx.tri[e];
        const uint32_t ia = o_pk + (ab & 0xFFu), ib = o_pk + (ab >> 8);
        double va[KMAX], vb[KMAX];
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const bool in = (uint32_t)k < K;
            va[k] = in ? ws[ia + k * S1] : 0.0;
            vb[k] = in ? ws[ib + k * S1] : 0.0;
        }
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; k += 2) {
            acc0 = __builtin_fma(va[k], vb[k], acc0);
            if (k + 1 < KMAX) acc1 = __builtin_fma(va[k + 1], vb[k + 1], acc1);
        }
        const double v = ws[o_u + e] - (acc0 + acc1);
        if (remote_parent)
            grid_store(cx.chunks + d.up_chunk + e, v, epoch);
        else
            ws[o_u + e] = v;
    }
}

// One trip of the workgroup's assembly stream with W operand words per entry: every load in flight at once.
template <int W>
__device__ __forceinline__ void asm_trip(double* ws, const uint32_t* st, int lane, uint32_t hdr, uint32_t o_j, uint32_t l_r, uint32_t o_pan,
                                         double lambda) {
    uint32_t op[W > 0 ? W : 1];
#pragma unroll
    for (int q = 0; q < W; ++q) op[q] = st[64 * (1 + q) + lane];
    const bool rhs = (hdr & FASM_RHS) != 0;
    const uint32_t o_b = rhs ? l_r : o_j;
    double va[W > 0 ? W : 1], vb[W > 0 ? W : 1];
#pragma unroll
    for (int q = 0; q < W; ++q) va[q] = ws[o_j + (op[q] & 0xFFFFu)], vb[q] = ws[o_b + (op[q] >> 16)];
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int q = 0; q < W; q += 2) {
        acc0 = __builtin_fma(va[q], vb[q], acc0);
        if (q + 1 < W) acc1 = __builtin_fma(va[q + 1], vb[q + 1], acc1);
    }
    double acc = acc0 + acc1;
    if (rhs) acc = -acc;
    if (hdr & FASM_DIAG) acc += lambda;
    if (!(hdr & FASM_NOP)) ws[o_pan + (hdr & 0xFFFFu)] = acc;
}
// The assembly of a linear solve (newton.rs:73-84: JtJ + lambda I and -Jt r), all wavefronts of the workgroup: panels and
// update matrices zeroed, then every element the workgroup's constraints contribute to, one lane per element.
__device__ __forceinline__ void assemble(const Ctx& cx, const FrontWg& W, double lambda, uint32_t l_r) {
    double* const ws = cx.ws;
    {
        double2* const Z = reinterpret_cast<double2*>(ws + W.l_panels);
        const uint32_t n2 = (W.ws_doubles - W.l_panels) / 2;
        const double2 z = {0.0, 0.0};
        for (uint32_t i = threadIdx.x; i < n2; i += blockDim.x) Z[i] = z;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t nwaves = blockDim.x >> 6, o_j = cx.l_jv, o_pan = W.l_panels;
    const uint32_t* const offs = cx.stream + W.asm_word0;
    for (uint32_t t = uni(threadIdx.x >> 6); t < W.asm_trips; t += nwaves) {
        const uint32_t* st = cx.stream + uni(offs[t]);
        const uint32_t hdr = st[lane];
        const uint32_t w = uni(hdr >> 24);
        switch (w) {
        case 0: asm_trip<0>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 1: asm_trip<1>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 2: asm_trip<2>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 3: asm_trip<3>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 4: asm_trip<4>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 5: asm_trip<5>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        case 6: asm_trip<6>(ws, st, lane, hdr, o_j, l_r, o_pan, lambda); break;
        default: {
            const bool rhs = (hdr & FASM_RHS) != 0;
            const uint32_t o_b = rhs ? l_r : o_j;
            double acc = 0.0;
            for (uint32_t q = 0; q < w; ++q) {
                const uint32_t op = st[64 * (1 + q) + lane];
                acc = __builtin_fma(ws[o_j + (op & 0xFFFFu)], ws[o_b + (op >> 16)], acc);
            }
            if (rhs) acc = -acc;
            if (hdr & FASM_DIAG) acc += lambda;
            if (!(hdr & FASM_NOP)) ws[o_pan + (hdr & 0xFFFFu)] = acc;
        }
        }
    }
    __syncthreads();
}

// One trip of a front's source stream with V source words (2 V sources) per entry.
template <int V>
__device__ __forceinline__ void src_trip(double* ws, const uint32_t* st, int lane, uint32_t hdr, uint32_t o_pan) {
    uint32_t x[V];
#pragma unroll
    for (int q = 0; q < V; ++q) x[q] = st[64 * (1 + q) + lane];
    const uint32_t dst = o_pan + (hdr & 0xFFFFu);
    double v0[V], v1[V];
#pragma unroll
    for (int q = 0; q < V; ++q) v0[q] = ws[o_pan + (x[q] & 0xFFFFu)], v1[q] = ws[o_pan + (x[q] >> 16)];
    double acc = ws[dst];
#pragma unroll
    for (int q = 0; q < V; ++q) acc += v0[q] + v1[q];
    if (!(hdr & FASM_NOP)) ws[dst] = acc;
}

__device__ __forceinline__ bool front_factor(const Ctx& cx, uint32_t k, int lane, uint32_t o_pan, unsigned int epoch) {
    const DescRegs d = load_desc(cx.descs + k);
    const uint32_t K = d.K, S = d.S, S1 = S + 1, R = S - K;
    // (every address an index from the workspace's base: a select between two LDS pointers would make them generic pointers)
    double* const ws = cx.ws;
    const uint32_t o_p = d.panel, o_u = d.upd;
    FRONT_CX_STAMP(cx, 100);
    // ---- extend-add: the elements of this workgroup's own children's update matrices, gathered by destination -----------------------
    {
        const uint32_t* st = cx.stream + d.src_off;
        for (uint32_t e0 = 0, tr = 0; e0 < d.src_n; e0 += 64, ++tr) {
            const uint32_t v = (d.src_v >> (8 * (tr < 3 ? tr : 3))) & 0xFFu;
            const uint32_t hdr = st[lane];
            switch (v) {
            case 1: src_trip<1>(ws, st, lane, hdr, o_pan); break;
            case 2: src_trip<2>(ws, st, lane, hdr, o_pan); break;
            case 3: src_trip<3>(ws, st, lane, hdr, o_pan); break;
            case 4: src_trip<4>(ws, st, lane, hdr, o_pan); break;
            default: {
                const uint32_t dst = o_pan + (hdr & 0xFFFFu);
                double acc = ws[dst];
                for (uint32_t q = 0; q < v; ++q) {
                    const uint32_t x = st[64 * (1 + q) + lane];
                    acc += ws[o_pan + (x & 0xFFFFu)] + ws[o_pan + (x >> 16)];
                }
                if (!(hdr & FASM_NOP)) ws[dst] = acc;
            }
            }
            st += 64 * (1 + v);
        }
    }
    wave_sync();
    FRONT_CX_STAMP(cx, 101);
    // ---- children in other workgroups: their update matrices arrive as chunks, one child after the other -------------------------------
    for (uint32_t c = 0; c < d.n_child; ++c) {
        const uint4 q = *reinterpret_cast<const uint4*>(cx.children + d.child0 + c);
        // (without the last entry: the right-hand side's row against itself is nobody's)
        const uint32_t upd = uni(q.x), rc1 = uni(q.y) & 0xFFFFu, n_c = rc1 * (rc1 + 1) / 2 - 1;
        const uint8_t* const map = cx.maps + uni(q.z);
        for (uint32_t e = lane; e < n_c; e += 64) {
            const uint32_t ab = cx.tri[e];
            const uint32_t i = map[ab & 0xFFu], j = map[ab >> 8];
            const double val = grid_wait(cx.chunks + upd + e, epoch, cx.dead);
            const uint32_t dst = j < K ? o_p + j * S1 + i : o_u + tri_index(i - K, j - K);
            ws[dst] += val;
        }
        wave_sync();
    }
    FRONT_CX_STAMP(cx, 102);
    // ---- the K pivot columns in registers ---------------------------------------------------------------------------------------------
    bool bad;
    if (K <= 4)
        bad = front_pivots<4>(ws + o_p, K, S, lane);
    else if (K <= 8)
        bad = front_pivots<8>(ws + o_p, K, S, lane);
    else
        bad = front_pivots<16>(ws + o_p, K, S, lane);
    wave_sync();
    FRONT_CX_STAMP(cx, 103);
    // ---- Schur complement of the rows below (row R = right-hand side) -------------------------------------------------------------------
    if (R) {
        if (K <= 4)
            front_schur<4>(cx, d, lane, epoch);
        else if (K <= 8)
            front_schur<8>(cx, d, lane, epoch);
        else
            front_schur<16>(cx, d, lane, epoch);
    }
    FRONT_CX_STAMP(cx, 104);
    return bad;
}

// Backward substitution of front k: its K unknowns from y (row S of the panel), the steps of the rows below and the panel.
template <int KMAX>
__device__ __forceinline__ double front_bwd_body(const Ctx& cx, const DescRegs& d, uint32_t K, uint32_t S, int lane, unsigned int epoch,
                                                 double dmax) {
    const uint32_t S1 = S + 1;
    const double* const P = cx.ws + d.panel;
    double* const dv = cx.ws + cx.l_d;
    const uint16_t* const frow = cx.rows + d.rows;
    const uint32_t r = (uint32_t)lane;
    const double xr = (r >= K && r < S) ? dv[frow[r]] : 0.0;
    const uint32_t kc = r < K ? r : K - 1;  // this lane's column
    const double* const col = P + kc * S1;
    double t = col[S];
    for (uint32_t rr = K; rr < S; ++rr) t = __builtin_fma(-col[rr], readlane_f64(xr, rr), t);
    double lcol[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) lcol[j] = (uint32_t)j < K ? col[j] : 0.0;  // (row j of column kc; its diagonal holds 1 / d)
    const double rinv = col[kc];
    double xown = 0.0;
#pragma unroll
    for (int j = KMAX - 1; j >= 0; --j) {
        if ((uint32_t)j < K) {
            const double xj = readlane_f64(t * rinv, j);
            if (r < (uint32_t)j) t = __builtin_fma(-lcol[j], xj, t);
            if (r == (uint32_t)j) xown = xj;
        }
    }
    if (r < K) {
        dv[frow[r]] = xown;
        dmax = fmax_abs(dmax, xown);
        if (d.flags & FRONT_EXPORTS) {
            const uint32_t ch = cx.exports[d.exp0 + r];
            if (ch != 0xFFFFFFFFu) grid_store(cx.chunks + ch, xown, epoch);
        }
    }
    return dmax;
}
__device__ __forceinline__ double front_bwd(const Ctx& cx, uint32_t k, int lane, unsigned int epoch, double dmax) {
    const DescRegs d = load_desc(cx.descs + k);
    const uint32_t K = d.K, S = d.S;
    if (K <= 4) return front_bwd_body<4>(cx, d, K, S, lane, epoch, dmax);
    if (K <= 8) return front_bwd_body<8>(cx, d, K, S, lane, epoch, dmax);
    return front_bwd_body<16>(cx, d, K, S, lane, epoch, dmax);
}

// Four workgroup- (and, on several workgroups, system-) wide reductions for one rendezvous: a sum, two NaN-ignoring maxima, a
// sum; every lane gets all four.  Fixed trees: deterministic.
struct Red {
    double* buf;  // LDS: 2 x 64 doubles
    int flip;
    unsigned char* scratch;  // the system's scratch (several workgroups), else null
    uint32_t G, wg;
    unsigned int seq;
    int* dead;
    __device__ __forceinline__ void reduce(double& v0, double& v1, double& v2, double& v3) {
        const int tid = threadIdx.x, wave = tid >> 6, nwaves = (int)(blockDim.x >> 6);
        v0 = reduce_wave_to_last_lane(v0, OpSum());
        v1 = reduce_wave_to_last_lane(v1, OpMax());
        v2 = reduce_wave_to_last_lane(v2, OpMax());
        v3 = reduce_wave_to_last_lane(v3, OpSum());
        double* b = buf + (flip ? 64 : 0);
        flip ^= 1;
        if ((tid & 63) == 63) b[wave] = v0, b[16 + wave] = v1, b[32 + wave] = v2, b[48 + wave] = v3;
        __syncthreads();
        v0 = b[0], v1 = b[16], v2 = b[32], v3 = b[48];
        for (int w = 1; w < nwaves; ++w) {
            v0 = v0 + b[w];
            v1 = fmax_nc(v1, b[16 + w]);
            v2 = fmax_nc(v2, b[32 + w]);
            v3 = v3 + b[48 + w];
        }
        if (G > 1) {
            const unsigned int s = ++seq, par = s & 1u;
            gridchunk_t* const arr = reinterpret_cast<gridchunk_t*>(scratch + sizeof(FrontScratchHead)) + par * kFrontRedValues * kFrontMaxWgs;
            gridchunk_t* const out = reinterpret_cast<gridchunk_t*>(scratch + sizeof(FrontScratchHead) + kFrontScratchRedBytes) +
                                     par * kFrontMaxWgs * kFrontRedValues;
            if (tid < 4) grid_store(arr + tid * kFrontMaxWgs + wg, tid == 0 ? v0 : tid == 1 ? v1 : tid == 2 ? v2 : v3, s);
            double* b2 = buf + (flip ? 64 : 0);
            flip ^= 1;
            if (wg == 0) {
                if (tid < 64) {
                    double w0 = 0.0, w1 = __builtin_nan(""), w2 = __builtin_nan(""), w3 = 0.0;
                    if ((uint32_t)tid < G) {
                        w0 = grid_wait(arr + 0 * kFrontMaxWgs + tid, s, dead);
                        w1 = grid_wait(arr + 1 * kFrontMaxWgs + tid, s, dead);
                        w2 = grid_wait(arr + 2 * kFrontMaxWgs + tid, s, dead);
                        w3 = grid_wait(arr + 3 * kFrontMaxWgs + tid, s, dead);
                    }
                    w0 = reduce_lanes<64>(w0, OpSum());
                    w1 = reduce_lanes<64>(w1, OpMax());
                    w2 = reduce_lanes<64>(w2, OpMax());
                    w3 = reduce_lanes<64>(w3, OpSum());
                    if ((uint32_t)tid < G && tid > 0) {
                        grid_store(out + tid * kFrontRedValues + 0, w0, s);
                        grid_store(out + tid * kFrontRedValues + 1, w1, s);
                        grid_store(out + tid * kFrontRedValues + 2, w2, s);
                        grid_store(out + tid * kFrontRedValues + 3, w3, s);
                    }
                    if (tid == 0) b2[0] = w0, b2[1] = w1, b2[2] = w2, b2[3] = w3;
                }
            } else {
                if (tid < 4) b2[tid] = grid_wait(out + wg * kFrontRedValues + tid, s, dead);
            }
            __syncthreads();
            v0 = b2[0], v1 = b2[1], v2 = b2[2], v3 = b2[3];
        }
    }
};

}  // namespace frontal

#ifdef EZPZ_STAMPS
#define FRONT_STAMP(id)                                                          \
    do {                                                                         \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && stamp_n < 2000) { \
            a.stamps[2 * stamp_n] = (id);                                        \
            a.stamps[2 * stamp_n + 1] = __builtin_readcyclecounter();            \
            ++stamp_n;                                                           \
        }                                                                        \
    } while (0)
#else
#define FRONT_STAMP(id) \
    do {                \
    } while (0)
#endif

// (512 lanes: two wavefronts per SIMD.  1024 -- four per SIMD at 128 registers, the non-linear evaluators spilling a little -- was
// measured twice and not kept: with a barrier per level of the tree, one solve of 300 variables 191 -> 193 us, 800: 2.09 -> 2.30 ms;
// with the wavefronts' schedules 50 / 150 / 300 / 500 / 800 / 2000 / 5000 / 10 000 variables 84 -> 107, 96 -> 114, 179 -> 171, 347 ->
// 413, 1844 -> 2038, 451 -> 509, 2659 -> 2703, 919 -> 798 us)
template <bool LIN>
__global__ void __launch_bounds__(512, 1) front_solve_kernel(const FrontArgs a) {
    using namespace frontal;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = (uint32_t)tid >> 6, nwaves = blockDim.x >> 6;
    const uint32_t G = a.n_wgs, wg = blockIdx.x % G, slot = blockIdx.x / G, n_slots = gridDim.x / G;
    const FrontWg& W = reinterpret_cast<const FrontWg*>(a.plan)[wg];
    // ---- LDS: [staged tables][workspace][reduction scratch 128 doubles][triangle table 2080 x u16][ints] -------------------------------
    unsigned char* const tab = reinterpret_cast<unsigned char*>(smem);
    double* const ws = smem + a.tab_lds_bytes / 8;
    double* const redbuf = ws + a.ws_doubles;
    uint16_t* const tri = reinterpret_cast<uint16_t*>(redbuf + 128);
    int* const ints = reinterpret_cast<int*>(tri + 2080);  // [0] warnings, [1] a pivot failed
    // per front: how many of its children (in this workgroup) have been factorised, all factorisations of this launch counted
    // together (never reset: a front starts at children x the number of the factorisation); the number of the factorisation whose
    // backward substitution the front has been through
    unsigned int* const arrived = reinterpret_cast<unsigned int*>(ints + 16);
    unsigned int* const bdone = arrived + W.n_fronts;
    for (uint32_t i = tid; i < 2 * W.n_fronts; i += blockDim.x) arrived[i] = 0;
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.plan + W.o_tables);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (uint32_t i = tid; i < W.tab_bytes / 16; i += blockDim.x) dst[i] = src[i];
        for (uint32_t ra = tid; ra < 64; ra += blockDim.x)
            for (uint32_t rb = 0; rb <= ra; ++rb) tri[tri_index(ra, rb)] = (uint16_t)(ra | (rb << 8));
    }
    Ctx cx;
    cx.ws = ws;
    cx.descs = reinterpret_cast<const FrontDesc*>(tab);
    cx.children = reinterpret_cast<const FrontChild*>(tab + W.t_children);
    cx.rows = reinterpret_cast<const uint16_t*>(tab + W.t_rows);
    cx.exports = reinterpret_cast<const uint32_t*>(tab + W.t_exports);
    cx.maps = reinterpret_cast<const uint8_t*>(tab + W.t_maps);
    cx.tri = tri;
    cx.stream = reinterpret_cast<const uint32_t*>(tab + W.t_stream);
    unsigned char* const scratch = G > 1 ? a.scratch + (size_t)slot * a.scratch_stride : nullptr;
    FrontScratchHead* const head = reinterpret_cast<FrontScratchHead*>(scratch);
    cx.chunks = G > 1 ? reinterpret_cast<gridchunk_t*>(scratch + sizeof(FrontScratchHead) + 2 * kFrontScratchRedBytes) : nullptr;
    cx.dead = G > 1 ? &head->dead : nullptr;
    cx.l_jv = W.l_jv;
    cx.l_d = W.l_d;
    cx.l_upool = W.l_upool;
    Red red;
    red.buf = redbuf;
    red.flip = 0;
    red.scratch = scratch;
    red.G = G;
    red.wg = wg;
    red.seq = 0;
    red.dead = cx.dead;
    const uint16_t* const sched = reinterpret_cast<const uint16_t*>(tab + W.t_sched);  // (FrontWg::t_sched)
    unsigned int fact_no = 0;  // factorisations of this launch so far
    unsigned int epoch = 0;  // tag of the chunks of one linear solve (update matrices up, steps down)
    if (G > 1) {             // continue the slot's sequence numbers where this workgroup's previous launch left them
        epoch = __hip_atomic_load(&head->hop[wg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red.seq = __hip_atomic_load(&head->red[wg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const uint32_t n_loc = W.n_loc, n_own = W.n_own, n_cons = W.n_cons, m = W.n_rows, zj = W.zj;
    const uint32_t* const var_glob = reinterpret_cast<const uint32_t*>(a.plan + W.o_var_glob);
    // (the constraint table: its copy in the staged tables when the planner found room for one, else global memory)
    const DevCon* const cons = W.t_cons != 0xFFFFFFFFu ? reinterpret_cast<const DevCon*>(tab + W.t_cons) : reinterpret_cast<const DevCon*>(a.plan + W.o_cons);
    const FrontGhost* const ghosts = reinterpret_cast<const FrontGhost*>(a.plan + W.o_ghosts);
    double* const xs = ws + W.l_x;
    double* const dv = ws + W.l_d;
    double* const jvp = ws + W.l_jv;
    uint32_t l_r = W.l_r, l_rn = W.l_rn;
    if (tid == 0) {  // the operands of padding pairs
        ws[W.l_r + m] = 0.0;
        ws[W.l_rn + m] = 0.0;
        jvp[zj] = 0.0;
        ws[W.l_panels] = 0.0;  // ... and of padding sources
    }
    const bool unit_w = a.unit_weights != 0;
    // null-space probes instead of the LM loop (FrontArgs::probe_m)
    const bool probing = a.probe_m != 0;
    const uint16_t* const slotmap = reinterpret_cast<const uint16_t*>(a.plan + W.o_slotmap);
    uint32_t sys_parity = 0;
    for (uint64_t sys = slot; sys < a.batch; sys += n_slots, sys_parity ^= 1u) {
#ifdef EZPZ_STAMPS
        int stamp_n = 0;
        cx.stamps = a.stamps;
        cx.stamp_n = &stamp_n;
#endif
        FRONT_STAMP(1);
        // (probes: a work item is ONE probe of one system -- item = system x probe_m + probe -- so that a system's probes run side by
        // side on as many workgroups)
        const double* const x0 = a.x0 + (probing ? sys / a.probe_m : sys) * a.n_vars;
        for (uint32_t i = tid; i < n_loc; i += blockDim.x) xs[i] = x0[var_glob[i]];
        int* const nwarn = G > 1 ? &head->nwarn[sys_parity] : &ints[0];
        if (tid == 0 && G == 1) ints[0] = 0;
        __syncthreads();
        FRONT_STAMP(2);
        enum { EVAL0 = 0, STEP = 1, FINAL = 2 };
        int mode = EVAL0;
        uint32_t pass = 0, it = 0;
        double residual_sq = 0.0, largest = 0.0, unsat_cnt = 0.0;
        bool r_is_at_x = true, all_satisfied = false;
        double lambda = a.initial_lambda;
        double step_inf_norm = 0.0;
        uint32_t iterations = a.max_iterations, converged = 0;
        double dmax = __builtin_nan("");  // (fmax drops NaN seeds; an all-NaN d stays NaN like reduce(fmax))
        double lambda_probe = 0.0;
        for (;;) {
            if (mode == STEP && probing) {
                // ---- probe j: w = pseudo-random entries, uniform in [-1, 1) (in x's place: the Jacobian is evaluated, x is not needed again), the
                //      right-hand side's residual vector J w (in r_next's place), lambda_p from J's largest entry ---------------------
                {
                    double m2 = __builtin_nan(""), z0 = 0.0, z1 = __builtin_nan(""), z2 = 0.0;
                    for (uint32_t i = tid; i < zj; i += blockDim.x) m2 = fmax_abs(m2, jvp[i]);
                    red.reduce(z0, m2, z1, z2);
                    lambda_probe = a.probe_scale * m2 * m2;
                }
                const uint32_t probe_j = (uint32_t)(sys % a.probe_m);
                for (uint32_t i = tid; i < n_loc; i += blockDim.x) {
                    if (a.probe_in) {
                        xs[i] = a.probe_in[sys * a.n_vars + var_glob[i]];
                        continue;
                    }
                    uint32_t h = var_glob[i] * 2654435761u ^ (probe_j * 0x9E3779B9u + 0x7F4A7C15u);
                    h ^= h >> 15;
                    h *= 0x2C1B3C6Du;
                    h ^= h >> 12;
                    h *= 0x297A2D39u;
                    h ^= h >> 15;
                    // (uniform in [-1, 1), not signs: a null vector like (1, -1) / sqrt 2 is orthogonal to every sign vector that gives
                    // its two variables the same sign -- all eight of them once in 256)
                    xs[i] = (double)h * (1.0 / 2147483648.0) - 1.0;
                }
                __syncthreads();
                for (uint32_t ci = tid; ci < n_cons; ci += blockDim.x) {
                    const DevCon c = load_con(cons + ci);
                    double r0 = 0.0, r1 = 0.0;
                    for (uint32_t q = 0; q < c.nslots; ++q) {
                        const uint32_t e = slotmap[c.jbase + q];
                        const double t = jvp[c.jbase + q] * xs[e & 0x7FFFu];
                        if (e >> 15)
                            r1 += t;
                        else
                            r0 += t;
                    }
                    ws[l_rn + c.row0] = r0;
                    if (c.nrows > 1) ws[l_rn + c.row0 + 1] = r1;
                }
                __syncthreads();
            }
            if (mode == STEP && !probing) {
                if (it >= a.max_iterations) {  // newton.rs:141-144
                    mode = FINAL;
                } else if (largest <= a.residual_tolerance) {  // newton.rs:50-60
                    iterations = it;
                    converged = 1;
                    mode = FINAL;
                }
            }
            if (mode == STEP) {
                // ---- (JtJ + lambda I) d = -Jt r: every wavefront its list of fronts (newton.rs:73-102) ---------------------------------
                ++epoch;
                if (tid == 0) ints[1] = 0;
                __syncthreads();
                assemble(cx, W, probing ? lambda_probe : lambda, probing ? l_rn : l_r);
                FRONT_STAMP(10);
                bool bad_here = false;
                ++fact_no;
                for (uint32_t i = uni(sched[wave]), i1 = uni(sched[wave + 1]); i < i1; ++i) {
                    const uint32_t k = uni(sched[i]);
                    const uint32_t kids = uni(cx.descs[k].n_kids_local), parent = uni(cx.descs[k].parent_local);
                    if (kids) {
                        const unsigned int want = kids * fact_no;
                        while (uni(__hip_atomic_load(&arrived[k], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != want) __builtin_amdgcn_s_sleep(1);
                    }
                    bad_here |= front_factor(cx, k, lane, W.l_panels, epoch);
                    if (parent != 0xFFFFFFFFu) {
                        wave_sync();
                        if (lane == 0) __hip_atomic_fetch_add(&arrived[parent], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                __syncthreads();
                FRONT_STAMP(1000);
                if (bad_here && lane == 0) ints[1] = 1;
                if (G > 1) {
                    // a pivot failed somewhere?  every workgroup tells workgroup 0, whose verdict travels with the steps
                    __syncthreads();
                    if (wg != 0) {
                        if (tid == 0) grid_store(cx.chunks + a.bad_chunk0 + wg, ints[1] ? 1.0 : 0.0, epoch);
                    } else {
                        if (tid > 0 && (uint32_t)tid < G && grid_wait(cx.chunks + a.bad_chunk0 + tid, epoch, cx.dead) != 0.0) ints[1] = 1;
                        __syncthreads();
                        if (tid == 0) grid_store(cx.chunks + a.verdict_chunk, ints[1] ? 1.0 : 0.0, epoch);
                    }
                    if (wg != 0) {
                        if (tid == 0 && grid_wait(cx.chunks + a.verdict_chunk, epoch, cx.dead) != 0.0) ints[1] = 1;
                    }
                }
                __syncthreads();
                const bool bad = ints[1] != 0;
                FRONT_STAMP(11);
                if (bad && probing) {  // (the host falls back to the pivoted QR)
                    for (uint32_t i = tid; i < n_own; i += blockDim.x) a.probe_out[sys * a.n_vars + var_glob[i]] = __builtin_nan("");
                    __syncthreads();
                    break;
                }
                if (bad) {  // numeric failure => lambda *= 10, burn the iteration (newton.rs:93-99)
                    lambda *= LM_LAMBDA_INCR;
                    ++it;
                    __syncthreads();  // (ints[1] is reset at the top of the next trip)
                    continue;
                }
                // ---- backward substitution, top down; the steps of variables other workgroups eliminate arrive as chunks -------------------
                if (G > 1 && wg != 0) {
                    for (uint32_t i = tid; i < W.n_ghost; i += blockDim.x) dv[ghosts[i].local] = grid_wait(cx.chunks + ghosts[i].chunk, epoch, cx.dead);
                    __syncthreads();
                }
                dmax = __builtin_nan("");
                for (uint32_t i = uni(sched[nwaves + 1 + wave]), i1 = uni(sched[nwaves + 2 + wave]); i < i1; ++i) {
                    const uint32_t k = uni(sched[i]);
                    const uint32_t parent = uni(cx.descs[k].parent_local);
                    if (parent != 0xFFFFFFFFu)
                        while (uni(__hip_atomic_load(&bdone[parent], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) != fact_no) __builtin_amdgcn_s_sleep(1);
                    dmax = front_bwd(cx, k, lane, epoch, dmax);
                    wave_sync();
                    if (lane == 0) __hip_atomic_store(&bdone[k], fact_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __syncthreads();
                FRONT_STAMP(2000);
                FRONT_STAMP(12);
                // ---- tentative step (newton.rs:111-114): own variables and ghosts alike (every workgroup moves its copy) -----------------
                for (uint32_t i = tid; i < n_loc; i += blockDim.x) xs[i] = xs[i] + dv[i];
                __syncthreads();
                FRONT_STAMP(14);
                if (probing) {  // y = w + d: the probe's answer for this workgroup's own variables
                    for (uint32_t i = tid; i < n_own; i += blockDim.x) a.probe_out[sys * a.n_vars + var_glob[i]] = xs[i];
                    __syncthreads();
                    break;
                }
            }
            if (mode == FINAL && r_is_at_x && unit_w) {
                // r already holds the unweighted residuals at this x (lib.rs:305-327, :358-370; see lm_kernel.hip.hpp)
                if (largest < EPS && !isnan(residual_sq)) {
                    all_satisfied = true;
                    if (a.unsat_mask && wg == 0)
                        for (uint32_t i = tid; i < a.n_cons; i += blockDim.x) a.unsat_mask[sys * a.n_cons + i] = 0;
                    break;
                }
                for (uint32_t ci = tid; ci < n_cons; ci += blockDim.x) {
                    const DevCon c = load_con(cons + ci);
                    bool sat = fabs(ws[l_r + c.row0]) < EPS;
                    if (c.nrows > 1) sat = sat && (fabs(ws[l_r + c.row0 + 1]) < EPS);
                    if (!sat) unsat_cnt += 1.0;
                    if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + c.pos] = sat ? 0 : 1;
                }
                break;
            }
            // ---- the residual sweep: EVAL0 -> r, STEP -> r_next, FINAL -> unsatisfied test on unweighted values ---------------------------
            const uint32_t l_dst = (mode == EVAL0) ? l_r : l_rn;
            double sq = 0.0, mx = __builtin_nan("");
            for (uint32_t ci = tid; ci < n_cons; ci += blockDim.x) {
                const DevCon c = load_con(cons + ci);
                double r0, r1;
                const bool deg = con_residual<LIN>(c, (const double*)xs, r0, r1);
                if (mode == FINAL) {
                    bool sat = fabs(r0) < EPS;
                    if (c.nrows > 1) sat = sat && (fabs(r1) < EPS);
                    if (!sat) unsat_cnt += 1.0;
                    if (a.unsat_mask) a.unsat_mask[sys * a.n_cons + c.pos] = sat ? 0 : 1;
                    continue;
                }
                const double wgt = unit_w ? 1.0 : c.weight;
                const double w0 = wgt * r0;
                ws[l_dst + c.row0] = w0;
                sq += w0 * w0;
                mx = fmax_abs(mx, w0);
                if (c.nrows > 1) {
                    const double w1 = wgt * r1;
                    ws[l_dst + c.row0 + 1] = w1;
                    sq += w1 * w1;
                    mx = fmax_abs(mx, w1);
                }
                if (deg) {  // Warning::Degenerate, every evaluation (solver.rs:340-346)
                    const int idx = atomicAdd(nwarn, 1);
                    if (a.warn_log && (uint32_t)idx < a.warn_cap) a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | c.pos;
                }
                if constexpr (LIN) {  // constant partials: the one Jacobian sweep of a linear-only system rides in eval()
                    if (mode == EVAL0) {
                        JacWriter<double*> w;
                        w.jv = jvp;
                        w.jbase = c.jbase;
                        const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
                        w.loc[0] = loc[0], w.loc[1] = loc[1], w.loc[2] = loc[2], w.loc[3] = loc[3];
                        w.weight = wgt;
                        (void)con_jacobian<LIN>(c, (const double*)xs, w);
                    }
                }
            }
            ++pass;
            __syncthreads();
            FRONT_STAMP(20);
            if (mode == FINAL) break;
            double dm = dmax, zero = 0.0;
            red.reduce(sq, mx, dm, zero);
            FRONT_STAMP(21);
            if (mode == STEP) step_inf_norm = a.n_vars > 0 ? dm : 0.0;  // newton.rs:108
            const bool accept = (mode == EVAL0) || (sq < residual_sq);  // strict, newton.rs:118
            if (accept) {
                if (mode == STEP) {
                    const uint32_t t = l_r;
                    l_r = l_rn;
                    l_rn = t;
                    lambda *= LM_LAMBDA_DECR;
                }
                if constexpr (!LIN) {
                    for (uint32_t ci = tid; ci < n_cons; ci += blockDim.x) {
                        const DevCon c = load_con(cons + ci);
                        JacWriter<double*> w;
                        w.jv = jvp;
                        w.jbase = c.jbase;
                        const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
                        w.loc[0] = loc[0], w.loc[1] = loc[1], w.loc[2] = loc[2], w.loc[3] = loc[3];
                        w.weight = unit_w ? 1.0 : c.weight;
                        const bool deg = con_jacobian<LIN>(c, (const double*)xs, w);
                        if (deg) {
                            const int idx = atomicAdd(nwarn, 1);
                            if (a.warn_log && (uint32_t)idx < a.warn_cap) a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | c.pos;
                        }
                    }
                }
                ++pass;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
            } else {  // reject: revert, raise lambda (newton.rs:124-131)
                r_is_at_x = false;
                for (uint32_t i = tid; i < n_loc; i += blockDim.x) xs[i] = xs[i] - dv[i];
                lambda *= LM_LAMBDA_INCR;
            }
            __syncthreads();
            FRONT_STAMP(22);
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
        if (probing) {  // (the probes' answers are written; nothing else is)
            if (G > 1 && tid == 0 && wg == 0) {
                if (__hip_atomic_load(cx.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) a.probe_out[sys * a.n_vars] = __builtin_nan("");
                __hip_atomic_store(nwarn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            continue;
        }
        // ---- write-back -------------------------------------------------------------------------------------------------------------------
        FRONT_STAMP(30);
        double d1 = __builtin_nan(""), d2 = __builtin_nan(""), d3 = 0.0;
        const bool skip_count = all_satisfied && (G == 1 || LIN);  // (several workgroups: the rendezvous also orders the warning counter)
        if (!skip_count) red.reduce(unsat_cnt, d1, d2, d3);
        double* const xo = a.x_out + sys * a.n_vars;
        for (uint32_t i = tid; i < n_own; i += blockDim.x) xo[var_glob[i]] = xs[i];
        if (tid == 0 && wg == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = G > 1 ? (uint32_t)__hip_atomic_load(nwarn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint32_t)*nwarn;
            st.final_residual_inf = largest;  // (a planned system has rows)
            st.final_lambda = lambda;
            if (G > 1 && __hip_atomic_load(cx.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st.iterations = EZPZ_ITERATIONS_TEAM_TIMEOUT;
                st.converged = 0;
            }
            a.status[sys] = st;
            // (several workgroups: this counter serves the slot's system after next; every workgroup passes a reduction of the
            // next system, which this thread joins only after the store, before it can get there)
            if (G > 1) __hip_atomic_store(nwarn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        FRONT_STAMP(32);
    }
    if (G > 1 && tid == 0) {
        __hip_atomic_store(&head->hop[wg], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&head->red[wg], red.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    publish_done(a.done);
}

}  // namespace ezpz
