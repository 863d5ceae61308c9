// The Levenberg-Marquardt solve kernel: one *team* of lanes owns one constraint system for the whole
// solve -- residual/Jacobian evaluation, normal equations, sparse Cholesky, triangular solves, step
// acceptance and convergence tests all run inside this one launch, with the system state resident in
// LDS.  It replaces, for a batch of systems sharing a topology,
//     Model::solve_levenberg_marquardt      reference ezpz/src/solver/newton.rs:29-145
//     Model::residual / refresh_jacobian    reference ezpz/src/solver.rs:318-440
//     faer's transpose/matmul/add/Llt/solve reference ezpz/src/solver/newton.rs:73-102
//     the unsatisfied check of solve_inner  reference ezpz/src/lib.rs:305-327
//
// Team modes (template):
//   SUB   TEAM in {1,2,4,...,64} lanes per system, 64/TEAM systems per 64-wide wavefront with their workspaces
//         interleaved element by element.  No s_barrier anywhere: lanes of a wave run in lockstep and the LDS
//         serves a wave's accesses in issue order; reductions are DPP shuffles.  Small programs are staged into
//         LDS once per workgroup (PLDS).
//   PART  one workgroup per system, every wavefront owns a *partition* (a balanced union of connected
//         components: its own constraints, variables, Jacobian slots, Cholesky columns).  All phases of an
//         LM iteration are wave-local; the only workgroup barriers are the two reductions per iteration that
//         the reference's global LM control needs (sum r^2 / max|r|, and Cholesky-failed / max|d|).
//   WGB   one workgroup per system, all lanes cooperate on one partition with s_barrier between phases
//         (systems dominated by one large connected component).
//   LDSWS=false  state lives in a per-workgroup global-memory workspace (systems too big for 160 KB of LDS).
//   grid team    PART with a.grid_wgs > 1: several workgroups share one system, each with its own partitions and
//         its share of the state in LDS; the two reductions per iteration cross workgroups (see GridScratch).
//   LIN   build without the sixteen non-linear kinds' evaluators, for topologies that do not use them.
// HBM traffic is only x0 in, x*/status/mask out (AoS rows, contiguous per team); the topology program is
// shared by every team and stays L2 (or LDS) resident.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "constraint_eval.hip.hpp"
#include "grid_ops.hip.hpp"
#include "launch_types.hpp"
#include "wave_ops.hip.hpp"

namespace ezpz {

// Typed pointers into the blob.  Index lists are 16-bit when they are staged in LDS (every count < 65536),
// 32-bit otherwise; the constraint table and the partition descriptors keep their layout.
template <class IDX>
struct Prog {
    const DevCon* cons;
    const PackedCon* pcons;       // packed form (same offset) + its side arrays
    const uint32_t* con_pos;
    const double* con_weight;
    const uint4* patterns;        // in the staged part of the blob
    const PartDesc* parts;
    const IDX *colj_ptr, *colj_items;
    const IDX *apair_ptr, *apairs;
    const IDX *lvl_cptr, *lvl_sptr, *l_col, *lvl_grp;
    const uint32_t* var_of;
    const IDX *lpair_ptr, *lpairs;
    const IDX *fwd_ptr, *fwd_items;
    const IDX *bwd_ptr, *bwd_items;
    const IDX *dense_col, *dense_slot, *dense_tab;
    const uint32_t *lvl_off, *lvl_stream;  // per-level blocks of the Cholesky lists (32-bit programs of one partition)
    const uint32_t *lvl_boff, *lvl_bstream;  // ... and of the backward substitution's
};

// `lists` addresses the index lists (LDS copy or global blob), `tables` the constraint table / partitions.
template <class IDX>
__device__ __forceinline__ Prog<IDX> make_prog(const ProgramView& v, const unsigned char* lists,
                                               const unsigned char* tables) {
    Prog<IDX> p;
    p.cons = reinterpret_cast<const DevCon*>(tables + v.o_cons);
    p.pcons = reinterpret_cast<const PackedCon*>(tables + v.o_cons);
    p.con_pos = reinterpret_cast<const uint32_t*>(tables + v.o_pos);
    p.con_weight = reinterpret_cast<const double*>(tables + v.o_weights);
    p.patterns = reinterpret_cast<const uint4*>(lists + v.o_patterns);
    p.parts = reinterpret_cast<const PartDesc*>(tables + v.o_parts);
    auto u = [&](uint32_t o) { return reinterpret_cast<const IDX*>(lists + o); };
    p.colj_ptr = u(v.o_colj_ptr);
    p.colj_items = u(v.o_colj_items);
    p.apair_ptr = u(v.o_apair_ptr);
    p.apairs = u(v.o_apairs);
    p.lvl_cptr = u(v.o_lvl_cptr);
    p.var_of = reinterpret_cast<const uint32_t*>(lists + v.o_var_of);
    p.lvl_sptr = u(v.o_lvl_sptr);
    p.l_col = u(v.o_l_col);
    p.lvl_grp = u(v.o_lvl_grp);
    p.lpair_ptr = u(v.o_lpair_ptr);
    p.lpairs = u(v.o_lpairs);
    p.fwd_ptr = u(v.o_fwd_ptr);
    p.fwd_items = u(v.o_fwd_items);
    p.bwd_ptr = u(v.o_bwd_ptr);
    p.bwd_items = u(v.o_bwd_items);
    p.dense_col = u(v.o_dense_col);
    p.dense_slot = u(v.o_dense_slot);
    p.dense_tab = u(v.o_dense_tab);
    p.lvl_off = reinterpret_cast<const uint32_t*>(lists + v.o_lvl_off);
    p.lvl_stream = reinterpret_cast<const uint32_t*>(lists + v.o_lvl_stream);
    p.lvl_boff = reinterpret_cast<const uint32_t*>(lists + v.o_lvl_boff);
    p.lvl_bstream = reinterpret_cast<const uint32_t*>(lists + v.o_lvl_bstream);
    return p;
}

// Packed record -> the in-register DevCon the evaluators take.  jloc is left to the Jacobian sweep (pattern index is
// parked in nslots); pos is fetched by the rare paths that need it.
__device__ __forceinline__ DevCon load_packed(const PackedCon* p, const double* weights, uint32_t ci, bool unit_weights) {
    const uint4* src = reinterpret_cast<const uint4*>(p + ci);
    const uint4 q0 = src[0], q1 = src[1];
    DevCon c;
    c.ids[0] = q0.x & 0xFFFFu;
    c.ids[1] = q0.x >> 16;
    c.ids[2] = q0.y & 0xFFFFu;
    c.ids[3] = q0.y >> 16;
    c.ids[4] = q0.z & 0xFFFFu;
    c.ids[5] = q0.z >> 16;
    c.ids[6] = q0.w & 0xFFFFu;
    c.ids[7] = q0.w >> 16;
    c.param = __hiloint2double((int)q1.y, (int)q1.x);
    c.row0 = q1.z & 0xFFFFu;
    c.jbase = q1.z >> 16;
    c.kind = (uint8_t)(q1.w & 0xFFu);
    c.tag = (uint8_t)((q1.w >> 8) & 0xFFu);
    c.nrows = (uint8_t)((q1.w >> 16) & 0xFFu);
    c.nslots = (uint8_t)(q1.w >> 24);
    c.weight = unit_weights ? 1.0 : weights[ci];
    c.pos = ci;
    return c;
}

#ifdef EZPZ_STAMPS
#define EZPZ_STAMP(id)                                                                     \
    do {                                                                                   \
        if (a.stamps && blockIdx.x == 0 && threadIdx.x == 0 && stamp_n < 1000) {            \
            a.stamps[2 * stamp_n] = (id);                                                  \
            a.stamps[2 * stamp_n + 1] = __builtin_readcyclecounter();                      \
            ++stamp_n;                                                                     \
        }                                                                                  \
    } while (0)
#define EZPZ_STAMP_DRAIN(id)                 \
    do {                                     \
        __builtin_amdgcn_s_waitcnt(0);       \
        EZPZ_STAMP(id);                      \
    } while (0)
#else
#define EZPZ_STAMP(id) \
    do {               \
    } while (0)
#define EZPZ_STAMP_DRAIN(id) \
    do {                     \
    } while (0)
#endif

namespace dev {

// A value every lane of the wavefront holds alike (level bounds, lanes per list: properties of the program, and every
// team of a wavefront runs the same program), moved to a scalar register: branches and loop bounds on it are scalar,
// and a `switch` on it is one jump instead of a walk through every case under an execution mask.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// One load for an (a, b) index pair of a list.
__device__ __forceinline__ void load_pair(const uint16_t* items, uint32_t q, uint32_t& a, uint32_t& b) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(items + 2 * q);
    a = w & 0xFFFFu;
    b = w >> 16;
}
__device__ __forceinline__ void load_pair(const uint32_t* items, uint32_t q, uint32_t& a, uint32_t& b) {
    const uint2 w = *reinterpret_cast<const uint2*>(items + 2 * q);
    a = w.x;
    b = w.y;
}

// Walks the pair list [q0, q1) LIST_CHUNK entries at a time (a constexpr in scope at the use site): the index
// loads of a chunk are independent, then its value loads are independent, and only then are the terms folded in list
// order (`fold(k)` for valid k).  This turns 1 + 2L dependent LDS hops per list into 3 per chunk while keeping every
// floating-point sum in exactly the same order.  Invalid tail entries still cost their instructions, so the chunk
// follows the typical list length: 4 for the dense little systems of sub-wavefront teams, 2 for the sparse
// systems of workgroup teams (1-2 entries per list in massive_parallel_system).
#define EZPZ_FOR_PAIRS4(items, q0, q1, A, B, LOADVALS, FOLD)                       \
    for (uint32_t q_ = (q0); q_ < (q1); q_ += LIST_CHUNK) {                        \
        uint32_t A[LIST_CHUNK], B[LIST_CHUNK];                                     \
        bool ok_[LIST_CHUNK];                                                      \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) {                   \
            ok_[k] = q_ + k < (q1);                                                \
            load_pair((items), ok_[k] ? q_ + k : q_, A[k], B[k]);                  \
        }                                                                          \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) { LOADVALS; }       \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) {                   \
            if (ok_[k]) { FOLD; }                                                  \
        }                                                                          \
    }

// The same walk by one lane of a group of `step` lanes that share the list: entries qb, qb + step, qb + 2 step, ...
// (qb = q0 + the lane's position in its group); the group adds its partial sums afterwards.
#define EZPZ_FOR_PAIRS_STRIDED(items, qb, q1, step, A, B, LOADVALS, FOLD)          \
    for (uint32_t q_ = (qb); q_ < (q1); q_ += LIST_CHUNK * (step)) {               \
        uint32_t A[LIST_CHUNK], B[LIST_CHUNK];                                     \
        bool ok_[LIST_CHUNK];                                                      \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) {                   \
            ok_[k] = q_ + k * (step) < (q1);                                       \
            load_pair((items), ok_[k] ? q_ + k * (step) : q_, A[k], B[k]);         \
        }                                                                          \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) { LOADVALS; }       \
        _Pragma("unroll") for (int k = 0; k < LIST_CHUNK; ++k) {                   \
            if (ok_[k]) { FOLD; }                                                  \
        }                                                                          \
    }

// ---- dense 8 x 8 linear solve in registers (teams of four lanes on systems of <= 8 variables) -----------------------
// Value of lane S (0..3) of every quad in all four lanes of the quad: one DPP move per 32-bit half, no LDS round trip.
__device__ __forceinline__ double quad_bcast(double v, int s) {
    switch (s & 3) {
    case 0: return dpp_move<0x00>(v);
    case 1: return dpp_move<0x55>(v);
    case 2: return dpp_move<0xAA>(v);
    default: return dpp_move<0xFF>(v);
    }
}

// (A + lambda I) d = b for a system of n <= 8 variables: A = D (diagonal, ws[o_d + r]) and the strictly-lower entries in
// the dense column layout of build_program (column c starts at slot lvl_sptr[c]); b in ws[o_v].  Lane l of the four
// holds rows l and l + 4 in registers; pivots and multipliers travel by quad broadcast.  Right-looking Cholesky with
// the forward substitution folded in, then the backward substitution: every entry receives its updates in ascending
// column order and is divided last, every substitution sum runs over ascending indices -- operation for operation
// what the level-scheduled list walk does on the same (full) pattern, without its ~10 LDS hops per level.
// Returns 1.0 when a pivot of the real system is not positive (LltError::Numeric), d in ws[o_v].
template <class WS, class LP>
__device__ __forceinline__ double dense8_solve(const WS& ws, uint32_t o_d, uint32_t o_l, uint32_t o_v, LP lvl_sptr,
                                               uint32_t n, int lane) {
    uint32_t cs[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) cs[c] = (uint32_t)c < n ? (uint32_t)lvl_sptr[c] : 0u;
    double a[2][8], bv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t r = (uint32_t)lane + 4u * q;
        const bool in = r < n;
        bv[q] = in ? ws[o_v + r] : 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            double v = 0.0;
            if (in && (uint32_t)c < r)
                v = ws[o_l + cs[c] + (r - c - 1)];
            else if ((uint32_t)c == r)
                v = in ? ws[o_d + r] : 1.0;  // rows beyond n: identity, they touch nothing
            a[q][c] = v;
        }
    }
    double bad = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const double pivot = quad_bcast(a[j >> 2][j], j & 3);
        if ((uint32_t)j < n && !(pivot > 0.0)) bad = 1.0;
        const double dj = sqrt(pivot);
        const double yj = quad_bcast(bv[j >> 2], j & 3) / dj;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = lane + 4 * q;
            if (r > j) {
                a[q][j] = a[q][j] / dj;
                bv[q] = bv[q] - a[q][j] * yj;
            } else if (r == j) {
                a[q][j] = dj;
                bv[q] = yj;
            }
        }
#pragma unroll
        for (int k = j + 1; k < 8; ++k) {
            const double lkj = quad_bcast(a[k >> 2][j], k & 3);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (lane + 4 * q >= k) a[q][k] = a[q][k] - a[q][j] * lkj;
        }
    }
#pragma unroll
    for (int j = 7; j >= 0; --j) {
        double acc = quad_bcast(bv[j >> 2], j & 3);
#pragma unroll
        for (int i = j + 1; i < 8; ++i) {
            const double p = a[i >> 2][j] * bv[i >> 2];  // meaningful on the lane that owns row i
            acc = acc - quad_bcast(p, i & 3);
        }
        const double xj = acc / quad_bcast(a[j >> 2][j], j & 3);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (lane + 4 * q == j) bv[q] = xj;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t r = (uint32_t)lane + 4u * q;
        if (r < n) ws[o_v + r] = bv[q];
    }
    return bad;
}

template <int TEAM, int MODE, bool GRID = false>
struct Team {
    int lane;     // lane inside the unit that walks a phase (team for SUB, wave for PART, workgroup for WGB)
    int stride;   // lanes in that unit
    double* red;  // PART/WGB: 2 (flip) x 2 (values) x 16 (waves) doubles of LDS scratch
    int red_flip;
    GridScratch* grid;  // grid teams only
    uint32_t grid_wgs, grid_wg;
    unsigned int grid_seq;  // sequence number of the last grid reduction (same on every workgroup of the system)

    // Orders one phase's LDS/global writes before the next phase's reads inside the unit.
    __device__ __forceinline__ void phase_sync() const {
        if constexpr (MODE == MODE_WGB) {
            __syncthreads();
        } else {
            // Lanes of one wavefront execute in lockstep and the LDS services a wave's accesses in issue
            // order, so no hardware barrier is needed; only the compiler must not move accesses across.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    // Whole-team rendezvous (cooperative loads/stores of x).
    __device__ __forceinline__ void team_sync() const {
        if constexpr (MODE == MODE_SUB)
            phase_sync();
        else
            __syncthreads();
    }

    // Two team-wide reductions for the price of one rendezvous; every lane gets both results.
    template <class OpA, class OpB>
    __device__ __forceinline__ void reduce2(double& a, double& b, OpA opa, OpB opb) {
        if constexpr (MODE == MODE_SUB) {
            // every lane of the team ends with the same bits: each step combines two values that are already
            // identical across the lanes that hold them, and the operators are commutative
            a = reduce_lanes<TEAM>(a, opa);
            b = reduce_lanes<TEAM>(b, opb);
        } else {
            a = reduce_wave_to_last_lane(a, opa);
            b = reduce_wave_to_last_lane(b, opb);
            double* buf = red + (red_flip ? 32 : 0);
            red_flip ^= 1;
            const int wave = threadIdx.x >> 6;
            const int nwaves = (blockDim.x + 63) >> 6;
            if ((threadIdx.x & 63) == 63) {
                buf[wave] = a;
                buf[16 + wave] = b;
            }
            __syncthreads();
            a = buf[0];
            b = buf[16];
            for (int w = 1; w < nwaves; ++w) {
                a = opa(a, buf[w]);
                b = opb(b, buf[16 + w]);
            }
            // the next reduction uses the other half of `red`, so no trailing barrier is needed
            if constexpr (GRID) {
                if (grid_wgs > 1) {  // grid team: gather at workgroup 0, fold in a fixed tree, scatter the result
                    const unsigned int seq = ++grid_seq;
                    const unsigned int par = seq & 1u;
                    // A workgroup publishes sequence number s+1 only after it has consumed the result of s, and
                    // workgroup 0 writes the result of s+2 only after every arrival for s+2: parities never collide.
                    if (threadIdx.x < 2) grid_store(&grid->arr[par][threadIdx.x][grid_wg], threadIdx.x ? b : a, seq);
                    double* buf2 = red + (red_flip ? 32 : 0);
                    red_flip ^= 1;
                    if (grid_wg == 0) {
                        a = opa.identity();
                        b = opb.identity();
                        if (threadIdx.x < grid_wgs) {
                            a = grid_wait(&grid->arr[par][0][threadIdx.x], seq, &grid->dead);
                            b = grid_wait(&grid->arr[par][1][threadIdx.x], seq, &grid->dead);
                        }
                        a = reduce_lanes<64>(a, opa);
                        b = reduce_lanes<64>(b, opb);
                        if ((threadIdx.x & 63) == 0) {
                            buf2[wave] = a;
                            buf2[16 + wave] = b;
                        }
                        __syncthreads();
                        a = buf2[0];
                        b = buf2[16];
                        for (int w = 1; w < nwaves; ++w) {
                            a = opa(a, buf2[w]);
                            b = opb(b, buf2[16 + w]);
                        }
                        if (threadIdx.x < grid_wgs && threadIdx.x > 0) {  // one line per workgroup
                            grid_store(&grid->out[par][threadIdx.x][0], a, seq);
                            grid_store(&grid->out[par][threadIdx.x][1], b, seq);
                        }
                    } else {
                        if (threadIdx.x < 2) buf2[16 * threadIdx.x] = grid_wait(&grid->out[par][grid_wg][threadIdx.x], seq, &grid->dead);
                        __syncthreads();
                        a = buf2[0];
                        b = buf2[16];
                    }
                }
            }
        }
    }
};

}  // namespace dev

// The per-system workspace as the phases index it.  Workgroup teams own one contiguous block of doubles (STRIDE 1).
// Sub-wavefront teams share a wavefront (64 / TEAM systems side by side) and interleave their workspaces element by
// element: element e of the wavefront's system t lives at word e * STRIDE + t, STRIDE = 64 / TEAM.  Lanes of
// different systems that walk the same list entry -- the common case, every system runs the same program -- then
// touch consecutive words instead of words a whole workspace apart (which collide on a few banks), and with one
// lane per system (TEAM 1) every access of the wavefront is one conflict-free 512-byte row.
template <int STRIDE>
struct WsRef {
    double* p;
    __device__ __forceinline__ double& operator[](uint32_t e) const { return p[(size_t)e * STRIDE]; }
    __device__ __forceinline__ WsRef operator+(uint32_t off) const {
        WsRef r;
        r.p = p + (size_t)off * STRIDE;
        return r;
    }
};

// Constraint record access: a wide by-value load when the table is in global memory, a plain reference when
// it sits in LDS (sub-wavefront teams with a staged program), where field-by-field reads are cheap.
// How a sweep gets at constraint `ci`: 0 = in place (table in LDS, sub-wavefront teams), 1 = one wide load of the
// 80-byte record, 2 = 32-byte packed record + side arrays.
template <int FORM, class PROG>
struct ConRef;
template <class PROG>
struct ConRef<0, PROG> {
    const DevCon* p;
    __device__ __forceinline__ ConRef(const PROG& P, uint32_t ci, bool) : p(P.cons + ci) {}
    __device__ __forceinline__ const DevCon& get() const { return *p; }
    __device__ __forceinline__ uint32_t pos(const PROG&, uint32_t) const { return p->pos; }
    __device__ __forceinline__ uint4 jloc(const PROG&) const { return *reinterpret_cast<const uint4*>(p->jloc); }
};
template <class PROG>
struct ConRef<1, PROG> {
    DevCon c;
    __device__ __forceinline__ ConRef(const PROG& P, uint32_t ci, bool) : c(load_con(P.cons + ci)) {}
    __device__ __forceinline__ const DevCon& get() const { return c; }
    __device__ __forceinline__ uint32_t pos(const PROG&, uint32_t) const { return c.pos; }
    __device__ __forceinline__ uint4 jloc(const PROG&) const { return *reinterpret_cast<const uint4*>(c.jloc); }
};
template <class PROG>
struct ConRef<2, PROG> {
    DevCon c;
    __device__ __forceinline__ ConRef(const PROG& P, uint32_t ci, bool unit_weights)
        : c(load_packed(P.pcons, P.con_weight, ci, unit_weights)) {}
    __device__ __forceinline__ const DevCon& get() const { return c; }
    __device__ __forceinline__ uint32_t pos(const PROG& P, uint32_t ci) const { return P.con_pos[ci]; }
    __device__ __forceinline__ uint4 jloc(const PROG& P) const { return P.patterns[c.nslots]; }
};

// LIN: every constraint of the topology is of a linear kind (see con_residual); the evaluators are built without
// the other sixteen kinds.
// GRID: the build for grid teams (several workgroups per system; MODE_PART with staged lists only).
// DENSE: tiny systems of sub-wavefront teams whose program was built with the dense factor layout (build_program):
// Cholesky and the substitutions are plain loops over rows and columns instead of level-by-level list walks.
// Occupancy hints: sub-wavefront teams are compiled for 4 workgroups per CU (128 VGPRs; measured against 3 and 2:
// +8 % on some topologies, -7 % on others), the register-resident dense solve for 2 (184 VGPRs, no spills: +12 %).
template <int TEAM, int MODE, bool LDSWS, bool PLDS, bool LIN, bool GRID = false, bool DENSE = false, int RECF = 0>
__global__ void __launch_bounds__(MODE == MODE_SUB ? 256 : (LIN && RECF == 0 ? 1024 : 512), MODE == MODE_SUB ? (DENSE ? 2 : 4) : 1)
    lm_solve_kernel(const SolveArgs a) {
    // RECF: the record walk's form -- 0 none, 1 state in LDS (16-bit addresses), 2 state in global memory (32-bit addresses)
    constexpr bool REC = RECF != 0;
    constexpr int RCH = RECF == 2 ? REC_WIDE_CHUNKS : REC_MAX_CHUNKS;  // chunks per lane and round
    static_assert(!DENSE || (MODE == MODE_SUB && TEAM == 4), "the register-resident dense solve is for teams of four");
    static_assert(!REC || (MODE == MODE_WGB && !GRID && !DENSE && (RECF == 1) == LDSWS), "the record walk is for one barrier workgroup");
    static_assert(!GRID || (MODE == MODE_PART && LDSWS && PLDS), "grid teams are partitioned teams with staged lists");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using namespace dev;
    Team<TEAM, MODE, GRID> tm;
    const int tid = threadIdx.x;
    tm.red_flip = 0;
    if constexpr (MODE == MODE_SUB) {
        tm.lane = tid % TEAM;
        tm.stride = TEAM;
    } else if constexpr (MODE == MODE_PART) {
        tm.lane = tid & 63;
        tm.stride = 64;
    } else {
        tm.lane = tid;
        tm.stride = (int)blockDim.x;
    }
    // lanes that cooperate on whole-system vectors (x load / store)
    const int tlane = (MODE == MODE_SUB) ? tm.lane : tid;
    const int tsize = (MODE == MODE_SUB) ? TEAM : (int)blockDim.x;
    const uint32_t teams_per_block = (MODE == MODE_SUB) ? (uint32_t)(blockDim.x / TEAM) : 1u;
    const uint32_t team_in_block = (MODE == MODE_SUB) ? (uint32_t)(tid / TEAM) : 0u;
    // The program this workgroup runs: the launch's, or -- in a grid team -- its own slice of it (its partitions,
    // renumbered from zero; see slice_program in records.cpp).  Rows of x0 / x_out and of the masks stay system-wide.
    constexpr bool GRID_OK = GRID;
    ProgramView pv = a.p;
    if (GRID_OK && a.grid_wgs > 1) pv = a.grid_views[blockIdx.x % a.grid_wgs];
    const uint32_t n = pv.n_vars, m = pv.n_rows, zj = pv.zj, zlo = pv.zlo;
    const uint32_t n_row = a.p.n_vars;  // values per system in x0 / x_out

    // ---- topology program: global/L2, or staged once per workgroup into LDS -------------------------------------
    // PLDS: the leading `stage_bytes` of the blob (the index lists; for sub-wavefront teams the whole blob incl. the
    // constraint table) are copied to LDS and the lists are 16-bit.
    using idx_t = typename std::conditional<PLDS, uint16_t, uint32_t>::type;
    const unsigned char* lbase = pv.base;
    const unsigned char* tbase = pv.base;
    if constexpr (PLDS) {
        const uint4* src = reinterpret_cast<const uint4*>(pv.base);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (uint32_t i = tid; i < pv.stage_bytes / 16; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        lbase = reinterpret_cast<const unsigned char*>(smem);
        if constexpr (MODE == MODE_SUB) tbase = lbase;
    }
    const Prog<idx_t> P = make_prog<idx_t>(pv, lbase, tbase);
    // constraint records: in place from LDS (sub-wavefront teams with a staged program), 32-byte packed records from
    // L2 (workgroup teams with staged lists; the host packs the table exactly when PLDS holds), else the wide record
    constexpr int LIST_CHUNK = (MODE == MODE_SUB) ? 4 : 2;
    constexpr int CON_FORM = PLDS ? (MODE == MODE_SUB ? 0 : 2) : 1;
    using CRef = ConRef<CON_FORM, Prog<idx_t>>;
    const bool unit_w = a.unit_weights != 0;

    // ---- grid team geometry (one system on several workgroups, each keeping its share of the state in LDS) ----------
    const uint32_t grid_wgs = GRID_OK ? a.grid_wgs : 1u;
    const uint32_t grid_wg = GRID_OK ? blockIdx.x % grid_wgs : 0u;  // this workgroup inside its system's group
    const uint32_t grid_slot = blockIdx.x / grid_wgs;               // which system-in-flight
    tm.grid_wgs = grid_wgs;
    tm.grid_wg = grid_wg;
    tm.grid = (GRID_OK && grid_wgs > 1) ? a.grid_scratch + grid_slot : nullptr;
    tm.grid_seq = 0;
    if (tm.grid) {  // continue the slot's sequence numbers where the previous launch left them (this workgroup's own)
        gridchunk_t c0, c1;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(c0)
                     : "v"(&tm.grid->arr[0][0][grid_wg])
                     : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=v"(c1)
                     : "v"(&tm.grid->arr[1][0][grid_wg])
                     : "memory");
        tm.grid_seq = c0.z > c1.z ? c0.z : c1.z;
    }

    // ---- workspace carve-up (doubles) ----------------------------------------------------------------------------
    constexpr int WS_STRIDE = (MODE == MODE_SUB) ? 64 / TEAM : 1;  // systems side by side in a wavefront
    static_assert(WS_STRIDE == 1 || LDSWS, "interleaved workspaces live in LDS");
    WsRef<WS_STRIDE> ws;
    if constexpr (MODE == MODE_SUB) {
        ws.p = smem + a.prog_lds_doubles + (size_t)(tid >> 6) * WS_STRIDE * a.ws_doubles + (tid & 63) / TEAM;
        tm.red = nullptr;
    } else if constexpr (LDSWS) {
        ws.p = smem + a.prog_lds_doubles;
        tm.red = smem + a.prog_lds_doubles + a.ws_doubles;
    } else {
        ws.p = a.gws + (size_t)blockIdx.x * a.ws_doubles;
        tm.red = smem;
    }
    // offsets of x r r_next Jv D L d inside the workspace (of this workgroup's program: a grid team's workgroup holds
    // its own slice only)
    const uint32_t o_x = 0;
    uint32_t o_r = n;
    uint32_t o_rn = n + m;
    bool rec_jglobal = false;
    if constexpr (RECF == 1) rec_jglobal = a.rec_jglobal != 0;
    double* const jglob = rec_jglobal ? a.gws + (size_t)blockIdx.x * a.rec_jstride : nullptr;
    const uint32_t o_j = n + 2 * m;
    const uint32_t o_d = o_j + (rec_jglobal ? 0u : zj);   // Cholesky diagonal, by variable
    const uint32_t o_l = o_d + n;    // strictly-lower L entries, (partition, level) grouped
    const uint32_t o_v = o_l + zlo;  // b, then y, then d (by variable)
    const uint32_t o_i = o_v + n;    // small int area
    int* nwarn;
    if constexpr (LDSWS) {
        nwarn = reinterpret_cast<int*>(&ws[o_i]);
    } else {
        nwarn = reinterpret_cast<int*>(smem + 64);  // LDS even when the bulk state is in global memory
    }

    // ---- this unit's partition --------------------------------------------------------------------------------------
    const PartDesc part = P.parts[(MODE == MODE_PART) ? (uint32_t)(tid >> 6) : 0u];
    const uint32_t con0 = uni(part.con0), con1 = uni(part.con1);
    const idx_t* lvl_cptr = P.lvl_cptr + part.lvl0;
    const idx_t* lvl_sptr = P.lvl_sptr + part.lvl0;
    const idx_t* lvl_grp = P.lvl_grp + part.lvl0;  // lanes per list, by level (fused levels only)
    const uint32_t nlev = uni(part.nlev);
    const uint32_t call0 = uni(lvl_cptr[0]), call1 = uni(lvl_cptr[nlev]);  // all of the partition's (internal) variables
    const uint32_t sall0 = uni(lvl_sptr[0]), sall1 = uni(lvl_sptr[nlev]);  // all of its strictly-lower L slots

    // ---- level staging ---------------------------------------------------------------------------------------------
    // A program read from global memory costs the Cholesky two dependent L2 round trips (ptr -> items) before the
    // first value load, twice per elimination level; with ~40 levels of a few columns each that chain is most of a
    // connected sketch's solve.  One wavefront / one barrier workgroup per system therefore copies each level's lists
    // (a contiguous block of lvl_stream, see pack_program) into LDS with one round of independent 16-byte loads and
    // walks them from there; the three level tables are copied once per workgroup.
    constexpr bool LVL_STAGE = !PLDS && !DENSE && !GRID && !REC && (MODE == MODE_WGB || (MODE == MODE_SUB && TEAM == 64));
    const uint32_t* lvl_tab = nullptr;  // [lvl_cptr | lvl_sptr | lvl_off | lvl_grp | lvl_boff], nlev + 1 words each
    uint32_t* lvl_buf = nullptr;
    if constexpr (LVL_STAGE) {
        if (a.lvl_buf_words) {
            uint32_t* tab = reinterpret_cast<uint32_t*>(smem + a.lvl_lds_off);
            for (uint32_t i = tid; i <= nlev; i += blockDim.x) {
                tab[i] = lvl_cptr[i];
                tab[nlev + 1 + i] = lvl_sptr[i];
                tab[2 * (nlev + 1) + i] = P.lvl_off[i];
                tab[3 * (nlev + 1) + i] = i < nlev ? lvl_grp[i] : 1u;
                tab[4 * (nlev + 1) + i] = P.lvl_boff[i];
            }
            __syncthreads();
            lvl_tab = tab;
            lvl_buf = tab + a.lvl_tab_words + (MODE == MODE_SUB ? (uint32_t)(tid >> 6) * a.lvl_buf_words : 0u);
        }
    }

    // (A variant that kept the first four rounds of constraint records in VGPRs across sweeps and systems was
    // measured slower -- 54 vs 38 us per 2000x2000 system -- and is not kept; records are re-read per sweep.)
    const uint64_t n_teams = (uint64_t)(gridDim.x / grid_wgs) * teams_per_block;
    uint32_t sys_parity = 0;  // grid teams: which of the slot's two warning counters this system uses
    double x_pre[4] = {0.0, 0.0, 0.0, 0.0};  // partitioned teams: the next system's first values per lane, fetched ahead
    bool x_have = false;
    uint32_t x_id[4] = {0, 0, 0, 0};  // ... and the caller's ids of this lane's first four variables (the same for every system)
    if constexpr (MODE == MODE_PART) {
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t ci = call0 + tm.lane + j * 64;
            x_id[j] = ci < call1 ? P.var_of[ci] : 0u;
        }
    }
    if constexpr (REC) {  // (ordered before their first use by every system's first rendezvous)
        if (tid == 0) ws[a.rec_zero] = 0.0;
        if (tid == 0 && rec_jglobal) jglob[zj] = 0.0;  // (the operand of a padding pair among the Jacobian's values)
        uint2* dst = reinterpret_cast<uint2*>(smem + a.rec_desc_off);
        const uint32_t nd = (a.rec_rounds + 2) * (blockDim.x >> 6);  // (two idle rounds behind the last)
        for (uint32_t i = tid; i < nd; i += blockDim.x) dst[i] = a.rec_desc[i];
    }
    const uint64_t n_sys = a.sys_count ? (uint64_t)min(*a.sys_count, (uint32_t)a.batch) : a.batch;
    // (a resident launch -- DoneWord::request, one workgroup -- serves one request after the other on the same buffers)
    // (the word through which thread 0 tells the others: the first double of the workspaces, free between two requests --
    // no static LDS in this kernel, whose dynamic allocation may take the CU's whole 160 KB)
    unsigned long long* const resident_word = reinterpret_cast<unsigned long long*>(LDSWS ? smem + a.prog_lds_doubles : smem);
    const unsigned long long born = wall_clock64();
    DoneWord done = a.done;
    do {
    x_have = false;
    for (uint64_t q = (uint64_t)grid_slot * teams_per_block + team_in_block; q < n_sys; q += n_teams, sys_parity ^= 1u) {
        const uint64_t sys = a.sys_list ? (uint64_t)a.sys_list[q] : q;
        // ---- load the initial values (AoS row, coalesced) ------------------------------------------------------------
#ifdef EZPZ_STAMPS
        int stamp_n = 0;
#endif
        EZPZ_STAMP(1);
        const bool resuming = a.resume != nullptr;
        const double* x0 = (resuming ? a.x_out : a.x0) + sys * n_row;
        if constexpr (MODE == MODE_PART) {
            // each wavefront loads (and later stores) its own partition's variables only: a wavefront that is already
            // on the next system never touches values another one has not stored yet.  The first four values per lane
            // were fetched while the previous system was being solved (the HBM round trip of this load was 5 % of a
            // 2000 x 2000 solve, on the critical path of whichever wavefront finished last).
            constexpr uint32_t XPRE = 4;
#pragma unroll
            for (uint32_t j = 0; j < XPRE; ++j) {
                const uint32_t ci = call0 + tm.lane + j * 64;
                if (ci < call1) ws[o_x + ci] = x_have ? x_pre[j] : x0[x_id[j]];
            }
            for (uint32_t ci = call0 + tm.lane + XPRE * 64; ci < call1; ci += tm.stride) ws[o_x + ci] = x0[P.var_of[ci]];
            x_have = q + n_teams < n_sys;
            if (x_have) {
                // (a resumed system continues from the values the lanes kernel left in x_out, like the load above)
                const double* x1 = (resuming ? a.x_out : a.x0) + (a.sys_list ? (uint64_t)a.sys_list[q + n_teams] : q + n_teams) * n_row;
#pragma unroll
                for (uint32_t j = 0; j < XPRE; ++j) {
                    const uint32_t ci = call0 + tm.lane + j * 64;
                    x_pre[j] = ci < call1 ? x1[x_id[j]] : 0.0;
                }
            }
        } else {
            for (uint32_t i = tlane; i < n; i += tsize) ws[o_x + i] = x0[P.var_of[i]];
        }
        const bool grid_team = GRID_OK && grid_wgs > 1;
        if (grid_team) {
            // shared by the system's workgroups: zeroed by the host / by workgroup 0 two systems ago (see write-back)
            // (grid teams never resume a system: launch.hip, launch_list_walk, passes `resume` to one-workgroup shapes only)
            nwarn = &tm.grid->nwarn[sys_parity];
        } else if (tlane == 0) {
            *nwarn = resuming ? (int)a.resume[q].nwarn : 0;
        }
        tm.team_sync();
        EZPZ_STAMP(2);

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
        double largest = 0.0;  // max |r| of the current residual vector
        double unsat_cnt = 0.0;
        bool r_is_at_x = true;  // the current r was evaluated at exactly the current x (false after a rejected step)
        bool all_satisfied = false;  // FINAL proved it from max |r| alone: no per-constraint count to reduce
        double lambda = a.initial_lambda;
        double step_inf_norm = 0.0;
        uint32_t iterations = a.max_iterations;
        uint32_t converged = 0;
        for (;;) {
            if (mode == STEP) {
                if (it >= a.max_iterations) {  // newton.rs:141-144
                    mode = FINAL;
                } else if (largest <= a.residual_tolerance) {  // newton.rs:50-60
                    iterations = it;
                    converged = 1;
                    mode = FINAL;
                }
            }
            if (mode == STEP) {
                // record walk: the first rounds' records do not depend on any value -- requested before the assembly
                uint4 rpa[RCH] = {}, rpb[RCH] = {};
                const uint32_t rec_nw = blockDim.x >> 6;
                const uint2* const rec_d = reinterpret_cast<const uint2*>(smem + a.rec_desc_off) + (tid >> 6);
                const uint4* const rec_c = a.rec_chunks + (tid & 63);
                // this wavefront's chunks of the round with descriptor `d` (0-3 of them)
                // (always three requests: a request under a condition makes its registers a merge of "loaded" and "kept", the
                // merge a copy, and the copy waits for the load right behind it; the chunks a wavefront does not have are read
                // from one 16-byte address instead -- a single line for all its lanes -- and never looked at)
                auto rec_load = [&](uint32_t fl, uint32_t chunk0, uint4 (&pr)[RCH]) __attribute__((always_inline)) {
                    const uint32_t nch = fl & REC_NCH_MASK;
                    const uint4* src = rec_c + (size_t)chunk0 * 64;
#pragma unroll
                    for (int c = 0; c < RCH; ++c) {
                        const uint4* from = (uint32_t)c < nch ? src + c * 64 : a.rec_chunks;
                        pr[c] = *from;
                    }
                };
                uint32_t rf0 = 0, rc0 = 0;  // this round's descriptor (scalars)
                uint2 rd1 = {0, 0};         // the next round's, as read from LDS
                if constexpr (REC) {  // (the descriptors end with two idle rounds: nothing here is conditional)
                    const uint2 d0 = rec_d[0];
                    rd1 = rec_d[rec_nw];
                    rf0 = uni(d0.x);
                    rc0 = uni(d0.y);
                    rec_load(rf0, rc0, rpa);
                }
                // ---- A = JtJ + lambda I (into L's storage) and b = Jt(-r)  (newton.rs:77-84) ---------------------
                // Record walk builds: from packed pair chunks (SolveArgs::rec_asm_*), one item per lane and trip, the next
                // trip's chunks requested before this trip's are used.
                // (AG / BG: the pair's first / second operand is a Jacobian value kept in global memory, SolveArgs::rec_jglobal)
                auto packed_pass = [&](auto kc, auto ag, auto bg, const uint4* base, uint32_t N, auto&& emit) __attribute__((always_inline)) {
                    constexpr int K = decltype(kc)::value;
                    constexpr bool AG = decltype(ag)::value, BG = decltype(bg)::value;
                    uint32_t i = (uint32_t)tid;
                    if (i >= N) return;
                    uint4 cur[K];
                    for (int k = 0; k < K; ++k) cur[k] = base[(size_t)k * N + i];
                    for (;;) {
                        const uint32_t in = i + (uint32_t)blockDim.x;
                        const bool more = in < N;
                        const uint32_t il = more ? in : i;
                        uint4 nxt[K];
                        for (int k = 0; k < K; ++k) nxt[k] = base[(size_t)k * N + il];
                        // (term by term in list order, products rounded before they are added, b's terms as J * -r: the
                        // list walk's sums bit for bit -- a padding pair adds 0 * 0)
                        double sd = 0.0, sp = 0.0, sn = 0.0;
                        for (int k = 0; k < K; ++k) {
                            const uint32_t w[4] = {cur[k].x, cur[k].y, cur[k].z, cur[k].w};
                            double va[4], vb[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                va[e] = AG ? jglob[w[e] & 0xFFFFu] : smem[w[e] & 0xFFFFu];
                                vb[e] = BG ? jglob[w[e] >> 16] : smem[w[e] >> 16];
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                sd += va[e] * va[e];
                                sp += va[e] * vb[e];
                                sn += va[e] * -vb[e];
                            }
                        }
                        emit(i, sd, sp, sn);
                        if (!more) break;
                        for (int k = 0; k < K; ++k) cur[k] = nxt[k];
                        i = in;
                    }
                };
                bool packed_asm = false;
                if constexpr (RECF == 1) packed_asm = a.rec_asm_kc != 0;
                if constexpr (RECF == 1) if (packed_asm) {
                    auto emit_col = [&](uint32_t v, double sd, double, double sn) {
                        ws[o_d + call0 + v] = sd + lambda;
                        ws[o_v + call0 + v] = sn;
                    };
                    auto emit_slot = [&](uint32_t sl, double, double sp, double) { ws[o_l + sall0 + sl] = sp; };
                    using T_ = std::true_type;
                    using F_ = std::false_type;
                    auto cols = [&](auto kc) {
                        if (rec_jglobal)
                            packed_pass(kc, T_{}, F_{}, a.rec_asm_cols, call1 - call0, emit_col);
                        else
                            packed_pass(kc, F_{}, F_{}, a.rec_asm_cols, call1 - call0, emit_col);
                    };
                    auto slots = [&](auto kc) {
                        if (rec_jglobal)
                            packed_pass(kc, T_{}, T_{}, a.rec_asm_slots, sall1 - sall0, emit_slot);
                        else
                            packed_pass(kc, F_{}, F_{}, a.rec_asm_slots, sall1 - sall0, emit_slot);
                    };
                    switch (a.rec_asm_kc) {
                    case 1: cols(std::integral_constant<int, 1>{}); break;
                    case 2: cols(std::integral_constant<int, 2>{}); break;
                    default: cols(std::integral_constant<int, 3>{}); break;
                    }
                    switch (a.rec_asm_ks) {
                    case 1: slots(std::integral_constant<int, 1>{}); break;
                    case 2: slots(std::integral_constant<int, 2>{}); break;
                    default: slots(std::integral_constant<int, 3>{}); break;
                    }
                }
                if (!packed_asm) {
                for (uint32_t ci = call0 + tm.lane; ci < call1; ci += tm.stride) {
                    const uint32_t v = ci;  // internal variable numbering = schedule order
                    double acc = 0.0, b = 0.0;
                    {
                        const uint32_t q0 = P.colj_ptr[v], q1 = P.colj_ptr[v + 1];
                        double jv[4], rv[4];
                        EZPZ_FOR_PAIRS4(P.colj_items, q0, q1, sl, rw,
                                        (jv[k] = ws[o_j + sl[k]], rv[k] = ws[o_r + rw[k]]),
                                        (acc += jv[k] * jv[k], b += jv[k] * -rv[k]))
                    }
                    ws[o_d + v] = acc + lambda;
                    ws[o_v + v] = b;
                }
                for (uint32_t s = sall0 + tm.lane; s < sall1; s += tm.stride) {
                    double acc = 0.0;
                    {
                        const uint32_t q0 = P.apair_ptr[s], q1 = P.apair_ptr[s + 1];
                        double va[4], vb[4];
                        EZPZ_FOR_PAIRS4(P.apairs, q0, q1, ia, ib, (va[k] = ws[o_j + ia[k]], vb[k] = ws[o_j + ib[k]]),
                                        (acc += va[k] * vb[k]))
                    }
                    ws[o_l + s] = acc;
                }
                }
                tm.phase_sync();
                EZPZ_STAMP(10);
                double bad = 0.0;
                double dmax = __builtin_nan("");  // fmax drops NaN seeds; an all-NaN d stays NaN like reduce(fmax)
                // the sum over a group of g = 2^lg lanes, in every lane of the group: one body, each step behind a scalar test
                // (a switch over g is a search through compares and one body per case: more instructions fetched per round)
                auto rec_group_sum = [&](double v, uint32_t lg) __attribute__((always_inline)) {
                    if (lg >= 1) v += dpp_move<0xB1>(v);
                    if (lg >= 2) v += dpp_move<0x4E>(v);
                    if (lg >= 3) v += dpp_move<0x141>(v);
                    if (lg >= 4) v += dpp_move<0x140>(v);
                    if (lg >= 5) {
                        double x, y;
                        lane_pairs<false>(v, x, y);
                        v = x + y;
                    }
                    if (lg >= 6) {
                        double x, y;
                        lane_pairs<true>(v, x, y);
                        v = x + y;
                    }
                    return v;
                };
                auto rec_group_sum2 = [&](double& u, double& w, uint32_t lg) __attribute__((always_inline)) {  // (two chains interleave)
                    if (lg >= 1) u += dpp_move<0xB1>(u), w += dpp_move<0xB1>(w);
                    if (lg >= 2) u += dpp_move<0x4E>(u), w += dpp_move<0x4E>(w);
                    if (lg >= 3) u += dpp_move<0x141>(u), w += dpp_move<0x141>(w);
                    if (lg >= 4) u += dpp_move<0x140>(u), w += dpp_move<0x140>(w);
                    if (lg >= 5) {
                        double x, y, p, q;
                        lane_pairs<false>(u, x, y);
                        lane_pairs<false>(w, p, q);
                        u = x + y, w = p + q;
                    }
                    if (lg >= 6) {
                        double x, y, p, q;
                        lane_pairs<true>(u, x, y);
                        lane_pairs<true>(w, p, q);
                        u = x + y, w = p + q;
                    }
                };
                if constexpr (REC) {
                    // ---- record walk (records.cpp: build_records) ---------------------------------------------------------------
                    // The factorisation and both substitutions of one connected system as ROUNDS: in a round a group of g
                    // lanes owns one item -- an entry l_ij, or a column's y_j, = (target - sum a_k b_k) / sqrt(A_jj - sum
                    // a_k^2) over ONE list (row j of L; b_k = l_ik or a zero for an entry, y_k for a column), or in the
                    // backward substitution x_j = (y_j - sum l_ij x_i) / d_j -- and each lane's share of the list comes as
                    // ready workspace addresses in records that are requested a round ahead: a round is value loads, a
                    // handful of multiply-adds, the group's sum, a square root, a divide and a store.  A level of the
                    // elimination tree is one or more rounds; the first one starts with the workgroup's rendezvous.
                    // Order inside a round: everything that reads the records of THIS round (the addresses of its operands)
                    // first, then the requests for the next round's records, then the work -- the counter that orders
                    // memory loads is in-order, so a wait for this round's records placed after the new requests would wait
                    // for those as well, a trip to L2 per round.
                    const uint32_t R = a.rec_rounds;
                    // a round's work for a wavefront whose lanes have NP operand pairs each (2, 6 or 10: one, two or three
                    // chunks) -- one straight-line body per count, so that nothing in it is conditional
                    // (an operand: LDS form -- an index into the __shared__ array, pointers into LDS that pass through selects become
                    // generic pointers and their loads flat loads that count with the record requests; wide form -- the workspace)
                    auto at = [&](uint32_t i) -> double& {
                        if constexpr (RECF == 2)
                            return ws[i];
                        else
                            return smem[i];
                    };
                    auto rec_body = [&](auto npc, const uint4 (&cur)[RCH], uint32_t fl, unsigned long long* tst) __attribute__((always_inline)) {
                        constexpr int NP = decltype(npc)::value;
                        const uint32_t lg = (fl >> REC_LG_SHIFT) & 7u;
                        uint32_t ia[NP + 1], ib[NP + 1], i_target, i_diag, i_dest, lane_fl;
                        if constexpr (RECF == 2) {
                            i_target = cur[0].x, i_diag = cur[0].y, i_dest = cur[0].z, lane_fl = cur[0].w;
#pragma unroll
                            for (int k = 0; k < NP; k += 2) {
                                ia[k] = cur[1 + k / 2].x, ib[k] = cur[1 + k / 2].y;
                                ia[k + 1] = cur[1 + k / 2].z, ib[k + 1] = cur[1 + k / 2].w;
                            }
                        } else {
                            // (addresses count doubles from the start of the LDS: the host knows where the workspace lies)
                            uint32_t word[NP];
                            word[0] = cur[0].z, word[1] = cur[0].w;
                            if constexpr (NP > 2) word[2] = cur[1].x, word[3] = cur[1].y, word[4] = cur[1].z, word[5] = cur[1].w;
                            if constexpr (NP > 6) word[6] = cur[2].x, word[7] = cur[2].y, word[8] = cur[2].z, word[9] = cur[2].w;
#pragma unroll
                            for (int k = 0; k < NP; ++k) ia[k] = word[k] & 0xFFFFu, ib[k] = word[k] >> 16;
                            i_target = cur[0].x & 0xFFFFu, i_diag = cur[0].x >> 16, i_dest = cur[0].y & 0xFFFFu, lane_fl = cur[0].y;
                        }
                        double va[NP + 1], vb[NP + 1];
#pragma unroll
                        for (int k = 0; k < NP; ++k) {
                            va[k] = at(ia[k]);
                            vb[k] = at(ib[k]);
                        }
                        const double target = at(i_target), diag = at(i_diag);
                        const bool writer = (lane_fl & REC_WRITER) != 0;
                        // (fused multiply-adds, two partial sums each: the group's sum reorders the terms anyway)
                        double sp0 = 0.0, sp1 = 0.0;
                        if constexpr (NP > 0) sp0 = va[0] * vb[0], sp1 = va[1] * vb[1];
#pragma unroll
                        for (int k = 2; k < NP; k += 2) {
                            sp0 = __builtin_fma(va[k], vb[k], sp0);
                            sp1 = __builtin_fma(va[k + 1], vb[k + 1], sp1);
                        }
                        double sp = sp0 + sp1;
#ifdef EZPZ_REC_TIMES
                        if (tst) { asm volatile("" : "+v"(sp)); tst[2] = __builtin_readcyclecounter(); }
#endif
                        if (fl & REC_BWD) {
                            sp = rec_group_sum(sp, lg);
                            const double res = (target - sp) * diag;  // (the factor's diagonal is kept as 1 / d_j)
                            if (writer) {
                                at(i_dest) = res;
                                dmax = fmax(dmax, fabs(res));
                            }
                        } else {
                            double sd0 = 0.0, sd1 = 0.0;
                            if constexpr (NP > 0) sd0 = va[0] * va[0], sd1 = va[1] * va[1];
#pragma unroll
                            for (int k = 2; k < NP; k += 2) {
                                sd0 = __builtin_fma(va[k], va[k], sd0);
                                sd1 = __builtin_fma(va[k + 1], va[k + 1], sd1);
                            }
                            double sd = sd0 + sd1;
                            rec_group_sum2(sd, sp, lg);
#ifdef EZPZ_REC_TIMES
                            if (tst) { asm volatile("" : "+v"(sp), "+v"(sd)); tst[3] = __builtin_readcyclecounter(); }
#endif
                            // 1 / sqrt(pivot) from the hardware's estimate and two coupled Newton steps (g -> sqrt, h -> 1 / (2
                            // sqrt)): nine instructions on the level's critical path where a correctly rounded square root
                            // followed by a correctly rounded division is forty; the last bit may differ from theirs
                            const double acc = diag - sd;
                            const double y0 = __builtin_amdgcn_rsq(acc);
                            const double g0 = acc * y0, h0 = 0.5 * y0;
                            const double r0 = __builtin_fma(-g0, h0, 0.5);
                            const double g1 = __builtin_fma(g0, r0, g0), h1 = __builtin_fma(h0, r0, h0);
                            const double r1 = __builtin_fma(-g1, h1, 0.5);
                            const double h2 = __builtin_fma(h1, r1, h1);
                            const double rinv = h2 + h2;
                            double res = (target - sp) * rinv;
#ifdef EZPZ_REC_TIMES
                            if (tst) { asm volatile("" : "+v"(res)); tst[4] = __builtin_readcyclecounter(); }
#endif
                            if (lane_fl & REC_ISCOL) {
                                if (!(acc > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                                if (writer) at(i_diag + a.rec_dd_delta) = rinv;
                            }
                            if (writer) at(i_dest) = res;
                        }
                    };
                    auto rec_round = [&](uint32_t rd, uint4 (&cur)[RCH], uint4 (&nxt)[RCH]) __attribute__((always_inline)) {
                        const uint32_t fl = rf0;
                        const uint32_t nch = fl & REC_NCH_MASK;
                        const uint32_t rf1 = uni(rd1.x), rc1 = uni(rd1.y);
#ifdef EZPZ_REC_TIMES
                        const bool stamping = a.stamps && blockIdx.x == 0 && tid == 0 && it == 1 && rd < 126;
                        unsigned long long* const tstamp = reinterpret_cast<unsigned long long*>(&ws[a.rec_zero + 2]) + 6 * rd;
                        if (stamping) tstamp[0] = __builtin_readcyclecounter();
#endif
                        // this round's records must have arrived BEFORE the next round's are requested: the counter that orders
                        // memory loads is in-order, a wait placed after the new requests would wait for those as well
                        // (unconditionally, all three chunks: a wait under a condition leaves the compiler's bookkeeping with
                        // "may be pending" at the join, and it waits again at the first use -- behind the new requests)
#pragma unroll
                        for (int c = 0; c < RCH; ++c) asm volatile("" : "+v"(cur[c].x), "+v"(cur[c].y), "+v"(cur[c].z), "+v"(cur[c].w));
                        __builtin_amdgcn_sched_barrier(0);
                        rec_load(rf1, rc1, nxt);
                        rd1 = rec_d[(rd + 2) * rec_nw];
                        __builtin_amdgcn_sched_barrier(0);
                        // (the rendezvous orders LDS traffic only: __syncthreads() would also wait for the record requests just
                        // made -- a release at workgroup scope drains the memory-load counter -- a trip to L2 in every round)
                        // (the wide form exchanges through global memory: there the release is what is needed)
                        if (fl & REC_BARRIER) {
                            if constexpr (RECF == 2)
                                __syncthreads();
                            else
                                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        }
#ifdef EZPZ_REC_TIMES
                        if (stamping) tstamp[1] = __builtin_readcyclecounter();
#endif
#ifdef EZPZ_REC_TIMES
#define REC_TST (stamping ? tstamp : nullptr)
#else
#define REC_TST nullptr
#endif
                        if constexpr (RECF == 2) {
                            switch (nch) {
                            case 1: rec_body(std::integral_constant<int, 0>{}, cur, fl, REC_TST); break;
                            case 2: rec_body(std::integral_constant<int, 2>{}, cur, fl, REC_TST); break;
                            case 3: rec_body(std::integral_constant<int, 4>{}, cur, fl, REC_TST); break;
                            case 4: rec_body(std::integral_constant<int, 6>{}, cur, fl, REC_TST); break;
                            case 5: rec_body(std::integral_constant<int, 8>{}, cur, fl, REC_TST); break;
                            default: break;
                            }
                        } else {
                            switch (nch) {
                            case 1: rec_body(std::integral_constant<int, 2>{}, cur, fl, REC_TST); break;
                            case 2: rec_body(std::integral_constant<int, 6>{}, cur, fl, REC_TST); break;
                            case 3: rec_body(std::integral_constant<int, 10>{}, cur, fl, REC_TST); break;
                            default: break;
                            }
                        }
#undef REC_TST
                        rf0 = rf1;
                        rc0 = rc1;
#ifdef EZPZ_REC_TIMES
                        if (stamping) tstamp[5] = __builtin_readcyclecounter();
#endif
                    };
                    for (uint32_t rd = 0; rd < R; rd += 2) {  // (R is even: build_records pads)
                        rec_round(rd, rpa, rpb);
                        rec_round(rd + 1, rpb, rpa);
                    }
                    // (the rendezvous of the reduction below orders the last round's stores before x + d)
#ifdef EZPZ_REC_TIMES
                    if (a.stamps && blockIdx.x == 0 && tid == 0 && it == 1)
                        for (uint32_t rd = 0; rd < 6 * R && rd < 6 * 126; ++rd)
                            a.stamps[4096 + rd] = reinterpret_cast<unsigned long long*>(&ws[a.rec_zero + 2])[rd];
#endif
                    EZPZ_STAMP(11);
                } else if constexpr (DENSE) {
                    // ---- <= 8 variables, four lanes: the whole linear solve in registers (dense8_solve) --------------------
                    bad = dense8_solve(ws, o_d, o_l, o_v, lvl_sptr, n, tm.lane);
                    tm.phase_sync();
                    EZPZ_STAMP(11);
                } else {
                // ---- level-scheduled sparse Cholesky + forward substitution (newton.rs:87-102) --------------------
                // One level = the columns [c0, c1) and the strictly-lower slots [s0, s1) of L.  `fptr` is indexed by
                // column and `lptr` / `lcol` by slot (absolute numbers: the staged copies are biased accordingly), the
                // list bounds they hold are relative to `fitems` / `lpairs`.
                // Teams that own a whole connected system (one wavefront, or a barrier workgroup) are bound by the chain
                // of dependent hops per level, so they run a level as ONE phase when it has no more columns than lanes:
                // the lane of column v takes d_v = sqrt(A_vv - sum l_vk^2) and y_v at once, and every lane of a slot
                // (i, j) recomputes d_j itself (same terms, same order, same bits) instead of waiting a phase for it.
                // d_v is stored after the rendezvous: until then A_vv is still being read, and nothing reads d_v before
                // the backward substitution (which the same lane does for v).
                constexpr bool FUSE_LEVEL = MODE == MODE_WGB || (MODE == MODE_SUB && TEAM == 64);
                // The top of an elimination tree is narrow and dense (a separator's columns: 1-10 columns whose lists
                // hold 20-50 terms), and a level lasts as long as its longest list.  There `g` lanes share every list
                // (the host picks g per level, choose_level_groups): each takes every g-th term, and the group adds
                // the partial sums in a fixed DPP / shuffle tree (the same on every run; the rounding differs from the
                // one-lane order, as it does between any two elimination orders).
                auto group_sum = [&](double v, uint32_t g) {
                    switch (g) {
                    case 2: return reduce_lanes<2>(v, OpSum());
                    case 4: return reduce_lanes<4>(v, OpSum());
                    case 8: return reduce_lanes<8>(v, OpSum());
                    case 16: return reduce_lanes<16>(v, OpSum());
                    case 32: return reduce_lanes<32>(v, OpSum());
                    default: return reduce_lanes<64>(v, OpSum());
                    }
                };
                // two sums of the same group in one basic block: the two chains of cross-lane moves interleave
                auto group_sum2 = [&](double& u, double& w, uint32_t g) {
                    switch (g) {
                    case 2: u = reduce_lanes<2>(u, OpSum()), w = reduce_lanes<2>(w, OpSum()); break;
                    case 4: u = reduce_lanes<4>(u, OpSum()), w = reduce_lanes<4>(w, OpSum()); break;
                    case 8: u = reduce_lanes<8>(u, OpSum()), w = reduce_lanes<8>(w, OpSum()); break;
                    case 16: u = reduce_lanes<16>(u, OpSum()), w = reduce_lanes<16>(w, OpSum()); break;
                    case 32: u = reduce_lanes<32>(u, OpSum()), w = reduce_lanes<32>(w, OpSum()); break;
                    default: u = reduce_lanes<64>(u, OpSum()), w = reduce_lanes<64>(w, OpSum()); break;
                    }
                };
                auto chol_level = [&](uint32_t c0, uint32_t c1, uint32_t s0, uint32_t s1, uint32_t g, auto fptr,
                                      auto fitems, auto lptr, auto lpairs, auto lcol) __attribute__((always_inline)) {
                    // One phase per level: its columns and its slots are work items of ONE walk.  Item (i, j) -- a slot, or
                    // a column's forward substitution with i = "y" -- is (target - sum of pairs) / sqrt(A_jj - sum l_jk^2):
                    // a column's pairs are (l_vk, y_k) on the list that also gives its diagonal, a slot's are the
                    // (l_ik, l_jk) of its own list.  Columns and slots used to run one after the other on the same
                    // lanes (two chains of hops, sqrt and divide per level: ~4 k cycles a level on a 300-variable sketch);
                    // as one walk a level is one chain.
                    // (Issuing the first round of both lists' hops together -- four round trips instead of seven -- was
                    // measured and not kept: 602 k vs 560 k cycles of factorisation per 300-variable solve; a level is
                    // bound by the ~400 instructions a wavefront issues for it, not by the hops.)
                    if (FUSE_LEVEL && g > 1) {  // (c1 - c0) * g <= lanes, by construction
                        const uint32_t lg = (uint32_t)__builtin_ctz(g);
                        const uint32_t sub = (uint32_t)tm.lane & (g - 1), grp = (uint32_t)tm.lane >> lg;
                        const uint32_t ngrp = (uint32_t)tm.stride >> lg;
                        const uint32_t ncol = c1 - c0, nitem = ncol + (s1 - s0);
                        double dv = 0.0;
                        for (uint32_t t = grp; t < nitem; t += ngrp) {
                            const bool iscol = t < ncol;
                            const uint32_t sl_ = s0 + (t - ncol);
                            uint32_t j = c0 + t;
                            if (!iscol) j = lcol[sl_];
                            double sd = 0.0, sp = 0.0;
                            {
                                const uint32_t q0 = fptr[j], q1 = fptr[j + 1];
                                double l[4], yk[4];
                                EZPZ_FOR_PAIRS_STRIDED(fitems, q0 + sub, q1, g, sl, vk,
                                                       (l[k] = ws[o_l + sl[k]], yk[k] = iscol ? ws[o_v + vk[k]] : 0.0),
                                                       (sd += l[k] * l[k], sp += iscol ? l[k] * yk[k] : 0.0))
                            }
                            if (!iscol) {
                                const uint32_t q0 = lptr[sl_], q1 = lptr[sl_ + 1];
                                double va[4], vb[4];
                                EZPZ_FOR_PAIRS_STRIDED(lpairs, q0 + sub, q1, g, ia, ib,
                                                       (va[k] = ws[o_l + ia[k]], vb[k] = ws[o_l + ib[k]]),
                                                       (sp += va[k] * vb[k]))
                            }
                            const double target = ws[iscol ? o_v + j : o_l + sl_];
                            const double diag = ws[o_d + j];
                            group_sum2(sd, sp, g);
                            const double acc = diag - sd;
                            const double dj = sqrt(acc);
                            const double res = (target - sp) / dj;
                            if (iscol) {
                                if (!(acc > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                                dv = dj;
                            }
                            if (sub == 0) {
                                if (iscol)
                                    ws[o_v + j] = res;
                                else
                                    ws[o_l + sl_] = res;
                            }
                        }
                        // d_v goes out after the rendezvous: until then A_vv is still being read
                        tm.phase_sync();
                        if (grp < ncol && sub == 0) ws[o_d + c0 + grp] = dv;
                        return;
                    }
                    if (FUSE_LEVEL && c1 - c0 <= (uint32_t)tm.stride) {
                        // one lane per item, every term subtracted from the entry in list order: bit for bit the
                        // textbook left-looking order
                        const uint32_t ncol = c1 - c0, nitem = ncol + (s1 - s0);
                        double dv = 0.0;
                        for (uint32_t t = tm.lane; t < nitem; t += tm.stride) {
                            const bool iscol = t < ncol;
                            const uint32_t sl_ = s0 + (t - ncol);
                            uint32_t j = c0 + t;
                            if (!iscol) j = lcol[sl_];
                            double dj = ws[o_d + j];
                            double acc = ws[iscol ? o_v + j : o_l + sl_];
                            {
                                const uint32_t q0 = fptr[j], q1 = fptr[j + 1];
                                double l[4], yk[4];
                                EZPZ_FOR_PAIRS4(fitems, q0, q1, sl, vk, (l[k] = ws[o_l + sl[k]], yk[k] = iscol ? ws[o_v + vk[k]] : 0.0),
                                                (dj -= l[k] * l[k], acc -= iscol ? l[k] * yk[k] : 0.0))
                            }
                            if (!iscol) {
                                const uint32_t q0 = lptr[sl_], q1 = lptr[sl_ + 1];
                                double va[4], vb[4];
                                EZPZ_FOR_PAIRS4(lpairs, q0, q1, ia, ib, (va[k] = ws[o_l + ia[k]], vb[k] = ws[o_l + ib[k]]),
                                                (acc -= va[k] * vb[k]))
                            }
                            const double root = sqrt(dj);
                            const double res = acc / root;
                            if (iscol) {
                                if (!(dj > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                                dv = root;
                                ws[o_v + j] = res;
                            } else {
                                ws[o_l + sl_] = res;
                            }
                        }
                        tm.phase_sync();
                        if ((uint32_t)tm.lane < ncol) ws[o_d + c0 + tm.lane] = dv;
                        return;
                    }
                    for (uint32_t ci = c0 + tm.lane; ci < c1; ci += tm.stride) {
                        const uint32_t v = ci;  // internal variable numbering = schedule order
                        double acc = ws[o_d + v];
                        {
                            // row v of L is complete (its entries belong to columns of earlier levels), so the forward
                            // substitution's sum for y_v rides on the same walk: one pass over the list instead of two
                            const uint32_t q0 = fptr[v], q1 = fptr[v + 1];
                            double y = ws[o_v + v];
                            double va[4], vb[4];
                            EZPZ_FOR_PAIRS4(fitems, q0, q1, sl, vk, (va[k] = ws[o_l + sl[k]], vb[k] = ws[o_v + vk[k]]),
                                            (acc -= va[k] * va[k], y -= va[k] * vb[k]))
                            if (!(acc > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                            const double dv = sqrt(acc);
                            ws[o_d + v] = dv;
                            ws[o_v + v] = y / dv;
                        }
                    }
                    for (uint32_t s = s0 + tm.lane; s < s1; s += tm.stride) {
                        double acc = ws[o_l + s];
                        {
                            const uint32_t q0 = lptr[s], q1 = lptr[s + 1];
                            double va[4], vb[4];
                            EZPZ_FOR_PAIRS4(lpairs, q0, q1, ia, ib, (va[k] = ws[o_l + ia[k]], vb[k] = ws[o_l + ib[k]]),
                                            (acc -= va[k] * vb[k]))
                        }
                        ws[o_l + s] = acc;
                    }
                    tm.phase_sync();
                    for (uint32_t s = s0 + tm.lane; s < s1; s += tm.stride)
                        ws[o_l + s] = ws[o_l + s] / ws[o_d + lcol[s]];
                    tm.phase_sync();
                };
                // ---- dense phases (Program::n_dense, records.cpp: make_dense_phases) ------------------------------------------
                // The last levels of such a program are PHASES: runs of whole levels of the elimination tree's top, whose
                // columns fall into independent blocks of K <= 16 columns (the last phase is the root block).  A block is a
                // dense panel of R rows in LDS: its K columns, the later columns that have entries in them, and b.
                // (1) All lanes, g per list: every structural entry of the phase minus its terms from the columns before
                // the phase (the Schur complement; structurally zero entries stay zero through the factorisation: no fill
                // is created that the symbolic phase had not found).  (2) One wavefront per block, lane = row, no barriers:
                // right-looking dense Cholesky in registers with the forward substitution as row R - 1, pivots and
                // multipliers by v_readlane -- each entry still receives its terms in ascending column order.  (3) All
                // lanes: the factor's entries and y back to the workspace for the phases above.  The backward
                // substitution runs the phases in reverse, one wavefront per block: the rows below the block first (their
                // d is final), then the block's triangle.  ~1 k cycles per column of the longest block instead of
                // ~4 k per level.  The sums run in a different order than the list walk's (as between any two elimination
                // orders).
                constexpr bool ROOT_OK = (MODE == MODE_WGB || (MODE == MODE_SUB && TEAM == 64)) && !GRID && !DENSE && !REC;
                // (one wavefront per system: every team has its own panels and walks all blocks of a phase itself)
                double* const dense_base = smem + a.dense_lds_off + (MODE == MODE_SUB ? (size_t)(tid >> 6) * a.dense_lds_doubles : 0);
                const uint32_t dense_wave = MODE == MODE_SUB ? 0u : (uint32_t)tid >> 6;
                auto readlane_f64 = [](double v, uint32_t l) {
                    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, (int)l);
                    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), (int)l);
                    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
                };
                const uint32_t dense_c0 = ROOT_OK && a.n_dense ? uni(lvl_cptr[a.dense_level0]) : 0u;
                const uint32_t dense_s0 = ROOT_OK && a.n_dense ? uni(lvl_sptr[a.dense_level0]) : 0u;
                auto dense_phase = [&](uint32_t lv) __attribute__((always_inline)) {
                    const idx_t* rec = P.dense_tab + uni(P.dense_tab[1 + (lv - a.dense_level0)]);
                    const uint32_t nb = uni(rec[0]);
                    double* DB = dense_base;
                    const uint32_t c0 = uni(lvl_cptr[lv]), s0 = uni(lvl_sptr[lv]);
                    const uint32_t ncol = uni(lvl_cptr[lv + 1]) - c0, nitem = ncol + (uni(lvl_sptr[lv + 1]) - s0);
                    const uint32_t g = uni(lvl_grp[lv]) & 0xFFu;
                    const uint32_t lg = (uint32_t)__builtin_ctz(g);
                    const uint32_t sub = (uint32_t)tm.lane & (g - 1), grp = (uint32_t)tm.lane >> lg;
                    const uint32_t ngrp = (uint32_t)tm.stride >> lg;
                    const bool last = lv + 1 == nlev;
                    // where an item's entry lives: (block, column, row) -> panel address; a column item also owns b's row
                    auto place = [&](bool iscol, uint32_t t, uint32_t& e_main, uint32_t& e_b) {
                        const uint32_t code = iscol ? P.dense_col[c0 + t - dense_c0] : P.dense_slot[s0 + (t - ncol) - dense_s0];
                        const idx_t* bk = rec + 1 + 5 * (code & 15u);
                        const uint32_t lc = (code >> 4) & 15u, lr = iscol ? lc : (code >> 8);
                        const uint32_t off = bk[2], st = bk[3];
                        e_main = off + lr * st + lc;
                        e_b = off + (bk[1] - 1u) * st + lc;
                    };
                    for (uint32_t t = grp; t < nitem; t += ngrp) {
                        const bool iscol = t < ncol;
                        const uint32_t sl_ = s0 + (t - ncol);
                        uint32_t q0, q1;
                        if (iscol) {
                            q0 = P.fwd_ptr[c0 + t];
                            q1 = P.fwd_ptr[c0 + t + 1];
                        } else {
                            q0 = P.lpair_ptr[sl_];
                            q1 = P.lpair_ptr[sl_ + 1];
                        }
                        const idx_t* lst = iscol ? P.fwd_items : P.lpairs;  // (l_jk, y_k) or (l_ik, l_jk)
                        const uint32_t o_b = iscol ? o_v : o_l;
                        double sp = 0.0, sd = 0.0;
                        double va[4], vb[4];
                        EZPZ_FOR_PAIRS_STRIDED(lst, q0 + sub, q1, g, ia, ib, (va[k] = ws[o_l + ia[k]], vb[k] = ws[o_b + ib[k]]),
                                               (sp += va[k] * vb[k], sd += va[k] * va[k]))
                        if (g > 1) group_sum2(sp, sd, g);
                        if (sub == 0) {
                            uint32_t e_main, e_b;
                            place(iscol, t, e_main, e_b);
                            if (iscol) {
                                DB[e_main] = ws[o_d + c0 + t] - sd;
                                DB[e_b] = ws[o_v + c0 + t] - sp;
                            } else {
                                DB[e_main] = ws[o_l + sl_] - sp;
                            }
                        }
                    }
                    tm.phase_sync();
                    for (uint32_t blk = uni(dense_wave); blk < nb; blk += (uint32_t)tm.stride >> 6) {
                        const idx_t* bk = rec + 1 + 5 * blk;
                        const uint32_t K = uni(bk[0]), R = uni(bk[1]), ST = uni(bk[3]);
                        double* D = DB + uni(bk[2]);
                        const uint32_t r = (uint32_t)tid & 63u, rr = r < R ? r : R - 1;
                        double rg[16];
#pragma unroll
                        for (int c = 0; c < 16; ++c) rg[c] = (uint32_t)c < K ? D[rr * ST + c] : 0.0;
#pragma unroll
                        for (int j = 0; j < 16; ++j) {
                            if ((uint32_t)j < K) {
                                const double piv = readlane_f64(rg[j], j);
                                if (!(piv > 0.0)) bad = 1.0;  // LltError::Numeric: non-positive pivot
                                const double dj = sqrt(piv);
                                const double l = rg[j] / dj;
                                if (r == (uint32_t)j)
                                    D[j * ST + j] = dj;
                                else if (r > (uint32_t)j && r < R)
                                    D[r * ST + j] = l;
                                // (a second, 32-column body beside this one costs the whole kernel its register
                                // allocation: one 300-variable solve 296 -> 368 us even when only this one runs)
#pragma unroll
                                for (int k = j + 1; k < 16; ++k) rg[k] -= l * readlane_f64(l, k);
                            }
                        }
                    }
                    tm.phase_sync();
                    if (!last) {  // the phases above read the factor's entries and y from the workspace
                        for (uint32_t t = (uint32_t)tm.lane; t < nitem; t += (uint32_t)tm.stride) {
                            const bool iscol = t < ncol;
                            uint32_t e_main, e_b;
                            place(iscol, t, e_main, e_b);
                            if (iscol)
                                ws[o_v + c0 + t] = DB[e_b];
                            else
                                ws[o_l + s0 + (t - ncol)] = DB[e_main];
                        }
                        tm.phase_sync();
                    }
                };
                auto dense_bwd = [&](uint32_t lv) __attribute__((always_inline)) {
                    const idx_t* rec = P.dense_tab + uni(P.dense_tab[1 + (lv - a.dense_level0)]);
                    const uint32_t nb = uni(rec[0]);
                    for (uint32_t blk = uni(dense_wave); blk < nb; blk += (uint32_t)tm.stride >> 6) {
                        const idx_t* bk = rec + 1 + 5 * blk;
                        const uint32_t K = uni(bk[0]), R = uni(bk[1]), ST = uni(bk[3]);
                        const double* D = dense_base + uni(bk[2]);
                        const idx_t* rowvar = P.dense_tab + uni(bk[4]);
                        const uint32_t r = (uint32_t)tid & 63u, rc = r < K ? r : K - 1;
                        double rem = D[(R - 1) * ST + rc], xr = 0.0;  // y_r minus the terms of the rows already solved
                        // lane q holds d of the q-th row below the block (final: its phase ran before this one)
                        const double xb = ws[o_v + rowvar[K + r < R - 1 ? K + r : K > 0 ? K - 1 : 0]];
                        for (uint32_t q = K; q + 1 < R; ++q) rem -= D[q * ST + rc] * readlane_f64(xb, q - K);
                        const double dr = D[rc * ST + rc];
                        double lrow[16];  // column r of the triangle: what row j subtracts from y_r
#pragma unroll
                        for (int j = 0; j < 16; ++j) lrow[j] = (uint32_t)j < K ? D[j * ST + rc] : 0.0;
#pragma unroll
                        for (int j = 15; j >= 0; --j) {
                            if ((uint32_t)j < K) {
                                // (rem * (1 / d_r) instead of the division was measured: 233.6 -> 228.1 us per 300-variable
                                // solve, not worth leaving the correctly rounded quotient)
                                const double xj = readlane_f64(rem / dr, j);
                                if (r < (uint32_t)j) rem -= lrow[j] * xj;
                                if (r == (uint32_t)j) xr = xj;
                            }
                        }
                        if (r < K) {
                            ws[o_v + rowvar[r]] = xr;
                            dmax = fmax(dmax, fabs(xr));
                        }
                    }
                    tm.phase_sync();
                };
                if constexpr (ROOT_OK) {
                    if (a.n_dense) {  // (the levels' rendezvous order this before the panels' entries are written)
                        for (uint32_t i = (uint32_t)tm.lane; i < a.dense_lds_doubles; i += (uint32_t)tm.stride) dense_base[i] = 0.0;
                    }
                }
                for (uint32_t lv = 0; lv < nlev; ++lv) {
                    if constexpr (ROOT_OK) {
                        if (a.n_dense && lv >= a.dense_level0) {
                            dense_phase(lv);
                            EZPZ_STAMP(1000 + lv);
                            continue;
                        }
                    }
                    if constexpr (LVL_STAGE) {
                        if (lvl_buf) {
                            const uint32_t w0 = uni(lvl_tab[2 * (nlev + 1) + lv]), w1 = uni(lvl_tab[2 * (nlev + 1) + lv + 1]);
                            if (w1 - w0 <= a.lvl_buf_words) {
                                const uint32_t c0 = uni(lvl_tab[lv]), c1 = uni(lvl_tab[lv + 1]);
                                const uint32_t s0 = uni(lvl_tab[nlev + 1 + lv]), s1 = uni(lvl_tab[nlev + 1 + lv + 1]);
                                const uint4* src = reinterpret_cast<const uint4*>(P.lvl_stream + w0);
                                uint4* dst = reinterpret_cast<uint4*>(lvl_buf);
                                for (uint32_t i = tm.lane; i < (w1 - w0) / 4; i += tm.stride) dst[i] = src[i];
                                tm.phase_sync();
                                const uint32_t n_fwd = uni(lvl_buf[0]), n_pairs = uni(lvl_buf[1]);
                                const uint32_t* fptr = lvl_buf + 2;
                                const uint32_t* fitems = fptr + ((c1 - c0 + 2) & ~1u);
                                const uint32_t* lptr = fitems + 2 * n_fwd;
                                const uint32_t* lpairs = lptr + ((s1 - s0 + 2) & ~1u);
                                const uint32_t* lcol = lpairs + 2 * n_pairs;
                                chol_level(c0, c1, s0, s1, uni(lvl_tab[3 * (nlev + 1) + lv]) & 0xFFu, fptr - c0, fitems, lptr - s0, lpairs,
                                           lcol - s0);
                                EZPZ_STAMP(2000 + lv);
                                continue;
                            }
                        }
                    }
                    chol_level(uni(lvl_cptr[lv]), uni(lvl_cptr[lv + 1]), uni(lvl_sptr[lv]), uni(lvl_sptr[lv + 1]),
                               FUSE_LEVEL ? (uni(lvl_grp[lv]) & 0xFFu) : 1u, P.fwd_ptr, P.fwd_items, P.lpair_ptr, P.lpairs, P.l_col);
                    EZPZ_STAMP(1000 + lv);
                }
                // a one-phase level stores its d_v after its rendezvous: order the last level's before the backward
                // substitution, whose lane for v may sit in another wavefront (different group sizes)
                if constexpr (FUSE_LEVEL) tm.phase_sync();
                EZPZ_STAMP(11);
                // ---- backward substitution (garbage but harmless if the factorisation failed) ---------------------------
                // Same two devices as the factorisation: the level's lists come from LDS when they were staged, and g
                // lanes share a column's list where the level has lanes to spare (high byte of lvl_grp).
                auto bwd_level = [&](uint32_t c0, uint32_t c1, uint32_t g, auto bptr, auto bitems)
                                     __attribute__((always_inline)) {
                    if (FUSE_LEVEL && g > 1) {  // (c1 - c0) * g <= lanes, by construction
                        const uint32_t lg = (uint32_t)__builtin_ctz(g);
                        const uint32_t sub = (uint32_t)tm.lane & (g - 1), v = c0 + ((uint32_t)tm.lane >> lg);
                        if (v < c1) {
                            double sy = 0.0;
                            const uint32_t q0 = bptr[v], q1 = bptr[v + 1];
                            double va[4], vb[4];
                            EZPZ_FOR_PAIRS_STRIDED(bitems, q0 + sub, q1, g, sl, vi,
                                                   (va[k] = ws[o_l + sl[k]], vb[k] = ws[o_v + vi[k]]), (sy += va[k] * vb[k]))
                            sy = group_sum(sy, g);
                            if (sub == 0) {
                                const double dval = (ws[o_v + v] - sy) / ws[o_d + v];
                                ws[o_v + v] = dval;
                                dmax = fmax(dmax, fabs(dval));
                            }
                        }
                    } else {
                        for (uint32_t ci = c0 + tm.lane; ci < c1; ci += tm.stride) {
                            const uint32_t v = ci;  // internal variable numbering = schedule order
                            double acc = ws[o_v + v];
                            {
                                const uint32_t q0 = bptr[v], q1 = bptr[v + 1];
                                double va[4], vb[4];
                                EZPZ_FOR_PAIRS4(bitems, q0, q1, sl, vi, (va[k] = ws[o_l + sl[k]], vb[k] = ws[o_v + vi[k]]),
                                                (acc -= va[k] * vb[k]))
                            }
                            const double dval = acc / ws[o_d + v];
                            ws[o_v + v] = dval;
                            dmax = fmax(dmax, fabs(dval));
                        }
                    }
                    tm.phase_sync();
                };
                for (uint32_t lv = nlev; lv-- > 0;) {
                    if constexpr (ROOT_OK) {
                        if (a.n_dense && lv >= a.dense_level0) {
                            dense_bwd(lv);
                            continue;
                        }
                    }
                    if constexpr (LVL_STAGE) {
                        if (lvl_buf) {
                            const uint32_t w0 = uni(lvl_tab[4 * (nlev + 1) + lv]), w1 = uni(lvl_tab[4 * (nlev + 1) + lv + 1]);
                            if (w1 - w0 <= a.lvl_buf_words) {
                                const uint32_t c0 = uni(lvl_tab[lv]), c1 = uni(lvl_tab[lv + 1]);
                                const uint4* src = reinterpret_cast<const uint4*>(P.lvl_bstream + w0);
                                uint4* dst = reinterpret_cast<uint4*>(lvl_buf);
                                for (uint32_t i = tm.lane; i < (w1 - w0) / 4; i += tm.stride) dst[i] = src[i];
                                tm.phase_sync();
                                const uint32_t* bptr = lvl_buf + 2;
                                const uint32_t* bitems = bptr + ((c1 - c0 + 2) & ~1u);
                                bwd_level(c0, c1, (uni(lvl_tab[3 * (nlev + 1) + lv]) >> 8) & 0xFFu, bptr - c0, bitems);
                                continue;
                            }
                        }
                    }
                    bwd_level(uni(lvl_cptr[lv]), uni(lvl_cptr[lv + 1]), FUSE_LEVEL ? ((uni(lvl_grp[lv]) >> 8) & 0xFFu) : 1u, P.bwd_ptr,
                              P.bwd_items);
                }
                }
                EZPZ_STAMP(12);
                // ---- ||d||_inf and "did any pivot fail": one rendezvous (newton.rs:96-99, :108) ------------------------------
                // (the list-walk builds took max |d_v| while the backward substitution produced d_v)
                if constexpr (DENSE) {
                    for (uint32_t ci = call0 + tm.lane; ci < call1; ci += tm.stride)
                        dmax = fmax(dmax, fabs(ws[o_v + ci]));
                }
                tm.reduce2(bad, dmax, OpMax(), OpMax());
                EZPZ_STAMP(13);
                if (bad > 0.0) {  // numeric failure => lambda *= 10, burn the iteration
                    lambda *= LM_LAMBDA_INCR;
                    ++it;
                    continue;
                }
                step_inf_norm = (n > 0) ? dmax : 0.0;
                // ---- tentative step (newton.rs:111-114) ---------------------------------------------------------------------
                for (uint32_t ci = call0 + tm.lane; ci < call1; ci += tm.stride) {
                    const uint32_t v = ci;  // internal variable numbering = schedule order
                    ws[o_x + v] = ws[o_x + v] + ws[o_v + v];
                }
                tm.phase_sync();
                EZPZ_STAMP(14);
            }

            if (mode == FINAL && r_is_at_x && a.unit_weights) {
                // every weight is 1 and r was evaluated at this x: r already holds the unweighted residuals, so the
                // unsatisfied check reads it instead of re-evaluating every constraint -- and when even the largest
                // |r| is below EPSILON (every converged solve) no constraint can be unsatisfied (lib.rs:358-370)
                // (fmax drops NaN residuals from `largest`, but is_satisfied's `abs() < EPSILON` is false for them: a NaN
                // residual makes the sum of squares NaN, and then every constraint takes the per-row test below)
                if (largest < EPS && !isnan(residual_sq)) {
                    all_satisfied = true;
                    if (a.unsat_mask && grid_wg == 0)
                        for (uint32_t i = tlane; i < a.p.n_cons; i += tsize) a.unsat_mask[sys * a.p.n_cons + i] = 0;
                    break;
                }
                for (uint32_t ci = con0 + tm.lane; ci < con1; ci += tm.stride) {
                    const CRef cref(P, ci, true);
                    const DevCon& c = cref.get();
                    const uint32_t row0 = c.row0;
                    bool sat = fabs(ws[o_r + row0]) < EPS;
                    if (c.nrows > 1) sat = sat && (fabs(ws[o_r + row0 + 1]) < EPS);
                    if (!sat) unsat_cnt += 1.0;
                    if (a.unsat_mask) a.unsat_mask[sys * a.p.n_cons + cref.pos(P, ci)] = sat ? 0 : 1;
                }
                break;
            }
            // ---- the one residual sweep: EVAL0 -> r, STEP -> r_next, FINAL -> unsatisfied test on unweighted values ---------------------
            // r[row] = weight * residual (solver.rs:327-355); one lane per constraint of the partition.
            const uint32_t o_dst = (mode == EVAL0) ? o_r : o_rn;
            double sq = 0.0;
            double mx = __builtin_nan("");
            auto residual_of = [&](const CRef& cref, uint32_t ci) {
                const DevCon& c = cref.get();
                double r0, r1;
                const bool deg = con_residual<LIN>(c, ws + o_x, r0, r1);
                if (mode == FINAL) {  // unsatisfied check on the unweighted residuals (lib.rs:305-327, :358-370)
                    bool sat = fabs(r0) < EPS;
                    if (c.nrows > 1) sat = sat && (fabs(r1) < EPS);
                    if (!sat) unsat_cnt += 1.0;
                    if (a.unsat_mask) a.unsat_mask[sys * a.p.n_cons + cref.pos(P, ci)] = sat ? 0 : 1;
                    return;
                }
                const double wgt = c.weight;
                const uint32_t row0 = c.row0;
                const double w0 = wgt * r0;
                ws[o_dst + row0] = w0;
                sq += w0 * w0;
                mx = fmax(mx, fabs(w0));
                if (c.nrows > 1) {
                    const double w1 = wgt * r1;
                    ws[o_dst + row0 + 1] = w1;
                    sq += w1 * w1;
                    mx = fmax(mx, fabs(w1));
                }
                if (deg && !(resuming && mode == EVAL0)) {  // Warning::Degenerate, every evaluation (solver.rs:340-346)
                    int idx = atomicAdd(nwarn, 1);
                    if (a.warn_log && (uint32_t)idx < a.warn_cap)
                        a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)pass << 32) | cref.pos(P, ci);
                }
            };
            // ---- the one Jacobian sweep (eval() and accepted steps, newton.rs:121; solver.rs:359-440) ------------------
            auto jacobian_of = [&](const CRef& cref, uint32_t ci) {
                const DevCon& c = cref.get();
                JacWriter<WsRef<WS_STRIDE>> w;
                w.jv = ws + o_j;
                if constexpr (RECF == 1) {
                    if (rec_jglobal) w.jv.p = jglob;
                }
                w.jbase = c.jbase;
                const uint4 loc = cref.jloc(P);
                w.loc[0] = loc.x;
                w.loc[1] = loc.y;
                w.loc[2] = loc.z;
                w.loc[3] = loc.w;
                w.weight = c.weight;
                const bool deg = con_jacobian<LIN>(c, ws + o_x, w);
                // (a resumed system's eval(): this sweep is the refresh its last accepted step still owed -- logged under that
                // step's pass number -- or one that was logged before the hand-over)
                const uint32_t log_pass = (resuming && mode == EVAL0) ? a.resume[q].jac_pass : pass;
                if (deg && log_pass != kNoPass) {
                    int idx = atomicAdd(nwarn, 1);
                    if (a.warn_log && (uint32_t)idx < a.warn_cap)
                        a.warn_log[sys * a.warn_cap + idx] = ((uint64_t)log_pass << 32) | cref.pos(P, ci);
                }
            };
            // A sweep over this unit's constraints.  (Fetching the next constraint's record before evaluating this one was
            // measured on the record-walk builds: one solve of 300 variables 192 -> 191 us, batches of 150 variables +13 %, of
            // 300 variables -16 %: the copies cost issue slots where four workgroups share a CU.  Not kept.)
            auto sweep = [&](auto&& f) __attribute__((always_inline)) {
                for (uint32_t ci = con0 + tm.lane; ci < con1; ci += tm.stride) {
                    const CRef cref(P, ci, unit_w);
                    f(cref, ci);
                }
            };
            sweep([&](const CRef& cref, uint32_t ci) {
                residual_of(cref, ci);
                // linear-only build: its one Jacobian sweep per solve (see below) shares eval()'s pass over the records
                if constexpr (LIN) {
                    if (mode == EVAL0) jacobian_of(cref, ci);
                }
            });
            ++pass;
            tm.phase_sync();
            EZPZ_STAMP(20);
            if (mode == FINAL) break;
            tm.reduce2(sq, mx, OpSum(), OpMax());
            EZPZ_STAMP(21);  // sum r^2 (newton.rs:116,:235) and max |r| (newton.rs:50-53)
            const bool accept = (mode == EVAL0) || (sq < residual_sq);  // strict, newton.rs:118
            if (accept) {
                if (mode == STEP) {
                    if (REC && a.rec_asm_kc != 0) {  // (the packed pairs hold r's addresses: the accepted residuals move there)
                        for (uint32_t i = tlane; i < m; i += tsize) ws[o_r + i] = ws[o_rn + i];
                    } else {
                        uint32_t t = o_r;
                        o_r = o_rn;
                        o_rn = t;
                    }
                    lambda *= LM_LAMBDA_DECR;
                }
                // The nine linear kinds have constant partials (weight x +-1 or +-0.5) and no degenerate guard: after
                // eval() the refresh of an accepted step (newton.rs:121) would store the same bits again, so the
                // linear-only build sweeps the Jacobian once per solve -- inside eval()'s residual sweep above.
                if constexpr (!LIN) sweep(jacobian_of);
                ++pass;
                residual_sq = sq;
                largest = mx;
                r_is_at_x = true;
            } else {  // reject: revert, raise lambda (newton.rs:124-131)
                r_is_at_x = false;  // x is now (x + d) - d, which may differ from the x of r in the last bit
                for (uint32_t ci = call0 + tm.lane; ci < call1; ci += tm.stride) {
                    const uint32_t v = ci;  // internal variable numbering = schedule order
                    ws[o_x + v] = ws[o_x + v] - ws[o_v + v];
                }
                lambda *= LM_LAMBDA_INCR;
            }
            tm.phase_sync();
            EZPZ_STAMP(22);
            if (mode == STEP) {
                if (step_inf_norm <= a.step_tolerance) {  // newton.rs:134-139
                    iterations = it;
                    converged = 1;
                    mode = FINAL;
                    continue;
                }
                ++it;
            }
            if (mode == EVAL0) {
                mode = STEP;
                if (resuming) {  // the loop goes on where the lanes kernel left it
                    lambda = a.resume[q].lambda;
                    it = a.resume[q].it;
                    pass = a.resume[q].pass;
                }
            }
        }

        // ---- write-back -----------------------------------------------------------------------------------------------
        double dummy = 0.0;
        EZPZ_STAMP(30);
        // the count of unsatisfied constraints; for workgroup teams also the rendezvous before the cooperative store
        // of x.  Wavefront-partitioned teams store their own partition's values and need neither when the count is
        // known to be zero (`all_satisfied` is uniform: it derives from the reduced max |r|).
        // Grid teams: the reduction's grid barrier is also what lets workgroup 0 read the shared warning counter, so it
        // is skipped only when no warning can exist (linear-only build).
        constexpr bool OWN_STORE = (MODE == MODE_PART);
        const bool skip_count = all_satisfied && (OWN_STORE || MODE == MODE_SUB) && (!grid_team || LIN);
        if (!skip_count) tm.reduce2(unsat_cnt, dummy, OpSum(), OpSum());
        EZPZ_STAMP(31);
        double* xo = a.x_out + sys * n_row;
        if constexpr (OWN_STORE) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t ci = call0 + tm.lane + j * 64;
                if (ci < call1) xo[x_id[j]] = ws[o_x + ci];
            }
            for (uint32_t ci = call0 + tm.lane + 4 * 64; ci < call1; ci += tm.stride) xo[P.var_of[ci]] = ws[o_x + ci];
        } else {
            for (uint32_t i = tlane; i < n; i += tsize) xo[P.var_of[i]] = ws[o_x + i];
        }
        if (tlane == 0 && grid_wg == 0) {
            EzpzStatus st;
            st.iterations = iterations;
            st.converged = converged;
            st.n_unsatisfied = (uint32_t)unsat_cnt;
            st.n_warnings = grid_team ? (uint32_t)__hip_atomic_load(nwarn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : (uint32_t)*nwarn;
            st.final_residual_inf = (m > 0) ? largest : 0.0;
            st.final_lambda = lambda;
            if (grid_team && __hip_atomic_load(&tm.grid->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                st.iterations = EZPZ_ITERATIONS_TEAM_TIMEOUT;
                st.converged = 0;
            }
            a.status[sys] = st;
            // grid teams: this counter serves the slot's system after next; every workgroup passes a grid reduction
            // of the next system (which this thread joins only after the store) before it can get there
            if (grid_team) __hip_atomic_store(nwarn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the workspace is reused by the next system of this team; wavefront-partitioned teams touch only their own
        // partition's part of it until the rendezvous that follows the next load
        if constexpr (MODE != MODE_PART) tm.team_sync();
        EZPZ_STAMP(32);
    }
    publish_done(done);
    } while (resident_next(done, born, resident_word));
}

// Evaluation-only kernel (K1): one workgroup per value vector, everything in global memory.
struct EvalArgs {
    ProgramView p;
    const double* x;
    double* r_out;
    double* jv_out;
    uint32_t* deg_out;
    uint64_t batch;
};

// (static: launch.hip and freedom.hip each take their own copy)
static __global__ void __launch_bounds__(256) eval_kernel(const EvalArgs e) {
    using namespace dev;
    __shared__ int nwarn;
    const Prog<uint32_t> P = make_prog<uint32_t>(e.p, e.p.base, e.p.base);
    for (uint64_t sys = blockIdx.x; sys < e.batch; sys += gridDim.x) {
        if (threadIdx.x == 0) nwarn = 0;
        __syncthreads();
        const double* xs = e.x + sys * e.p.n_vars;
        double* r = e.r_out + sys * e.p.n_rows;
        double* jv = e.jv_out + sys * e.p.zj;
        for (uint32_t ci = threadIdx.x; ci < e.p.n_cons; ci += blockDim.x) {
            DevCon c;
            if (e.p.packed) {
                c = load_packed(P.pcons, P.con_weight, ci, false);
                *reinterpret_cast<uint4*>(c.jloc) = P.patterns[c.nslots];
            } else {
                c = load_con(P.cons + ci);
            }
            double r0, r1;
            if (con_residual(c, xs, r0, r1)) atomicAdd(&nwarn, 1);
            if (e.r_out) {
                r[c.row0] = c.weight * r0;
                if (c.nrows > 1) r[c.row0 + 1] = c.weight * r1;
            }
            JacWriter<double*> w;
            w.jv = jv;
            w.jbase = c.jbase;
            const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
            w.loc[0] = loc[0];
            w.loc[1] = loc[1];
            w.loc[2] = loc[2];
            w.loc[3] = loc[3];
            w.weight = c.weight;
            if (con_jacobian(c, xs, w)) atomicAdd(&nwarn, 1);
        }
        __syncthreads();
        if (threadIdx.x == 0 && e.deg_out) e.deg_out[sys] = (uint32_t)nwarn;
        __syncthreads();
    }
}

}  // namespace ezpz
