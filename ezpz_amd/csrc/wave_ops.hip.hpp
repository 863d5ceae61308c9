// Wavefront-level building blocks shared by the LM kernels (lm_kernel.hip.hpp, comp_kernel.hip.hpp): the LM damping
// constants, the two reduction operators of the LM control (sum of squares, NaN-ignoring maximum) and cross-lane
// reductions on DPP moves (no LDS crossbar traffic) for gfx950's 64-wide wavefronts.
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif
#include "dev_types.hpp"

namespace ezpz {
namespace dev {

// The last thing a kernel does when its launch carries a completion word (DoneWord, dev_types.hpp): every thread of the
// workgroup waits until its stores are acknowledged (the workgroup barrier does not wait for them, and the release below only
// waits for its own wavefront's); one thread per workgroup counts it in, and the last workgroup's release store at system scope
// publishes everything the launch wrote before the word itself.
__device__ __forceinline__ void publish_done(const DoneWord& w) {
    if (!w.flag) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        bool last = true;
        if (gridDim.x > 1) {
            last = __hip_atomic_fetch_add(w.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
            if (last) __hip_atomic_store(w.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (last) __hip_atomic_store(w.flag, w.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

constexpr double LM_LAMBDA_INCR = 10.0;  // newton.rs:15
constexpr double LM_LAMBDA_DECR = 0.1;   // newton.rs:16

struct OpSum {
    __device__ __forceinline__ double operator()(double a, double b) const { return a + b; }
    __device__ __forceinline__ double identity() const { return 0.0; }
};
// fmax as ONE instruction.  The compiler's fmax is v_max_f64 behind a canonicalisation (v_max_f64 x, x, x) of every
// operand it cannot prove free of signalling NaNs -- anything that came through a select, a DPP move or a load -- i.e.
// three instructions per maximum in the reductions and accumulators of the LM control.  v_max_f64 itself ignores a
// quiet NaN operand like libm's fmax (newton.rs:53,:108), and nothing here is a signalling NaN: every operand is the
// result of arithmetic or the quiet NaN that starts a maximum.  fmax_abs is fmax(a, |b|).
__device__ __forceinline__ double fmax_nc(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double fmax_abs(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
struct OpMax {  // libm::fmax (NaN-ignoring), newton.rs:53,:108
    __device__ __forceinline__ double operator()(double a, double b) const { return fmax_nc(a, b); }
    __device__ __forceinline__ double identity() const { return __builtin_nan(""); }  // dropped by fmax
};

// A resident launch (DoneWord::request): every thread of the (one) workgroup calls this after publish_done; true = another
// request is to be served (w.seq is its sequence number), false = the kernel ends (no request within the lease, the host's
// "leave", or the launch's lifetime is over) -- after it has said so to the host.  `born` = wall_clock64() at kernel start.
__device__ __forceinline__ bool resident_next(DoneWord& w, unsigned long long born, unsigned long long* shared_word) {
    if (!w.request) return false;
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        unsigned long long next = 0;
        for (;;) {
            const unsigned long long v = __hip_atomic_load(w.request, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v != w.seq) {
                // a request carries the generation of the launch it is meant for in its upper 24 bits: anything else in
                // the word -- "leave" (~0), or the request of the launch that replaces this one -- ends this kernel
                next = (v >> 40) == (w.seq >> 40) ? v : 0;
                break;
            }
            const unsigned long long now = wall_clock64();
            if (now - t0 > w.lease_ticks || now - born > w.life_ticks) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (next == 0) __hip_atomic_store(w.gone, w.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        *shared_word = next;
    }
    __syncthreads();
    const unsigned long long next = *shared_word;
    __syncthreads();
    if (next == 0) return false;
    w.seq = next;
    return true;
}

// Cross-lane moves inside a row of 16 lanes without touching the LDS crossbar (DPP modifiers on v_mov):
// quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror.  After the four steps every lane of a row holds
// the row's reduction (the two mirror steps work because all lanes of a quad / half-row already agree).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    // bound_ctrl: every lane of these permutations has a valid source, and with it the compiler need not
    // materialise the `old` operand (one v_mov less per 32-bit half)
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// Lane l with lane l ^ 16 (HALVES false) or l ^ 32 (true) on gfx950's v_permlane16_swap / v_permlane32_swap: `a` and `b`
// are the pair's two values in every lane of the pair (lower one in `a`), two VALU moves per 32-bit half instead of a
// ds_bpermute round trip through the LDS crossbar.
template <bool HALVES>
__device__ __forceinline__ void lane_pairs(double v, double& a, double& b) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned ulo = (unsigned)u, uhi = (unsigned)(u >> 32);
    if constexpr (HALVES) {
        const auto lo = __builtin_amdgcn_permlane32_swap(ulo, ulo, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(uhi, uhi, false, false);
        a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
        b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap(ulo, ulo, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(uhi, uhi, false, false);
        a = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
        b = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
    }
}
template <int WIDTH, class Op>
__device__ __forceinline__ double reduce_lanes(double v, Op op) {
    if constexpr (WIDTH >= 2) v = op(v, dpp_move<0xB1>(v));   // quad_perm [1,0,3,2]
    if constexpr (WIDTH >= 4) v = op(v, dpp_move<0x4E>(v));   // quad_perm [2,3,0,1]
    if constexpr (WIDTH >= 8) v = op(v, dpp_move<0x141>(v));  // row_half_mirror
    if constexpr (WIDTH >= 16) v = op(v, dpp_move<0x140>(v)); // row_mirror
    if constexpr (WIDTH >= 32) {
        double a, b;
        lane_pairs<false>(v, a, b);
        v = op(a, b);
    }
    if constexpr (WIDTH >= 64) {
        double a, b;
        lane_pairs<true>(v, a, b);
        v = op(a, b);
    }
    return v;
}

// Reduction of a whole wavefront into its LAST lane without the LDS crossbar: rows of 16 as above, then the two
// wave-level DPP broadcasts of GFX9 (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3).  Lanes the
// row mask leaves out keep the operator's identity, so one unconditional op() per step serves every lane.  Same
// pairing as the xor shuffles ((r0 + r1) + (r2 + r3)), so the same bits.
// (Only the last lane's value is meaningful afterwards: the broadcasts go to every row that has a source -- row 1 takes
// r0, row 2 r1, row 3 r2, then rows 2 and 3 take lane 31 = r1 + r0 -- instead of to rows 1 / 3 and 2 / 3 only with the
// operator's identity materialised for the others; rows without a source read zero, and their lanes are not read.)
template <int CTRL>
__device__ __forceinline__ double dpp_move_rows(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <class Op>
__device__ __forceinline__ double reduce_wave_to_last_lane(double v, Op op) {
    v = reduce_lanes<16>(v, op);
    v = op(v, dpp_move_rows<0x142>(v));  // row_bcast:15: lane 63 = r3 + r2, lane 31 = r1 + r0
    v = op(v, dpp_move_rows<0x143>(v));  // row_bcast:31: lane 63 = (r3 + r2) + (r1 + r0)
    return v;
}

}  // namespace dev
}  // namespace ezpz
