// Device helpers shared by the kernels that read constraint records from global memory and by those whose workgroups
// exchange values inside a launch (lm_kernel.hip.hpp: grid teams; front_kernel.hip.hpp: fronts across workgroups).
#pragma once
#include <hip/hip_runtime.h>

#include "dev_types.hpp"
#include "launch_types.hpp"

namespace ezpz {

// One wide, fully parallel load of a constraint record (5 x 16 B in flight) instead of field-by-field
// dependent loads: the sweeps are latency bound on exactly this.
__device__ __forceinline__ DevCon load_con(const DevCon* p) {
    union {
        DevCon c;
        uint4 q[5];
    } u;
    const uint4* src = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 5; ++i) u.q[i] = src[i];
    return u.c;
}

// One large system on many workgroups ("grid team", MODE_PART, each workgroup's share of the state in its LDS):
// workgroup g of the G that share a system owns partitions [g*W, (g+1)*W) (W wavefronts), and the two scalar
// reductions of an LM iteration go through this per-system scratch.  Every workgroup publishes its two partials;
// workgroup 0 alone polls them, folds them in a fixed tree and writes the result to one 64-byte line per workgroup;
// every other workgroup polls only its own line.  A value travels as a 16-byte (value, sequence number) chunk moved
// by one device-coherent (sc0 sc1) 128-bit access, so it validates itself and no release/acquire fence (an L2
// write-back / L1 invalidate each) is needed.  Measured alternatives on 256 workgroups, all 25 us per reduction: a
// central atomic counter + generation word (256 cross-XCD atomics serialise on one word), everyone polling
// everyone's flag (all pollers hit the same few lines), release/acquire flags (10 us just to scatter 255 lines).
// All G workgroups must be resident at once: see launch_grid_kernel in launch.hip.
__device__ __forceinline__ void grid_store(gridchunk_t* p, double v, unsigned int seq) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    gridchunk_t c;
    c.x = (unsigned int)u;
    c.y = (unsigned int)(u >> 32);
    c.z = seq;
    c.w = 0;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory");
}
// Spins until the chunk carries `seq`, returns its value.  The spin is bounded (~1 s): if a workgroup of the team is
// not resident (a device with fewer free CUs than the launch was sized for, e.g. another process's grid teams on the
// same device) the waiters give up, flag the slot dead and return NaN instead of hanging the GPU; the solve's status
// then carries EZPZ_ITERATIONS_TEAM_TIMEOUT.
constexpr unsigned int kGridSpinLimit = 1u << 21;
__device__ __forceinline__ double grid_wait(const gridchunk_t* p, unsigned int seq, int* dead) {
    gridchunk_t c;
    for (unsigned int spins = 0;; ++spins) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(p) : "memory");
        if (c.z == seq) break;
        if ((spins & 1023u) == 1023u &&
            (spins >= kGridSpinLimit || __hip_atomic_load(dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            __hip_atomic_store(dead, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return __builtin_nan("");
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return __builtin_bit_cast(double, ((unsigned long long)c.y << 32) | c.x);
}

}  // namespace ezpz
