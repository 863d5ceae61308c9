// Plain-data records shared by the host symbolic phase and the device kernels.  No standard-library dependency: this
// header is also compiled at run time (hiprtc, see jit.cpp) as part of the class-specialised kernels.
#pragma once
#ifdef __HIPCC_RTC__
#include "ezpz_amd.h"
#else
#include <cstdint>

#include "../../include/ezpz_amd.h"
#endif

namespace ezpz {

// One constraint as the kernel sees it.  80 bytes.
struct alignas(16) DevCon {
    uint32_t ids[8];
    double param;
    double weight;
    uint32_t row0;   // first residual row (rows are numbered in request order, solver.rs:226-253)
    uint32_t jbase;  // first Jacobian slot owned by this constraint
    uint32_t pos;    // position in the caller's constraint list (for unsat mask / warnings)
    uint8_t kind, tag, nrows, nslots;
    uint8_t jloc[16];  // per emitted partial: slot offset from jbase; bit 7 = accumulate into an earlier entry's slot
};
static_assert(sizeof(DevCon) == 80, "DevCon layout");

// The same constraint in 32 bytes, for programs whose every count fits 16 bits and whose constraint table is read
// from global memory / L2 by workgroup teams: the table is re-read by every residual and Jacobian sweep of every
// system, and its bytes (not HBM's) are what the massive_parallel_system launch is bound by.  weight, pos and the
// jloc pattern live in side arrays (weights are read only when some weight != 1, pos only when a mask or a warning
// is written, the handful of distinct jloc patterns sit in LDS).
struct alignas(16) PackedCon {
    uint16_t ids[8];
    double param;
    uint16_t row0, jbase;
    uint8_t kind, tag, nrows, pattern;
};
static_assert(sizeof(PackedCon) == 32, "PackedCon layout");

// A partition of the system: a union of connected components that one wavefront can own end to end
// (its constraints, variables, Jacobian slots and Cholesky columns are disjoint from every other partition's).
struct PartDesc {
    uint32_t con0, con1;  // constraint range (table is sorted by partition, then kind)
    uint32_t lvl0, nlev;  // this partition's slice of lvl_cptr / lvl_sptr (nlev + 1 entries each)
};

// The LM state with which a system changes kernels in mid-solve: a straggler of a lanes-across-the-batch launch
// (batch_kernel.hip.hpp) goes on in the per-system teams (lm_kernel.hip.hpp) from its current values.
struct LmResume {
    double lambda;
    uint32_t it, pass, nwarn;
    uint32_t jac_pass;  // pass number under which the resuming eval()'s Jacobian sweep logs (the refresh of the last accepted
                        // step was still owed), or kNoPass: that refresh was logged before the hand-over
};
constexpr uint32_t kNoPass = 0xFFFFFFFFu;

// The completion word of a one-call launch (pipeline.cpp: system_solve_one; ezpz_solve is one launch per tier).  After its
// last store the launch writes `seq` to `flag`, a word of host memory mapped into the device that the calling thread
// polls: the call returns ~7 us sooner than through the runtime's completion signal (tools/launch_floor.hip:
// 6.0 us launch-to-flag against 13.4 us launch-to-hipStreamQuery).  `counter` (device memory, zero between launches)
// counts the workgroups of a launch that has several.  flag == null: an ordinary launch.
// RESIDENT launches (one workgroup; pipeline.cpp: system_solve_one): after publishing, the kernel does not end but waits for the
// host's next request on the same buffers -- `request` is a word of device memory the host stores into through the PCIe
// BAR: the tag of the latest request (the launch's generation in the upper 24 bits, a sequence number below; any other
// generation, e.g. ~0, means "leave") -- for at most `lease_ticks` of the 100 MHz clock, and for `life_ticks` in all;
// when it ends it stores `generation` to `gone` (mapped host memory).  `seq` is the tag of the request being served.  A solve() call that finds its
// topology's kernel resident costs a store, the solve and a poll: ~3 us of round trip instead of the ~8 us of a launch
// (tools/launch_floor.hip).  request == null: an ordinary launch.
struct DoneWord {
    unsigned long long* flag;
    unsigned long long seq;
    unsigned int* counter;
    const unsigned long long* request;
    unsigned long long* gone;
    unsigned long long generation;
    unsigned int lease_ticks, life_ticks;
};

}  // namespace ezpz
