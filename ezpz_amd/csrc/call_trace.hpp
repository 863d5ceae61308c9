// Per-stage time stamps of one ezpz_solve call (diagnostic; tools/solve_call_breakdown.py).  The calling thread hands the
// library a buffer with ezpz_debug_call_trace; while it is set, every stage boundary of the one-call path appends an
// (id, CLOCK_MONOTONIC nanoseconds) pair.  Without a buffer a stamp is one thread-local load and a branch.
#pragma once
#include <time.h>

#include <cstddef>
#include <cstdint>

namespace ezpz {

enum CallStage : uint32_t {
    CALL_ENTER = 1,        // ezpz_solve entered
    CALL_PLAN = 2,         // request recognised (compare with the cached plan / scans + symbolic phase on a miss)
    CALL_SIDES = 3,        // sides inferred from the guesses, tier chosen
    CALL_LOCKED = 4,       // system lock + device selected
    CALL_STAGED = 5,       // guesses where the kernel reads them
    CALL_LAUNCHED = 6,     // kernel enqueued
    CALL_COMPLETE = 7,     // completion seen by the host
    CALL_UNPACKED = 8,     // values / status copied out of the staging buffers
    CALL_FINISHED = 9,     // unsatisfied list, warnings, outcome filled in
    CALL_RETURN = 10,
    // a request the process has not seen (or ezpz_cache_clear): the symbolic phase, between CALL_ENTER and CALL_SIDES
    COLD_PLAN_BUILT = 20,     // tiers, lint, validation (build_plan)
    COLD_ANALYSED = 21,       // analyze_into: components / classes / elimination order / lists (Model::new's counterpart)
    COLD_UPLOADED = 22,       // device allocations + program upload
    COLD_KERNEL_FOUND = 23,   // the specialised kernel's registry entry (source text compared)
};

struct CallTrace {
    uint64_t* buf = nullptr;
    size_t cap = 0, n = 0;
};
extern thread_local CallTrace t_call_trace;

inline void call_stamp(uint32_t id) {
    CallTrace& t = t_call_trace;
    if (__builtin_expect(t.buf != nullptr, 0) && t.n + 2 <= t.cap) {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        t.buf[t.n++] = id;
        t.buf[t.n++] = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
    }
}

}  // namespace ezpz
