// The device half of ONE solve() call (reference protocol: ezpz-cli/src/main.rs:86-100, ezpz/benches/solver_bench.rs:15-24
// time exactly one `solve` per iteration).  Implemented in pipeline.cpp, used by solve.cpp.
#pragma once
#include <cstdint>

#include "../../include/ezpz_amd.h"

namespace ezpz {

// One system, host pointers: Model::solve_levenberg_marquardt (newton.rs:29-145) + the unsatisfied check (lib.rs:305-327)
// as ONE launch whose completion the calling thread sees in a word of mapped host memory (DoneWord, dev_types.hpp).
//   * the guesses go straight into device memory through the PCIe BAR when the device has a large BAR (host stores,
//     no DMA descriptor, no read across the link by the kernel), else into mapped host memory;
//   * values and status come back through mapped host memory, written by the kernel;
//   * unsat_mask [n_cs] and warn_log [warn_cap] stay on the device and are fetched only when the status says some
//     constraint is unsatisfied / some evaluation warned (they are left untouched otherwise).
// No allocation after the calling thread's first call on a device (grow-only per-thread buffers).
int system_solve_one(EzpzSystem* sys, const double* x0, const EzpzConfig* cfg, double* x_out, EzpzStatus* status,
                     uint8_t* unsat_mask, uint64_t* warn_log, uint32_t warn_cap);

// The calling thread's one-call kernel may still be on the device, waiting for the thread's next request (a resident
// launch, dev_types.hpp: DoneWord): whatever else the thread is about to enqueue there would queue behind it until its
// lease (EZPZ_RESIDENT_US) runs out -- it is told to leave first.  Cheap when there is none.
void release_thread_kernel(int device);

}  // namespace ezpz
