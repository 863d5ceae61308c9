// Launch of the component-resident LM kernel (comp_kernel.hip.hpp) -- its own translation unit so that the two kernel
// builds compile beside the list-walk kernels of launch.hip.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <algorithm>
#include <atomic>

#include "batch_kernel.hip.hpp"
#include "comp_kernel.hip.hpp"

namespace ezpz {

namespace {

template <bool LIN>
int launch_build(const CompPlan& plan, const CompArgs& args, int device, int cus, size_t lds_limit, hipStream_t stream) {
    auto kernel = comp_solve_kernel<LIN>;
    static std::atomic<bool> raised[16];  // per kernel build and device (see launch_kernel in launch.hip)
    if (plan.lds_bytes > 48 * 1024 && !raised[device & 15].load(std::memory_order_acquire)) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_limit) != hipSuccess) {
            (void)hipGetLastError();
            return EZPZ_ERR_HIP;
        }
        raised[device & 15].store(true, std::memory_order_release);
    }
    const uint32_t threads = plan.n_waves * 64;
    // workgroups: eight times what the device holds at once (LDS and the 2048 lanes of a CU), each walks the batch with the grid's
    // stride -- the dispatcher hands a free place to the next one, whatever the systems before it took: 2000 x 2000 on the
    // interpreter at x1 / x2 / x8: 18.7 / 19.2 / 20.2 M solves/s
    const uint64_t per_cu = std::max<uint64_t>(1, std::min<uint64_t>(lds_limit / std::max<uint32_t>(plan.lds_bytes, 1), 2048 / threads));  // (a CU holds 2048 lanes)
    const uint32_t grid = (uint32_t)std::min<uint64_t>(args.batch, (uint64_t)cus * per_cu * 8u);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), plan.lds_bytes, stream, args);
    if (hipGetLastError() != hipSuccess) return EZPZ_ERR_HIP;
    return EZPZ_OK;
}

}  // namespace

int comp_launch(const CompPlan& plan, const uint32_t* dev_blob, const CompLaunch& L, int device, int cus, size_t lds_limit,
                void* stream) {
    if (L.batch == 0) return EZPZ_OK;
    CompArgs a{};
    a.prog = dev_blob;
    a.o_waves = plan.o_waves;
    a.o_chunks = plan.o_chunks;
    a.n_row = plan.n_vars;
    a.n_cons = plan.n_cons;
    a.n_rows_total = plan.n_rows;
    a.x0 = L.x0;
    a.x_out = L.x_out;
    a.status = L.status;
    a.unsat_mask = L.unsat_mask;
    a.warn_log = L.warn_cap ? L.warn_log : nullptr;
    a.warn_cap = L.warn_cap;
    a.batch = L.batch;
    a.max_iterations = L.max_iterations;
    a.unit_weights = plan.unit_weights ? 1u : 0u;
    a.residual_tolerance = L.residual_tolerance;
    a.step_tolerance = L.step_tolerance;
    a.initial_lambda = L.initial_lambda;
    a.done = L.done;
    a.scratch_row0 = plan.rows_persistent;
    a.scratch_rows = plan.scratch_rows;
    a.red_row0 = plan.rows_persistent + plan.n_waves * plan.scratch_rows;
    return plan.linear ? launch_build<true>(plan, a, device, cus, lds_limit, static_cast<hipStream_t>(stream))
                       : launch_build<false>(plan, a, device, cus, lds_limit, static_cast<hipStream_t>(stream));
}

uint64_t batch_launch_waves(int cus) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, batch_lane_kernel, 256, 0) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    return (uint64_t)cus * (uint64_t)per_cu * 4;
}

// EZPZ_LANES_STRAGGLERS: the working lanes (of 64) at which a wavefront hands its systems to the teams, 0 = never.  262 144 jittered
// 300-variable sketches, the list sized for every wavefront handing over that many (launch.hip): 16 / 24 / 28 / 32 / 36 / 40 / 48 / 64
// lanes 7.85 / 8.13 / 8.34 / 8.38 / 7.85 / 8.18 / 8.04 / 3.67 M solves/s (the teams walk records: they resume a system faster than a
// thinning wavefront finishes it, until they are handed most of the batch); a list that overflows leaves the lanes their tail: 5.4.
uint32_t batch_straggler_lanes() {
    static const int env_strag = [] { const char* e = std::getenv("EZPZ_LANES_STRAGGLERS"); return e ? std::atoi(e) : 32; }();
    return (uint32_t)std::min(std::max(env_strag, 0), 64);
}

int batch_launch(const BatchPlan& plan, const uint32_t* dev_blob, double* dev_ws, uint64_t ws_waves, uint32_t n_cons, const CompLaunch& L,
                 void* stream, uint32_t* strag_list, uint32_t* strag_count, uint32_t strag_cap, LmResume* strag_state) {
    if (L.batch == 0) return EZPZ_OK;
    BatchArgs a{};
    a.prog = dev_blob;
    a.nv = plan.nv, a.m = plan.m, a.zj = plan.zj, a.zlo = plan.zlo, a.ncons = plan.ncons;
    a.n_ops = plan.n_ops, a.ops_off = plan.ops_off, a.cons_off = plan.cons_off, a.var_off = plan.var_off, a.inv_off = plan.inv_off;
    a.o_d = plan.o_d, a.o_r = plan.o_r, a.o_rn = plan.o_rn, a.o_j = plan.o_j, a.o_dg = plan.o_dg, a.o_l = plan.o_l, a.rows = plan.rows;
    a.n_cons = n_cons;
    a.x0 = L.x0;
    a.x_out = L.x_out;
    a.status = L.status;
    a.unsat_mask = L.unsat_mask;
    a.warn_log = L.warn_cap ? L.warn_log : nullptr;
    a.warn_cap = L.warn_cap;
    a.max_iterations = L.max_iterations;
    a.unit_weights = plan.unit_weights ? 1u : 0u;
    // (4 ... 32 measured on 262 144 jittered systems of 300 variables, 4-19 iterations each: no difference -- at one system
    // per lane a wavefront lasts as long as its slowest lane)
    static const int env_refill = [] { const char* e = std::getenv("EZPZ_LANES_REFILL"); return e ? std::atoi(e) : 0; }();
    a.refill_lanes = env_refill > 0 ? (uint32_t)env_refill : 8;  // (524 288 jittered systems of 300 variables: 22 -> 8 lanes +9 %)
    a.batch = L.batch;
    a.residual_tolerance = L.residual_tolerance;
    a.step_tolerance = L.step_tolerance;
    a.initial_lambda = L.initial_lambda;
    a.ws = dev_ws;
    // a wavefront down to half of its lanes working with nothing left to take gives them up: continuing costs the whole
    // wavefront a round per iteration, while the teams resume a system where it stands (batch_straggler_lanes)
    const int env_strag = (int)batch_straggler_lanes();
    a.strag_list = env_strag > 0 && strag_cap && strag_state ? strag_list : nullptr;
    a.strag_state = strag_state;
    a.strag_count = strag_count;
    a.strag_cap = strag_cap;
    a.strag_lanes = (uint32_t)std::max(env_strag, 0);
    // persistent lanes: as many wavefronts as have a workspace (and systems to solve)
    static const int env_waves = [] { const char* e = std::getenv("EZPZ_LANES_WAVES"); return e ? std::atoi(e) : 0; }();
    uint64_t blocks = std::min<uint64_t>((L.batch + 255) / 256, ws_waves / 4);
    if (env_waves > 0) blocks = std::min<uint64_t>(blocks, (uint64_t)env_waves / 4);
    if (blocks == 0) return EZPZ_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(batch_lane_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    if (hipGetLastError() != hipSuccess) return EZPZ_ERR_HIP;
    return EZPZ_OK;
}

}  // namespace ezpz
