// Launch of the FRONTAL shape (front_kernel.hip.hpp; plan: fronts.cpp): one workgroup per system, or G workgroups per system
// that must all be resident (they wait for each other's chunks), as many systems in flight as the device holds.
#include "system.hpp"

#include "front_kernel.hip.hpp"

using namespace ezpz;

namespace ezpz {

extern std::mutex g_grid_mu;          // launch.hip: launches whose workgroups wait for each other are chained per device
extern hipEvent_t g_grid_event[16];

// What a probe launch adds to a solve launch (FrontArgs::probe_*).
struct FrontProbe {
    uint32_t m = 0;
    double* out = nullptr;
    const double* in = nullptr;
    double scale = 1e-11;
};

template <bool LIN>
static int front_launch_kernel(EzpzSystem& s, FrontArgs& fa, hipStream_t stream) {
    const FrontPlan& plan = *s.fronts;
    auto kernel = front_solve_kernel<LIN>;
    const uint32_t G = plan.n_wgs;
    if (s.front_capacity == 0) {  // once per system: these runtime calls cost more than a small solve
        if (plan.lds_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lim.lds_bytes));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)plan.threads, plan.lds_bytes));
        s.front_capacity = (uint64_t)s.lim.cus * (uint64_t)std::max(per_cu, 1);
    }
    if (G == 1) {
        // persistent workgroups: a few per CU's worth of the batch
        const uint32_t grid = (uint32_t)std::min<uint64_t>(fa.batch, s.front_capacity * 2);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(plan.threads), plan.lds_bytes, stream, fa);
        HIP_TRY(hipGetLastError());
        return EZPZ_OK;
    }
    if (s.front_capacity < G) return EZPZ_ERR_TOO_LARGE;
    const uint32_t slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(fa.batch, s.front_capacity / G));
    const uint32_t stride = front_scratch_bytes(plan.n_chunks);
    if (s.front_scratch.cap < (size_t)slots * stride) {
        int rc = s.front_scratch.ensure((size_t)slots * stride);
        if (rc != EZPZ_OK) return rc;
        HIP_TRY(hipMemsetAsync(s.front_scratch.p, 0, s.front_scratch.cap, stream));
    }
    fa.scratch = s.front_scratch.p;
    fa.scratch_stride = stride;
    fa.done.request = nullptr;
    // every workgroup of the launch must become resident: slots x G never exceeds what the device holds, and launches of
    // this kind are chained on one event per device (launch.hip: launch_grid_kernel)
    std::lock_guard<std::mutex> lock(g_grid_mu);
    hipEvent_t& ev = g_grid_event[s.device & 15];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else
        HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
    // (per LM iteration a factorisation's hops up and down the tree and the reductions of the LM control -- at most a few per level;
    // the sequence numbers of the scratch start again before they wrap: system.hpp)
    if (seq_budget_spent(s.front_seq_used, fa.batch, 16ull * ((uint64_t)fa.max_iterations + 4) * std::max<uint32_t>(1, plan.n_levels)))
        HIP_TRY(hipMemsetAsync(s.front_scratch.p, 0, s.front_scratch.cap, stream));
    hipLaunchKernelGGL(kernel, dim3(slots * G), dim3(plan.threads), plan.lds_bytes, stream, fa);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev, stream));
    return EZPZ_OK;
}

static int front_launch_with(EzpzSystem& s, SolveArgs& args, hipStream_t stream, const FrontProbe& probe) {
    const FrontPlan& plan = *s.fronts;
    if (!s.dev_fronts) return EZPZ_ERR_INVALID_ARGUMENT;
    FrontArgs fa{};
    fa.plan = static_cast<const unsigned char*>(s.dev_fronts);
    fa.n_wgs = plan.n_wgs;
    fa.n_vars = plan.n_vars;
    fa.n_cons = plan.n_cons;
    fa.x0 = args.x0;
    fa.x_out = args.x_out;
    fa.status = args.status;
    fa.unsat_mask = args.unsat_mask;
    fa.warn_log = args.warn_log;
    fa.warn_cap = args.warn_cap;
    fa.max_iterations = args.max_iterations;
    fa.batch = args.batch;
    fa.residual_tolerance = args.residual_tolerance;
    fa.step_tolerance = args.step_tolerance;
    fa.initial_lambda = args.initial_lambda;
    fa.unit_weights = plan.unit_weights ? 1u : 0u;
    fa.tab_lds_bytes = (plan.tab_bytes_max + 15u) & ~15u;
    fa.ws_doubles = plan.ws_doubles_max;
    fa.n_chunks = plan.n_chunks;
    fa.bad_chunk0 = plan.bad_chunk0;
    fa.verdict_chunk = plan.verdict_chunk;
    fa.scratch = nullptr;
    fa.scratch_stride = 0;
    fa.probe_m = probe.m;  // (front_launch_probe)
    fa.probe_out = probe.out;
    fa.probe_in = probe.in;
    fa.probe_scale = probe.scale;
    fa.stamps = args.stamps;
    fa.done = args.done;
    fa.done.request = nullptr;  // (this kernel does not stay resident between calls)
    args.done.request = nullptr;
    return plan.linear_only ? front_launch_kernel<true>(s, fa, stream) : front_launch_kernel<false>(s, fa, stream);
}

int front_launch(EzpzSystem& s, SolveArgs& args, hipStream_t stream) { return front_launch_with(s, args, stream, FrontProbe{}); }

// The null-space probes of FreedomAnalysis (FrontArgs::probe_m): `m` probes of `batch` systems at the values x_dev ([batch][n_vars],
// caller order), answers to y_dev ([batch][m][n_vars]); w_dev: the probes' vectors ([batch][m][n_vars]), or null = pseudo-random entries, uniform in [-1, 1).
int front_launch_probe(EzpzSystem& s, const double* x_dev, size_t batch, double* y_dev, uint32_t m, hipStream_t stream, const double* w_dev,
                       double lambda_scale) {
    if (!s.fronts || !s.dev_fronts || !m) return EZPZ_ERR_INVALID_ARGUMENT;
    SolveArgs args{};
    args.x0 = x_dev;
    args.batch = batch * m;  // (a work item is one probe of one system: front_kernel.hip.hpp)
    args.max_iterations = 0;
    args.residual_tolerance = 0.0;
    args.step_tolerance = 0.0;
    args.initial_lambda = 0.0;
    // (like launch(): what a launch creates on first use -- the occupancy figure, the scratch of several workgroups -- under the lock)
    std::lock_guard<std::mutex> launch_lock(s.launch_mu);
    FrontProbe probe;
    probe.m = m;
    probe.out = y_dev;
    probe.in = w_dev;
    probe.scale = lambda_scale;
    return front_launch_with(s, args, stream, probe);
}

}  // namespace ezpz
