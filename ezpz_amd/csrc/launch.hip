// Kernel dispatch: which instantiation of lm_solve_kernel / which component, lane or specialised kernel a launch of an
// EzpzSystem takes (launch), the device-pointer entry point of the C ABI and the evaluation-only kernel.  The only
// translation unit that instantiates the list-walk / record-walk kernels (lm_kernel.hip.hpp).
#include "system.hpp"

#include "lm_kernel.hip.hpp"

using namespace ezpz;

namespace {

template <int TEAM, int MODE, bool LDSWS, bool PLDS, bool LIN, bool DENSE = false, int REC = 0>
int launch_kernel(EzpzSystem& s, const SolveArgs& args, uint32_t grid, hipStream_t stream) {
    auto kernel = lm_solve_kernel<TEAM, MODE, LDSWS, PLDS, LIN, false, DENSE, REC>;
    // hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel, not to the system: raised once per kernel
    // build and device, to everything the device allows, so that systems of different sizes sharing a build never
    // lower each other's limit
    static std::atomic<bool> raised[16];
    if (s.lds_bytes > 48 * 1024 && !raised[s.device & 15].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)s.lim.lds_bytes));
        raised[s.device & 15].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(s.block_threads), s.lds_bytes, stream, args);
    HIP_TRY(hipGetLastError());
    return EZPZ_OK;
}

// Every team shape comes in two builds: all 25 kinds, or the nine linear kinds only (`linear_only` topologies).
template <int TEAM, int MODE, bool LDSWS, bool PLDS>
int launch_variant(EzpzSystem& s, const SolveArgs& args, uint32_t grid, hipStream_t stream) {
    if (s.linear_only) return launch_kernel<TEAM, MODE, LDSWS, PLDS, true>(s, args, grid, stream);
    return launch_kernel<TEAM, MODE, LDSWS, PLDS, false>(s, args, grid, stream);
}

template <int TEAM>
int launch_sub(EzpzSystem& s, const SolveArgs& args, uint32_t grid, hipStream_t stream) {
    if constexpr (TEAM == 4) {  // <= 8 variables: dense factor layout, solved in registers (always staged)
        if (s.counts.dense)
            return s.linear_only ? launch_kernel<TEAM, MODE_SUB, true, true, true, true>(s, args, grid, stream)
                                 : launch_kernel<TEAM, MODE_SUB, true, true, false, true>(s, args, grid, stream);
    }
    return s.prog_in_lds ? launch_variant<TEAM, MODE_SUB, true, true>(s, args, grid, stream)
                         : launch_variant<TEAM, MODE_SUB, true, false>(s, args, grid, stream);
}

}  // namespace
namespace ezpz {
thread_local uint64_t t_call_batch = 0;
std::mutex g_grid_mu;
hipEvent_t g_grid_event[16] = {};  // per device: completion of the last grid-team launch of this process (front.hip's too)
}  // namespace ezpz
namespace {

// Grid team: G workgroups per system, all of a launch's workgroups resident at once, as many systems in flight as
// the device holds.
bool stream_capturing(hipStream_t stream);

template <bool LIN>
int launch_grid_kernel(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
    // (launches whose workgroups wait for each other are chained on a process-wide event and zero their scratch on first use: neither
    // may end up inside a stream capture, which would be invalidated and leave the event unusable -- refused up front)
    if (stream_capturing(stream)) return EZPZ_ERR_INVALID_ARGUMENT;
    auto kernel = lm_solve_kernel<64, MODE_PART, true, true, LIN, true>;
    if (s.grid_capacity == 0) {  // once per system: these two runtime calls cost more than the solve
        if (s.lds_bytes > 48 * 1024)
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)s.lim.lds_bytes));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)s.block_threads, s.lds_bytes));
        s.grid_capacity = (uint64_t)s.lim.cus * (uint64_t)std::max(per_cu, 1);
    }
    const uint64_t capacity = s.grid_capacity;
    if (capacity < s.grid_wgs) return EZPZ_ERR_TOO_LARGE;
    const uint32_t slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(args.batch, capacity / s.grid_wgs));
    int rc;
    if (!s.dev_grid_blob) {  // first launch: the workgroups' sub-programs and their views
        HIP_TRY(hipMalloc(&s.dev_grid_blob, s.grid_blob.size()));
        HIP_TRY(hipMemcpy(s.dev_grid_blob, s.grid_blob.data(), s.grid_blob.size(), hipMemcpyHostToDevice));
        std::vector<ProgramView> views = s.host_grid_views;
        for (ProgramView& pv : views) {
            pv.base = static_cast<const unsigned char*>(s.dev_grid_blob) + pv.blob_bytes;
            pv.blob_bytes = 0;
        }
        if ((rc = s.grid_views.ensure(views.size())) != EZPZ_OK) return rc;
        HIP_TRY(hipMemcpy(s.grid_views.p, views.data(), views.size() * sizeof(ProgramView), hipMemcpyHostToDevice));
    }
    if (s.grid_scratch.cap < slots) {
        if ((rc = s.grid_scratch.ensure(slots)) != EZPZ_OK) return rc;
        HIP_TRY(hipMemsetAsync(s.grid_scratch.p, 0, s.grid_scratch.cap * sizeof(GridScratch), stream));
    }
    args.grid_scratch = s.grid_scratch.p;
    args.grid_views = s.grid_views.p;
    args.grid_wgs = s.grid_wgs;
    // Every workgroup of the launch must become resident (they wait for each other).  slots * G never exceeds what
    // the device holds, and grid-team launches of this process are chained on one event per device, so two of them
    // are never half-resident at the same time whatever streams they were enqueued on; other kernels only delay
    // residency.  (hipLaunchCooperativeKernel gives the same guarantee across processes but costs 21 us per launch,
    // more than a third of a 200 000-variable solve; another process running grid teams on the same device at the
    // same time is not supported.)
    {
        std::lock_guard<std::mutex> lock(g_grid_mu);
        hipEvent_t& ev = g_grid_event[s.device & 15];
        if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        else HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
        // (three exchanges per LM iteration and a few around them; the sequence numbers start again before they wrap: system.hpp)
        if (seq_budget_spent(s.grid_seq_used, args.batch, 3ull * ((uint64_t)args.max_iterations + 4)))
            HIP_TRY(hipMemsetAsync(s.grid_scratch.p, 0, s.grid_scratch.cap * sizeof(GridScratch), stream));
        hipLaunchKernelGGL(kernel, dim3(slots * s.grid_wgs), dim3(s.block_threads), s.lds_bytes, stream, args);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev, stream));
    }
    return EZPZ_OK;
}

int launch_grid_team(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
    return s.linear_only ? launch_grid_kernel<true>(s, args, stream) : launch_grid_kernel<false>(s, args, stream);
}

// (a launch that is being recorded into a graph never takes the `_fast` entry: its redo list is zeroed by the NEXT call's launches,
// which a replay does not run)
bool stream_capturing(hipStream_t stream) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
        (void)hipGetLastError();
        capturing = hipStreamCaptureStatusNone;
    }
    return capturing != hipStreamCaptureStatusNone;
}

// Launches of one system that share device state between them -- the counters its workgroups draw their systems from, the redo lists of
// its `_fast` entry, whose count a call's last launch zeroes for the NEXT call -- run one after the other whatever streams they are
// enqueued on: a launch on another stream than the last one waits for everything enqueued there.  (An event per launch instead cost the
// back-to-back launches of one stream two runtime calls and a barrier packet each.)  False: could not be arranged.
bool chain_launches(EzpzSystem& s, hipStream_t stream) {
    bool ok = true;
    if (!s.ticket_done) ok = hipEventCreateWithFlags(&s.ticket_done, hipEventDisableTiming) == hipSuccess;
    if (ok && s.ticket_used && s.ticket_stream != stream) {
        ok = hipEventRecord(s.ticket_done, s.ticket_stream) == hipSuccess && hipStreamWaitEvent(stream, s.ticket_done, 0) == hipSuccess;
        if (!ok) {  // (the old stream is gone: whatever ran on it is awaited the blunt way)
            (void)hipGetLastError();
            ok = hipDeviceSynchronize() == hipSuccess;
        }
    }
    if (!ok) (void)hipGetLastError();
    return ok;
}
// The counters a specialised kernel's workgroups draw their systems from (jit_kernel.hip.hpp: JitArgs::ticket), for a launch on
// `stream`: created on first use; launches that share them are chained.  False: the launch keeps fixed shares.
bool prepare_tickets(EzpzSystem& s, hipStream_t stream) {
    static const bool tickets_enabled = [] {  // EZPZ_TICKETS=0: fixed shares (A/B runs: 101.1 -> 110.0 M solves/s with them)
        const char* e = std::getenv("EZPZ_TICKETS");
        return !(e && e[0] == '0');
    }();
    // (a launch that is being recorded into a graph keeps fixed shares: a replay would find the counters elsewhere)
    if (!tickets_enabled || stream_capturing(stream)) return false;
    bool ok = s.ticket.p != nullptr;
    if (!ok && s.ticket.ensure(8 * 1024) == EZPZ_OK) {  // (jit_kernel.hip.hpp: kTicketStride words apart)
        ok = hipMemset(s.ticket.p, 0, 8 * 1024 * sizeof(unsigned int)) == hipSuccess;
        for (unsigned int& b : s.ticket_base) b = 0;
    }
    ok = ok && chain_launches(s, stream);
    if (!ok) (void)hipGetLastError();
    return ok;
}
// ... and what a launch of `workgroups` workgroups over `batch` systems has drawn from each: counter c hands out its share of the
// systems beyond the workgroups' own, and `in_vain` values more to each of its workgroups (the loop kernel draws once per system
// it solves, the last time in vain; the kernel that asks for its guesses a system ahead draws a system ahead: twice in vain)
void advance_tickets(EzpzSystem& s, hipStream_t stream, uint64_t workgroups, uint64_t batch, unsigned int in_vain) {
    for (uint64_t c = 0; c < 8; ++c) {
        const uint64_t wgs_c = (workgroups + 7 - c) / 8, beyond = batch - workgroups;
        s.ticket_base[c] += (unsigned int)(in_vain * wgs_c + (beyond > c ? (beyond - c + 7) / 8 : 0));
    }
    s.ticket_stream = stream;
    s.ticket_used = true;
}

// The two redo lists of a system whose specialised kernel has a `_fast` entry (system.hpp: jit_redo), for `batch` systems, and the
// mapped word the loop's launch leaves its count in.
int jit_redo_lists(EzpzSystem& s, uint64_t batch, hipStream_t stream) {
    for (auto& list : s.jit_redo)
        if (list.cap < batch + 1) {
            int rc = list.ensure(batch + 1);  // (synchronises the device: nobody reads the old one any more)
            if (rc != EZPZ_OK) return rc;
            HIP_TRY(hipMemsetAsync(list.p, 0, sizeof(unsigned int), stream));
        }
    if (!s.jit_redo_seen) {
        HIP_TRY(hipHostMalloc((void**)&s.jit_redo_seen, sizeof(unsigned int), hipHostMallocMapped));
        *s.jit_redo_seen = 0;
        HIP_TRY(hipHostGetDevicePointer((void**)&s.jit_redo_seen_dev, s.jit_redo_seen, 0));
    }
    return EZPZ_OK;
}

// The class-specialised kernel of a system spread over several workgroups (CompPlan::jit_wgs > 1): as many systems in
// flight as the device holds whole teams of; every workgroup of the launch must be resident (they wait for each other),
// so launches of this kind are chained like the list-walk grid teams' (launch_grid_kernel).
int launch_jit_grid(EzpzSystem& s, const CompLaunch& L, hipStream_t stream) {
    if (stream_capturing(stream)) return EZPZ_ERR_INVALID_ARGUMENT;  // (as launch_grid_kernel)
    const uint32_t G = s.comp->jit_wgs;
    const uint64_t capacity = comp_jit_capacity(s.jit, *s.comp, s.device, s.lim.cus);
    if (capacity < G) return EZPZ_ERR_TOO_LARGE;
    const uint32_t slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(L.batch, capacity / G));
    // (a linear system: the kernel that does not wait for its verdicts goes first -- leaner, so more systems in flight -- and the
    // loop solves what it lists)
    const uint32_t fast_slots = (uint32_t)(comp_jit_capacity_fast(s.jit, *s.comp, s.device, s.lim.cus) / G);
    const uint32_t most = std::max(slots, fast_slots);
    if (s.jit_scratch.cap < (size_t)most * kJitGridScratchBytes) {
        // (sized for the device, not the call: a later, larger call must find the sequence numbers the slots have reached)
        const uint32_t all = (uint32_t)std::max<uint64_t>(most, std::max<uint64_t>(capacity / G, fast_slots));
        int rc = s.jit_scratch.ensure((size_t)all * kJitGridScratchBytes);
        if (rc != EZPZ_OK) return rc;
        HIP_TRY(hipMemsetAsync(s.jit_scratch.p, 0, s.jit_scratch.cap, stream));
    }
    const bool lists = fast_slots && comp_jit_fast_ok(s.jit, *s.comp, L, s.device, s.lim.cus) && !stream_capturing(stream);
    if (lists) {
        int rc = jit_redo_lists(s, L.batch, stream);
        if (rc != EZPZ_OK) return rc;
    }
    std::lock_guard<std::mutex> lock(g_grid_mu);
    hipEvent_t& ev = g_grid_event[s.device & 15];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else
        HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
    // (the loop's three exchanges per LM iteration, the ring's one per system: the sequence numbers start again before they wrap)
    if (seq_budget_spent(s.jit_seq_used, L.batch, 3ull * ((uint64_t)L.max_iterations + 4)))
        HIP_TRY(hipMemsetAsync(s.jit_scratch.p, 0, s.jit_scratch.cap, stream));
    const unsigned int turn = s.jit_redo_turn;
    int rc = comp_jit_launch(s.jit, *s.comp, s.dev_comp, L, s.device, s.lim.cus, stream, s.jit_scratch.p, slots, lists ? fast_slots : 0,
                             lists ? s.jit_redo[turn].p : nullptr, lists ? s.jit_redo[turn ^ 1u].p : nullptr, s.jit_redo_seen_dev,
                             s.jit_redo_seen);
    if (lists) s.jit_redo_turn = turn ^ 1u;
    if (rc != EZPZ_OK) return rc;
    HIP_TRY(hipEventRecord(ev, stream));
    return EZPZ_OK;
}


CompLaunch comp_launch_args(const SolveArgs& args) {
    CompLaunch L{};
    L.x0 = args.x0;
    L.x_out = args.x_out;
    L.status = args.status;
    L.unsat_mask = args.unsat_mask;
    L.warn_log = args.warn_log;
    L.warn_cap = args.warn_cap;
    L.batch = args.batch;
    L.max_iterations = args.max_iterations;
    L.residual_tolerance = args.residual_tolerance;
    L.step_tolerance = args.step_tolerance;
    L.initial_lambda = args.initial_lambda;
    L.done = args.done;
    return L;
}


// The list-walk teams of a system (lm_kernel.hip.hpp), whatever their shape: sub-wavefront teams, workgroups with their
// workspace in LDS or in global memory, grid teams.  (launch() holds the system's launch lock.)
int launch_list_walk(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
    // a resident launch (DoneWord::request) is one workgroup that keeps nothing another launch of this system waits for:
    // not a grid team, not a shape whose workspace or Jacobian lives in the system's one global scratch
    if (s.grid_wgs > 1 || args.batch != 1 || (s.mode != MODE_SUB && (!s.lds_ws || (s.rec && s.rec_jglobal)))) args.done.request = nullptr;
    uint32_t grid;
    if (s.mode == MODE_SUB) {
        const uint32_t tpb = s.block_threads / s.team_size;
        uint64_t blocks = (args.batch + tpb - 1) / tpb;
        grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)s.lim.cus * 32);
        switch (s.team_size) {
        case 1: return launch_sub<1>(s, args, grid, stream);
        case 2: return launch_sub<2>(s, args, grid, stream);
        case 4: return launch_sub<4>(s, args, grid, stream);
        case 8: return launch_sub<8>(s, args, grid, stream);
        case 16: return launch_sub<16>(s, args, grid, stream);
        case 32: return launch_sub<32>(s, args, grid, stream);
        default: return launch_sub<64>(s, args, grid, stream);
        }
    }
    if (s.grid_wgs > 1) {
        // (a grid team starts every system from its guesses: its shared warning counter has no resumed value)
        if (args.resume) return EZPZ_ERR_INVALID_ARGUMENT;
        return launch_grid_team(s, args, stream);
    }
    const uint32_t per_cu = s.lds_ws ? (uint32_t)std::max<size_t>(1, s.lim.lds_bytes / std::max<size_t>(s.lds_bytes, 1))
                                     : 2048u / s.block_threads;
    // Workgroups: twice what the device holds at once where a workgroup serves several systems side by side (sub-wavefront teams)
    // or owns a workspace in global memory; a workgroup per system -- up to 32 times what the device holds -- where it solves one
    // system at a time in its LDS: the dispatcher then hands a free place the next system, whatever the systems before it took
    // (a jittered batch's systems take 4 to 10 iterations) and whoever else occupies places on the device -- sketch150 x 32 768
    // at x1 / x2 / x4 / x8 / x32: 3.58 / 3.62 / 3.70 / 3.82 / 3.91 M solves/s; starting a workgroup costs a few microseconds
    // against the ~250 of a system.
    // (a list of systems on the device -- the lanes' stragglers: `batch` is the list's capacity, the systems are a few hundred)
    const uint32_t rounds = (s.lds_ws && s.mode != MODE_SUB && !args.sys_list) ? 32u : 2u;
    grid = (uint32_t)std::min<uint64_t>(args.batch, (uint64_t)s.lim.cus * std::min<uint32_t>(per_cu, 8) * rounds);
    if (s.rec && s.rec_jglobal)  // (a workgroup's Jacobian values in global memory: at most 256 MiB of them per system object)
        grid = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(grid, (1ull << 25) / ((s.counts.zj + 2) & ~1ull)));
    if (!s.lds_ws) {
        int rc = s.gws_dev.ensure((size_t)grid * s.ws_doubles);
        if (rc != EZPZ_OK) return rc;
        args.gws = s.gws_dev.p;
    }
    const bool staged = s.prog_in_lds;
    if (!s.lds_ws) {
        // the workspace in global memory is one per system object: launches on different streams are chained on an event
        // (like the lanes kernel's), never overlapped
        if (!s.lanes_done)
            HIP_TRY(hipEventCreateWithFlags(&s.lanes_done, hipEventDisableTiming));
        else
            HIP_TRY(hipStreamWaitEvent(stream, s.lanes_done, 0));
        const int rc = s.mode == MODE_PART ? launch_variant<64, MODE_PART, false, false>(s, args, grid, stream)
                       : !s.rec            ? launch_variant<64, MODE_WGB, false, false>(s, args, grid, stream)
                       : s.linear_only     ? launch_kernel<64, MODE_WGB, false, false, true, false, 2>(s, args, grid, stream)
                                           : launch_kernel<64, MODE_WGB, false, false, false, false, 2>(s, args, grid, stream);
        if (rc == EZPZ_OK) HIP_TRY(hipEventRecord(s.lanes_done, stream));
        return rc;
    }
    if (s.mode == MODE_PART)
        return staged ? launch_variant<64, MODE_PART, true, true>(s, args, grid, stream)
                      : launch_variant<64, MODE_PART, true, false>(s, args, grid, stream);
    if (s.rec) {  // one connected system, its linear solve as a record walk
        if (s.rec_jglobal) {
            // the Jacobian's values of every workgroup in global memory: one array per system object, launches on different
            // streams chained on an event (like the other per-system device scratch)
            const size_t stride = (s.counts.zj + 2) & ~1u;
            int rc = s.gws_dev.ensure((size_t)grid * stride);
            if (rc != EZPZ_OK) return rc;
            args.gws = s.gws_dev.p;
            if (!s.lanes_done)
                HIP_TRY(hipEventCreateWithFlags(&s.lanes_done, hipEventDisableTiming));
            else
                HIP_TRY(hipStreamWaitEvent(stream, s.lanes_done, 0));
            rc = s.linear_only ? (staged ? launch_kernel<64, MODE_WGB, true, true, true, false, 1>(s, args, grid, stream)
                                         : launch_kernel<64, MODE_WGB, true, false, true, false, 1>(s, args, grid, stream))
                               : (staged ? launch_kernel<64, MODE_WGB, true, true, false, false, 1>(s, args, grid, stream)
                                         : launch_kernel<64, MODE_WGB, true, false, false, false, 1>(s, args, grid, stream));
            if (rc == EZPZ_OK) HIP_TRY(hipEventRecord(s.lanes_done, stream));
            return rc;
        }
        if (s.linear_only)
            return staged ? launch_kernel<64, MODE_WGB, true, true, true, false, 1>(s, args, grid, stream)
                          : launch_kernel<64, MODE_WGB, true, false, true, false, 1>(s, args, grid, stream);
        return staged ? launch_kernel<64, MODE_WGB, true, true, false, false, 1>(s, args, grid, stream)
                      : launch_kernel<64, MODE_WGB, true, false, false, false, 1>(s, args, grid, stream);
    }
    return staged ? launch_variant<64, MODE_WGB, true, true>(s, args, grid, stream)
                  : launch_variant<64, MODE_WGB, true, false>(s, args, grid, stream);
}

int launch(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
    if (args.batch == 0) return EZPZ_OK;
    // enqueueing on one EzpzSystem from several threads (each on its own stream) is allowed: what a launch creates on
    // first use -- workspaces, events, occupancy figures -- is created under this lock, and launches that share a
    // workspace are chained on an event below
    std::lock_guard<std::mutex> launch_lock(s.launch_mu);
    constexpr uint64_t kNoLanesWorkspace = ~0ull;  // the allocation failed once: not tried again on every call
    if (args.batch != 1) args.done.request = nullptr;  // (residency is for one-call launches: one system, one workgroup)
    if (s.lanes && args.batch >= s.lanes_min) {
        args.done.request = nullptr;  // a device-filling batch of one connected sketch: lanes across the batch
        if (s.lanes_ws_waves == 0) {
            // one workspace per wavefront the device holds (capped at 24 GiB of the 288: fewer wavefronts then)
            uint64_t waves = batch_launch_waves(s.lim.cus);
            const uint64_t per = (uint64_t)s.lanes->rows * 512;
            while (waves > 4 && waves * per > (24ull << 30)) waves /= 2;
            s.lanes_ws_waves = s.lanes_ws.ensure((size_t)(waves * per / 8)) == EZPZ_OK ? waves : kNoLanesWorkspace;
        }
        if (s.lanes_ws_waves != kNoLanesWorkspace) {
            // one workspace per system object: launches on different streams are chained, never overlapped
            if (!s.lanes_done)
                HIP_TRY(hipEventCreateWithFlags(&s.lanes_done, hipEventDisableTiming));
            else
                HIP_TRY(hipStreamWaitEvent(stream, s.lanes_done, 0));
            // the systems the lanes give up (stragglers, batch_kernel.hip.hpp) are listed on the device and resumed by this
            // system's list-walk teams right after: an indirect batch whose count stays on the device
            // (room for every wavefront handing over its threshold's worth of lanes once: a list that overflows leaves the lanes their tail)
            const uint64_t strag_most = std::min<uint64_t>(s.lanes_ws_waves, (args.batch + 63) / 64) * batch_straggler_lanes();
            const uint32_t strag_cap = args.batch < (1ull << 32) && args.batch >= 256 && strag_most
                                           ? (uint32_t)std::min<uint64_t>(args.batch, std::max<uint64_t>(4096, strag_most)) : 0u;
            bool list_ok = strag_cap && s.strag_list.ensure(strag_cap) == EZPZ_OK && s.strag_count.ensure(1) == EZPZ_OK &&
                           s.strag_state.ensure(strag_cap) == EZPZ_OK;
            if (list_ok && hipMemsetAsync(s.strag_count.p, 0, sizeof(uint32_t), stream) != hipSuccess) {
                (void)hipGetLastError();
                list_ok = false;
            }
            if (batch_launch(*s.lanes, s.dev_lanes, s.lanes_ws.p, s.lanes_ws_waves, s.counts.n_cons, comp_launch_args(args), stream,
                             list_ok ? s.strag_list.p : nullptr, list_ok ? s.strag_count.p : nullptr, list_ok ? strag_cap : 0u,
                             list_ok ? s.strag_state.p : nullptr) == EZPZ_OK) {
                int rc = EZPZ_OK;
                if (list_ok) {
                    args.sys_list = s.strag_list.p;
                    args.sys_count = s.strag_count.p;
                    args.resume = s.strag_state.p;  // (the teams go on from the values the lanes left in x_out)
                    args.batch = strag_cap;
                    rc = launch_list_walk(s, args, stream);
                }
                // (after the teams: the next launch of this system, on whatever stream, resets the list's count)
                HIP_TRY(hipEventRecord(s.lanes_done, stream));
                return rc;
            }
        }
    }
    // one connected sketch as a tree of dense fronts (fronts.cpp): every call of a system created for one solve, the small calls
    // of a system created for batches
    if (s.fronts && !args.resume && !args.sys_list && (t_call_batch ? t_call_batch : args.batch) <= s.front_max_batch) {
        // (a launch that is being recorded into a graph: fronts on several workgroups allocate and zero their scratch on first use and
        // chain their launches on the process-wide event of the grid teams -- neither may end up inside a capture, which would also
        // leave that event unusable for the launches after it.  A system created for batches has its other shapes and takes them; a
        // system the fronts alone serve says so)
        if (s.fronts->n_wgs > 1 && stream_capturing(stream)) {
            if (s.front_max_batch == 0xFFFFFFFFu) return EZPZ_ERR_INVALID_ARGUMENT;
        } else {
            return front_launch(s, args, stream);
        }
    }
    if (s.jit && s.launches.load(std::memory_order_relaxed) == 0) comp_jit_probe(s.jit);  // the kernel may be in the on-disk cache
    if (s.lane && s.wave_jit && args.batch <= (uint64_t)s.lim.cus) {
        // one solve (or a few) of a small system built for latency: one wavefront per system, sweeps and assembly across its
        // lanes (jit_kernel.hip.hpp: wave_kernel), compiled like the lane kernel
        if (s.launches.load(std::memory_order_relaxed) == 0) comp_jit_probe(s.wave_jit);
        int st = comp_jit_state(s.wave_jit);
        if (st == 0 && (jit_sync() || s.launches.load(std::memory_order_relaxed) >= s.lim.policy.jit_after_launches)) st = comp_jit_request(s.wave_jit, jit_sync());
        if (st == 2 && wave_jit_launch(s.wave_jit, *s.lane, comp_launch_args(args), s.device, s.lim.cus, stream) == EZPZ_OK) return EZPZ_OK;
    }
    if (s.lane && s.jit) {  // a small system: one lane per system once the specialised kernel is compiled
        int st = comp_jit_state(s.jit);
        const EzpzLaunchPolicy& pol = s.lim.policy;
        if (st == 0 && (args.batch >= pol.jit_lane_min_batch || jit_sync() || s.launches.fetch_add(1) >= pol.jit_after_launches))
            st = comp_jit_request(s.jit, jit_sync());
        if (st == 2 && lane_jit_launch(s.jit, *s.lane, comp_launch_args(args), s.device, s.lim.cus, stream) == EZPZ_OK) return EZPZ_OK;
    }
    if (s.comp) {  // many small components in few classes: one lane per component (comp_kernel.hip.hpp)
        const CompLaunch L = comp_launch_args(args);
        // the class-specialised kernel once it is compiled; large batches start its compilation (background thread)
        if (s.jit) {
            const bool sync = jit_sync();
            int st = comp_jit_state(s.jit);
            const EzpzLaunchPolicy& pol = s.lim.policy;
            const bool big = args.batch >= pol.jit_comp_min_batch || args.batch * (uint64_t)s.counts.n_vars >= pol.jit_comp_min_values;
            if (st == 0 && (big || sync || s.launches.fetch_add(1) >= pol.jit_after_launches)) st = comp_jit_request(s.jit, sync);
            if (st == 2) {
                if (s.comp->jit_wgs <= 1 && comp_jit_fast_ok(s.jit, *s.comp, L, s.device, s.lim.cus) && !stream_capturing(stream) &&
                    jit_redo_lists(s, L.batch, stream) == EZPZ_OK && chain_launches(s, stream)) {
                    // a linear system: the kernel that does not wait for the LM control's verdicts, then the loop over the systems it
                    // lists (jit_kernel.hip.hpp: solve_kernel_fast)
                    const unsigned int turn = s.jit_redo_turn;
                    const uint32_t fast_wgs = (uint32_t)std::min<uint64_t>(comp_jit_capacity_fast(s.jit, *s.comp, s.device, s.lim.cus), 0xFFFFFFFFull);
                    CompLaunch Lt = L;
                    if (L.batch > fast_wgs && prepare_tickets(s, stream)) Lt.ticket = s.ticket.p, Lt.ticket_base = s.ticket_base;
                    if (comp_jit_launch(s.jit, *s.comp, s.dev_comp, Lt, s.device, s.lim.cus, stream, nullptr, 0, fast_wgs, s.jit_redo[turn].p,
                                        s.jit_redo[turn ^ 1u].p, s.jit_redo_seen_dev, s.jit_redo_seen) == EZPZ_OK) {
                        if (Lt.ticket) advance_tickets(s, stream, fast_wgs, L.batch, 2);
                        s.ticket_stream = stream;  // (the redo lists are shared with the next call: chain_launches)
                        s.ticket_used = true;
                        s.jit_redo_turn = turn ^ 1u;
                        return EZPZ_OK;
                    }
                } else if (s.comp->jit_wgs <= 1) {
                    // a batch beyond the launch's workgroups: the workgroups draw their systems from the system's counter, and
                    // launches that share the counter are chained
                    const uint64_t capacity = L.done.flag ? 0 : comp_jit_capacity(s.jit, *s.comp, s.device, s.lim.cus);
                    CompLaunch Lt = L;
                    if (capacity && L.batch > capacity && L.batch < (1ull << 32) && prepare_tickets(s, stream)) Lt.ticket = s.ticket.p, Lt.ticket_base = s.ticket_base;
                    if (comp_jit_launch(s.jit, *s.comp, s.dev_comp, Lt, s.device, s.lim.cus, stream) == EZPZ_OK) {
                        if (Lt.ticket) advance_tickets(s, stream, capacity, L.batch, 1);
                        return EZPZ_OK;
                    }
                } else {
                    CompLaunch Lg = L;
                    Lg.done.request = nullptr;  // (several workgroups per system: never resident)
                    if (launch_jit_grid(s, Lg, stream) == EZPZ_OK) {
                        args.done.request = nullptr;
                        return EZPZ_OK;
                    }
                }
            }
        }
        if (s.comp->interpretable) return comp_launch(*s.comp, s.dev_comp, L, s.device, s.lim.cus, s.lim.lds_bytes, stream);
        // (a system too large for the interpreter's LDS state: the list-walk grid team below until the specialised
        // kernel is ready)
    }
    return launch_list_walk(s, args, stream);
}


void fill_cfg(SolveArgs& a, const EzpzConfig* cfg) {
    EzpzConfig d;
    ezpz_default_config(&d);
    if (!cfg) cfg = &d;
    a.max_iterations = (uint32_t)std::min<uint64_t>(cfg->max_iterations, 0xFFFFFFFFull);
    a.residual_tolerance = cfg->residual_tolerance;
    a.step_tolerance = cfg->step_tolerance;
    a.initial_lambda = cfg->initial_lambda;
}

}  // namespace

namespace ezpz {

unsigned long long* g_stamps = nullptr;

// (`resident`: whether the launch stays on the device for further requests, DoneWord::request)
int solve_batch_device_impl(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg, double* x_out_dev,
                                   EzpzStatus* status_dev, uint8_t* unsat_mask_dev, uint64_t* warn_log_dev, uint32_t warn_cap,
                                   void* stream, const DoneWord& done, bool* resident) {
    if (!sys || (batch && (!x_out_dev || !status_dev))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch && sys->counts.n_vars && !x0_dev) return EZPZ_ERR_INVALID_ARGUMENT;
    EZPZ_ON_DEVICE(sys->device);
    SolveArgs a{};
    a.p = sys->view;
    a.x0 = x0_dev;
    a.x_out = x_out_dev;
    a.status = status_dev;
    a.unsat_mask = unsat_mask_dev;
    a.warn_log = warn_cap ? warn_log_dev : nullptr;
    a.warn_cap = warn_cap;
    a.gws = nullptr;
    a.batch = batch;
    a.ws_doubles = sys->ws_doubles;
    a.prog_lds_doubles = sys->prog_lds_doubles;
    a.lvl_lds_off = sys->lvl_lds_off;
    a.lvl_tab_words = sys->lvl_tab_words;
    a.lvl_buf_words = sys->lvl_buf_words;
    a.n_dense = sys->n_dense;
    a.dense_level0 = sys->dense_level0;
    a.dense_lds_off = sys->dense_lds_off;
    a.dense_lds_doubles = sys->dense_lds_doubles;
    a.stamps = g_stamps;
    a.unit_weights = sys->unit_weights ? 1u : 0u;
    a.grid_wgs = 1;
    a.grid_scratch = nullptr;
    a.grid_views = nullptr;
    a.sys_list = nullptr;
    a.sys_count = nullptr;
    a.resume = nullptr;
    a.done = done;
    if (sys->rec) {
        const unsigned char* base = static_cast<const unsigned char*>(sys->dev_program);
        a.rec_desc = reinterpret_cast<const uint2*>(base + sys->rec_desc_off);
        a.rec_chunks = reinterpret_cast<const uint4*>(base + sys->rec_chunks_off);
        a.rec_rounds = sys->rec_rounds;
        a.rec_desc_off = sys->rec_desc_lds_off;
        if (sys->rec_asm_kc) {
            a.rec_asm_cols = reinterpret_cast<const uint4*>(base + sys->rec_asm_cols_off);
            a.rec_asm_slots = reinterpret_cast<const uint4*>(base + sys->rec_asm_slots_off);
            a.rec_asm_kc = sys->rec_asm_kc;
            a.rec_asm_ks = sys->rec_asm_ks;
        }
        const uint32_t n = sys->counts.n_vars, m = sys->counts.n_rows;
        const uint32_t o_d = n + 2 * m + (sys->rec_jglobal ? 0u : sys->counts.zj), o_dd = rec_ws_base(sys->counts, sys->rec_jglobal);
        a.rec_dd_delta = o_dd - o_d;
        a.rec_zero = o_dd + n;
        a.rec_jglobal = sys->rec_jglobal ? 1u : 0u;
        a.rec_jstride = (sys->counts.zj + 2) & ~1u;  // (the values, the zero of padding pairs)
    }
    fill_cfg(a, cfg);
    const int rc = launch(*sys, a, static_cast<hipStream_t>(stream));
    if (resident) *resident = rc == EZPZ_OK && a.done.request != nullptr;
    return rc;
}


// The systems of one topology inside a heterogeneous batch, solved IN PLACE by the lane-per-system kernel (mixed.hip): lane i
// reads its values at row_offset[i] of the caller's ragged buffer, writes them back there and its status to
// status[sys_of[i]] -- no gather into a block, no scatter back.  EZPZ_OK when it ran that way; 1 when this topology is not
// (yet) served by that kernel (not a small system, kernel not compiled): the caller gathers, calls the ordinary entry, scatters.
int lane_indexed_launch(EzpzSystem* sys, const double* x_ragged, const uint64_t* row_offset_dev, const uint32_t* sys_of_dev,
                        uint64_t count, const EzpzConfig* cfg, double* x_out_ragged, EzpzStatus* status_all, void* stream) {
    if (!sys || !sys->lane || !sys->jit || count == 0) return 1;
    std::lock_guard<std::mutex> launch_lock(sys->launch_mu);
    int st = comp_jit_state(sys->jit);
    if (st == 0 && (count >= sys->lim.policy.jit_lane_min_batch || jit_sync())) st = comp_jit_request(sys->jit, jit_sync());
    if (st != 2) return 1;
    SolveArgs a{};
    fill_cfg(a, cfg);
    CompLaunch L = comp_launch_args(a);
    L.x0 = x_ragged;
    L.x_out = x_out_ragged;
    L.status = status_all;
    L.batch = count;
    L.row_offset = row_offset_dev;
    L.sys_of = sys_of_dev;
    return lane_jit_launch(sys->jit, *sys->lane, L, sys->device, sys->lim.cus, static_cast<hipStream_t>(stream)) == EZPZ_OK ? EZPZ_OK : 1;
}

// The pipelined host-to-host path's copy out, as a kernel (pipeline.cpp): `bytes` (a multiple of 8: rows of doubles, 32-byte
// statuses) from device memory into registered host memory through its device address.  16 bytes per lane and store where both
// addresses and the length allow it; a piece of an odd number of doubles, or one that starts on an odd double (odd n_vars and
// an odd number of systems before it), goes 8 bytes at a time.  Which engine moves a hipMemcpyAsync is the runtime's choice;
// this one is ours (tools/pcie_duplex.hip: a copy kernel out beside copies in keeps 41-43 GB/s each way whatever moves the
// copies in).
template <class T>
__global__ void __launch_bounds__(256) copy_out_kernel(const T* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_copy_out(void* dst_host_as_device, const void* src_dev, size_t bytes, void* stream) {
    const bool wide = ((reinterpret_cast<uintptr_t>(dst_host_as_device) | reinterpret_cast<uintptr_t>(src_dev) | bytes) & 15u) == 0;
    if (wide)
        hipLaunchKernelGGL(copy_out_kernel<uint4>, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(src_dev),
                           static_cast<uint4*>(dst_host_as_device), bytes / 16);
    else
        hipLaunchKernelGGL(copy_out_kernel<uint2>, dim3(64), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint2*>(src_dev),
                           static_cast<uint2*>(dst_host_as_device), bytes / 8);
}

// The evaluation-only kernel (K1: residuals and Jacobian values at given points, internal numbering) on `stream`.
void launch_eval(EzpzSystem* sys, const double* x_int_dev, size_t batch, double* r_out_dev, double* jv_out_dev, uint32_t* deg_out_dev,
                 uint32_t grid, hipStream_t stream) {
    EvalArgs e{};
    e.p = sys->view;
    e.x = x_int_dev;
    e.r_out = r_out_dev;
    e.jv_out = jv_out_dev;
    e.deg_out = deg_out_dev;
    e.batch = batch;
    hipLaunchKernelGGL(eval_kernel, dim3(grid), dim3(256), 0, stream, e);
}

}  // namespace ezpz

extern "C" {

int ezpz_system_solve_batch_device(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg,
                                   double* x_out_dev, EzpzStatus* status_dev, uint8_t* unsat_mask_dev,
                                   uint64_t* warn_log_dev, uint32_t warn_cap, void* stream) {
    if (sys) release_thread_kernel(sys->device);
    return solve_batch_device_impl(sys, x0_dev, batch, cfg, x_out_dev, status_dev, unsat_mask_dev, warn_log_dev, warn_cap, stream,
                                   DoneWord{nullptr, 0, nullptr});
}

int ezpz_system_eval_batch(EzpzSystem* sys, const double* x, size_t batch, double* r_out, double* jv_out,
                           uint32_t* degenerate_count_out) {
    if (!sys || !x || !r_out || !jv_out) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    release_thread_kernel(sys->device);
    if (int rc0 = ensure_program(sys)) return rc0;
    std::lock_guard<std::mutex> lock(sys->mu);
    EZPZ_ON_DEVICE(sys->device);
    return ezpz::eval_batch_locked(sys, x, batch, r_out, jv_out, degenerate_count_out);
}

}  // extern "C"

// (the caller holds sys->mu, the system's device is current and its program is there: ensure_program)
int ezpz::eval_batch_locked(EzpzSystem* sys, const double* x, size_t batch, double* r_out, double* jv_out, uint32_t* degenerate_count_out) {
    const size_t n = sys->counts.n_vars, m = sys->counts.n_rows, zj = sys->counts.zj;
    DevBuf<double> xd, rd, jd;
    DevBuf<uint32_t> dd;
    int rc;
    if ((rc = xd.ensure(batch * std::max<size_t>(n, 1))) != EZPZ_OK) return rc;
    if ((rc = rd.ensure(batch * std::max<size_t>(m, 1))) != EZPZ_OK) return rc;
    if ((rc = jd.ensure(batch * std::max<size_t>(zj, 1))) != EZPZ_OK) return rc;
    if ((rc = dd.ensure(batch)) != EZPZ_OK) return rc;
    // the evaluators address values / rows by the program's internal numbering
    std::vector<double> xin(batch * std::max<size_t>(n, 1)), rin(batch * std::max<size_t>(m, 1));
    for (size_t b = 0; b < batch; ++b)
        for (size_t k = 0; k < n; ++k) xin[b * n + k] = x[b * n + sys->host_var_of[k]];
    HIP_TRY(hipMemcpy(xd.p, xin.data(), batch * n * sizeof(double), hipMemcpyHostToDevice));
    EvalArgs e{};
    e.p = sys->view;
    e.x = xd.p;
    e.r_out = rd.p;
    e.jv_out = jd.p;
    e.deg_out = dd.p;
    e.batch = batch;
    uint32_t grid = (uint32_t)std::min<size_t>(batch, 4096);
    hipLaunchKernelGGL(eval_kernel, dim3(grid), dim3(256), 0, nullptr, e);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(rin.data(), rd.p, batch * m * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < batch; ++b)
        for (size_t k = 0; k < m; ++k) r_out[b * m + sys->host_row_of[k]] = rin[b * m + k];
    HIP_TRY(hipMemcpy(jv_out, jd.p, batch * zj * sizeof(double), hipMemcpyDeviceToHost));
    if (degenerate_count_out)
        HIP_TRY(hipMemcpy(degenerate_count_out, dd.p, batch * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return EZPZ_OK;
}

extern "C" {


}  // extern "C"
