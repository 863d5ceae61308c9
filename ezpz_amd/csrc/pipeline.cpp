// Host-pointer entry points: how guesses get to the device and results back.  ONE solve() call (system_solve_one: the
// guesses through the PCIe BAR, the completion word in mapped host memory, the kernel resident between calls), small
// calls through mapped host memory, registered buffers through the three-stage pipeline, pageable buffers in chunks.
// No kernels here: everything is enqueued through solve_batch_device_impl (launch.hip).
#include "system.hpp"

using namespace ezpz;

namespace {

// Host ranges the caller has registered (ezpz_host_register): page-locked, so batch calls can DMA straight from / to
// them with asynchronous copies that overlap the kernels.
std::mutex g_host_mu;
std::map<uintptr_t, size_t> g_host_ranges;  // start -> bytes

}  // namespace

bool ezpz::host_range_registered(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lock(g_host_mu);
    auto it = g_host_ranges.upper_bound(a);
    if (it == g_host_ranges.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}

namespace {

// The staging buffer of the zero-copy path belongs to the calling thread (one per device), not to the system: a
// solve() on a new topology then does not pay a hipHostMalloc (~200 us) for its first launch, and threads never share
// one.  The thread's solve has synchronised its stream before it returns, so the buffer is free for its next call.
thread_local PinnedBuf t_pinned[16];

}  // namespace

extern "C" {

int ezpz_system_solve_batch(EzpzSystem* sys, const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out,
                            EzpzStatus* status, uint8_t* unsat_mask, uint64_t* warn_log, uint32_t warn_cap) {
    if (!sys) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    release_thread_kernel(sys->device);
    std::lock_guard<std::mutex> lock(sys->mu);
    EZPZ_ON_DEVICE(sys->device);
    // (the launch shape is chosen for the call, not for each of the pieces it may be fed to the device in)
    struct CallBatch {
        explicit CallBatch(size_t b) { t_call_batch = b; }
        ~CallBatch() { t_call_batch = 0; }
    } call_batch(batch);
    const size_t n = sys->counts.n_vars, C = sys->counts.n_cons;
    const bool want_log = warn_log && warn_cap;
    const size_t x_bytes = batch * std::max<size_t>(n, 1) * sizeof(double);
    const size_t st_bytes = batch * sizeof(EzpzStatus);
    const size_t mask_bytes = unsat_mask ? ((batch * std::max<size_t>(C, 1) + 15) & ~size_t(15)) : 0;
    const size_t log_bytes = want_log ? batch * (size_t)warn_cap * sizeof(uint64_t) : 0;
    int rc;
    call_stamp(CALL_LOCKED);
    if (x_bytes + st_bytes + mask_bytes <= sys->lim.policy.zero_copy_max_bytes) {
        // Small call (the solve() case): no DMA at all.  The kernel reads the guesses from, and writes the
        // results to, pinned host memory mapped into the device address space; one launch + one stream sync.
        const size_t total = x_bytes + st_bytes + mask_bytes + log_bytes;
        PinnedBuf& pinned = t_pinned[sys->device & 15];
        if ((rc = pinned.ensure(total)) != EZPZ_OK) return rc;
        unsigned char* h = pinned.p;
        double* hx = reinterpret_cast<double*>(h);
        EzpzStatus* hst = reinterpret_cast<EzpzStatus*>(h + x_bytes);
        uint8_t* hmask = h + x_bytes + st_bytes;
        uint64_t* hlog = reinterpret_cast<uint64_t*>(h + x_bytes + st_bytes + mask_bytes);
        if (n) std::memcpy(hx, x0, batch * n * sizeof(double));
        call_stamp(CALL_STAGED);
        // on the calling thread's own stream: solve() calls from different threads (on different systems) overlap on
        // the device instead of queueing behind each other on the null stream
        rc = ezpz_system_solve_batch_device(sys, hx, batch, cfg, hx, hst, unsat_mask ? hmask : nullptr,
                                            want_log ? hlog : nullptr, warn_cap, hipStreamPerThread);
        if (rc != EZPZ_OK) return rc;
        call_stamp(CALL_LAUNCHED);
        // a solve() call is over in tens of microseconds: poll the stream for a while before blocking on it (the
        // blocking wait sleeps on an interrupt and comes back ~10 us late)
        {
            const auto t0 = std::chrono::steady_clock::now();
            hipError_t q;
            while ((q = hipStreamQuery(hipStreamPerThread)) == hipErrorNotReady) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
            }
            if (q != hipSuccess) {
                (void)hipGetLastError();
                HIP_TRY(hipStreamSynchronize(hipStreamPerThread));
            }
        }
        call_stamp(CALL_COMPLETE);
        std::memcpy(status, hst, st_bytes);
        if (can_time_out(*sys))
            for (size_t b2 = 0; b2 < batch; ++b2)
                if (status[b2].iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT) return EZPZ_ERR_HIP;
        if (n) std::memcpy(x_out, hx, batch * n * sizeof(double));
        if (unsat_mask && C) std::memcpy(unsat_mask, hmask, batch * C);
        if (want_log) {
            // only the entries the kernel wrote are meaningful: n_warnings per system, capped
            for (size_t b = 0; b < batch; ++b) {
                size_t cnt = std::min<size_t>(hst[b].n_warnings, warn_cap);
                std::memcpy(warn_log + b * warn_cap, hlog + b * warn_cap, cnt * sizeof(uint64_t));
            }
        }
        call_stamp(CALL_UNPACKED);
        return EZPZ_OK;
    }
    // Registered (page-locked) caller buffers: the batch moves through a three-stage pipeline -- copies in, kernels, copies
    // out, one stream each, pieces of 8 MB through a ring of four device buffers -- so that the link carries guesses in and
    // results out at the same time.  What the link gives (tools/pcie_duplex.hip, profiles/r04_pcie_duplex.txt): 56 GB/s one
    // way alone; both ways at once 46-48 GB/s each when every direction is ONE queue of pieces of >= 8 MB, 39 with 2 MB
    // pieces, 34-40 with three queues per direction (round 3's shape: three streams each doing in / kernel / out in turn
    // with 2 MB pieces: 32 GB/s each way).  All kernels of the call run on one stream in order, so every launch shape may
    // use it (per-system device scratch is never shared by two kernels in flight).  Calls without mask / warning log.
    // (a system that runs lanes across the batch from a few systems on -- EZPZ_TEAM_BATCH_LANES -- has no piece size below
    // its threshold worth pipelining: it takes the chunked path below)
    const bool lanes_always = sys->lanes && sys->lanes_min <= std::min<size_t>(batch, 8);
    if (n && !lanes_always && !unsat_mask && !want_log && host_range_registered(x0, x_bytes) &&
        host_range_registered(x_out, x_bytes)) {
        const size_t row = n * sizeof(double);
        static const size_t piece_env = [] {  // (EZPZ_H2H_PIECE_MB: measurements)
            const char* e = std::getenv("EZPZ_H2H_PIECE_MB");
            return (size_t)(e && std::atoi(e) > 0 ? std::atoi(e) : 0) << 20;
        }();
        // Pieces of a sixteenth of the call, between 4 and 16 MB: filling and draining the pipeline costs one piece each
        // way, and the link moves 2 / 4 / 8 / 16 MB pieces at 33 / 39 / 42 / 43 GB/s each way (2000 x 2000, 16 384 systems).
        // Big systems at least 8 to a piece (a launch needs several of them to use the device).
        const EzpzLaunchPolicy& pol = sys->lim.policy;
        const size_t piece_bytes = piece_env ? piece_env
                                             : std::min<size_t>(pol.h2h_piece_max_bytes, std::max<size_t>(pol.h2h_piece_min_bytes, x_bytes / pol.h2h_pieces_per_call));
        size_t piece = std::max<size_t>(std::min<size_t>(batch, 8), std::min<size_t>(piece_bytes / row, (batch + 7) / 8));
        // (the lanes-across-the-batch kernel is for device-filling calls: the pieces stay below its threshold and run on the
        // teams, which resume nothing and keep their state in LDS)
        if (sys->lanes && piece >= sys->lanes_min) piece = std::max<size_t>(1, (size_t)sys->lanes_min - 1);
        EzpzSystem::Pipe& P = sys->pipe;
        constexpr int K = EzpzSystem::Pipe::kSlots;
        if (!P.in) {
            bool ok = hipStreamCreateWithFlags(&P.in, hipStreamNonBlocking) == hipSuccess &&
                      hipStreamCreateWithFlags(&P.run, hipStreamNonBlocking) == hipSuccess &&
                      hipStreamCreateWithFlags(&P.out, hipStreamNonBlocking) == hipSuccess;
            for (int k = 0; k < K && ok; ++k)
                ok = hipEventCreateWithFlags(&P.arrived[k], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&P.solved[k], hipEventDisableTiming) == hipSuccess &&
                     hipEventCreateWithFlags(&P.left[k], hipEventDisableTiming) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                return EZPZ_ERR_HIP;
            }
        }
        for (int k = 0; k < K; ++k)  // (the previous call drained its streams: nothing is using the buffers)
            if ((rc = P.x[k].ensure(piece * n)) != EZPZ_OK) return rc;
        // the statuses of the whole call collect in one device buffer; they follow each piece out when the caller's status
        // array is registered too (32 bytes per system: half of the traffic of an 8-variable system), else come back in one
        // copy at the end
        if ((rc = sys->st_dev.ensure(batch)) != EZPZ_OK) return rc;
        const bool st_registered = host_range_registered(status, st_bytes);
        // whatever happens after the first copy is enqueued, nothing returns while a copy may still be reading or
        // writing the caller's buffers
        auto drain = [&](int result) {
            for (hipStream_t st : {P.in, P.run, P.out}) {
                hipError_t q;
                while ((q = hipStreamQuery(st)) == hipErrorNotReady) __builtin_ia32_pause();
                if (q != hipSuccess) {
                    (void)hipGetLastError();
                    if (hipStreamSynchronize(st) != hipSuccess) (void)hipGetLastError();
                    if (result == EZPZ_OK) result = EZPZ_ERR_HIP;
                }
            }
            return result;
        };
        static const bool h2h_debug = debug_topic("h2h");
        // How the results leave: copies by the runtime (hipMemcpyAsync), or a copy kernel of ours into the registered buffer's
        // device address (EZPZ_H2H_OUT=dma / kernel).  Both directions as runtime copies is the fastest pair where the
        // runtime gives each direction an SDMA engine (46-48 GB/s each way, ROCm 7.2); the runtime PyTorch bundles (7.0)
        // moves the copies in with a shader when a kernel's results wait on another stream, and a shader in beside SDMA out
        // is the slowest pair there is (27-29 GB/s each way, tools/pcie_duplex.hip) -- a copy kernel out keeps 41-43 either way.
        static const int out_env = [] {
            const char* e = std::getenv("EZPZ_H2H_OUT");
            return !e ? 0 : e[0] == 'k' ? 1 : e[0] == 'd' ? 2 : 0;
        }();
        static const bool out_auto_kernel = [] {
            int v = 0;
            return hipRuntimeGetVersion(&v) == hipSuccess && v < 70200000;  // (HIP_VERSION: major * 10^7 + minor * 10^5 + patch)
        }();
        void *x_out_mapped = nullptr, *st_mapped = nullptr;
        bool out_by_kernel = (out_env == 1 || (out_env == 0 && out_auto_kernel)) &&
                             hipHostGetDevicePointer(&x_out_mapped, x_out, 0) == hipSuccess;
        if (out_by_kernel && st_registered && hipHostGetDevicePointer(&st_mapped, status, 0) != hipSuccess) out_by_kernel = false;
        if (!out_by_kernel) (void)hipGetLastError();
        const auto t_enq0 = std::chrono::steady_clock::now();
        size_t k = 0;
        for (size_t off = 0; off < batch; off += piece, ++k) {
            const int sl = (int)(k % K);
            const size_t nb = std::min(piece, batch - off);
            double* xd = P.x[sl].p;
            // The buffer is free again when the results of the piece that used it last have left.  The HOST waits for that:
            // it then never runs more than four pieces ahead of the device -- with a hundred pieces queued up front the
            // runtime's enqueue calls slow down tenfold and the streams' cross-dependencies halve the link's rate (126
            // pieces of 8 MB: 20 GB/s each way against 42 for 32 pieces).
            // (polled, not hipEventSynchronize: in a process whose runtime waits on interrupts -- torch sets the device up that
            // way -- every blocking wait wakes ~100 us late, a third of a piece's transfer: 2.7 -> 1.9 M solves/s)
            if (k >= (size_t)K) {
                hipError_t q;
                while ((q = hipEventQuery(P.left[sl])) == hipErrorNotReady) __builtin_ia32_pause();
                if (q != hipSuccess) {
                    (void)hipGetLastError();
                    return drain(EZPZ_ERR_HIP);
                }
            }
            if (hipMemcpyAsync(xd, x0 + off * n, nb * row, hipMemcpyHostToDevice, P.in) != hipSuccess ||
                hipEventRecord(P.arrived[sl], P.in) != hipSuccess || hipStreamWaitEvent(P.run, P.arrived[sl], 0) != hipSuccess)
                return drain(EZPZ_ERR_HIP);
            rc = ezpz_system_solve_batch_device(sys, xd, nb, cfg, xd, sys->st_dev.p + off, nullptr, nullptr, 0, P.run);
            if (rc != EZPZ_OK) return drain(rc);
            if (hipEventRecord(P.solved[sl], P.run) != hipSuccess || hipStreamWaitEvent(P.out, P.solved[sl], 0) != hipSuccess) return drain(EZPZ_ERR_HIP);
            if (out_by_kernel) {
                launch_copy_out(reinterpret_cast<char*>(x_out_mapped) + off * row, xd, nb * row, P.out);
                if (st_registered) launch_copy_out(reinterpret_cast<char*>(st_mapped) + off * sizeof(EzpzStatus), sys->st_dev.p + off, nb * sizeof(EzpzStatus), P.out);
                if (hipGetLastError() != hipSuccess) return drain(EZPZ_ERR_HIP);
            } else if (hipMemcpyAsync(x_out + off * n, xd, nb * row, hipMemcpyDeviceToHost, P.out) != hipSuccess ||
                       (st_registered && hipMemcpyAsync(status + off, sys->st_dev.p + off, nb * sizeof(EzpzStatus), hipMemcpyDeviceToHost, P.out) != hipSuccess)) {
                return drain(EZPZ_ERR_HIP);
            }
            if (hipEventRecord(P.left[sl], P.out) != hipSuccess) return drain(EZPZ_ERR_HIP);
        }
        const auto t_enq1 = std::chrono::steady_clock::now();
        if ((rc = drain(EZPZ_OK)) != EZPZ_OK) return rc;
        const auto t_enq2 = std::chrono::steady_clock::now();
        if (!st_registered) HIP_TRY(hipMemcpy(status, sys->st_dev.p, batch * sizeof(EzpzStatus), hipMemcpyDeviceToHost));
        if (h2h_debug)
            std::fprintf(stderr, "[ezpz h2h] %zu pieces of %zu systems: enqueue %.0f us, drain %.0f us, statuses %.0f us\n", k, piece,
                         std::chrono::duration<double, std::micro>(t_enq1 - t_enq0).count(),
                         std::chrono::duration<double, std::micro>(t_enq2 - t_enq1).count(),
                         std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enq2).count());
        if (can_time_out(*sys))
            for (size_t b2 = 0; b2 < batch; ++b2)
                if (status[b2].iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT) return EZPZ_ERR_HIP;
        return EZPZ_OK;
    }
    // Larger calls: DMA in pieces of <= 16 MB of guesses (pageable copies of that size run at ~43 GB/s on this
    // platform, 64 MB ones at ~20 GB/s), each piece H2D -> solve -> D2H through the same device buffers.  (Two sets of
    // buffers on two streams with hipMemcpyAsync were measured slower, 1.01 vs 1.39 M solves/s on the 2000x2000
    // system: copies from and to pageable memory do not overlap, they only add stream bookkeeping.)
    const size_t row_bytes = std::max<size_t>(n, 1) * sizeof(double);
    const size_t piece = std::max<size_t>(1, std::min<size_t>(batch, (16u << 20) / row_bytes));
    if ((rc = sys->x_dev.ensure(piece * std::max<size_t>(n, 1))) != EZPZ_OK) return rc;
    if ((rc = sys->xo_dev.ensure(piece * std::max<size_t>(n, 1))) != EZPZ_OK) return rc;
    if ((rc = sys->st_dev.ensure(piece)) != EZPZ_OK) return rc;
    if (unsat_mask && (rc = sys->mask_dev.ensure(piece * std::max<size_t>(C, 1))) != EZPZ_OK) return rc;
    if (want_log && (rc = sys->log_dev.ensure(piece * (size_t)warn_cap)) != EZPZ_OK) return rc;
    for (size_t off = 0; off < batch; off += piece) {
        const size_t nb = std::min(piece, batch - off);
        if (n) HIP_TRY(hipMemcpy(sys->x_dev.p, x0 + off * n, nb * n * sizeof(double), hipMemcpyHostToDevice));
        rc = ezpz_system_solve_batch_device(sys, sys->x_dev.p, nb, cfg, sys->xo_dev.p, sys->st_dev.p,
                                            unsat_mask ? sys->mask_dev.p : nullptr, want_log ? sys->log_dev.p : nullptr,
                                            warn_cap, nullptr);
        if (rc != EZPZ_OK) return rc;
        HIP_TRY(hipMemcpy(status + off, sys->st_dev.p, nb * sizeof(EzpzStatus), hipMemcpyDeviceToHost));
        if (can_time_out(*sys))
            for (size_t b2 = 0; b2 < nb; ++b2)
                if (status[off + b2].iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT) return EZPZ_ERR_HIP;
        if (n) HIP_TRY(hipMemcpy(x_out + off * n, sys->xo_dev.p, nb * n * sizeof(double), hipMemcpyDeviceToHost));
        if (unsat_mask && C) HIP_TRY(hipMemcpy(unsat_mask + off * C, sys->mask_dev.p, nb * C, hipMemcpyDeviceToHost));
        if (want_log) {
            // the log's capacity is sized for the worst case (every constraint warning in every sweep): bring back only
            // what each system wrote, or everything when that is small anyway
            const size_t bytes = nb * (size_t)warn_cap * sizeof(uint64_t);
            if (bytes <= (1u << 20)) {
                HIP_TRY(hipMemcpy(warn_log + off * warn_cap, sys->log_dev.p, bytes, hipMemcpyDeviceToHost));
            } else {
                for (size_t b2 = 0; b2 < nb; ++b2) {
                    const size_t cnt = std::min<size_t>(status[off + b2].n_warnings, warn_cap);
                    if (cnt)
                        HIP_TRY(hipMemcpy(warn_log + (off + b2) * warn_cap, sys->log_dev.p + b2 * warn_cap,
                                          cnt * sizeof(uint64_t), hipMemcpyDeviceToHost));
                }
            }
        }
    }
    return EZPZ_OK;
}


}  // extern "C"

// ---- one solve() call ------------------------------------------------------------------------------------------------------------
namespace {

// What one thread's one-call launches on one device go through (grow-only; a thread's call has seen its completion word
// before it returns, so the buffers are free for its next call).
struct CallBufs {
    // mapped host memory: [completion word, 64 B][the resident kernel's "gone" word, 64 B][status, 64 B][values out]
    // [values in, no BAR][short unsatisfied mask][short warning log]
    unsigned char* host = nullptr;
    size_t host_cap = 0;
    // fine-grained device memory the host stores into through the BAR: [request word, 64 B][guesses], else null
    unsigned char* bar_mem = nullptr;
    size_t bar_cap = 0;
    int bar = -1;  // -1 not asked yet, 0 no (the kernel reads the guesses from mapped host memory), 1 yes
    DevBuf<uint8_t> mask;
    DevBuf<uint64_t> log;
    DevBuf<unsigned int> counter;
    uint64_t seq = 0;
    // the resident kernel of this thread's last one-call launch, if it stayed (DoneWord::request)
    bool res_alive = false;
    EzpzSystem* res_sys = nullptr;
    uint64_t res_generation = 0;
    EzpzConfig res_cfg{};
    int res_stage = 0;         // which of the topology's kernels it is: 0 interpreting, 1 specialised, 2 one wavefront per system
    uint32_t res_warn_cap = 0;
    bool res_log = false;
    ~CallBufs() {
        if (res_alive && bar_mem) {  // (thread exit: the kernel is told to leave before its buffers go)
            std::atomic_thread_fence(std::memory_order_seq_cst);
            *reinterpret_cast<volatile uint64_t*>(bar_mem) = ~0ull;
            std::atomic_thread_fence(std::memory_order_seq_cst);
        }
        if (host) (void)hipHostFree(host);
        if (bar_mem) (void)hipFree(bar_mem);
    }
};
thread_local CallBufs t_call[16];
constexpr size_t kCallHeader = 192;

bool device_has_large_bar(int device) {
    static const bool allowed = [] {
        const char* e = std::getenv("EZPZ_BAR");  // EZPZ_BAR=0: stage the guesses in mapped host memory (A/B runs)
        return !(e && e[0] == '0');
    }();
    int v = 0;
    if (!allowed || hipDeviceGetAttribute(&v, hipDeviceAttributeIsLargeBar, device) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return v != 0;
}

// EZPZ_RESIDENT_US: how long a one-call kernel waits on the device for the calling thread's next request before it ends
// (0 = never resident; default 200).  A solve() loop -- the reference's benchmark protocol, an interactive drag -- keeps
// its kernel; anything that synchronises the whole device waits at most this long for it; no kernel stays longer than 50 ms.
unsigned resident_lease_us() {
    static const unsigned us = [] {
        const char* e = std::getenv("EZPZ_RESIDENT_US");
        return e ? (unsigned)std::max(0, std::atoi(e)) : 200u;
    }();
    return us;
}

void store_request(CallBufs& cb, uint64_t v) {  // through the BAR, after everything stored before it
    std::atomic_thread_fence(std::memory_order_seq_cst);
    *reinterpret_cast<volatile uint64_t*>(cb.bar_mem) = v;
    std::atomic_thread_fence(std::memory_order_seq_cst);
}

void dismiss_resident(CallBufs& cb) {  // "leave": the kernel ends within a poll; nothing waits for it (its stream runs in order)
    if (cb.res_alive && cb.bar_mem) store_request(cb, ~0ull);
    cb.res_alive = false;
    cb.res_sys = nullptr;
}

}  // namespace

int ezpz::system_solve_one(EzpzSystem* sys, const double* x0, const EzpzConfig* cfg, double* x_out, EzpzStatus* status,
                           uint8_t* unsat_mask, uint64_t* warn_log, uint32_t warn_cap) {
    if (!sys || !status) return EZPZ_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(sys->mu);
    EZPZ_ON_DEVICE(sys->device);
    call_stamp(CALL_LOCKED);
    const size_t n = sys->counts.n_vars, C = sys->counts.n_cons;
    if (n && (!x0 || !x_out)) return EZPZ_ERR_INVALID_ARGUMENT;
    CallBufs& cb = t_call[sys->device & 15];
    int rc;
    const size_t x_bytes = (std::max<size_t>(n, 1) * sizeof(double) + 63) & ~size_t(63);
    if (cb.bar < 0) cb.bar = device_has_large_bar(sys->device) ? 1 : 0;
    // A short unsatisfied mask / warning log is written straight to mapped host memory (a few byte / word stores across
    // the link); long ones stay on the device and are fetched when the status says there is something in them.
    const bool want_log = warn_log && warn_cap;
    const bool host_mask = C <= sys->lim.policy.one_call_host_mask_max_constraints,
               host_log = want_log && warn_cap <= sys->lim.policy.one_call_host_log_max_entries;
    const size_t mask_bytes = host_mask ? 256 : 0, log_bytes = host_log ? (size_t)warn_cap * sizeof(uint64_t) : 0;
    if (cb.host_cap < kCallHeader + 2 * x_bytes + mask_bytes + log_bytes) {
        dismiss_resident(cb);  // (it writes into the buffer that goes away)
        if (cb.host) (void)hipHostFree(cb.host);
        cb.host = nullptr;
        cb.host_cap = 0;
        const size_t want = std::max<size_t>(kCallHeader + 3 * x_bytes + 256 + 2 * log_bytes, 64 * 1024);
        HIP_TRY(hipHostMalloc((void**)&cb.host, want, hipHostMallocMapped));
        std::memset(cb.host, 0, kCallHeader);
        cb.host_cap = want;
        cb.seq = 0;
    }
    if (cb.bar == 1 && cb.bar_cap < 64 + x_bytes) {
        dismiss_resident(cb);
        if (cb.bar_mem) (void)hipFree(cb.bar_mem);
        cb.bar_mem = nullptr;
        cb.bar_cap = 0;
        const size_t want = std::max<size_t>(64 + x_bytes + x_bytes / 2, 64 * 1024);
        if (hipExtMallocWithFlags((void**)&cb.bar_mem, want, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            cb.bar_mem = nullptr;
            cb.bar = 0;  // the kernel reads the guesses from mapped host memory instead
        } else {
            cb.bar_cap = want;
        }
    }
    if (cb.counter.cap == 0) {
        if ((rc = cb.counter.ensure(16)) != EZPZ_OK) return rc;
        HIP_TRY(hipMemset(cb.counter.p, 0, 16 * sizeof(unsigned int)));
    }
    if (!host_mask && cb.mask.cap < C) {
        dismiss_resident(cb);
        if ((rc = cb.mask.ensure(C)) != EZPZ_OK) return rc;
    }
    if (want_log && !host_log && cb.log.cap < warn_cap) {
        dismiss_resident(cb);
        if ((rc = cb.log.ensure(warn_cap)) != EZPZ_OK) return rc;
    }
    volatile uint64_t* word = reinterpret_cast<volatile uint64_t*>(cb.host);
    volatile uint64_t* gone = reinterpret_cast<volatile uint64_t*>(cb.host + 64);
    EzpzStatus* hst = reinterpret_cast<EzpzStatus*>(cb.host + 128);
    double* hx_out = reinterpret_cast<double*>(cb.host + kCallHeader);
    double* hx_in = reinterpret_cast<double*>(cb.host + kCallHeader + x_bytes);
    uint8_t* hmask = cb.host + kCallHeader + 2 * x_bytes;
    uint64_t* hlog = reinterpret_cast<uint64_t*>(cb.host + kCallHeader + 2 * x_bytes + mask_bytes);
    double* x_in = cb.bar == 1 ? reinterpret_cast<double*>(cb.bar_mem + 64) : hx_in;
    EzpzConfig dcfg;
    if (!cfg) {
        ezpz_default_config(&dcfg);
        cfg = &dcfg;
    }
    // ---- the topology's kernel still on the device from this thread's previous call? ------------------------------------------
    const unsigned lease_us = cb.bar == 1 ? resident_lease_us() : 0;
    // (which of the topology's kernels a launch would take now: a resident one of an earlier stage makes room for it)
    auto kernel_stage = [&] {
        return sys->wave_jit && comp_jit_state(sys->wave_jit) == 2 ? 2 : sys->jit && comp_jit_state(sys->jit) == 2 ? 1 : 0;
    };
    const int stage_now = kernel_stage();
    bool resident = cb.res_alive && cb.res_sys == sys && lease_us && std::memcmp(&cb.res_cfg, cfg, sizeof(EzpzConfig)) == 0 &&
                    cb.res_stage == stage_now && cb.res_log == want_log && (!want_log || cb.res_warn_cap == warn_cap);
    if (cb.res_alive && !resident) dismiss_resident(cb);
    // (what launch() does for a topology solved again and again: its specialised kernels are asked for after so many solves)
    if (!jit_sync() && sys->launches.load(std::memory_order_relaxed) >= sys->lim.policy.jit_after_launches)
        for (CompJit* j : {sys->jit, sys->wave_jit})
            if (j && comp_jit_state(j) == 0) (void)comp_jit_request(j, false);
    if (n) std::memcpy(x_in, x0, n * sizeof(double));
    // the request's tag: the generation of the launch that is to serve it (a resident kernel of an earlier launch that
    // still polls the word leaves when it sees another generation) and a sequence number
    constexpr uint64_t kSeqMask = (1ull << 40) - 1;
    ++cb.seq;
    if (!resident) ++cb.res_generation;
    const uint64_t generation = cb.res_generation & 0xFFFFFFull;
    const uint64_t seq = (generation << 40) | (cb.seq & kSeqMask);
    call_stamp(CALL_STAGED);
    if (resident) {
        sys->launches.fetch_add(1, std::memory_order_relaxed);
        store_request(cb, seq);  // (the guesses above are write-combined stores through the BAR: drained first)
        call_stamp(CALL_LAUNCHED);
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t spins = 0;
        while (*word != seq) {
            if (*gone == generation) {  // the lease ran out between the calls: an ordinary launch serves this request
                resident = false;
                break;
            }
            __builtin_ia32_pause();
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                // (the kernel neither answered nor left: its stream says what happened)
                const hipError_t q = hipStreamQuery(hipStreamPerThread);
                if (q != hipErrorNotReady) {
                    (void)hipGetLastError();
                    resident = false;
                    if (q != hipSuccess) {
                        cb.res_alive = false;
                        return EZPZ_ERR_HIP;
                    }
                    break;
                }
            }
        }
        if (!resident) {
            cb.res_alive = false;
            cb.res_sys = nullptr;
        }
    }
    if (!resident) {
        // (a resident kernel that left between the calls: the request above carried ITS generation; this launch gets a new
        // one, and the request is stored again under it)
        uint64_t tag = seq;
        if ((cb.res_generation & 0xFFFFFFull) == generation && *gone == generation) {
            ++cb.res_generation;
            tag = ((cb.res_generation & 0xFFFFFFull) << 40) | (cb.seq & kSeqMask);
        }
        // (the stores above are write-combined when they go through the BAR: drained before the doorbell write of the launch)
        DoneWord done{const_cast<unsigned long long*>(reinterpret_cast<volatile unsigned long long*>(word)), tag, cb.counter.p};
        if (lease_us) {
            *gone = 0;
            store_request(cb, tag);  // (the request word reads this launch's own tag when the kernel first polls it: nothing new yet)
            done.request = reinterpret_cast<const unsigned long long*>(cb.bar_mem);
            done.gone = const_cast<unsigned long long*>(reinterpret_cast<volatile unsigned long long*>(gone));
            done.generation = cb.res_generation & 0xFFFFFFull;
            done.lease_ticks = lease_us * 100u;
            done.life_ticks = 50000u * 100u;
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
        bool stays = false;
        rc = solve_batch_device_impl(sys, x_in, 1, cfg, hx_out, hst, host_mask ? hmask : cb.mask.p,
                                     !want_log ? nullptr : host_log ? hlog : cb.log.p, warn_cap, hipStreamPerThread, done, &stays);
        if (rc != EZPZ_OK) return rc;
        if (stays) {
            cb.res_alive = true;
            cb.res_sys = sys;
            cb.res_cfg = *cfg;
            cb.res_stage = stage_now;  // (a kernel that became ready during the launch is noticed by the next call)
            cb.res_log = want_log;
            cb.res_warn_cap = warn_cap;
        }
        call_stamp(CALL_LAUNCHED);
        // The completion word first; a launch that never writes it (a shape without the epilogue, a failed kernel) is caught by
        // the stream's own state, asked every few microseconds once the word is overdue.
        const auto t0 = std::chrono::steady_clock::now();
        auto next_query = t0 + std::chrono::microseconds(100);
        uint32_t spins = 0;
        while (*word != tag) {
            __builtin_ia32_pause();
            if ((++spins & 63u) != 0) continue;
            const auto now = std::chrono::steady_clock::now();
            if (now < next_query) continue;
            const hipError_t q = hipStreamQuery(hipStreamPerThread);
            if (q == hipSuccess) break;  // the stream is idle: the launch is over, word or no word
            if (q != hipErrorNotReady) {
                (void)hipGetLastError();
                cb.res_alive = false;
                return EZPZ_ERR_HIP;
            }
            next_query = now + std::chrono::microseconds(now - t0 > std::chrono::milliseconds(2) ? 200 : 5);
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    call_stamp(CALL_COMPLETE);
    *status = *hst;
    if (status->iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT && can_time_out(*sys)) return EZPZ_ERR_HIP;
    if (n) std::memcpy(x_out, hx_out, n * sizeof(double));
    const bool fetch_mask = unsat_mask && C && status->n_unsatisfied > 0;
    const size_t n_log = want_log ? std::min<size_t>(status->n_warnings, warn_cap) : 0;
    if (fetch_mask && host_mask) std::memcpy(unsat_mask, hmask, C);
    if (n_log && host_log) std::memcpy(warn_log, hlog, n_log * sizeof(uint64_t));
    if ((fetch_mask && !host_mask) || (n_log && !host_log)) {
        // (copies on the thread's stream would queue behind a resident kernel: it leaves first)
        dismiss_resident(cb);
        if (fetch_mask && !host_mask) HIP_TRY(hipMemcpyAsync(unsat_mask, cb.mask.p, C, hipMemcpyDeviceToHost, hipStreamPerThread));
        if (n_log && !host_log)
            HIP_TRY(hipMemcpyAsync(warn_log, cb.log.p, n_log * sizeof(uint64_t), hipMemcpyDeviceToHost, hipStreamPerThread));
        HIP_TRY(hipStreamSynchronize(hipStreamPerThread));
    }
    call_stamp(CALL_UNPACKED);
    return EZPZ_OK;
}

// Anything else the calling thread is about to enqueue on this device -- a batch on its per-thread stream, copies on the
// null stream -- would queue behind its resident kernel until the lease runs out: the kernel is told to leave first.
void ezpz::release_thread_kernel(int device) {
    if (device < 0) return;
    CallBufs& cb = t_call[device & 15];
    if (cb.res_alive) dismiss_resident(cb);
}

// A system that goes away takes its resident kernel along: the calling thread's is told to leave (another thread's runs out
// of its lease; hipFree waits for the device either way).
void ezpz::dismiss_resident_of(EzpzSystem* sys) {
    if (sys->device < 0) return;
    CallBufs& cb = t_call[sys->device & 15];
    if (cb.res_alive && cb.res_sys == sys) dismiss_resident(cb);
}


extern "C" {

int ezpz_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return EZPZ_ERR_INVALID_ARGUMENT;
    if (ezpz_device_count() < 1) return EZPZ_ERR_NO_DEVICE;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped) != hipSuccess) {
        (void)hipGetLastError();
        return EZPZ_ERR_HIP;
    }
    std::lock_guard<std::mutex> lock(g_host_mu);
    g_host_ranges[reinterpret_cast<uintptr_t>(p)] = bytes;
    return EZPZ_OK;
}

int ezpz_host_unregister(void* p) {
    {
        std::lock_guard<std::mutex> lock(g_host_mu);
        auto it = g_host_ranges.find(reinterpret_cast<uintptr_t>(p));
        if (it == g_host_ranges.end()) return EZPZ_ERR_INVALID_ARGUMENT;
        g_host_ranges.erase(it);
    }
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return EZPZ_ERR_HIP;
    }
    return EZPZ_OK;
}


}  // extern "C"
