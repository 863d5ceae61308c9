// Internal to the library (never installed): one analysed topology resident on a device -- EzpzSystem -- and what the
// translation units that work on it share.  The C ABI of include/ezpz_amd.h is implemented in
//   api.hip       system lifetime (create / destroy / info / analyze), small entry points, kernel specialisation
//   shape.cpp     launch-shape selection: analyze_into (Model::new's counterpart: which kernel family, team size, LDS plan)
//   records.cpp   the topology program as device data: pack_program, grid slices, the record walk's rounds, dense phases
//   launch.hip    kernel dispatch: launch(), the device-pointer entry point, the evaluation-only kernel
//   pipeline.cpp  host-pointer entry points: the one-call path (resident kernels), zero-copy, the three-stage pipeline
//   freedom.hip   FreedomAnalysis
//   mixed.hip, multi.cpp, solve.cpp  on top of the C ABI (heterogeneous batches, several devices, solve / solve_inner)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "call_trace.hpp"
#include "comp_program.hpp"
#include "fronts.hpp"
#include "kinds.hpp"
#include "launch_types.hpp"
#include "one_call.hpp"
#include "policy.hpp"
#include "program.hpp"

namespace ezpz {


// What the launch shapes are sized for.  Queried from the device the system is created on (a partitioned MI355X --
// CPX / DPX -- or a CU-masked process sees fewer CUs than the full chip's 256); the host-only analysis
// (ezpz_analyze, no device) assumes the full MI355X.
struct DeviceLimits {
    int cus = 256;                  // compute units
    size_t lds_bytes = 160 * 1024;  // LDS one workgroup may allocate (MI355X: 160 KiB per CU)
    EzpzLaunchPolicy policy = launch_policy_for(256);  // the thresholds of policy.hpp at this CU count
};
inline const DeviceLimits& device_limits(int device) {
    static const DeviceLimits full_chip;
    static std::mutex mu;
    static DeviceLimits cache[16];
    static bool have[16] = {};
    if (device < 0 || device >= 16) return full_chip;
    std::lock_guard<std::mutex> lock(mu);
    if (!have[device]) {
        DeviceLimits d;
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) d.cus = v;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
            d.lds_bytes = (size_t)v;
        d.policy = launch_policy_for(d.cus);
        (void)hipGetLastError();
        cache[device] = d;
        have[device] = true;
    }
    return cache[device];
}

// (EZPZ_DEBUG=hip: the failing call and the runtime's message on stderr)
inline bool hip_debug() {
    static const bool on = debug_topic("hip");
    return on;
}
#define HIP_TRY(expr)                                                                                             \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) {                                                                                   \
            if (hip_debug()) std::fprintf(stderr, "[ezpz hip] %s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            (void)hipGetLastError();                                                                              \
            return EZPZ_ERR_HIP;                                                                                  \
        }                                                                                                         \
    } while (0)

// Makes `device` the calling thread's current HIP device for a scope and puts the caller's own back at its end: a host entry
// point must not leave a torch caller on an 8-GPU node on another device than it was on (every entry that touches the runtime
// on behalf of a system opens one).
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
        if (prev == device) {
            prev = -1;  // nothing to put back
        } else if (hipSetDevice(device) != hipSuccess) {
            (void)hipGetLastError();
            ok = false;
        }
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define EZPZ_ON_DEVICE(device)              \
    ezpz::DeviceGuard device_guard_(device); \
    if (!device_guard_.ok) return EZPZ_ERR_HIP

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    int ensure(size_t count) {
        if (count <= cap) return EZPZ_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(count, 16);
        if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) {
            (void)hipGetLastError();
            return EZPZ_ERR_HIP;
        }
        cap = want;
        return EZPZ_OK;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

// Grow-only pinned, device-mapped host buffer (zero-copy path of small solves).
struct PinnedBuf {
    unsigned char* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return EZPZ_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max<size_t>(bytes + bytes / 2, 64 * 1024);
        if (hipHostMalloc((void**)&p, want, hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            return EZPZ_ERR_HIP;
        }
        cap = want;
        return EZPZ_OK;
    }
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
};

}  // namespace ezpz

// (the opaque type of the C ABI lives in the global namespace; this header is internal to the library)
using namespace ezpz;

struct EzpzSystem {
    int device = -1;      // -1: host-only analysis
    DeviceLimits lim;     // of `device`
    ProgramCounts counts;
    EzpzSystemInfo info{};
    void* dev_program = nullptr;  // single allocation holding every list
    ProgramView view{};
    uint32_t team_size = 0;
    int mode = MODE_SUB;  // TeamMode
    bool lds_ws = true;
    bool prog_in_lds = false;
    bool unit_weights = true;
    bool linear_only = false;  // every constraint is of a linear kind: the evaluators are built without the others
    // component-resident launch shape (comp_program.hpp): the plan and its device copy; when present, solves run on
    // comp_solve_kernel and the list-walk program above serves only evaluation / FreedomAnalysis
    std::unique_ptr<CompPlan> comp;
    uint32_t* dev_comp = nullptr;
    // A latency-shaped block system (ezpz_solve) is ready to solve as soon as its component plan is: the list-walk program of
    // the whole system -- which then serves only evaluation, FreedomAnalysis and the sizes of EzpzSystemInfo -- is built
    // and uploaded when one of those asks for it (ensure_program): 2000 x 2000, a request the process has not seen, 513 -> ~250 us.
    std::atomic<bool> program_deferred{false};
    std::vector<EzpzConstraint> deferred_cs;
    std::mutex defer_mu;
    CompJit* jit = nullptr;  // the plan's class-specialised kernel (run-time compiled), when it has one
    DevBuf<unsigned char> jit_scratch;  // ... and, when it spreads a system over several workgroups, their reduction scratch
    DevBuf<unsigned int> jit_redo[2];   // ... and the lists of systems its `_fast` entry leaves to the loop (count, then systems):
                                        // calls alternate between the two, each zeroes the other's count (launch.hip)
    unsigned int jit_redo_turn = 0;
    unsigned int* jit_redo_seen = nullptr;      // mapped host memory: the count the loop's last launch found (a hint, jit.cpp)
    unsigned int* jit_redo_seen_dev = nullptr;  // ... its device address
    std::unique_ptr<LanePlan> lane;  // small systems: one lane per system, run-time compiled (jit stands for it then)
    CompJit* wave_jit = nullptr;     // ... and, for the latency of one solve, the same class on one wavefront per system
    // connected sketches in large batches: one lane per system, uniform program, state in global memory (batch_kernel.hip.hpp)
    std::unique_ptr<BatchPlan> lanes;
    uint32_t* dev_lanes = nullptr;
    DevBuf<double> lanes_ws;
    DevBuf<uint32_t> strag_list, strag_count;  // the systems a lanes launch hands over to the teams (device-side list + count)
    DevBuf<LmResume> strag_state;              // ... and the LM state each had reached
    uint64_t lanes_ws_waves = 0;
    // the specialised kernels' work counters (JitArgs::ticket): never reset -- what a launch draws from each follows from its batch and
    // its workgroups, the host keeps the running totals -- so launches that use them must not overlap: on one stream they do not
    // anyway; when the stream CHANGES, an event recorded on the old one is waited for on the new one (under launch_mu)
    DevBuf<unsigned int> ticket;
    unsigned int ticket_base[8] = {};
    hipEvent_t ticket_done = nullptr;
    hipStream_t ticket_stream = nullptr;  // the stream of the last launch that drew from the counters
    bool ticket_used = false;
    hipEvent_t lanes_done = nullptr;  // completion of this system's last launch that used its global-memory workspace (lanes
                                      // kernel, list walk with the workspace in global memory): the next one, on any stream, waits for it
    uint64_t lanes_min = ~0ull;  // systems per call from which `lanes` serves the call
    std::atomic<uint32_t> launches{0};  // a topology solved again and again (an interactive sketch) earns its specialised kernel
    // frontal launch shape (fronts.hpp, front_kernel.hip.hpp): the plan, its device copy, the scratch of systems that several
    // workgroups share, and how many of the kernel's workgroups the device holds at once
    std::unique_ptr<FrontPlan> fronts;
    void* dev_fronts = nullptr;
    DevBuf<unsigned char> front_scratch;
    uint64_t front_capacity = 0;
    uint64_t front_max_batch = 0;  // calls of up to this many systems take the fronts (~0: every call; shape.cpp)
    uint32_t grid_wgs = 1;     // grid team: workgroups that share one system (each keeps its share of the state in LDS)
    uint32_t grid_ws_doubles = 0;
    DevBuf<GridScratch> grid_scratch;
    std::vector<unsigned char> grid_blob;       // the workgroups' sub-programs, one after the other
    std::vector<ProgramView> host_grid_views;   // per workgroup; blob_bytes = offset of its slice in grid_blob
    size_t grid_stage_bytes = 0;
    uint64_t grid_capacity = 0;  // workgroups of the grid build the device holds at once (0 = not asked yet)
    // The scratch areas through which the workgroups of one system talk inside a launch carry 32-bit sequence numbers that go on from
    // launch to launch (grid_ops.hip.hpp, jit_kernel.hip.hpp, front_kernel.hip.hpp).  An upper bound of what the launches so far
    // have used of them; past kSeqBudget the launch code zeroes the area first (ordered behind the last launch that used it), and
    // the numbers start again -- a process that solves one such system for days never meets the wrap.
    uint64_t grid_seq_used = 0, jit_seq_used = 0, front_seq_used = 0;
    void* dev_grid_blob = nullptr;
    DevBuf<ProgramView> grid_views;
    uint32_t prog_lds_doubles = 0;
    uint32_t lvl_lds_off = 0, lvl_tab_words = 0, lvl_buf_words = 0;  // level staging of the Cholesky lists (finish_team)
    uint32_t lvl_nlev = 0;
    uint32_t n_dense = 0, dense_level0 = 0, dense_lds_off = 0, dense_lds_doubles = 0;  // dense phases (make_dense_phases)
    bool lean_lds = false;  // batch-throughput workgroup: keep LDS per workgroup small (no whole-list staging)
    // record walk (build_records): the linear solve of one connected system on a barrier workgroup as rounds of per-lane
    // records; rec_extra = doubles behind the workspace proper (the factor's diagonal, one zero), offsets into the blob
    bool rec = false, rec_wide = false, rec_jglobal = false;
    uint32_t rec_extra = 0, rec_rounds = 0, rec_desc_lds_off = 0;
    size_t rec_desc_off = 0, rec_chunks_off = 0, rec_asm_cols_off = 0, rec_asm_slots_off = 0;
    uint32_t rec_asm_kc = 0, rec_asm_ks = 0;
    uint32_t ws_doubles = 0;
    uint32_t block_threads = 256;
    size_t lds_bytes = 0;
    std::mutex launch_mu;  // launch(): lazily created per-system state
    // grow-only scratch for the host-pointer entry points
    std::mutex mu;
    DevBuf<double> x_dev;
    DevBuf<double> xo_dev;  // ... and their results (pageable host buffers: out of place, so that a linear block system's call may take
                            // the kernel that stores its values before the LM control's verdicts are in, jit_kernel.hip.hpp)
    DevBuf<EzpzStatus> st_dev;
    DevBuf<uint8_t> mask_dev;
    DevBuf<uint64_t> log_dev;
    DevBuf<double> gws_dev;
    // the pipelined host-to-host path (registered caller buffers): one stream per stage -- copies in, kernels, copies
    // out -- and a ring of device buffers, each with an event per stage
    struct Pipe {
        static constexpr int kSlots = 4;
        hipStream_t in = nullptr, run = nullptr, out = nullptr;
        DevBuf<double> x[kSlots];
        hipEvent_t arrived[kSlots] = {}, solved[kSlots] = {}, left[kSlots] = {};
        ~Pipe() {
            for (hipStream_t st : {in, run, out})
                if (st) (void)hipStreamDestroy(st);
            for (int k = 0; k < kSlots; ++k)
                for (hipEvent_t e : {arrived[k], solved[k], left[k]})
                    if (e) (void)hipEventDestroy(e);
        }
    } pipe;
    std::vector<uint32_t> host_var_of, host_row_of, host_slot_row, host_slot_col;  // internal -> caller numbering
    // FreedomAnalysis program (built on first use) and its scratch
    struct Freedom {
        bool built = false;
        bool lane = false;
        uint32_t ncomp = 0, ws = 0, max_n = 0, group = 1, threads = 64;
        DevBuf<FreedomComp> comps;
        DevBuf<uint32_t> lists;  // items | comp_vars | col_ptr | col_slots
        uint32_t o_vars = 0, o_col_ptr = 0, o_col_slots = 0;
        DevBuf<double> x_in, x_int, jv, part, gws, step_tau, probe, probe_w;
        DevBuf<uint32_t> step_done;
        FreedomComp comp0{};  // host copy of the largest component (the wide QR path factorises it over the whole device)
        uint32_t big = 0;     // ... and its index
        DevBuf<uint8_t> mask;
        DevBuf<uint32_t> count;
    } freedom;
    ~EzpzSystem() {
        if (dev_program) (void)hipFree(dev_program);
        if (dev_grid_blob) (void)hipFree(dev_grid_blob);
        if (dev_comp) (void)hipFree(dev_comp);
        if (dev_lanes) (void)hipFree(dev_lanes);
        if (dev_fronts) (void)hipFree(dev_fronts);
        if (lanes_done) (void)hipEventDestroy(lanes_done);
        if (ticket_done) (void)hipEventDestroy(ticket_done);
        if (jit_redo_seen) (void)hipHostFree(jit_redo_seen);
        comp_jit_destroy(jit);
        comp_jit_destroy(wave_jit);
    }
};

namespace ezpz {

// A system spread over several workgroups that wait for each other (a grid team, a multi-workgroup specialised kernel, fronts across
// workgroups): its rendezvous can time out, and the status then says so (EZPZ_ITERATIONS_TEAM_TIMEOUT) -- every host entry turns
// that into EZPZ_ERR_HIP.
constexpr uint64_t kSeqBudget = 1ull << 31;
// `used` += what a launch of `batch` systems with `per_system` exchanges each may add; true: the area is to be zeroed first.
// (EZPZ_SEQ_BUDGET: a smaller budget, for the test that wants to see the numbers start again)
inline bool seq_budget_spent(uint64_t& used, uint64_t batch, uint64_t per_system) {
    static const uint64_t budget = [] {
        const char* e = std::getenv("EZPZ_SEQ_BUDGET");
        const long long v = e ? std::atoll(e) : 0;
        return v > 0 ? (uint64_t)v : kSeqBudget;
    }();
    const uint64_t add = batch * per_system;
    if (used + add > budget) {
        used = add;
        return true;
    }
    used += add;
    return false;
}
inline bool can_time_out(const EzpzSystem& s) {
    return s.grid_wgs > 1 || (s.comp && s.comp->jit_wgs > 1) || (s.fronts && s.fronts->n_wgs > 1);
}

inline uint32_t pow2_ceil(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

constexpr size_t kProgLdsMax = 24 * 1024;  // sub-wavefront teams: stage the whole program into LDS when it is this small

inline uint32_t workspace_doubles(const ProgramCounts& c) {
    const uint64_t doubles = 3ull * c.n_vars + 2ull * c.n_rows + c.zj + c.zlo + 2;
    return (uint32_t)((doubles + 1) & ~1ull);
}

// The list walk's part of a record-walk workspace: without the Jacobian's values when those live in global memory.
inline uint32_t rec_ws_base(const ProgramCounts& c, bool jglobal) {
    const uint64_t doubles = 3ull * c.n_vars + 2ull * c.n_rows + (jglobal ? 0u : c.zj) + c.zlo + 2;
    return (uint32_t)((doubles + 1) & ~1ull);
}

inline bool jit_sync() {
    static const bool sync = [] {
        const char* e = std::getenv("EZPZ_JIT");
        return e && std::strcmp(e, "sync") == 0;
    }();
    return sync;
}

template <class T>
size_t append(std::vector<unsigned char>& blob, const std::vector<T>& v) {
    size_t off = (blob.size() + 15) & ~size_t(15);
    blob.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
    if (!v.empty()) std::memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}


// ---- shape.cpp ------------------------------------------------------------------------------------------------------------
// `may_defer`: a latency shape whose component plan is interpretable returns with that plan alone (EzpzSystem::program_deferred);
// `keep_comp`: the system already has its component plan (ensure_program: the deferred rest).
int analyze_into(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint32_t team_size, EzpzSystem& s, Program& P,
                 std::vector<unsigned char>& blob, int32_t* err_constraint, int64_t* err_variable, bool may_defer = false,
                 bool keep_comp = false);

// ---- records.cpp ----------------------------------------------------------------------------------------------------------
struct RecPlan {
    std::vector<uint32_t> desc, chunks;
    uint32_t rounds = 0;
    // packed assembly (SolveArgs::rec_asm_*): chunks per column / per entry of the strict lower part (0: none, the lists are walked)
    std::vector<uint32_t> asm_cols, asm_slots;
    uint32_t asm_kc = 0, asm_ks = 0;
};
extern const uint32_t kRecMaxComponents;
size_t pack_program(const Program& P, bool idx16, bool pack_table, std::vector<unsigned char>& blob, ProgramView& v);
bool pack_grid_slices(EzpzSystem& s, const Program& P, uint32_t G, uint32_t W);
bool build_records(const Program& P, uint32_t T, uint32_t lds_base, bool wide, bool jglobal, RecPlan& out);
void choose_level_groups(Program& P, const EzpzSystem& s);
bool make_dense_phases(Program& P, uint32_t n_waves, size_t lds_room_bytes);

// ---- launch.hip -------------------------------------------------------------------------------------------------------------
extern unsigned long long* g_stamps;  // diagnostic builds only (tools/stamps.py sets it through ezpz_debug_set_stamps)
// (`done`: the completion word of a one-call launch, null for every other caller; `resident`: whether the launch stays on the
// device for further requests, DoneWord::request)
int solve_batch_device_impl(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg, double* x_out_dev,
                            EzpzStatus* status_dev, uint8_t* unsat_mask_dev, uint64_t* warn_log_dev, uint32_t warn_cap, void* stream,
                            const DoneWord& done, bool* resident = nullptr);

// The whole call's systems while a host entry of this thread feeds them to launch() in pieces (pipeline.cpp): the launch shape is
// chosen once per call, not per piece -- a short last piece does not change shape (0: the piece is the call).
extern thread_local uint64_t t_call_batch;
int front_launch(EzpzSystem& s, SolveArgs& args, hipStream_t stream);
int front_launch_probe(EzpzSystem& s, const double* x_dev, size_t batch, double* y_dev, uint32_t m, hipStream_t stream,
                       const double* w_dev = nullptr, double lambda_scale = 1e-11);  // front.hip  // front.hip: the frontal shape (EzpzSystem::fronts)

void launch_eval(EzpzSystem* sys, const double* x_int_dev, size_t batch, double* r_out_dev, double* jv_out_dev, uint32_t* deg_out_dev,
                 uint32_t grid, hipStream_t stream);  // the evaluation-only kernel (values in internal numbering)

int lane_indexed_launch(EzpzSystem* sys, const double* x_ragged, const uint64_t* row_offset_dev, const uint32_t* sys_of_dev, uint64_t count,
                        const EzpzConfig* cfg, double* x_out_ragged, EzpzStatus* status_all, void* stream);  // launch.hip
void launch_copy_out(void* dst_host_as_device, const void* src_dev, size_t bytes, void* stream);  // a copy kernel into mapped host memory

// ---- api.hip ----------------------------------------------------------------------------------------------------------------
int ensure_program(EzpzSystem* sys);  // the rest of a deferred analysis (EzpzSystem::program_deferred)
// ezpz_system_eval_batch for a caller that holds sys->mu already (launch.hip; FreedomAnalysis by probes checks its null vectors
// against the Jacobian)
int eval_batch_locked(EzpzSystem* sys, const double* x, size_t batch, double* r_out, double* jv_out, uint32_t* degenerate_count_out);

// ---- pipeline.cpp -----------------------------------------------------------------------------------------------------------
bool host_range_registered(const void* p, size_t bytes);  // inside a range the caller registered (ezpz_host_register)
void dismiss_resident_of(EzpzSystem* sys);                 // a system that goes away takes its resident kernel along

}  // namespace ezpz
