// C ABI implementation (include/ezpz_amd.h), device side: system lifetime (Model::new, reference
// ezpz/src/solver.rs:192-300, as a cached topology program), launch-shape selection and kernel dispatch for the LM
// solve (newton.rs:29-145), the evaluation-only kernel, and FreedomAnalysis (solver/find_dof.rs).  The host
// orchestration above it (solve, solve_inner, priority tiers, lint) is in solve.cpp.
// All numeric work runs in kernels on the GPU; there is no CPU solver in this library.
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
#include "freedom.hip.hpp"
#include "kinds.hpp"
#include "lm_kernel.hip.hpp"
#include "one_call.hpp"
#include "policy.hpp"
#include "program.hpp"

using namespace ezpz;

namespace {

// What the launch shapes are sized for.  Queried from the device the system is created on (a partitioned MI355X --
// CPX / DPX -- or a CU-masked process sees fewer CUs than the full chip's 256); the host-only analysis
// (ezpz_analyze, no device) assumes the full MI355X.
struct DeviceLimits {
    int cus = 256;                  // compute units
    size_t lds_bytes = 160 * 1024;  // LDS one workgroup may allocate (MI355X: 160 KiB per CU)
    EzpzLaunchPolicy policy = launch_policy_for(256);  // the thresholds of policy.hpp at this CU count
};
const DeviceLimits& device_limits(int device) {
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

// (EZPZ_HIP_DEBUG=1: the failing call and the runtime's message on stderr)
inline bool hip_debug() {
    static const bool on = std::getenv("EZPZ_HIP_DEBUG") != nullptr;
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

// Host ranges the caller has registered (ezpz_host_register): page-locked, so batch calls can DMA straight from / to
// them with asynchronous copies that overlap the kernels.
std::mutex g_host_mu;
std::map<uintptr_t, size_t> g_host_ranges;  // start -> bytes
bool host_range_registered(const void* p, size_t bytes) {
    if (!p || !bytes) return false;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lock(g_host_mu);
    auto it = g_host_ranges.upper_bound(a);
    if (it == g_host_ranges.begin()) return false;
    --it;
    return a >= it->first && a + bytes <= it->first + it->second;
}


// The staging buffer of the zero-copy path belongs to the calling thread (one per device), not to the system: a
// solve() on a new topology then does not pay a hipHostMalloc (~200 us) for its first launch, and threads never share
// one.  The thread's solve has synchronised its stream before it returns, so the buffer is free for its next call.
thread_local PinnedBuf t_pinned[16];

}  // namespace

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
    std::unique_ptr<LanePlan> lane;  // small systems: one lane per system, run-time compiled (jit stands for it then)
    CompJit* wave_jit = nullptr;     // ... and, for the latency of one solve, the same class on one wavefront per system
    // connected sketches in large batches: one lane per system, uniform program, state in global memory (batch_kernel.hip.hpp)
    std::unique_ptr<BatchPlan> lanes;
    uint32_t* dev_lanes = nullptr;
    DevBuf<double> lanes_ws;
    DevBuf<uint32_t> strag_list, strag_count;  // the systems a lanes launch hands over to the teams (device-side list + count)
    DevBuf<LmResume> strag_state;              // ... and the LM state each had reached
    uint64_t lanes_ws_waves = 0;
    hipEvent_t lanes_done = nullptr;  // completion of this system's last launch that used its global-memory workspace (lanes
                                      // kernel, list walk with the workspace in global memory): the next one, on any stream, waits for it
    uint64_t lanes_min = ~0ull;  // systems per call from which `lanes` serves the call
    std::atomic<uint32_t> launches{0};  // a topology solved again and again (an interactive sketch) earns its specialised kernel
    uint32_t grid_wgs = 1;     // grid team: workgroups that share one system (each keeps its share of the state in LDS)
    uint32_t grid_ws_doubles = 0;
    DevBuf<GridScratch> grid_scratch;
    std::vector<unsigned char> grid_blob;       // the workgroups' sub-programs, one after the other
    std::vector<ProgramView> host_grid_views;   // per workgroup; blob_bytes = offset of its slice in grid_blob
    size_t grid_stage_bytes = 0;
    uint64_t grid_capacity = 0;  // workgroups of the grid build the device holds at once (0 = not asked yet)
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
        DevBuf<double> x_int, jv, part, gws, step_tau;
        DevBuf<uint32_t> step_done;
        FreedomComp comp0{};  // host copy of the first component (the wide QR path runs on one-component systems)
        DevBuf<uint8_t> mask;
        DevBuf<uint32_t> count;
    } freedom;
    ~EzpzSystem() {
        if (dev_program) (void)hipFree(dev_program);
        if (dev_grid_blob) (void)hipFree(dev_grid_blob);
        if (dev_comp) (void)hipFree(dev_comp);
        if (dev_lanes) (void)hipFree(dev_lanes);
        if (lanes_done) (void)hipEventDestroy(lanes_done);
        comp_jit_destroy(jit);
        comp_jit_destroy(wave_jit);
    }
};

namespace {

uint32_t pow2_ceil(uint32_t v) {
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

constexpr size_t kProgLdsMax = 24 * 1024;  // sub-wavefront teams: stage the whole program into LDS when it is this small

uint32_t workspace_doubles(const ProgramCounts& c) {
    const uint64_t doubles = 3ull * c.n_vars + 2ull * c.n_rows + c.zj + c.zlo + 2;
    return (uint32_t)((doubles + 1) & ~1ull);
}

// The list walk's part of a record-walk workspace: without the Jacobian's values when those live in global memory.
uint32_t rec_ws_base(const ProgramCounts& c, bool jglobal) {
    const uint64_t doubles = 3ull * c.n_vars + 2ull * c.n_rows + (jglobal ? 0u : c.zj) + c.zlo + 2;
    return (uint32_t)((doubles + 1) & ~1ull);
}

// Sub-wavefront team for small systems: lanes per system.
// Lanes per system for sub-wavefront teams.  Every system of a batch runs the same program, so with few lanes per
// system the constraints a wavefront evaluates in one round are of few kinds (less divergence) while each lane's
// serial share of a phase grows.  Measured on 65 536-system batches, solves/s by lanes per system 1/2/4/8/16:
//   arc_radius     (cost  5) 1.16/1.58/1.41/0.85/-    G      circle_tangent (cost  9) 0.85/1.00/0.91/0.57/0.32 G
//   parallelogram  (cost 15) 182/224/278/237/-        M      square         (cost 20) 86/112/140/123/71        M
//   two_rectangles (cost 24) -/432/545/584/363        M
// with cost = sum over constraints of 1 (linear kinds), 3 (hypot kinds) or 4 (angle / arc kinds): the best team is
// the power of two nearest to cost / 4, never below 2.
uint32_t auto_sub_team(const EzpzConstraint* cs, size_t n_cs) {
    uint32_t cost = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        const uint32_t k = cs[i].kind;
        const bool angle = k == EZPZ_LINES_AT_ANGLE || k == EZPZ_ARC_ANGLE || k == EZPZ_POINTS_AT_ANGLE ||
                           k == EZPZ_POINT_ARC_COINCIDENT || k == EZPZ_ARC_LENGTH;
        cost += kind_is_linear(k) ? 1u : angle ? 4u : 3u;
    }
    uint32_t team = 2;
    while (team < 64 && (double)cost / 4.0 > 1.41421356 * team) team <<= 1;  // nearest power of two on a log scale
    return team;
}
// Workgroup size for large systems.
uint32_t auto_wg_team(uint32_t width) { return std::min<uint32_t>(512, std::max<uint32_t>(128, pow2_ceil((width + 3) / 4))); }

// Fixes the launch shape once the program (and so the workspace size) is known.  `stage_bytes` > 0 means
// that many leading bytes of the blob are copied to LDS by every workgroup (16-bit index lists).
// `panel_bytes`: LDS every team needs on top of its workspace (dense phases), counted when the workgroup is sized.
void finish_team(EzpzSystem& s, size_t stage_bytes, size_t panel_bytes = 0) {
    s.ws_doubles = rec_ws_base(s.counts, s.rec_jglobal) + s.rec_extra;
    const size_t ws_bytes = (size_t)s.ws_doubles * 8;
    s.prog_in_lds = stage_bytes > 0;
    s.prog_lds_doubles = (uint32_t)((stage_bytes + 15) / 16 * 2);
    const size_t prog_bytes = (size_t)s.prog_lds_doubles * 8;
    // Level staging (lm_kernel.hip.hpp, Cholesky loop): programs read from global memory on one wavefront or one
    // barrier workgroup per system get LDS for the three level tables and for one level of lists (levels wider than
    // the buffer are walked from global memory as before).
    s.lvl_lds_off = s.lvl_tab_words = s.lvl_buf_words = 0;
    const bool lvl_ok = s.view.lvl_words_max > 0 && !s.prog_in_lds && s.grid_wgs <= 1 && s.rec_extra == 0 &&
                        ((s.mode == MODE_SUB && s.team_size == 64) || s.mode == MODE_WGB);
    const uint32_t lvl_tab_words = (5 * (s.lvl_nlev + 1) + 3) & ~3u;
    if (s.mode == MODE_SUB) {
        const uint32_t team = s.team_size;
        s.lds_ws = true;
        size_t buf_bytes = 0;
        if (lvl_ok) {  // one buffer per wavefront, at most a quarter of what the workspace takes
            buf_bytes = std::min<size_t>((size_t)s.view.lvl_words_max * 4, std::max<size_t>(ws_bytes / 4, 2048));
            buf_bytes &= ~size_t(15);
        }
        // 256 lanes unless the workspaces would not fit; measured: smaller workgroups (more resident wavefronts for
        // big workspaces) are never faster, the kernels are issue-bound
        uint32_t threads = 256;
        while (threads > 64 && prog_bytes + (size_t)(threads / team) * (ws_bytes + buf_bytes + panel_bytes) > 64 * 1024) threads >>= 1;
        s.block_threads = std::max(threads, team);
        s.lds_bytes = prog_bytes + (size_t)(s.block_threads / team) * ws_bytes + 16;
        if (lvl_ok && buf_bytes >= 1024) {
            s.lvl_lds_off = (uint32_t)((s.lds_bytes + 15) / 16 * 2);
            s.lvl_tab_words = lvl_tab_words;
            s.lvl_buf_words = (uint32_t)(buf_bytes / 4);
            s.lds_bytes = (size_t)s.lvl_lds_off * 8 + (size_t)lvl_tab_words * 4 + (size_t)(s.block_threads / 64) * buf_bytes;
        }
    } else {
        s.block_threads = s.team_size;
        if (s.grid_wgs > 1) {  // every workgroup stages its own sub-program's lists
            s.ws_doubles = s.grid_ws_doubles;
            s.lds_ws = true;
            s.prog_in_lds = true;
            s.prog_lds_doubles = (uint32_t)((s.grid_stage_bytes + 15) / 16 * 2);
            s.lds_bytes = (size_t)s.prog_lds_doubles * 8 + (size_t)s.ws_doubles * 8 + 64 * 8 + 16;
        } else {
            s.lds_ws = prog_bytes + ws_bytes + 1024 <= s.lim.lds_bytes;
            s.lds_bytes = s.lds_ws ? prog_bytes + ws_bytes + 64 * 8 + 16 : prog_bytes + 80 * 8;
            if (lvl_ok) {
                const size_t base = (s.lds_bytes + 15) & ~size_t(15);
                // (4 KB stay free for the dense root block, analyze_into)
                const size_t room = s.lim.lds_bytes - 5120 > base + (size_t)lvl_tab_words * 4
                                        ? s.lim.lds_bytes - 5120 - base - (size_t)lvl_tab_words * 4 : 0;
                size_t buf_bytes = std::min<size_t>((size_t)s.view.lvl_words_max * 4, std::min<size_t>(room, 48 * 1024));
                if (s.lean_lds) buf_bytes = std::min<size_t>(buf_bytes, std::max<size_t>(ws_bytes / 4, 2048));
                buf_bytes &= ~size_t(15);
                if (buf_bytes >= 1024) {
                    s.lvl_lds_off = (uint32_t)(base / 8);
                    s.lvl_tab_words = lvl_tab_words;
                    s.lvl_buf_words = (uint32_t)(buf_bytes / 4);
                    s.lds_bytes = base + (size_t)lvl_tab_words * 4 + buf_bytes;
                }
            }
        }
    }
}

bool sub_team_fits(const ProgramCounts& c, uint32_t team) {
    return team <= 64 && (size_t)workspace_doubles(c) * 8 * (64 / team) <= 60 * 1024;
}

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

std::mutex g_grid_mu;
hipEvent_t g_grid_event[16] = {};  // per device: completion of the last grid-team launch of this process

// Grid team: G workgroups per system, all of a launch's workgroups resident at once, as many systems in flight as
// the device holds.
template <bool LIN>
int launch_grid_kernel(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
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
        hipLaunchKernelGGL(kernel, dim3(slots * s.grid_wgs), dim3(s.block_threads), s.lds_bytes, stream, args);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev, stream));
    }
    return EZPZ_OK;
}

int launch_grid_team(EzpzSystem& s, SolveArgs& args, hipStream_t stream) {
    return s.linear_only ? launch_grid_kernel<true>(s, args, stream) : launch_grid_kernel<false>(s, args, stream);
}

// The class-specialised kernel of a system spread over several workgroups (CompPlan::jit_wgs > 1): as many systems in
// flight as the device holds whole teams of; every workgroup of the launch must be resident (they wait for each other),
// so launches of this kind are chained like the list-walk grid teams' (launch_grid_kernel).
int launch_jit_grid(EzpzSystem& s, const CompLaunch& L, hipStream_t stream) {
    const uint32_t G = s.comp->jit_wgs;
    const uint64_t capacity = comp_jit_capacity(s.jit, *s.comp, s.device, s.lim.cus);
    if (capacity < G) return EZPZ_ERR_TOO_LARGE;
    const uint32_t slots = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(L.batch, capacity / G));
    if (s.jit_scratch.cap < (size_t)slots * kJitGridScratchBytes) {
        int rc = s.jit_scratch.ensure((size_t)slots * kJitGridScratchBytes);
        if (rc != EZPZ_OK) return rc;
        HIP_TRY(hipMemsetAsync(s.jit_scratch.p, 0, s.jit_scratch.cap, stream));
    }
    std::lock_guard<std::mutex> lock(g_grid_mu);
    hipEvent_t& ev = g_grid_event[s.device & 15];
    if (!ev)
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    else
        HIP_TRY(hipStreamWaitEvent(stream, ev, 0));
    int rc = comp_jit_launch(s.jit, *s.comp, s.dev_comp, L, s.device, s.lim.cus, stream, s.jit_scratch.p, slots);
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

bool jit_sync() {
    static const bool sync = [] {
        const char* e = std::getenv("EZPZ_JIT");
        return e && std::strcmp(e, "sync") == 0;
    }();
    return sync;
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
    grid = (uint32_t)std::min<uint64_t>(args.batch, (uint64_t)s.lim.cus * std::min<uint32_t>(per_cu, 8) * 2);
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
                if (s.comp->jit_wgs <= 1) {
                    if (comp_jit_launch(s.jit, *s.comp, s.dev_comp, L, s.device, s.lim.cus, stream) == EZPZ_OK) return EZPZ_OK;
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

unsigned long long* g_stamps = nullptr;  // diagnostic builds only (tools/stamps.py sets it through ezpz_debug_set_stamps)

void fill_cfg(SolveArgs& a, const EzpzConfig* cfg) {
    EzpzConfig d;
    ezpz_default_config(&d);
    if (!cfg) cfg = &d;
    a.max_iterations = (uint32_t)std::min<uint64_t>(cfg->max_iterations, 0xFFFFFFFFull);
    a.residual_tolerance = cfg->residual_tolerance;
    a.step_tolerance = cfg->step_tolerance;
    a.initial_lambda = cfg->initial_lambda;
}

template <class T>
size_t append(std::vector<unsigned char>& blob, const std::vector<T>& v) {
    size_t off = (blob.size() + 15) & ~size_t(15);
    blob.resize(off + std::max<size_t>(v.size() * sizeof(T), 16));
    if (!v.empty()) std::memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

}  // namespace

extern "C" {

void ezpz_default_config(EzpzConfig* cfg) {
    cfg->max_iterations = 35;  // solver.rs:72-81
    cfg->residual_tolerance = 1e-8;
    cfg->step_tolerance = 1e-12;
    cfg->initial_lambda = 1e-9;
}

int ezpz_launch_policy(int compute_units, EzpzLaunchPolicy* out) {
    if (!out || compute_units < 0) return EZPZ_ERR_INVALID_ARGUMENT;
    *out = launch_policy_for(compute_units > 0 ? compute_units : 256);
    return EZPZ_OK;
}

int ezpz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int ezpz_current_device(void) {
    int dev = -1;
    if (ezpz_device_count() < 1 || hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return dev;
}

const char* ezpz_error_string(int err) {
    switch (err) {
    case EZPZ_OK: return "ok";
    case EZPZ_ERR_NOT_FOUND: return "ID not found";
    case EZPZ_ERR_WRONG_NUMBER_GUESSES: return "There should be exactly 1 guess per variable";
    case EZPZ_ERR_MISSING_GUESS: return "Constraint references a variable that does not appear in the initial guesses";
    case EZPZ_ERR_MATRIX: return "Could not create matrix: variable id out of range";
    case EZPZ_ERR_EMPTY_SYSTEM: return "Cannot solve an empty system";
    case EZPZ_ERR_NO_DEVICE: return "no HIP device available (this library has no CPU fallback)";
    case EZPZ_ERR_HIP: return "HIP runtime error";
    case EZPZ_ERR_TOO_LARGE: return "system too large for this build";
    case EZPZ_ERR_INVALID_ARGUMENT: return "invalid argument";
    case EZPZ_ERR_KERNEL_BUDGET: return "the process already holds its budget of specialised kernels";
    case EZPZ_ERR_PARSE: return "could not parse problem text";
    case EZPZ_ERR_TEXT_MISSING_GUESS: return "No guess was given for a point";
    case EZPZ_ERR_TEXT_UNUSED_GUESSES: return "You gave a guess for points which weren't defined";
    case EZPZ_ERR_TEXT_UNDEFINED_POINT: return "You referred to a point that was never defined";
    default: return "unknown error";
    }
}

// Serialises a program: [index lists][jloc patterns] | [partitions][constraint table (+ side arrays)].  The lists go
// first so that a workgroup can stage exactly them in LDS; `idx16` stores them as 16-bit indices.  `pack_table` turns
// the constraint table into 32-byte PackedCon records + side arrays (workgroup teams read it from L2 in every sweep);
// it is honoured only with idx16 and at most 256 distinct jloc patterns.  Returns the byte size of the leading
// (stageable) part; fills every offset of `v` (not base / stage_bytes).
static size_t pack_program(const Program& P, bool idx16, bool pack_table, std::vector<unsigned char>& blob, ProgramView& v) {
    blob.clear();
    auto put = [&](const std::vector<uint32_t>& src) -> uint32_t {
        if (!idx16) return (uint32_t)append(blob, src);
        std::vector<uint16_t> t(src.begin(), src.end());
        return (uint32_t)append(blob, t);
    };
    v.o_colj_ptr = put(P.colj_ptr);
    v.o_colj_items = put(P.colj_items);
    v.o_apair_ptr = put(P.apair_ptr);
    v.o_apairs = put(P.apairs);
    v.o_lvl_cptr = put(P.lvl_cptr);
    v.o_lvl_sptr = put(P.lvl_sptr);
    v.o_l_col = put(P.l_col);
    {
        std::vector<uint32_t> grp = P.lvl_grp;
        grp.resize(P.lvl_cptr.size(), 1u | (1u << 8));
        v.o_lvl_grp = put(grp);
    }
    v.o_lpair_ptr = put(P.lpair_ptr);
    v.o_lpairs = put(P.lpairs);
    v.o_fwd_ptr = put(P.fwd_ptr);
    v.o_fwd_items = put(P.fwd_items);
    v.o_bwd_ptr = put(P.bwd_ptr);
    v.o_bwd_items = put(P.bwd_items);
    v.o_dense_col = put(P.dense_col);
    v.o_dense_slot = put(P.dense_slot);
    v.o_dense_tab = put(P.dense_tab);
    v.o_var_of = (uint32_t)append(blob, P.var_of);
    blob.resize((blob.size() + 15) & ~size_t(15));
    v.packed = 0;
    v.o_pos = v.o_weights = v.o_patterns = 0;
    std::vector<PackedCon> packed;
    if (idx16 && pack_table) {
        std::vector<std::array<uint8_t, 16>> patterns;
        packed.resize(P.cons.size());
        bool ok = true;
        for (size_t i = 0; i < P.cons.size() && ok; ++i) {
            const DevCon& d = P.cons[i];
            std::array<uint8_t, 16> pat;
            std::memcpy(pat.data(), d.jloc, 16);
            size_t k = 0;
            while (k < patterns.size() && patterns[k] != pat) ++k;
            if (k == patterns.size()) patterns.push_back(pat);
            if (k > 255) ok = false;
            PackedCon& q = packed[i];
            for (int e = 0; e < 8; ++e) q.ids[e] = (uint16_t)d.ids[e];
            q.param = d.param;
            q.row0 = (uint16_t)d.row0;
            q.jbase = (uint16_t)d.jbase;
            q.kind = d.kind;
            q.tag = d.tag;
            q.nrows = d.nrows;
            q.pattern = (uint8_t)k;
        }
        if (ok) {
            v.packed = 1;
            v.o_patterns = (uint32_t)append(blob, patterns);
            blob.resize((blob.size() + 15) & ~size_t(15));
        }
    }
    const size_t lists_bytes = blob.size();
    v.o_parts = (uint32_t)append(blob, P.parts);
    blob.resize((blob.size() + 15) & ~size_t(15));
    if (v.packed) {
        v.o_cons = (uint32_t)append(blob, packed);
        std::vector<uint32_t> pos(P.cons.size());
        std::vector<double> weights(P.cons.size());
        for (size_t i = 0; i < P.cons.size(); ++i) {
            pos[i] = P.cons[i].pos;
            weights[i] = P.cons[i].weight;
        }
        v.o_pos = (uint32_t)append(blob, pos);
        blob.resize((blob.size() + 15) & ~size_t(15));
        v.o_weights = (uint32_t)append(blob, weights);
    } else {
        v.o_cons = (uint32_t)append(blob, P.cons);
    }
    blob.resize((blob.size() + 15) & ~size_t(15));
    // Programs read from global memory (32-bit lists) of one partition: the lists one elimination level walks,
    // gathered into one contiguous block per level with level-relative list bounds, so that a team can bring a whole
    // level into LDS with one round of independent loads instead of chasing ptr -> items -> values through L2 twice
    // per level.  Block layout (32-bit words, every array padded to an even count, the block to a multiple of 4):
    //   [n_fwd, n_pairs] [fwd_ptr - fwd_ptr[c0] : ncols + 1] [fwd_items : 2 n_fwd]
    //   [lpair_ptr - lpair_ptr[s0] : nslots + 1] [lpairs : 2 n_pairs] [l_col : nslots]
    v.o_lvl_off = v.o_lvl_stream = v.o_lvl_boff = v.o_lvl_bstream = v.lvl_words_max = 0;
    if (!idx16 && P.c.n_parts == 1 && !P.c.dense && !P.parts.empty()) {
        const uint32_t lvl0 = P.parts[0].lvl0, nlev = P.parts[0].nlev;
        std::vector<uint32_t> off(nlev + 1), stream;
        auto pad = [&](size_t to) {
            while (stream.size() % to) stream.push_back(0);
        };
        uint32_t widest = 0;
        // (dense phases read their lists in place: no blocks for them, and their width does not size the level buffer)
        const uint32_t nwalk = P.n_dense ? P.dense_level0 : nlev;
        for (uint32_t lv = 0; lv < nlev; ++lv) {
            off[lv] = (uint32_t)stream.size();
            if (lv >= nwalk) continue;
            const uint32_t c0 = P.lvl_cptr[lvl0 + lv], c1 = P.lvl_cptr[lvl0 + lv + 1];
            const uint32_t s0 = P.lvl_sptr[lvl0 + lv], s1 = P.lvl_sptr[lvl0 + lv + 1];
            const uint32_t fq0 = P.fwd_ptr[c0], fq1 = P.fwd_ptr[c1], lq0 = P.lpair_ptr[s0], lq1 = P.lpair_ptr[s1];
            stream.push_back(fq1 - fq0);
            stream.push_back(lq1 - lq0);
            for (uint32_t k = c0; k <= c1; ++k) stream.push_back(P.fwd_ptr[k] - fq0);
            pad(2);
            stream.insert(stream.end(), P.fwd_items.begin() + 2 * (size_t)fq0, P.fwd_items.begin() + 2 * (size_t)fq1);
            for (uint32_t k = s0; k <= s1; ++k) stream.push_back(P.lpair_ptr[k] - lq0);
            pad(2);
            stream.insert(stream.end(), P.lpairs.begin() + 2 * (size_t)lq0, P.lpairs.begin() + 2 * (size_t)lq1);
            stream.insert(stream.end(), P.l_col.begin() + s0, P.l_col.begin() + s1);
            pad(4);
            widest = std::max<uint32_t>(widest, (uint32_t)stream.size() - off[lv]);
        }
        off[nlev] = (uint32_t)stream.size();
        std::vector<uint32_t> boff(nlev + 1), bstream;
        for (uint32_t lv = 0; lv < nlev; ++lv) {
            boff[lv] = (uint32_t)bstream.size();
            if (lv >= nwalk) continue;
            const uint32_t c0 = P.lvl_cptr[lvl0 + lv], c1 = P.lvl_cptr[lvl0 + lv + 1];
            const uint32_t q0 = P.bwd_ptr[c0], q1 = P.bwd_ptr[c1];
            bstream.push_back(q1 - q0);
            bstream.push_back(0);
            for (uint32_t k = c0; k <= c1; ++k) bstream.push_back(P.bwd_ptr[k] - q0);
            while (bstream.size() % 2) bstream.push_back(0);
            bstream.insert(bstream.end(), P.bwd_items.begin() + 2 * (size_t)q0, P.bwd_items.begin() + 2 * (size_t)q1);
            while (bstream.size() % 4) bstream.push_back(0);
            widest = std::max<uint32_t>(widest, (uint32_t)bstream.size() - boff[lv]);
        }
        boff[nlev] = (uint32_t)bstream.size();
        if (stream.size() < (1u << 30) && bstream.size() < (1u << 30)) {
            v.o_lvl_off = (uint32_t)append(blob, off);
            blob.resize((blob.size() + 15) & ~size_t(15));
            v.o_lvl_stream = (uint32_t)append(blob, stream);
            blob.resize((blob.size() + 15) & ~size_t(15));
            v.o_lvl_boff = (uint32_t)append(blob, boff);
            blob.resize((blob.size() + 15) & ~size_t(15));
            v.o_lvl_bstream = (uint32_t)append(blob, bstream);
            blob.resize((blob.size() + 15) & ~size_t(15));
            v.lvl_words_max = widest;
        }
    }
    v.blob_bytes = (uint32_t)blob.size();
    v.n_cons = P.c.n_cons;
    v.n_vars = P.c.n_vars;
    v.n_rows = P.c.n_rows;
    v.zj = P.c.zj;
    v.zlo = P.c.zlo;
    v.n_parts = P.c.n_parts;
    return lists_bytes;
}

// The program of partitions [p0, p1) alone, renumbered from zero.  The internal numbering is partition-major in every
// index space (variables, rows, Jacobian slots, L slots, constraints), so a run of partitions is a contiguous range of
// each and the slice is the same lists minus the range's first index.  A grid team's workgroup runs exactly like a
// workgroup team on its slice.
static Program slice_program(const Program& P, uint32_t p0, uint32_t p1) {
    Program S;
    const PartDesc& first = P.parts[p0];
    const PartDesc& last = P.parts[p1 - 1];
    const uint32_t C = P.c.n_cons;
    const uint32_t v0 = P.lvl_cptr[first.lvl0], v1 = P.lvl_cptr[last.lvl0 + last.nlev];
    const uint32_t l0 = P.lvl_sptr[first.lvl0], l1 = P.lvl_sptr[last.lvl0 + last.nlev];
    const uint32_t c0 = first.con0, c1 = last.con1;
    const uint32_t r0 = c0 < C ? P.cons[c0].row0 : P.c.n_rows, r1 = c1 < C ? P.cons[c1].row0 : P.c.n_rows;
    const uint32_t j0 = c0 < C ? P.cons[c0].jbase : P.c.zj, j1 = c1 < C ? P.cons[c1].jbase : P.c.zj;
    const uint32_t lvl_a = first.lvl0, lvl_b = last.lvl0 + last.nlev + 1;  // this run's entries of lvl_cptr / lvl_sptr
    S.c = P.c;
    S.c.n_cons = c1 - c0;
    S.c.n_vars = v1 - v0;
    S.c.n_rows = r1 - r0;
    S.c.zj = j1 - j0;
    S.c.zlo = l1 - l0;
    S.c.n_parts = p1 - p0;
    for (uint32_t p = p0; p < p1; ++p) {
        PartDesc d = P.parts[p];
        d.con0 -= c0;
        d.con1 -= c0;
        d.lvl0 -= lvl_a;
        S.parts.push_back(d);
    }
    for (uint32_t k = lvl_a; k < lvl_b; ++k) {
        S.lvl_cptr.push_back(P.lvl_cptr[k] - v0);
        S.lvl_sptr.push_back(P.lvl_sptr[k] - l0);
    }
    // CSR slices: ptr[a..b] rebased, items (x - bx, y - by)
    auto csr = [](const std::vector<uint32_t>& ptr, const std::vector<uint32_t>& items, uint32_t a, uint32_t b,
                  uint32_t bx, uint32_t by, std::vector<uint32_t>& optr, std::vector<uint32_t>& oitems) {
        const uint32_t q0 = ptr[a], q1 = ptr[b];
        optr.resize(b - a + 1);
        for (uint32_t k = a; k <= b; ++k) optr[k - a] = ptr[k] - q0;
        oitems.resize(2 * (size_t)(q1 - q0));
        for (uint32_t q = q0; q < q1; ++q) {
            oitems[2 * (q - q0)] = items[2 * q] - bx;
            oitems[2 * (q - q0) + 1] = items[2 * q + 1] - by;
        }
    };
    csr(P.colj_ptr, P.colj_items, v0, v1, j0, r0, S.colj_ptr, S.colj_items);
    csr(P.apair_ptr, P.apairs, l0, l1, j0, j0, S.apair_ptr, S.apairs);
    csr(P.lpair_ptr, P.lpairs, l0, l1, l0, l0, S.lpair_ptr, S.lpairs);
    csr(P.fwd_ptr, P.fwd_items, v0, v1, l0, v0, S.fwd_ptr, S.fwd_items);
    csr(P.bwd_ptr, P.bwd_items, v0, v1, l0, v0, S.bwd_ptr, S.bwd_items);
    S.c.n_apairs = S.apairs.size() / 2;
    S.c.n_lpairs = S.lpairs.size() / 2;
    S.l_col.assign(P.l_col.begin() + l0, P.l_col.begin() + l1);
    for (uint32_t& v : S.l_col) v -= v0;
    S.var_of.assign(P.var_of.begin() + v0, P.var_of.begin() + v1);
    S.cons.assign(P.cons.begin() + c0, P.cons.begin() + c1);
    for (DevCon& d : S.cons) {
        const KindInfo& K = kKinds[d.kind];
        for (int k = 0; k < K.n_ids; ++k) d.ids[k] = d.ids[k] >= v0 && d.ids[k] < v1 ? d.ids[k] - v0 : 0;
        d.row0 -= r0;
        d.jbase -= j0;
    }
    return S;
}

// Grid team: one sub-program per workgroup (slice_program), packed and staged like a workgroup team's.  False when some
// slice does not fit a CU's LDS (state + staged lists) or cannot be packed; `s` is then left without grid data.
static bool pack_grid_slices(EzpzSystem& s, const Program& P, uint32_t G, uint32_t W) {
    s.grid_blob.clear();
    s.host_grid_views.clear();
    s.grid_ws_doubles = 0;
    s.grid_stage_bytes = 0;
    std::vector<unsigned char> sub;
    for (uint32_t g = 0; g < G; ++g) {
        const Program S = slice_program(P, g * W, g * W + W);
        ProgramView sv{};
        const bool fits16 = S.c.n_vars < 65536 && S.c.n_rows < 65536 && S.c.zj < 65536 && S.c.zlo < 65536 &&
                            S.c.n_apairs < 65536 && S.c.n_lpairs < 65536 && S.c.n_cons < 65536;
        const size_t lists_bytes = fits16 ? pack_program(S, true, true, sub, sv) : 0;
        const uint32_t wsd = workspace_doubles(S.c);
        if (!fits16 || !sv.packed || lists_bytes + (size_t)wsd * 8 + 2048 > s.lim.lds_bytes ||
            s.grid_blob.size() + sub.size() > 0xFFFFFF00ull) {
            s.grid_blob.clear();
            s.host_grid_views.clear();
            return false;
        }
        sv.stage_bytes = (uint32_t)lists_bytes;
        sv.blob_bytes = (uint32_t)s.grid_blob.size();  // for a grid view: byte offset of this slice in the grid blob
        s.grid_blob.insert(s.grid_blob.end(), sub.begin(), sub.end());
        s.grid_blob.resize((s.grid_blob.size() + 255) & ~size_t(255));
        s.host_grid_views.push_back(sv);
        s.grid_ws_doubles = std::max(s.grid_ws_doubles, wsd);
        s.grid_stage_bytes = std::max<size_t>(s.grid_stage_bytes, lists_bytes);
    }
    return true;
}

// Record walk (lm_kernel.hip.hpp, REC builds): the factorisation, the forward and the backward substitution of ONE connected
// system on a barrier workgroup of T lanes, as rounds.  In a round a group of g lanes owns one item:
//   factor, entry (i, j):  l_ij = (A_ij - sum_k l_ik l_jk) / sqrt(A_jj - sum_k l_jk^2)   over row j of L (k < j); where row i
//                          has no entry in column k the pair's second operand is a double that stays zero
//   factor, column j:      y_j  = (b_j  - sum_k l_jk y_k ) / sqrt(A_jj - sum_k l_jk^2)   and 1 / d_j (kept beside A_jj: the entries of
//                          column j read A_jj in the same round)
//   backward, column j:    x_j  = (y_j  - sum_i l_ij x_i ) / d_j                          over column j of L (i > j)
// -- one list per item (every lane of a column's entries recomputes d_j from the same terms in the same order), cut
// into the lanes' shares at build time: a lane's record is ready workspace addresses, nothing is looked up on the device.
// A level of the elimination tree takes ceil(items x g / T) rounds, longest lists first; g (a power of two per level)
// minimises rounds x (a round's fixed cost + its longest share + the group's sum).
// Layout: desc[(round x wavefronts + wavefront) x 2] = flags (chunks to load: 0 = nothing to do; log2 g; rendezvous first;
// backward), first chunk; chunks[((chunk + c) x 64 + lane of the wavefront) x 4]: chunk 0 = target | diagonal << 16, destination
// | lane flags, two (a | b << 16) pairs; chunks 1 and 2 = four pairs each (REC_* in lm_kernel.hip.hpp).  Only a wavefront that
// has an item in a round has chunks for it, as many as its longest share needs.
// (analyze_into: up to this many components walk records as one partition; EZPZ_REC_MAX_COMPONENTS for A/B runs)
static const uint32_t kRecMaxComponents = [] {
    const char* e = std::getenv("EZPZ_REC_MAX_COMPONENTS");
    return e ? (uint32_t)std::atol(e) : launch_policy_for(256).rec_max_components;  // (from 128 the component-resident shape may take the system)
}();
struct RecPlan {
    std::vector<uint32_t> desc, chunks;
    uint32_t rounds = 0;
    // packed assembly (SolveArgs::rec_asm_*): chunks per column / per entry of the strict lower part (0: none, the lists are walked)
    std::vector<uint32_t> asm_cols, asm_slots;
    uint32_t asm_kc = 0, asm_ks = 0;
};
// `wide`: the workspace lives in global memory -- 32-bit addresses counted from its start (lds_base = 0), chunk 0 = target,
// diagonal, destination, lane flags, then up to four chunks of two (a, b) pairs; no packed assembly.
// `jglobal` (LDS form): the Jacobian's values live in global memory (SolveArgs::rec_jglobal): no room for them in the workspace, and
// the packed assembly's J operands are plain slot numbers (padding: slot zJ, a zero behind the values).
static bool build_records(const Program& P, uint32_t T, uint32_t lds_base, bool wide, bool jglobal, RecPlan& out) {
    if (P.c.n_parts != 1 || P.parts.size() != 1 || P.c.dense || P.n_dense || T < 64 || T % 64) return false;
    const uint32_t n = P.c.n_vars, m = P.c.n_rows, zj = P.c.zj, zlo = P.c.zlo;
    // (addresses in the records count doubles from the start of the LDS; the workspace begins `lds_base` doubles in)
    const uint32_t o_d = lds_base + n + 2 * m + (jglobal ? 0u : zj), o_l = o_d + n, o_v = o_l + zlo,
                   o_dd = lds_base + rec_ws_base(P.c, jglobal), o_zero = o_dd + n;
    if (!wide && o_zero >= 65536) return false;  // 16-bit addresses
    const uint32_t lvl0 = P.parts[0].lvl0, nlev = P.parts[0].nlev, n_waves = T / 64;
    const uint32_t kMaxShare = wide ? REC_WIDE_PAIRS : REC_MAX_PAIRS;
    struct Item {
        uint32_t target, diag, dest;
        bool col;
        std::vector<std::pair<uint32_t, uint32_t>> list;
    };
    const uint32_t zero_pair = wide ? o_zero : o_zero | (o_zero << 16);  // (wide: every word of a chunk is an address)
    auto emit_level = [&](std::vector<Item>& items, bool bwd, bool barrier) {
        if (items.empty()) return;
        std::stable_sort(items.begin(), items.end(), [](const Item& x, const Item& y) { return x.list.size() > y.list.size(); });
        const uint32_t longest = (uint32_t)items[0].list.size();
        uint32_t best_g = 0, best_lg = 0;
        double best = 0.0;
        for (uint32_t g = 1, lg = 0; g <= 64; g <<= 1, ++lg) {
            if ((longest + g - 1) / g > kMaxShare) continue;
            const uint32_t ngrp = T / g;
            double cost = 0.0;
            // (measured and not kept: no rendezvous before a round whose items all sit in wavefront 0 while no other wavefront has
            // stored since the last one -- half of a sketch's rounds -- changes nothing: the rendezvous is not what a round costs)
            for (size_t t = 0; t < items.size(); t += ngrp)
                cost += 500.0 + 20.0 * (double)((items[t].list.size() + g - 1) / g) + (g > 1 ? 30.0 * lg : 0.0);
            if (!best_g || cost < best - 1e-9) best = cost, best_g = g, best_lg = lg;
            if ((uint64_t)items.size() * g >= T && g >= longest) break;  // more lanes per list buy nothing
        }
        if (!best_g) {
            out.rounds = 0xFFFFFFFFu;  // a list longer than 64 lanes x 10 pairs
            return;
        }
        const uint32_t g = best_g, ngrp = T / g;
#ifdef EZPZ_STAMPS
        std::fprintf(stderr, "rounds %3u..: %s level of %5zu items, longest list %3u, %2u lanes per list\n", out.rounds, bwd ? "bwd" : "fac",
                     items.size(), longest, g);
#endif
        for (size_t t0 = 0; t0 < items.size(); t0 += ngrp) {
            for (uint32_t w = 0; w < n_waves; ++w) {
                // this wavefront's lanes: groups [w * 64 / g, (w + 1) * 64 / g)
                uint32_t nch = 0;
                for (uint32_t l = 0; l < 64; ++l) {
                    const size_t t = t0 + (w * 64 + l) / g;
                    if (t >= items.size()) continue;
                    const uint32_t sub = l & (g - 1), len = (uint32_t)items[t].list.size();
                    const uint32_t share = len > sub ? (len - sub + g - 1) / g : 0;
                    nch = std::max(nch, wide ? 1u + (share + 1) / 2 : share <= 2 ? 1u : 1u + (share - 2 + 3) / 4);
                }
                const uint32_t chunk0 = (uint32_t)(out.chunks.size() / (64 * 4));
                out.desc.push_back(nch | (best_lg << REC_LG_SHIFT) | (barrier && t0 == 0 ? REC_BARRIER : 0u) | (bwd ? REC_BWD : 0u));
                out.desc.push_back(chunk0);
                out.chunks.resize(out.chunks.size() + (size_t)nch * 64 * 4, zero_pair);
                if (!nch) continue;
                for (uint32_t l = 0; l < 64; ++l) {
                    uint32_t* c0 = &out.chunks[((size_t)chunk0 * 64 + l) * 4];
                    const size_t t = t0 + (w * 64 + l) / g;
                    // (an idle lane of a working wavefront reads zeros and writes nothing)
                    if (wide) {
                        c0[0] = c0[1] = c0[2] = o_zero;
                        c0[3] = 0u;
                    } else {
                        c0[0] = zero_pair;
                        c0[1] = o_zero;
                    }
                    if (t >= items.size()) continue;
                    const Item& it = items[t];
                    const uint32_t sub = l & (g - 1);
                    const uint32_t lane_flags = (sub == 0 ? REC_WRITER : 0u) | (it.col ? REC_ISCOL : 0u);
                    if (wide) {
                        c0[0] = it.target, c0[1] = it.diag, c0[2] = it.dest, c0[3] = lane_flags;
                    } else {
                        c0[0] = it.target | (it.diag << 16);
                        c0[1] = it.dest | lane_flags;
                    }
                    uint32_t k = 0;
                    for (size_t q = sub; q < it.list.size(); q += g, ++k) {
                        if (wide) {
                            uint32_t* c = &out.chunks[((size_t)(chunk0 + 1 + k / 2) * 64 + l) * 4 + 2 * (k % 2)];
                            c[0] = it.list[q].first;
                            c[1] = it.list[q].second;
                            continue;
                        }
                        const uint32_t word = it.list[q].first | (it.list[q].second << 16);
                        if (k < 2)
                            c0[2 + k] = word;
                        else
                            out.chunks[((size_t)(chunk0 + 1 + (k - 2) / 4) * 64 + l) * 4 + (k - 2) % 4] = word;
                    }
                }
            }
            ++out.rounds;
        }
    };
    out.desc.clear();
    out.chunks.clear();
    out.rounds = 0;
    {  // packed assembly: every column's (J slot, row of r) pairs and every lower entry's (J slot, J slot) pairs, four to a chunk
        const uint32_t o_r = lds_base + n, o_j = lds_base + n + 2 * m;
        const uint32_t call0 = P.lvl_cptr[lvl0], call1 = P.lvl_cptr[lvl0 + nlev], sall0 = P.lvl_sptr[lvl0], sall1 = P.lvl_sptr[lvl0 + nlev];
        auto pack = [&](const std::vector<uint32_t>& ptr, const std::vector<uint32_t>& items, uint32_t i0, uint32_t i1, uint32_t off_a,
                        uint32_t off_b, uint32_t zero_pair, std::vector<uint32_t>& dst) -> uint32_t {
            uint32_t longest = 0;
            for (uint32_t i = i0; i < i1; ++i) longest = std::max(longest, ptr[i + 1] - ptr[i]);
            const uint32_t K = std::max(1u, (longest + 3) / 4), N = i1 - i0;
            if (K > 3) return 0;
            dst.assign((size_t)K * N * 4 + 4, zero_pair);
            for (uint32_t i = i0; i < i1; ++i)
                for (uint32_t q = ptr[i], e = 0; q < ptr[i + 1]; ++q, ++e)
                    dst[((size_t)(e / 4) * N + (i - i0)) * 4 + e % 4] = (off_a + items[2 * q]) | ((off_b + items[2 * q + 1]) << 16);
            return K;
        };
        // (J operands: LDS addresses, or -- jglobal -- slot numbers with slot zJ as the zero)
        const uint32_t ja = jglobal ? 0u : o_j, jz = jglobal ? zj : o_zero;
        out.asm_kc = wide ? 0 : pack(P.colj_ptr, P.colj_items, call0, call1, ja, o_r, jz | (o_zero << 16), out.asm_cols);
        out.asm_ks = out.asm_kc ? pack(P.apair_ptr, P.apairs, sall0, sall1, ja, ja, jz | (jz << 16), out.asm_slots) : 0;
        if (!out.asm_ks) out.asm_kc = 0;
    }
    std::vector<Item> items;
    std::vector<uint32_t> other(zlo, 0xFFFFFFFFu);  // per slot (j, k) of the current column's row: the slot (i, k), if any
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        const uint32_t c0 = P.lvl_cptr[lvl0 + lv], c1 = P.lvl_cptr[lvl0 + lv + 1];
        const uint32_t s0 = P.lvl_sptr[lvl0 + lv], s1 = P.lvl_sptr[lvl0 + lv + 1];
        items.clear();
        for (uint32_t j = c0; j < c1; ++j) {
            Item it{o_v + j, o_d + j, o_v + j, true, {}};
            for (uint32_t q = P.fwd_ptr[j]; q < P.fwd_ptr[j + 1]; ++q)
                it.list.push_back({o_l + P.fwd_items[2 * q], o_v + P.fwd_items[2 * q + 1]});
            items.push_back(std::move(it));
        }
        for (uint32_t sl = s0; sl < s1; ++sl) {
            const uint32_t j = P.l_col[sl];
            if (j < c0 || j >= c1) return false;
            // (slot_ik, slot_jk) pairs of this entry: which of the two lies in row j tells them apart
            std::vector<uint32_t> touched;
            for (uint32_t q = P.fwd_ptr[j]; q < P.fwd_ptr[j + 1]; ++q) other[P.fwd_items[2 * q]] = 0xFFFFFFFEu;
            bool ok = true;
            for (uint32_t q = P.lpair_ptr[sl]; q < P.lpair_ptr[sl + 1]; ++q) {
                const uint32_t u = P.lpairs[2 * q], w = P.lpairs[2 * q + 1];
                if (w < zlo && other[w] == 0xFFFFFFFEu)
                    other[w] = u;
                else if (u < zlo && other[u] == 0xFFFFFFFEu)
                    other[u] = w;
                else
                    ok = false;
            }
            Item it{o_l + sl, o_d + j, o_l + sl, false, {}};
            for (uint32_t q = P.fwd_ptr[j]; q < P.fwd_ptr[j + 1]; ++q) {
                const uint32_t sjk = P.fwd_items[2 * q];
                it.list.push_back({o_l + sjk, other[sjk] < zlo ? o_l + other[sjk] : o_zero});
                other[sjk] = 0xFFFFFFFFu;
            }
            if (!ok) return false;
            items.push_back(std::move(it));
        }
        emit_level(items, false, lv > 0);  // (the assembly ends with a rendezvous of its own)
        if (out.rounds == 0xFFFFFFFFu) return false;
    }
    for (uint32_t lv = nlev; lv-- > 0;) {
        const uint32_t c0 = P.lvl_cptr[lvl0 + lv], c1 = P.lvl_cptr[lvl0 + lv + 1];
        items.clear();
        for (uint32_t j = c0; j < c1; ++j) {
            Item it{o_v + j, o_dd + j, o_v + j, true, {}};
            for (uint32_t q = P.bwd_ptr[j]; q < P.bwd_ptr[j + 1]; ++q)
                it.list.push_back({o_l + P.bwd_items[2 * q], o_v + P.bwd_items[2 * q + 1]});
            items.push_back(std::move(it));
        }
        emit_level(items, true, true);
        if (out.rounds == 0xFFFFFFFFu) return false;
    }
    // an even number of rounds (the kernel alternates between two sets of registers), then two idle ones: the requests a round
    // makes for the next round's records need no condition
    const uint32_t idle = 2 + (out.rounds & 1u);
    out.desc.resize(out.desc.size() + (size_t)idle * n_waves * 2, 0u);
    out.rounds += out.rounds & 1u;
    out.chunks.resize(out.chunks.size() + 64 * 4, zero_pair);
    return out.rounds > 0 && out.chunks.size() / (64 * 4) < 0xFFFFFFF0ull;
}

// Symbolic phase + launch-shape decision shared by ezpz_system_create and ezpz_analyze.
// Lanes per list, level by level, for the teams that run a level as one phase (one wavefront or one barrier workgroup
// on a one-partition program): a level lasts as long as its longest list, and the top levels of an elimination tree are
// a few columns with long lists, so there g lanes share each list.  g minimises passes x (rounds per list + the group's
// reduction), in units of one chunk's round trip.
static void choose_level_groups(Program& P, const EzpzSystem& s) {
    P.lvl_grp.assign(P.lvl_cptr.size(), 1u | (1u << 8));
    const bool fused = (s.mode == MODE_WGB || (s.mode == MODE_SUB && s.team_size == 64)) && s.grid_wgs <= 1;
    if (!fused || P.c.n_parts != 1 || P.c.dense || P.parts.empty()) return;
    const uint32_t lanes = s.team_size, chunk = s.mode == MODE_SUB ? 4u : 2u;
    const uint32_t lvl0 = P.parts[0].lvl0, nlev = P.parts[0].nlev;
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        const uint32_t c0 = P.lvl_cptr[lvl0 + lv], c1 = P.lvl_cptr[lvl0 + lv + 1];
        const uint32_t s0 = P.lvl_sptr[lvl0 + lv], s1 = P.lvl_sptr[lvl0 + lv + 1];
        if (c1 - c0 > lanes) continue;  // wider than the team: the two-phase walk
        uint32_t need = 0;
        for (uint32_t c = c0; c < c1; ++c) need = std::max(need, P.fwd_ptr[c + 1] - P.fwd_ptr[c]);
        for (uint32_t k = s0; k < s1; ++k) need = std::max(need, P.lpair_ptr[k + 1] - P.lpair_ptr[k]);
        double best = 0.0;
        uint32_t best_g = 1;
        // (measured on 150 / 300 / 800 variables, one solve, groups capped at 1 / 2 / 4 / 8 / 16 / 64 lanes: 370 / 290 / 245 /
        // 230 / 222 / 223 us, 641 / 483 / 393 / 360 / 347 / 346 us, 10.7 / 7.9 / 6.7 / 6.3 / 6.1 / 6.1 ms)
        for (uint32_t g = 1, lg = 0; g <= 64 && (uint64_t)(c1 - c0) * g <= lanes; g <<= 1, ++lg) {
            // (columns and slots are items of one walk: lm_kernel.hip.hpp, chol_level)
            const double passes = std::max<double>(1.0, std::ceil((double)((c1 - c0) + (s1 - s0)) * g / lanes));
            const double rounds = std::ceil((double)need / (g * chunk));
            const double cost = passes * (rounds + (g > 1 ? 0.3 + 0.25 * lg : 0.0));
            if (g == 1 || cost < best - 1e-9) best = cost, best_g = g;
        }
        // backward substitution: one list per column, so g is bounded by the lanes per column only
        uint32_t bneed = 0, bg = 1;
        for (uint32_t c = c0; c < c1; ++c) bneed = std::max(bneed, P.bwd_ptr[c + 1] - P.bwd_ptr[c]);
        while (bg < 64 && (uint64_t)(c1 - c0) * (bg * 2) <= lanes && bg * chunk < bneed) bg <<= 1;
        P.lvl_grp[lvl0 + lv] = best_g | (bg << 8);
#ifdef EZPZ_STAMPS
        std::fprintf(stderr, "level %3u: columns %4u slots %5u longest list %3u (bwd %3u) lanes/list %2u (bwd %2u)\n", lv, c1 - c0, s1 - s0,
                     need, bneed, best_g, bg);
#endif
    }
}

// Dense phases of a one-partition program of one connected component (Program::n_dense).  The top of a connected
// sketch's elimination tree is a tree of separators: chains of one or two columns per level whose lists hold 20-40 terms,
// each level a full round of dependent hops, a reduction, a square root and a divide for a handful of entries (~3 k
// cycles a level in the factorisation, ~1.2 k in the backward substitution).  From the top down, runs of whole levels
// become phases: the columns of a phase fall into the connected pieces of the elimination tree inside it (at most one per
// wavefront, <= 16 columns and <= 63 panel rows each), every piece a dense panel (lm_kernel.hip.hpp, dense phases).  The
// last phase is the root block (the last <= 16 columns).  Returns false -- program untouched -- when the root block is
// not worth it (fewer than 5 levels), or for anything but one connected component in one partition.
static bool make_dense_phases(Program& P, uint32_t n_waves, size_t lds_room_bytes) {
    if (P.c.n_parts != 1 || P.c.n_components != 1 || P.c.dense || P.parts.size() != 1 || P.n_dense) return false;
    const uint32_t lvl0 = P.parts[0].lvl0, nlev = P.parts[0].nlev, n = P.c.n_vars, zlo = P.c.zlo;
    if (lvl0 != 0 || nlev < 6 || n_waves == 0) return false;
    constexpr uint32_t kMaxCols = 16, kMaxRows = 63, kMaxPhases = 4, NONE = 0xFFFFFFFFu;
    n_waves = std::min(n_waves, 8u);
    std::vector<uint32_t> level(n), parent(n, NONE);
    for (uint32_t lv = 0; lv < nlev; ++lv)
        for (uint32_t j = P.lvl_cptr[lv]; j < P.lvl_cptr[lv + 1]; ++j) level[j] = lv;
    for (uint32_t j = 0; j < n; ++j)
        for (uint32_t q = P.bwd_ptr[j]; q < P.bwd_ptr[j + 1]; ++q) {
            const uint32_t sl = P.bwd_items[2 * q], i = P.bwd_items[2 * q + 1];
            if (sl >= zlo || i <= j || i >= n || P.l_col[sl] != j) return false;
            parent[j] = std::min(parent[j], i);  // the first row below the diagonal is the parent in the elimination tree
        }
    struct Block {
        std::vector<uint32_t> cols, below;  // ascending
    };
    struct Phase {
        uint32_t la, lb;
        std::vector<Block> blocks;
    };
    // the blocks of the levels [la, lb): connected pieces of the tree inside them, each with the later rows it touches
    auto cut = [&](uint32_t la, uint32_t lb, std::vector<Block>& out) -> bool {
        const uint32_t c0 = P.lvl_cptr[la], c1 = P.lvl_cptr[lb];
        std::vector<uint32_t> top(c1 - c0);
        // (parents come later in the numbering: one pass from the top labels every column with its piece's top column)
        for (uint32_t j = c1; j-- > c0;) top[j - c0] = (parent[j] != NONE && parent[j] < c1) ? top[parent[j] - c0] : j;
        std::vector<uint32_t> tops;
        for (uint32_t j = c0; j < c1; ++j)
            if (top[j - c0] == j) tops.push_back(j);
        if (tops.size() > std::min(16u, 2 * n_waves)) return false;  // at most two blocks per wavefront
        out.assign(tops.size(), Block());
        for (uint32_t j = c0; j < c1; ++j) {
            const size_t b = std::lower_bound(tops.begin(), tops.end(), top[j - c0]) - tops.begin();
            out[b].cols.push_back(j);
            for (uint32_t q = P.bwd_ptr[j]; q < P.bwd_ptr[j + 1]; ++q) {
                const uint32_t i = P.bwd_items[2 * q + 1];
                if (i >= c1)
                    out[b].below.push_back(i);
                else if (top[i - c0] != top[j - c0])
                    return false;  // (cannot happen: a row of column j is an ancestor of j)
            }
        }
        for (Block& b : out) {
            std::sort(b.below.begin(), b.below.end());
            b.below.erase(std::unique(b.below.begin(), b.below.end()), b.below.end());
            if (b.cols.size() > kMaxCols || b.cols.size() + b.below.size() + 1 > kMaxRows) return false;
        }
        return true;
    };
    auto lds_doubles = [](const std::vector<Block>& bs) {
        size_t d = 0;
        for (const Block& b : bs) d += (b.cols.size() + b.below.size() + 1) * (b.cols.size() | 1u);
        return d;
    };
    std::vector<Phase> phases;  // from the top down
    size_t lds_used = 0;
    {
        uint32_t la = nlev;
        while (la > 1 && n - P.lvl_cptr[la - 1] <= kMaxCols) --la;
        Phase root{la, nlev, {}};
        if (nlev - la < 5 || !cut(la, nlev, root.blocks)) return false;
        if (root.blocks.size() != 1) {  // several tree tops among the last columns: still one panel (no rows below it)
            Block all;
            for (uint32_t j = P.lvl_cptr[la]; j < n; ++j) all.cols.push_back(j);
            root.blocks.assign(1, all);
        }
        lds_used = lds_doubles(root.blocks) * 8;
        if (lds_used > lds_room_bytes) return false;
        phases.push_back(std::move(root));
    }
    static const uint32_t max_phases = [] {
        const char* e = std::getenv("EZPZ_DENSE_PHASES");
        return e ? std::min<uint32_t>(4u, (uint32_t)std::atoi(e)) : 4u;
    }();
    while (phases.size() < max_phases) {
        const uint32_t lb = phases.back().la;
        // How far down?  A walked level costs ~4.1 k cycles (2.9 k in the factorisation, 1.2 k in the backward substitution).
        // A phase costs ~9 k for its gather, write-back and rendezvous, ~3 k per round of blocks (one block per wavefront
        // and round, the largest blocks first) and ~0.5 k per column of a round's largest block (stamps on the 300-variable
        // sketch: 16 blocks of <= 3 columns 12.3 k + 6.7 k cycles, 4 blocks of <= 14: 14.6 k + 6.7 k, the root block of 16:
        // 12.7 k + 6.7 k; the constants swept on 150-2000 variables, one solve: 0.8 k per column keeps 800 and 2000 variables
        // at two phases, 4.80 / 2.17 ms, 0.5 k gives them a third, 4.49 / 2.02 ms; a fixed cost of 4 k instead of 9 k costs
        // 300 variables 234 -> 247 us): the cut that saves most.
        uint32_t la = lb, best_la = lb;
        double best_saving = 0.0;
        std::vector<Block> best, trial;
        // (at most 32 levels per phase: every trial re-scans the whole run)
        while (la > 1 && lb - la < 32 && cut(la - 1, lb, trial) && lds_used + lds_doubles(trial) * 8 <= lds_room_bytes) {
            --la;
            std::sort(trial.begin(), trial.end(), [](const Block& x, const Block& y) { return x.cols.size() > y.cols.size(); });
            double cost = 9000.0;
            for (size_t b = 0; b < trial.size(); b += n_waves) cost += 3000.0 + 500.0 * (double)trial[b].cols.size();
            const double saving = 4100.0 * (lb - la) - cost;
            if (saving > best_saving) best_saving = saving, best_la = la, best = trial;
        }
        if (best_la == lb) break;
        la = best_la;
        lds_used += lds_doubles(best) * 8;
        phases.push_back(Phase{la, lb, std::move(best)});
    }
    std::reverse(phases.begin(), phases.end());  // in the order they run
    if (std::getenv("EZPZ_DENSE_DEBUG")) {
        for (const Phase& ph : phases) {
            std::fprintf(stderr, "dense phase: levels [%u, %u) of %u:", ph.la, ph.lb, nlev);
            for (const Block& b : ph.blocks) std::fprintf(stderr, " %zu cols + %zu rows below;", b.cols.size(), b.below.size());
            std::fprintf(stderr, "\n");
        }
        std::vector<Block> t;
        const uint32_t lb = phases.front().la;
        for (uint32_t la = lb; la-- > 0 && lb - la <= 8;) {
            const bool ok = cut(la, lb, t);
            std::fprintf(stderr, "  next phase [%u, %u): %s, %zu blocks:", la, lb, ok ? "ok" : "no", t.size());
            for (const Block& b : t) std::fprintf(stderr, " %zu+%zu", b.cols.size(), b.below.size());
            std::fprintf(stderr, "\n");
        }
    }
    const uint32_t lw = phases.front().la, dc0 = P.lvl_cptr[lw], ds0 = P.lvl_sptr[lw];
    // ---- tables ---------------------------------------------------------------------------------------------------------------
    std::vector<uint32_t> dcol(n - dc0, 0), dslot(zlo - ds0, 0), cutcol(n - dc0, 0), tab(1 + phases.size(), 0);
    tab[0] = (uint32_t)phases.size();
    std::vector<uint32_t> lrow_of(n, NONE);  // scratch: local row of a variable inside the block being emitted
    uint32_t lds_off = 0;
    for (size_t p = 0; p < phases.size(); ++p) {
        const Phase& ph = phases[p];
        tab[1 + p] = (uint32_t)tab.size();
        const size_t rec = tab.size();
        tab.push_back((uint32_t)ph.blocks.size());
        tab.resize(tab.size() + 5 * ph.blocks.size(), 0);
        for (size_t b = 0; b < ph.blocks.size(); ++b) {
            const Block& blk = ph.blocks[b];
            const uint32_t K = (uint32_t)blk.cols.size(), R = K + (uint32_t)blk.below.size() + 1, st = K | 1u;
            uint32_t* t = &tab[rec + 1 + 5 * b];
            t[0] = K, t[1] = R, t[2] = lds_off, t[3] = st;
            const uint32_t rv = (uint32_t)tab.size();
            tab[rec + 1 + 5 * b + 4] = rv;  // (t is stale after the pushes below)
            lds_off += R * st;
            uint32_t lr = 0;
            for (uint32_t j : blk.cols) lrow_of[j] = lr++, tab.push_back(j);
            for (uint32_t i : blk.below) lrow_of[i] = lr++, tab.push_back(i);
            for (uint32_t lc = 0; lc < K; ++lc) {
                const uint32_t j = blk.cols[lc];
                dcol[j - dc0] = (uint32_t)b | lc << 4;
                cutcol[j - dc0] = P.lvl_cptr[ph.la];
                for (uint32_t q = P.bwd_ptr[j]; q < P.bwd_ptr[j + 1]; ++q) {
                    const uint32_t sl = P.bwd_items[2 * q], i = P.bwd_items[2 * q + 1];
                    if (sl < ds0 || lrow_of[i] == NONE) return false;
                    dslot[sl - ds0] = (uint32_t)b | lc << 4 | lrow_of[i] << 8;
                }
            }
            for (uint32_t j : blk.cols) lrow_of[j] = NONE;
            for (uint32_t i : blk.below) lrow_of[i] = NONE;
        }
    }
    for (uint32_t sl = ds0; sl < zlo; ++sl)
        if (P.l_col[sl] < dc0) return false;  // (level-major numbering: the slots of the phases' columns are the last)
    // ---- every list of a phase keeps the terms of the columns before the phase, in their order ----------------------------------
    {
        std::vector<uint32_t> ptr(P.fwd_ptr.begin(), P.fwd_ptr.begin() + dc0 + 1), items(P.fwd_items.begin(), P.fwd_items.begin() + 2 * (size_t)P.fwd_ptr[dc0]);
        for (uint32_t j = dc0; j < n; ++j) {
            for (uint32_t q = P.fwd_ptr[j]; q < P.fwd_ptr[j + 1]; ++q)
                if (P.fwd_items[2 * q + 1] < cutcol[j - dc0]) items.push_back(P.fwd_items[2 * q]), items.push_back(P.fwd_items[2 * q + 1]);
            ptr.push_back((uint32_t)(items.size() / 2));
        }
        P.fwd_ptr.swap(ptr);
        P.fwd_items.swap(items);
    }
    {
        std::vector<uint32_t> ptr(P.lpair_ptr.begin(), P.lpair_ptr.begin() + ds0 + 1), items(P.lpairs.begin(), P.lpairs.begin() + 2 * (size_t)P.lpair_ptr[ds0]);
        for (uint32_t sl = ds0; sl < zlo; ++sl) {
            const uint32_t cutc = cutcol[P.l_col[sl] - dc0];
            for (uint32_t q = P.lpair_ptr[sl]; q < P.lpair_ptr[sl + 1]; ++q)
                if (P.l_col[P.lpairs[2 * q]] < cutc) items.push_back(P.lpairs[2 * q]), items.push_back(P.lpairs[2 * q + 1]);
            ptr.push_back((uint32_t)(items.size() / 2));
        }
        P.lpair_ptr.swap(ptr);
        P.lpairs.swap(items);
        P.c.n_lpairs = P.lpairs.size() / 2;
    }
    {  // the phases' backward substitution is dense: no lists
        const uint32_t keep = P.bwd_ptr[dc0];
        P.bwd_items.resize(2 * (size_t)keep);
        for (uint32_t j = dc0 + 1; j <= n; ++j) P.bwd_ptr[j] = keep;
    }
    {  // one level per phase
        std::vector<uint32_t> cptr(P.lvl_cptr.begin(), P.lvl_cptr.begin() + lw + 1), sptr(P.lvl_sptr.begin(), P.lvl_sptr.begin() + lw + 1);
        for (const Phase& ph : phases) cptr.push_back(P.lvl_cptr[ph.lb]), sptr.push_back(P.lvl_sptr[ph.lb]);
        P.lvl_cptr.swap(cptr);
        P.lvl_sptr.swap(sptr);
    }
    P.parts[0].nlev = lw + (uint32_t)phases.size();
    P.c.n_levels = P.parts[0].nlev;
    P.n_dense = (uint32_t)phases.size();
    P.dense_level0 = lw;
    P.dense_lds_doubles = lds_off;
    P.dense_col.swap(dcol);
    P.dense_slot.swap(dslot);
    P.dense_tab.swap(tab);
    return true;
}

// What EzpzSystemInfo says about a system that runs component-resident.
static void comp_info(EzpzSystemInfo& info, const CompPlan& plan) {
    info.team_mode = 3;
    info.team_size = plan.n_waves * 64;
    info.n_partitions = plan.n_chunks;
    info.workspace_bytes = plan.lds_bytes;
    info.workspace_in_lds = 1;
    info.program_in_lds = 0;
    info.grid_workgroups = 1;
}

// `may_defer`: a latency shape whose component plan is interpretable returns with that plan alone (EzpzSystem::program_deferred);
// `keep_comp`: the system already has its component plan (ensure_program: the deferred rest).
static int analyze_into(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint32_t team_size, EzpzSystem& s,
                        Program& P, std::vector<unsigned char>& blob, int32_t* err_constraint, int64_t* err_variable,
                        bool may_defer = false, bool keep_comp = false) {
    BuildError be;
    static const bool comp_enabled0 = [] {
        const char* e = std::getenv("EZPZ_COMP");
        return !(e && e[0] == '0');
    }();
    static const bool defer_enabled = [] {
        const char* e = std::getenv("EZPZ_DEFER");  // EZPZ_DEFER=0: every system is analysed whole at creation (A/B runs)
        return !(e && e[0] == '0');
    }();
    if (may_defer && (team_size == EZPZ_TEAM_AUTO_LATENCY || team_size == EZPZ_TEAM_LATENCY_WAVE) && comp_enabled0 && defer_enabled && !keep_comp) {
        std::unique_ptr<CompPlan> plan(new CompPlan());
        CompLimits cl;
        cl.lds_bytes = s.lim.lds_bytes;
        if (comp_plan_build(cs, n_cs, n_vars, cl, *plan) && plan->interpretable) {
            s.counts = ProgramCounts();
            s.counts.n_cons = plan->n_cons;
            s.counts.n_vars = plan->n_vars;
            s.counts.n_rows = plan->n_rows;
            s.unit_weights = plan->unit_weights;
            EzpzSystemInfo& info = s.info;
            std::memset(&info, 0, sizeof(info));
            info.n_constraints = plan->n_cons;
            info.n_vars = plan->n_vars;
            info.n_rows = plan->n_rows;
            info.program_bytes = plan->blob.size() * 4;
            comp_info(info, *plan);
            s.comp = std::move(plan);
            s.deferred_cs.assign(cs, cs + n_cs);
            s.program_deferred.store(true);
            blob.clear();
            return EZPZ_OK;
        }
    }
    const uint32_t width = (uint32_t)std::max<size_t>(1, std::max(n_cs, n_vars));
    auto fail = [&]() {
        if (err_constraint) *err_constraint = be.constraint;
        if (err_variable) *err_variable = be.variable;
        return be.code;
    };
    const bool latency_phases = team_size == EZPZ_TEAM_LATENCY_PHASES;
    if (team_size == EZPZ_TEAM_LATENCY_WAVE) team_size = EZPZ_TEAM_AUTO_LATENCY;  // (the wavefront form is chosen by ezpz_system_create)
    const bool for_latency = team_size == EZPZ_TEAM_AUTO_LATENCY || latency_phases;
    const bool batch_lanes = team_size == EZPZ_TEAM_BATCH_LANES;
    const bool auto_shape = team_size == 0 || for_latency || batch_lanes;
    const bool lists_only = team_size == EZPZ_TEAM_AUTO_LISTS;  // the list-walk shapes as they are chosen for batches, dense phases included
    if (for_latency || batch_lanes || lists_only) team_size = 0;
    bool want_sub = team_size ? team_size <= 64 : width <= 64;
    if (want_sub) {
        uint32_t team = team_size ? pow2_ceil(team_size) : auto_sub_team(cs, n_cs);
        // Systems of <= 8 variables (the typical sketch fixture) on teams of four lanes solve their normal equations
        // in registers: a level-by-level list walk costs an 8 x 8 system 27 k cycles per factorisation (~10 LDS hops
        // per level, 8 levels), the register version ~3 k.
        const bool allow_dense = n_vars <= 8 && n_vars >= 2 && (team_size == 0 || team == 4);
        if (!build_program(cs, n_cs, n_vars, P, be, 1, allow_dense)) return fail();
        const bool dense = P.c.dense != 0;  // granted only when JtJ is mostly full
        // (16 lanes for one solve -- sweeps and assembly in one round, the first quad factorising -- was measured: `square`
        // 108 -> 148 us per call, `parallelogram` 57 -> 76: the quads stay)
        if (dense) team = 4;
        // one solve of a system too large for the register solve: a whole wavefront (its levels run as one phase each, the
        // lists shared by groups of lanes, the top of the elimination tree as dense phases)
        // (not below 17: `square`, `two_rectangles` ... take as long on a wavefront with one dense phase as on their quads
        // with the register solve, 109 / 51 us per call: an iteration is a dozen phases of 1.5-3 k cycles either way)
        if (for_latency && !dense && width > 16) team = 64;
        // 64 / team workspaces share a wavefront: keep a wavefront's share of the LDS <= 32 KiB when choosing
        // automatically (>= 4 wavefronts per CU), and inside the hard limit in any case
        if (!team_size)
            while (team < 64 && (size_t)workspace_doubles(P.c) * 8 * (64 / team) > 32 * 1024) team <<= 1;
        while (team < 64 && !sub_team_fits(P.c, team)) team <<= 1;
        if (dense && team != 4) {  // many constraints on few variables pushed the team up: the list-walk build after all
            if (!build_program(cs, n_cs, n_vars, P, be, 1, false)) return fail();
        }
        // One connected system walks records (build_records) from 25 variables for one solve and from 57 in batches, on one
        // wavefront: one solve of 32 / 50 / 64 variables 61 -> 54, 142 -> 112, 119 -> 82 us; batches of 64 variables 17.0 -> 23.1 M
        // solves/s, but of 50 variables 19.2 -> 16.8 M and of 32 variables 60 -> 37 M (two to four systems share a wavefront
        // there).  EZPZ_REC_SMALL = that bound for both (A/B runs), 0 = the sub-wavefront teams always.
        static const int rec_small_env = [] {
            const char* e = std::getenv("EZPZ_REC_SMALL");
            return e ? std::atoi(e) : -1;
        }();
        const int rec_small = rec_small_env >= 0 ? rec_small_env
                                                 : (int)(for_latency ? s.lim.policy.rec_min_vars_one_solve : s.lim.policy.rec_min_vars_batch) - 1;
        static const bool rec_on = [] {
            const char* e = std::getenv("EZPZ_REC");
            return !(e && e[0] == '0');
        }();
        const bool walk_records = rec_on && rec_small > 0 && !team_size && !lists_only && !latency_phases && P.c.n_components == 1 &&
                                  !dense && (int)width > rec_small;
        if (sub_team_fits(P.c, team) && !walk_records) {
            s.mode = MODE_SUB;
            s.team_size = team;
        } else {
            want_sub = false;
            team_size = 0;
        }
    }
    if (!want_sub) {
        // the general build is compiled for <= 512 lanes (145 VGPRs), the linear-only build for <= 1024 (59 VGPRs)
        bool lin = true;
        for (size_t i = 0; i < n_cs; ++i) lin = lin && kind_is_linear(cs[i].kind);
        const uint32_t max_team = lin ? 1024 : 512;
        uint32_t team = team_size ? std::min<uint32_t>(max_team, (std::max<uint32_t>(team_size, 128) + 63) & ~63u)
                                  : auto_wg_team(width);
        // One large system whose state cannot live in a CU's LDS: spread it over G workgroups (a "grid team"), each
        // owning team/64 partitions, ~64+ variables per wavefront; falls back to one workgroup when the components
        // cannot be balanced over that many partitions.  Obviously large systems go straight to the grid build; the
        // others are built for one workgroup first and rebuilt only if their exact state turns out not to fit.
        const uint32_t W = team / 64;
        // (independent pieces of the system, by union-find over the constraints: a grid team needs at least two per
        // wavefront to balance, and a single connected sketch must not pay for build attempts that cannot succeed)
        size_t n_pieces = 0;
        {
            std::vector<uint32_t> parent(n_vars);
            for (size_t v = 0; v < n_vars; ++v) parent[v] = (uint32_t)v;
            auto find = [&](uint32_t a) {
                while (parent[a] != a) a = parent[a] = parent[parent[a]];
                return a;
            };
            std::vector<char> used(n_vars, 0);
            for (size_t i = 0; i < n_cs; ++i) {
                if (cs[i].kind >= EZPZ_NUM_KINDS) continue;
                const KindInfo& K = kKinds[cs[i].kind];
                uint32_t first = UINT32_MAX;
                for (int r = 0; r < K.n_rows; ++r)
                    for (int e = 0; e < K.n_nz[r]; ++e) {
                        const uint32_t v = cs[i].ids[K.nz[r][e]];
                        if (v >= n_vars) continue;  // reported by build_program
                        used[v] = 1;
                        if (first == UINT32_MAX)
                            first = v;
                        else
                            parent[find(v)] = find(first);
                    }
            }
            for (size_t v = 0; v < n_vars; ++v) n_pieces += (!used[v] || find((uint32_t)v) == v) ? 1 : 0;
        }
        auto grid_wgs_for = [&]() {
            uint32_t g = 1;
            // every workgroup of a grid team must be resident at once: never more of them than the device has CUs
            while (g < (uint32_t)kGridMaxWgs && g * 2 <= (uint32_t)s.lim.cus && (uint64_t)g * 2 * W * 64 <= n_vars &&
                   (uint64_t)g * 2 * W * 2 <= n_pieces)
                g <<= 1;
            return g;
        };
        auto build_grid = [&](uint32_t g0) -> int {  // > 1: workgroups of the grid team now in P; 0: none works; -1: error
            // unbalanced -> fewer, larger shares; a share too big for a CU's LDS -> more, smaller ones
            bool grow = false;
            for (uint32_t g = g0; g > 1 && g <= (uint32_t)kGridMaxWgs;) {
                Program Q;
                BuildError qe;
                if (!build_program(cs, n_cs, n_vars, Q, qe, g * W)) {
                    be = qe;
                    return -1;
                }
                if (Q.c.n_parts != g * W) {
                    if (grow) return 0;
                    g >>= 1;
                    continue;
                }
                if (pack_grid_slices(s, Q, g, W)) {
                    P = std::move(Q);
                    return (int)g;
                }
                grow = true;
                g <<= 1;
            }
            return 0;
        };
        uint32_t G = 1;
        bool have_program = false;
        if (!team_size && (3 * n_vars + 2 * n_cs) * 8 > s.lim.lds_bytes) {
            const int r = build_grid(grid_wgs_for());
            if (r < 0) return fail();
            if (r > 1) {
                G = (uint32_t)r;
                have_program = true;
            }
        }
        if (!have_program) {
            // A FEW components (a document of several sketches) are one partition for the record walk, which needs levels, not
            // connectivity: batches of 4 x 150 / 8 x 80 / 3 x 300 variables 0.64 -> 1.73, 0.68 -> 1.63, 0.57 -> 1.28 M solves/s against a
            // wavefront per balanced share of the components.  (Many small components are the component-resident shape's.)
            static const bool rec_multi = [] {
                const char* e = std::getenv("EZPZ_REC_MULTI");
                return !(e && e[0] == '0');
            }();
            static const bool rec_on2 = [] {
                const char* e = std::getenv("EZPZ_REC");
                return !(e && e[0] == '0');
            }();
            // (one solve of such a system too: 4 x 150 / 8 x 80 / 6 x 40 variables 790 -> 237, 727 -> 237, 224 -> 109 us)
            const bool few = rec_on2 && rec_multi && !team_size && !latency_phases && !lists_only && n_pieces >= 2 &&
                             n_pieces <= kRecMaxComponents;
            if (!build_program(cs, n_cs, n_vars, P, be, few ? 1u : W)) return fail();
            if (!team_size && (size_t)workspace_doubles(P.c) * 8 + 4096 > s.lim.lds_bytes && grid_wgs_for() > 1) {
                Program one = std::move(P);
                const int r = build_grid(grid_wgs_for());
                if (r < 0) return fail();
                if (r > 1)
                    G = (uint32_t)r;
                else
                    P = std::move(one);
            }
        }
        s.grid_wgs = G;
        s.mode = P.c.n_parts > 1 ? MODE_PART : MODE_WGB;
        // one partition: the lanes are not tied to partitions, and the levels' long lists are shared by groups of
        // lanes (choose_level_groups), so more lanes shorten every level (800 variables: 11.1 / 8.7 / 7.1 ms per 60
        // iterations on 128 / 256 / 512 lanes)
        if (!team_size && G == 1 && P.c.n_parts == 1) team = std::min<uint32_t>(512, std::max(team, pow2_ceil(width)));
        s.team_size = team;
        // One connected component is a chain of elimination levels, each a few dependent memory hops and a divide: a
        // workgroup's extra lanes mostly wait at its barriers.  For batches one wavefront per system (no barriers, 2-4
        // systems per CU) gives 2-2.6x the rate at 150-300 variables; one solve alone takes ~25 % longer that way.
        // Larger states (from ~250 variables) do better on a 128-lane workgroup that does NOT stage its lists whole (two
        // or three workgroups per CU instead of one; levels are staged one at a time).  Measured, one wavefront vs this:
        // 200 variables 3.06 / 3.02 M solves/s, 250: 1.66 / 1.92, 300: 1.04 / 1.39, 400: 0.56 / 0.76, 500: 0.20 / 0.28.
        if (!team_size && !for_latency && G == 1 && P.c.n_parts == 1) {
            const size_t wsb = (size_t)workspace_doubles(P.c) * 8;
            if (sub_team_fits(P.c, 64) && wsb <= 24 * 1024) {
                s.mode = MODE_SUB;
                s.team_size = 64;
            } else if (wsb <= 56 * 1024) {
                s.team_size = 128;
                s.lean_lds = true;
            }
        }
    }
    s.counts = P.c;
    s.unit_weights = true;
    s.linear_only = true;
    for (const DevCon& d : P.cons) {
        if (d.weight != 1.0) s.unit_weights = false;
        if (!kind_is_linear(d.kind)) s.linear_only = false;
    }
    s.host_var_of = P.var_of;
    s.host_row_of = P.row_of;
    s.host_slot_row = P.slot_row;
    s.host_slot_col = P.slot_col;

    choose_level_groups(P, s);
    // ---- pack the blob (pack_program) ----------------------------------------------------------------------------
    const bool small_counts = P.c.n_vars < 65536 && P.c.n_rows < 65536 && P.c.zj < 65536 && P.c.zlo < 65536 &&
                              P.c.n_apairs < 65536 && P.c.n_lpairs < 65536 && P.c.n_cons < 65536;
    ProgramView& v = s.view;
    size_t stage_bytes = 0;
    auto pack_and_shape = [&](bool may_stage, size_t panel_bytes = 0) {
        stage_bytes = 0;
        if (small_counts && may_stage) {
            const size_t lists_bytes = pack_program(P, true, s.mode != MODE_SUB, blob, v);
            const size_t ws_bytes = ((size_t)rec_ws_base(P.c, s.rec_jglobal) + s.rec_extra) * 8;
            if (s.mode == MODE_SUB) {
                if (blob.size() <= kProgLdsMax) stage_bytes = blob.size();  // lists and constraint table
            } else if (s.grid_wgs == 1 && v.packed && lists_bytes + ws_bytes + 2048 <= s.lim.lds_bytes && !s.lean_lds) {
                stage_bytes = lists_bytes;
            }
        }
        if (stage_bytes == 0) pack_program(P, false, false, blob, v);
        v.stage_bytes = (uint32_t)stage_bytes;
        s.lvl_nlev = P.parts.empty() ? 0 : P.parts[0].nlev;
        finish_team(s, stage_bytes, panel_bytes);
    };
    // ---- one solve of one connected system on a barrier workgroup: the linear solve as a record walk (build_records) -----------
    static const bool rec_enabled = [] {
        const char* e = std::getenv("EZPZ_REC");
        return !(e && e[0] == '0');
    }();
    RecPlan rec;
    s.rec = s.rec_wide = s.rec_jglobal = false;
    s.rec_rounds = 0;
    s.rec_extra = 0;
    // Batches of one connected sketch on the per-system teams take the record walk as well -- one wavefront up to 160 variables,
    // then 128 / 256 / 512 lanes as four / two or three / one workgroup fit a CU -- instead of one wavefront / a lean workgroup walking level lists
    // with dense phases on top: 100 / 150 / 200 / 300 / 500 / 800 variables 10.5 -> 23.0, 4.3 -> 8.7, 4.0 -> 7.5, 1.49 -> 3.10,
    // 0.33 -> 0.61 M solves/s, 57 -> 86 k (EZPZ_REC_BATCH = lanes for A/B runs, 0 = the shapes above).
    static const int rec_batch_lanes = [] {
        const char* e = std::getenv("EZPZ_REC_BATCH");
        return e ? std::atoi(e) : -1;
    }();
    const int saved_mode = s.mode;
    const uint32_t saved_team = s.team_size;
    const bool saved_lean = s.lean_lds;
    const bool rec_batch = rec_enabled && rec_batch_lanes != 0 && auto_shape && team_size == 0 && !for_latency && !want_sub &&
                           s.grid_wgs == 1 && P.c.n_parts == 1 && P.c.n_components >= 1 && P.c.n_components <= kRecMaxComponents && !P.c.dense;
    // Batches keep the Jacobian's values in global memory (SolveArgs::rec_jglobal) when the assembly can read them from packed
    // pairs (no list of more than twelve): a fifth of a system's LDS, one more workgroup per CU.  EZPZ_REC_JGLOBAL=0: in the LDS.
    static const bool jglobal_enabled = [] {
        const char* e = std::getenv("EZPZ_REC_JGLOBAL");
        return !(e && e[0] == '0');
    }();
    bool jglobal = rec_batch && jglobal_enabled && P.c.zj < 65535 && !P.parts.empty();
    if (jglobal) {
        const uint32_t l0 = P.parts[0].lvl0, nl = P.parts[0].nlev;
        for (uint32_t v = P.lvl_cptr[l0]; v < P.lvl_cptr[l0 + nl] && jglobal; ++v) jglobal = P.colj_ptr[v + 1] - P.colj_ptr[v] <= 12;
        for (uint32_t sl = P.lvl_sptr[l0]; sl < P.lvl_sptr[l0 + nl] && jglobal; ++sl) jglobal = P.apair_ptr[sl + 1] - P.apair_ptr[sl] <= 12;
    }
    if (rec_batch) {
        // (about eight wavefronts per CU: 300 variables, four workgroups per CU, 3.19 M solves/s on 128 lanes against 2.94 M on 256;
        // 500 variables, two per CU, 0.68 against 0.91 M; 800 variables, one per CU, 53 k / 73 k / 91 k on 128 / 256 / 512 lanes)
        auto shape_for = [&](bool jg, uint32_t& per_cu) {
            const size_t ws_b = ((size_t)rec_ws_base(P.c, jg) + P.c.n_vars + 4) * 8;
            per_cu = (uint32_t)std::max<size_t>(1, s.lim.lds_bytes / (ws_b + 4096));
            return P.c.n_vars <= s.lim.policy.rec_one_wavefront_max_vars ? 64u : std::min(512u, std::max(128u, pow2_ceil(512u / per_cu)));
        };
        uint32_t per_cu = 1, per_cu_j = 1;
        uint32_t t = shape_for(false, per_cu);
        if (jglobal) {
            // ... and J out of the LDS where that puts more wavefronts on a CU, or as many in more systems: 300 / 500 / 800 variables
            // 3.15 -> 3.68, 0.90 -> 1.16 M solves/s, 88 -> 114 k (four -> five, two -> three, one -> two workgroups per CU); not at 400
            // (three of 256 lanes -> four of 128: 2.48 -> 2.34 M), 600 (two either way: 1.44 -> 1.33 M) or on one wavefront (100: -8 %)
            const uint32_t tj = shape_for(true, per_cu_j);
            const uint32_t w = per_cu * t, wj = per_cu_j * tj;
            jglobal = P.c.n_vars > 160 && (wj > w || (wj == w && per_cu_j > per_cu));
            if (jglobal) t = tj;
        }
        if (rec_batch_lanes >= 64 && rec_batch_lanes <= 512 && rec_batch_lanes % 64 == 0) t = (uint32_t)rec_batch_lanes;
        s.mode = MODE_WGB;
        s.team_size = t;
        s.lean_lds = true;  // (its lists stay in L2: the LDS is for as many systems as fit)
    }
    const bool rec_try = rec_enabled && auto_shape && ((for_latency && !latency_phases) || rec_batch) && s.mode == MODE_WGB &&
                         s.grid_wgs == 1 && P.c.n_parts == 1 && P.c.n_components >= 1 && P.c.n_components <= kRecMaxComponents && !P.c.dense;
    if (rec_try) {
        s.rec_jglobal = jglobal;
        s.rec_extra = (P.c.n_vars + 2 + 1) & ~1u;
#ifdef EZPZ_REC_TIMES
        s.rec_extra += 6 * 128;  // (diagnostic build: six cycle stamps per round of the second iteration's walk, behind the zero)
#endif
        if (const char* e = std::getenv("EZPZ_REC_LANES")) {  // (A/B runs)
            const uint32_t t = (uint32_t)std::atoi(e);
            if (t >= 64 && t <= 512 && t % 64 == 0) s.team_size = t;
        }
    }
    pack_and_shape(true);
    bool rec_wide = false;
    if (rec_try && !s.lds_ws) {
        // no room in the LDS with the walk's extra doubles: if the state fits without them the list walk keeps it there (with its dense
        // phases); a state that lives in global memory anyway walks records in the wide form
        static const bool wide_enabled = [] {
            const char* e = std::getenv("EZPZ_REC_WIDE");
            return !(e && e[0] == '0');
        }();
        const uint32_t extra = s.rec_extra;
        s.rec_extra = 0;
        s.rec_jglobal = false;  // (J in global memory is for states that fit the LDS with it)
        pack_and_shape(true);
        // (batches: 4194 systems of 2000 variables 106 -> 114 k solves/s, 1677 of 5000 variables 6.3 -> 8.4 k; a round through
        // global memory is a store's acknowledgement, a rendezvous and a trip to L2)
        static const uint32_t wide_one_solve_max = [] {  // (A/B runs: one solve walks wide records up to this many variables)
            const char* e = std::getenv("EZPZ_REC_WIDE_LATENCY");
            return e ? (uint32_t)std::atol(e) : launch_policy_for(256).rec_wide_one_solve_max_vars;
        }();
        // (one solve of 1600 / 2000 / 3000 / 4000 / 5000 variables: 1.23 -> 1.16, 2.02 -> 1.82, 1.77 -> 1.49, 2.47 -> 2.47, 25.0 -> 26.9 ms)
        if (!s.lds_ws && wide_enabled && (rec_batch || P.c.n_vars <= wide_one_solve_max)) {
            rec_wide = true;
            s.rec_extra = extra;
            pack_and_shape(true);
        }
    }
    if (rec_try) {
        if ((s.lds_ws || rec_wide) && s.rec_extra && build_records(P, s.team_size, rec_wide ? 0u : s.prog_lds_doubles, rec_wide, s.rec_jglobal, rec) &&
            (!s.rec_jglobal || rec.asm_kc)) {
            s.rec = true;
            s.rec_rounds = rec.rounds;
            s.rec_desc_lds_off = (uint32_t)((s.lds_bytes + 15) / 16 * 2);  // the descriptors' copy in LDS, behind everything else
            s.lds_bytes = (size_t)s.rec_desc_lds_off * 8 + rec.desc.size() * 4;
            if (s.lds_bytes > s.lim.lds_bytes) s.rec = false;
        }
        s.rec_wide = s.rec && rec_wide;
        if (!s.rec) {
            s.rec_extra = 0;
            s.rec_jglobal = false;
            if (rec_batch) {  // no room: the shape chosen before
                s.mode = saved_mode;
                s.team_size = saved_team;
                s.lean_lds = saved_lean;
            }
            pack_and_shape(true);
        }
    }
    // ---- dense phases: the top of a connected sketch's elimination tree on a barrier workgroup ------------------------------
    s.n_dense = s.dense_level0 = s.dense_lds_off = s.dense_lds_doubles = 0;
    static const bool root_enabled = [] {
        const char* e = std::getenv("EZPZ_ROOT");
        return !(e && e[0] == '0');
    }();
    // (256-512 lanes: 800 variables 41.7 -> 49.7 k/s, 2000: 90 -> 97 k/s; the lean 128-lane batch shape only out of the LDS
    // slack that keeps its workgroups per CU: with 6 KB of panels 300 variables fell 1.50 -> 1.32 M solves/s, 400 rose
    // 0.83 -> 0.92)
    // One wavefront per system (batches of 100-220 variables): every team of the workgroup has its own panels, at most 6 KB.
    // ... out of the LDS its workgroup leaves unused at the number of workgroups a CU holds now: the panels must not cost a
    // batch its occupancy (150 variables: 4 workgroups of 2 teams -> 3 with 6 KB of panels per team: -5 % despite the
    // shorter solve; with the root block alone in the 2.5 KB of slack per team the count stays).
    const bool wave_teams = s.mode == MODE_SUB && s.team_size == 64;
    const uint32_t teams_now = wave_teams ? s.block_threads / 64 : 1;
    size_t dense_room = 0;
    if (!for_latency && (wave_teams || (s.mode == MODE_WGB && s.team_size < 256))) {  // (the lean 128-lane batch shape as well)
        const size_t per_cu = std::max<size_t>(1, s.lim.lds_bytes / std::max<size_t>(s.lds_bytes, 1));
        const size_t slack = s.lim.lds_bytes / per_cu > s.lds_bytes + 64 ? s.lim.lds_bytes / per_cu - s.lds_bytes - 64 : 0;
        dense_room = slack / teams_now;
    } else if (s.lds_bytes + 4096 * teams_now <= s.lim.lds_bytes) {  // (one solve: occupancy does not matter)
        dense_room = std::min<size_t>((s.lim.lds_bytes - s.lds_bytes - 1024) / teams_now, 48 * 1024);
    }
    if (root_enabled && (auto_shape || lists_only) && s.grid_wgs == 1 && !s.rec &&
        ((s.mode == MODE_WGB && (for_latency || s.team_size >= 128)) || wave_teams) && dense_room >= 1024 &&
        make_dense_phases(P, wave_teams ? 1 : s.team_size / 64, dense_room)) {
        const int mode_before = s.mode;
        choose_level_groups(P, s);
        s.counts = P.c;
        // (the lists only got shorter: the same shape again -- but a program that did not fit the LDS beside its workspace
        // before must not move in now and take the panels' room; a workgroup of wavefront teams is sized with its panels)
        pack_and_shape(stage_bytes > 0, wave_teams ? (size_t)P.dense_lds_doubles * 8 : 0);
        const uint32_t teams = wave_teams ? s.block_threads / 64 : 1;
        {  // the level staging buffer is optional space (levels wider than it are walked in place): the panels come first
            const size_t need = s.lds_bytes + (size_t)P.dense_lds_doubles * 8 * teams + 64;
            if (need > s.lim.lds_bytes && s.mode == MODE_WGB && s.lvl_buf_words) {
                const size_t over = (need - s.lim.lds_bytes + 15) & ~size_t(15);
                if ((size_t)s.lvl_buf_words * 4 >= over + 1024) {
                    s.lvl_buf_words -= (uint32_t)(over / 4);
                    s.lds_bytes -= over;
                } else {  // no staging at all: its tables and buffer go
                    s.lds_bytes = (size_t)s.lvl_lds_off * 8;
                    s.lvl_lds_off = s.lvl_tab_words = s.lvl_buf_words = 0;
                }
            }
        }
        if (s.mode == mode_before && s.lds_bytes + (size_t)P.dense_lds_doubles * 8 * teams + 64 <= s.lim.lds_bytes) {
            s.n_dense = P.n_dense;
            s.dense_level0 = P.dense_level0;
            s.dense_lds_doubles = P.dense_lds_doubles;
            s.dense_lds_off = (uint32_t)((s.lds_bytes + 15) / 16 * 2);
            s.lds_bytes = (size_t)s.dense_lds_off * 8 + (size_t)P.dense_lds_doubles * 8 * teams;
        } else {
            if (std::getenv("EZPZ_DENSE_DEBUG"))
                std::fprintf(stderr, "dense phases: mode %d -> %d, threads %u, lds %zu + %zu x %u of %zu\n", mode_before, (int)s.mode,
                             s.block_threads, s.lds_bytes, (size_t)P.dense_lds_doubles * 8, teams, s.lim.lds_bytes);
            be.code = EZPZ_ERR_TOO_LARGE;  // cannot happen: the same program with shorter lists
            return fail();
        }
    }
    if (s.rec) {
        s.rec_desc_off = append(blob, rec.desc);
        s.rec_chunks_off = append(blob, rec.chunks);
        s.rec_asm_kc = rec.asm_kc;
        s.rec_asm_ks = rec.asm_ks;
        if (rec.asm_kc) {
            s.rec_asm_cols_off = append(blob, rec.asm_cols);
            s.rec_asm_slots_off = append(blob, rec.asm_slots);
        }
        if (std::getenv("EZPZ_REC_DEBUG"))
            std::fprintf(stderr, "record walk: %u rounds on %u lanes, %zu KB of descriptors, %zu KB of records\n", rec.rounds, s.team_size,
                         rec.desc.size() * 4 / 1024, rec.chunks.size() * 4 / 1024);
    }
    if (blob.size() > 0xFFFFFFF0ull) {
        be.code = EZPZ_ERR_TOO_LARGE;
        return fail();
    }

    EzpzSystemInfo& info = s.info;
    std::memset(&info, 0, sizeof(info));
    info.n_constraints = P.c.n_cons;
    info.n_vars = P.c.n_vars;
    info.n_rows = P.c.n_rows;
    info.nnz_j = P.c.zj;
    info.nnz_a = P.c.za;
    info.nnz_l = (uint64_t)P.c.zlo + P.c.n_vars;
    info.n_levels = P.c.n_levels;
    info.n_components = P.c.n_components;
    info.program_bytes = blob.size();
    info.workspace_bytes = (uint64_t)s.ws_doubles * 8;
    info.team_size = s.team_size;
    info.workspace_in_lds = s.lds_ws ? 1 : 0;
    info.team_mode = s.rec ? 4u : (uint32_t)s.mode;
    info.n_partitions = P.c.n_parts;
    info.program_in_lds = s.prog_in_lds ? 1 : 0;
    info.grid_workgroups = s.grid_wgs;

    // ---- component-resident launch shape ------------------------------------------------------------------------------
    // Systems of many small independent components (>= 128 of them, in a few isomorphism classes, state within the LDS)
    // run one lane per component instead of walking per-system lists; chosen automatically only (an explicit team size
    // asks for one of the list-walk shapes; EZPZ_COMP=0 in the environment turns the shape off for A/B runs).
    const bool comp_enabled = comp_enabled0;
    if (keep_comp && s.comp) {
        info.program_bytes += s.comp->blob.size() * 4;
        if (s.comp->interpretable) comp_info(info, *s.comp);
    } else if (auto_shape && comp_enabled) {
        s.comp.reset();
        std::unique_ptr<CompPlan> plan(new CompPlan());
        CompLimits cl;
        cl.lds_bytes = s.lim.lds_bytes;
        const bool planned = comp_plan_build(cs, n_cs, n_vars, cl, *plan);
        if (planned && !plan->interpretable) {
            // too much state for the interpreter: the list-walk shape chosen above serves until (and unless) the
            // specialised multi-workgroup kernel is compiled
            info.program_bytes += plan->blob.size() * 4;
            s.comp = std::move(plan);
        } else if (planned) {
            comp_info(info, *plan);
            info.program_bytes += plan->blob.size() * 4;
            s.comp = std::move(plan);
        }
    } else {
        s.comp.reset();
    }
    // ---- one lane per system: small systems that are not block systems (sub-wavefront teams otherwise) -----------------------
    s.lane.reset();
    if (auto_shape && comp_enabled && !s.comp && s.mode == MODE_SUB) {
        std::unique_ptr<LanePlan> lp(new LanePlan());
        if (lane_plan_build(cs, n_cs, n_vars, *lp)) s.lane = std::move(lp);
    }
    // ---- lanes across the batch: one connected sketch too large for a lane's registers.  A lane walks its system alone, every
    //      operand a trip to L2 / HBM, so the shape pays once the batch gives half of a CU's SIMDs a wavefront (64 x 2 x CUs
    //      systems: 32 768 on the MI355X; measured at 16 384 / 24 576 / 32 768 / 65 536 / 262 144 systems of 300 variables:
    //      1.06 / 1.53 / 2.00 / 3.70 / 9.4 M solves/s against the teams' 1.50 M); smaller batches keep the teams.
    s.lanes.reset();
    // (sketches of up to 64 variables: twice that -- their rounds are short, the lanes' time is a latency floor of ~3 ms
    // whatever the batch, and the teams run them at 12-19 M solves/s: 32 768 systems of 50 variables 9.8 M/s on the lanes)
    // (and up to 600 variables since the teams walk records: 32 768 systems of 100 / 150 / 300 / 500 variables 15.2 / 5.9 / 2.3 / 0.69
    // M solves/s on the lanes, 22.4 / 8.6 / 3.1 / 0.61 on the teams; 65 536: 26.4 / 10.5 / 4.3 / 1.27 against 22.8 / 8.7 / 3.1 / 0.61)
    // (500 variables: 0.91 M on the teams whatever the batch, 0.69 / 1.27 M on the lanes at 32 768 / 65 536)
    s.lanes_min = batch_lanes ? 1 : n_vars < s.lim.policy.lanes_large_from_vars ? s.lim.policy.lanes_min_systems_small
                                                                                 : s.lim.policy.lanes_min_systems_large;
    static const bool lanes_enabled = [] {
        const char* e = std::getenv("EZPZ_LANES");
        return !(e && e[0] == '0');
    }();
    if (auto_shape && lanes_enabled && !s.comp && !s.lane && s.grid_wgs == 1 && P.c.n_parts == 1 && n_vars > 20) {
        std::unique_ptr<BatchPlan> bp(new BatchPlan());
        if (batch_plan_build(cs, n_cs, n_vars, *bp)) s.lanes = std::move(bp);
    }
    return EZPZ_OK;
}

int ezpz_system_create(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, int device, uint32_t team_size,
                       EzpzSystem** out, int32_t* err_constraint, int64_t* err_variable) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<EzpzSystem> s(new EzpzSystem());
    Program P;
    std::vector<unsigned char> blob;
    // launch shapes are sized for the device the system will live on; request errors (MissingGuess ...) are reported
    // before the absence of a device, like the reference reports them before any numeric work
    const bool have_device = device >= 0 && ezpz_device_count() > device;
    if (have_device) {
        s->device = device;
        s->lim = device_limits(device);
    }
    int rc = analyze_into(cs, n_cs, n_vars, team_size, *s, P, blob, err_constraint, err_variable, /*may_defer=*/true);
    call_stamp(COLD_ANALYSED);
    if (rc != EZPZ_OK) return rc;
    if (!have_device) return EZPZ_ERR_NO_DEVICE;
    HIP_TRY(hipSetDevice(device));
    if (!s->program_deferred.load()) {
        HIP_TRY(hipMalloc(&s->dev_program, blob.size()));
        HIP_TRY(hipMemcpy(s->dev_program, blob.data(), blob.size(), hipMemcpyHostToDevice));
        s->view.base = static_cast<const unsigned char*>(s->dev_program);
    }
    if (s->comp) {
        HIP_TRY(hipMalloc((void**)&s->dev_comp, s->comp->blob.size() * 4));
        HIP_TRY(hipMemcpy(s->dev_comp, s->comp->blob.data(), s->comp->blob.size() * 4, hipMemcpyHostToDevice));
        call_stamp(COLD_UPLOADED);
        s->jit = comp_jit_create(*s->comp);
        call_stamp(COLD_KERNEL_FOUND);
    } else if (s->lane) {
        s->jit = comp_jit_create_source(s->lane->jit_source, "ezpz_jit_lane");
        static const bool wave_enabled = [] {
            const char* e = std::getenv("EZPZ_JIT_WAVE");  // EZPZ_JIT_WAVE=0: one solve of a small system stays on one lane (A/B runs)
            return !(e && e[0] == '0');
        }();
        // one solve of a small system on one wavefront per system: where the sweeps are worth spreading (measured: `square`,
        // 8 variables / 4 non-linear constraints, 47 -> 40 us; 14 variables / 4 distances 63 -> 53; arc_radius, 8 / 1, 7.2 -> 8.5;
        // a 4-variable linear system 12 -> 16) -- or always, when the caller asks for the form (EZPZ_TEAM_LATENCY_WAVE)
        size_t non_linear = 0;
        for (size_t i = 0; i < n_cs; ++i) non_linear += kind_is_linear(cs[i].kind) ? 0 : 1;
        const bool pays = n_vars >= 8 && non_linear >= 3;
        if (wave_enabled && !s->lane->wave_source.empty() &&
            (team_size == EZPZ_TEAM_LATENCY_WAVE || (team_size == EZPZ_TEAM_AUTO_LATENCY && pays)))
            s->wave_jit = comp_jit_create_source(s->lane->wave_source, "ezpz_jit_wave");
    }
    if (s->lanes) {
        HIP_TRY(hipMalloc((void**)&s->dev_lanes, s->lanes->blob.size() * 4));
        HIP_TRY(hipMemcpy(s->dev_lanes, s->lanes->blob.data(), s->lanes->blob.size() * 4, hipMemcpyHostToDevice));
    }
    *out = s.release();
    return EZPZ_OK;
}

}  // extern "C"

// The rest of a deferred analysis (EzpzSystem::program_deferred): the list-walk program of the whole system, for
// evaluation, FreedomAnalysis and the sizes of EzpzSystemInfo.  Solves never wait for it.
static void dismiss_resident_of(EzpzSystem* sys);  // (the one-call section below)

static int ensure_program(EzpzSystem* sys) {
    if (!sys->program_deferred.load(std::memory_order_acquire)) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->defer_mu);
    if (!sys->program_deferred.load()) return EZPZ_OK;
    Program P;
    std::vector<unsigned char> blob;
    int rc = analyze_into(sys->deferred_cs.data(), sys->deferred_cs.size(), sys->counts.n_vars, EZPZ_TEAM_AUTO_LATENCY, *sys, P, blob,
                          nullptr, nullptr, false, /*keep_comp=*/true);
    if (rc != EZPZ_OK) return rc;
    if (sys->device >= 0) {
        HIP_TRY(hipSetDevice(sys->device));
        HIP_TRY(hipMalloc(&sys->dev_program, blob.size()));
        HIP_TRY(hipMemcpy(sys->dev_program, blob.data(), blob.size(), hipMemcpyHostToDevice));
        sys->view.base = static_cast<const unsigned char*>(sys->dev_program);
    }
    sys->deferred_cs.clear();
    sys->deferred_cs.shrink_to_fit();
    sys->program_deferred.store(false, std::memory_order_release);
    return EZPZ_OK;
}

extern "C" {

void ezpz_system_destroy(EzpzSystem* sys) {
    if (!sys) return;
    // the system's allocations are freed with its device current; the caller's own current device is put back (a torch
    // caller on an 8-GPU node must not find itself on device 7 after a handle was collected)
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) {
        (void)hipGetLastError();
        prev = -1;
    }
    const int device = sys->device;
    if (device >= 0) (void)hipSetDevice(device);
    dismiss_resident_of(sys);
    delete sys;
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
}

int ezpz_system_info(const EzpzSystem* sys, EzpzSystemInfo* info) {
    if (!sys || !info) return EZPZ_ERR_INVALID_ARGUMENT;
    if (int rc = ensure_program(const_cast<EzpzSystem*>(sys))) return rc;
    *info = sys->info;
    return EZPZ_OK;
}

int ezpz_analyze(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, EzpzSystemInfo* info, int32_t* err_constraint,
                 int64_t* err_variable) {
    if (!info) return EZPZ_ERR_INVALID_ARGUMENT;
    EzpzSystem tmp;
    Program P;
    std::vector<unsigned char> blob;
    int rc = analyze_into(cs, n_cs, n_vars, 0, tmp, P, blob, err_constraint, err_variable);
    if (rc != EZPZ_OK) return rc;
    *info = tmp.info;
    return EZPZ_OK;
}

int ezpz_system_jacobian_pattern(const EzpzSystem* sys, uint32_t* rows, uint32_t* cols) {
    if (!sys || !rows || !cols) return EZPZ_ERR_INVALID_ARGUMENT;
    if (int rc = ensure_program(const_cast<EzpzSystem*>(sys))) return rc;
    for (uint32_t s = 0; s < sys->counts.zj; ++s) {
        rows[s] = sys->host_slot_row[s];
        cols[s] = sys->host_slot_col[s];
    }
    return EZPZ_OK;
}

int ezpz_system_eval_batch(EzpzSystem* sys, const double* x, size_t batch, double* r_out, double* jv_out,
                           uint32_t* degenerate_count_out) {
    if (!sys || !x || !r_out || !jv_out) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    release_thread_kernel(sys->device);
    if (int rc0 = ensure_program(sys)) return rc0;
    std::lock_guard<std::mutex> lock(sys->mu);
    HIP_TRY(hipSetDevice(sys->device));
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

}  // extern "C"

// (done: the completion word of a one-call launch, system_solve_one; null for every other caller)
// (`resident`: whether the launch stays on the device for further requests, DoneWord::request)
static int solve_batch_device_impl(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg, double* x_out_dev,
                                   EzpzStatus* status_dev, uint8_t* unsat_mask_dev, uint64_t* warn_log_dev, uint32_t warn_cap,
                                   void* stream, const DoneWord& done, bool* resident = nullptr) {
    if (!sys || (batch && (!x_out_dev || !status_dev))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch && sys->counts.n_vars && !x0_dev) return EZPZ_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(sys->device));
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
        static const bool packed = [] {  // (A/B runs)
            const char* e = std::getenv("EZPZ_REC_ASM");
            return !(e && e[0] == '0');
        }();
        if ((packed || sys->rec_jglobal) && sys->rec_asm_kc) {
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

extern "C" {

int ezpz_system_solve_batch_device(EzpzSystem* sys, const double* x0_dev, size_t batch, const EzpzConfig* cfg,
                                   double* x_out_dev, EzpzStatus* status_dev, uint8_t* unsat_mask_dev,
                                   uint64_t* warn_log_dev, uint32_t warn_cap, void* stream) {
    if (sys) release_thread_kernel(sys->device);
    return solve_batch_device_impl(sys, x0_dev, batch, cfg, x_out_dev, status_dev, unsat_mask_dev, warn_log_dev, warn_cap, stream,
                                   DoneWord{nullptr, 0, nullptr});
}

int ezpz_system_solve_batch(EzpzSystem* sys, const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out,
                            EzpzStatus* status, uint8_t* unsat_mask, uint64_t* warn_log, uint32_t warn_cap) {
    if (!sys) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    release_thread_kernel(sys->device);
    std::lock_guard<std::mutex> lock(sys->mu);
    HIP_TRY(hipSetDevice(sys->device));
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
        if (sys->grid_wgs > 1)
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
        static const bool h2h_debug = std::getenv("EZPZ_H2H_DEBUG") != nullptr;
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
            if (hipEventRecord(P.solved[sl], P.run) != hipSuccess || hipStreamWaitEvent(P.out, P.solved[sl], 0) != hipSuccess ||
                hipMemcpyAsync(x_out + off * n, xd, nb * row, hipMemcpyDeviceToHost, P.out) != hipSuccess ||
                (st_registered && hipMemcpyAsync(status + off, sys->st_dev.p + off, nb * sizeof(EzpzStatus), hipMemcpyDeviceToHost, P.out) != hipSuccess) ||
                hipEventRecord(P.left[sl], P.out) != hipSuccess)
                return drain(EZPZ_ERR_HIP);
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
        if (sys->grid_wgs > 1 || (sys->comp && sys->comp->jit_wgs > 1))  // a system spread over several workgroups: its rendezvous can time out
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
    if ((rc = sys->st_dev.ensure(piece)) != EZPZ_OK) return rc;
    if (unsat_mask && (rc = sys->mask_dev.ensure(piece * std::max<size_t>(C, 1))) != EZPZ_OK) return rc;
    if (want_log && (rc = sys->log_dev.ensure(piece * (size_t)warn_cap)) != EZPZ_OK) return rc;
    for (size_t off = 0; off < batch; off += piece) {
        const size_t nb = std::min(piece, batch - off);
        if (n) HIP_TRY(hipMemcpy(sys->x_dev.p, x0 + off * n, nb * n * sizeof(double), hipMemcpyHostToDevice));
        rc = ezpz_system_solve_batch_device(sys, sys->x_dev.p, nb, cfg, sys->x_dev.p, sys->st_dev.p,
                                            unsat_mask ? sys->mask_dev.p : nullptr, want_log ? sys->log_dev.p : nullptr,
                                            warn_cap, nullptr);
        if (rc != EZPZ_OK) return rc;
        HIP_TRY(hipMemcpy(status + off, sys->st_dev.p, nb * sizeof(EzpzStatus), hipMemcpyDeviceToHost));
        if (sys->grid_wgs > 1)
            for (size_t b2 = 0; b2 < nb; ++b2)
                if (status[off + b2].iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT) return EZPZ_ERR_HIP;
        if (n) HIP_TRY(hipMemcpy(x_out + off * n, sys->x_dev.p, nb * n * sizeof(double), hipMemcpyDeviceToHost));
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
    HIP_TRY(hipSetDevice(sys->device));
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
    if (status->iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT && (sys->grid_wgs > 1 || (sys->comp && sys->comp->jit_wgs > 1))) return EZPZ_ERR_HIP;
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
static void dismiss_resident_of(EzpzSystem* sys) {
    if (sys->device < 0) return;
    CallBufs& cb = t_call[sys->device & 15];
    if (cb.res_alive && cb.res_sys == sys) dismiss_resident(cb);
}

extern "C" {

int ezpz_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return EZPZ_ERR_INVALID_ARGUMENT;
    if (ezpz_device_count() < 1) return EZPZ_ERR_NO_DEVICE;
    if (hipHostRegister(p, bytes, hipHostRegisterPortable) != hipSuccess) {
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

int ezpz_system_specialize(EzpzSystem* sys, int wait) {
    if (!sys) return EZPZ_ERR_INVALID_ARGUMENT;
    if (!sys->jit) return 0;
    const int st = comp_jit_request(sys->jit, wait != 0);
    if (st == kJitBudgetExhausted) return EZPZ_ERR_KERNEL_BUDGET;
    // (a latency-shaped small system has a second kernel -- one wavefront per system: requested along; the state returned is
    // the first one's unless the second failed)
    if (st >= 0 && sys->wave_jit) {
        const int sw = comp_jit_request(sys->wave_jit, wait != 0);
        if (sw < 0 && sw != kJitBudgetExhausted) return EZPZ_ERR_HIP;
    }
    return st < 0 ? EZPZ_ERR_HIP : st;
}

long ezpz_specialized_source(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, int compile, char* buf, size_t cap) {
    CompPlan plan;
    CompLimits cl;
    LanePlan lane;
    std::string text;
    const bool wave = (compile & 4) != 0;  // the one-wavefront-per-system form of a small system instead of its lane form
    compile &= 3;
    if (comp_plan_build(cs, n_cs, n_vars, cl, plan))
        text = wave ? std::string() : plan.jit_source;
    else if (lane_plan_build(cs, n_cs, n_vars, lane))
        text = wave ? lane.wave_source : lane.jit_source;
    if (text.empty()) return 0;
    const std::string source = text;
    long rc = (long)text.size();
    if (compile) {
        std::vector<char> code;
        std::string log;
        // compile == 2: through the on-disk cache of code objects, like the solve entry points; otherwise a real compilation
        if ((compile == 2 ? comp_jit_compile(source, code, log) : comp_jit_compile_uncached(source, code, log)) != EZPZ_OK) {
            text = log;
            rc = EZPZ_ERR_HIP;
        }
    }
    if (buf && cap) {
        const size_t k = std::min(cap - 1, text.size());
        std::memcpy(buf, text.data(), k);
        buf[k] = 0;
    }
    return rc;
}

#ifdef EZPZ_STAMPS
void ezpz_debug_set_stamps(unsigned long long* dev_buf) { g_stamps = dev_buf; }
#endif

}  // extern "C"

// ---- FreedomAnalysis (solver/find_dof.rs, analysis.rs) ----------------------------------------------------------------
namespace {

// Connected components of the Jacobian's row/variable graph and, per component, the dense placement of its slots.
int build_freedom(EzpzSystem* sys) {
    auto& F = sys->freedom;
    if (F.built) return EZPZ_OK;
    const uint32_t n = sys->counts.n_vars, m = sys->counts.n_rows, zj = sys->counts.zj;
    std::vector<uint32_t> parent(n + m);
    for (uint32_t i = 0; i < n + m; ++i) parent[i] = i;
    auto find = [&](uint32_t a) {
        while (parent[a] != a) {
            parent[a] = parent[parent[a]];
            a = parent[a];
        }
        return a;
    };
    for (uint32_t s = 0; s < zj; ++s) {
        uint32_t a = find(sys->host_slot_col[s]), b = find(n + sys->host_slot_row[s]);
        if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
    std::vector<uint32_t> comp_of(n + m, UINT32_MAX), lidx(n + m, 0);
    std::vector<FreedomComp> comps;
    std::vector<uint32_t> col_count(n, 0);
    for (uint32_t s = 0; s < zj; ++s) col_count[sys->host_slot_col[s]]++;
    for (uint32_t v = 0; v < n; ++v) {  // components in order of their smallest variable; local columns ascending
        if (!col_count[v]) continue;
        uint32_t r = find(v);
        if (comp_of[r] == UINT32_MAX) {
            comp_of[r] = (uint32_t)comps.size();
            comps.push_back(FreedomComp{0, 0, 0, 0, 0, 0});
        }
        comp_of[v] = comp_of[r];
        lidx[v] = comps[comp_of[v]].n++;
    }
    for (uint32_t r = 0; r < m; ++r) {
        uint32_t root = find(n + r);
        if (comp_of[root] == UINT32_MAX) continue;  // a row without entries
        comp_of[n + r] = comp_of[root];
        lidx[n + r] = comps[comp_of[root]].m++;
    }
    uint32_t var_total = 0, ws = 0, max_n = 0;
    for (auto& c : comps) {
        c.var0 = var_total;
        var_total += c.n;
        const uint64_t w = (uint64_t)c.m * c.n + 2ull * c.n * c.n + 2ull * c.n;
        if (w > (1ull << 31)) return EZPZ_ERR_TOO_LARGE;
        ws = std::max<uint32_t>(ws, (uint32_t)w);
        max_n = std::max(max_n, c.n);
    }
    std::vector<uint32_t> comp_vars(std::max<uint32_t>(var_total, 1));
    for (uint32_t v = 0; v < n; ++v)
        if (col_count[v]) comp_vars[comps[comp_of[v]].var0 + lidx[v]] = v;
    // slots grouped by component
    std::vector<uint32_t> per_comp(comps.size() + 1, 0);
    for (uint32_t s = 0; s < zj; ++s) per_comp[comp_of[sys->host_slot_col[s]] + 1]++;
    for (size_t c = 0; c < comps.size(); ++c) per_comp[c + 1] += per_comp[c];
    for (size_t c = 0; c < comps.size(); ++c) {
        comps[c].item0 = per_comp[c];
        comps[c].item1 = per_comp[c];
    }
    std::vector<uint32_t> items(2 * std::max<uint32_t>(zj, 1));
    for (uint32_t s = 0; s < zj; ++s) {
        const uint32_t v = sys->host_slot_col[s], r = sys->host_slot_row[s];
        FreedomComp& c = comps[comp_of[v]];
        items[2 * c.item1] = s;
        items[2 * c.item1 + 1] = lidx[v] * c.m + lidx[n + r];
        c.item1++;
    }
    std::vector<uint32_t> col_ptr(n + 1, 0), col_slots(std::max<uint32_t>(zj, 1));
    for (uint32_t v = 0; v < n; ++v) col_ptr[v + 1] = col_ptr[v] + col_count[v];
    {
        std::vector<uint32_t> next(col_ptr.begin(), col_ptr.end() - 1);
        for (uint32_t s = 0; s < zj; ++s) col_slots[next[sys->host_slot_col[s]]++] = s;
    }
    // one allocation for the four index lists
    std::vector<uint32_t> lists;
    lists.insert(lists.end(), items.begin(), items.end());
    F.o_vars = (uint32_t)lists.size();
    lists.insert(lists.end(), comp_vars.begin(), comp_vars.end());
    F.o_col_ptr = (uint32_t)lists.size();
    lists.insert(lists.end(), col_ptr.begin(), col_ptr.end());
    F.o_col_slots = (uint32_t)lists.size();
    lists.insert(lists.end(), col_slots.begin(), col_slots.end());
    int rc;
    if ((rc = F.comps.ensure(std::max<size_t>(comps.size(), 1))) != EZPZ_OK) return rc;
    if ((rc = F.lists.ensure(lists.size())) != EZPZ_OK) return rc;
    if (!comps.empty())
        HIP_TRY(hipMemcpy(F.comps.p, comps.data(), comps.size() * sizeof(FreedomComp), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(F.lists.p, lists.data(), lists.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    F.ncomp = (uint32_t)comps.size();
    if (!comps.empty()) F.comp0 = comps[0];
    F.ws = std::max<uint32_t>(ws, 1);
    F.max_n = max_n;
    // LANE: a lane per (system, component) with 128 private workspaces in <= 64 KiB of LDS
    F.lane = F.ws <= 64;
    if (F.lane) {
        F.threads = 128;
        F.group = F.ncomp >= 128 ? 1 : 128 / std::max<uint32_t>(F.ncomp, 1);
    } else {
        F.threads = max_n <= 64 ? 64 : 256;
        F.group = 1;
    }
    F.built = true;
    return EZPZ_OK;
}

// x_dev: final values, caller order.  Everything on `stream`.
int freedom_device(EzpzSystem* sys, const double* x_dev, size_t batch, uint8_t* mask_dev, double* part_dev,
                   uint32_t* count_dev, hipStream_t stream) {
    auto& F = sys->freedom;
    release_thread_kernel(sys->device);
    int rc = ensure_program(sys);
    if (rc != EZPZ_OK) return rc;
    const size_t n = sys->counts.n_vars, zj = sys->counts.zj;
    if (n == 0 || sys->counts.n_rows == 0) return EZPZ_ERR_EMPTY_SYSTEM;  // find_dof.rs:43-44
    rc = build_freedom(sys);
    if (rc != EZPZ_OK) return rc;
    if ((rc = F.x_int.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.jv.ensure(batch * std::max<size_t>(zj, 1))) != EZPZ_OK) return rc;
    if (!part_dev) {
        if ((rc = F.part.ensure(batch * n)) != EZPZ_OK) return rc;
        part_dev = F.part.p;
    }
    const uint32_t* var_of = reinterpret_cast<const uint32_t*>(sys->view.base + sys->view.o_var_of);
    const uint64_t total = (uint64_t)batch * n;
    hipLaunchKernelGGL(gather_values_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0,
                       stream, x_dev, var_of, F.x_int.p, (uint32_t)n, total);
    EvalArgs e{};
    e.p = sys->view;
    e.x = F.x_int.p;
    e.r_out = nullptr;
    e.jv_out = F.jv.p;
    e.deg_out = nullptr;
    e.batch = batch;
    hipLaunchKernelGGL(eval_kernel, dim3((uint32_t)std::min<size_t>(batch, 8192)), dim3(256), 0, stream, e);
    FreedomArgs a{};
    a.jv = F.jv.p;
    a.comps = F.comps.p;
    a.items = F.lists.p;
    a.comp_vars = F.lists.p + F.o_vars;
    a.col_ptr = F.lists.p + F.o_col_ptr;
    a.col_slots = F.lists.p + F.o_col_slots;
    a.part = part_dev;
    a.mask = mask_dev;
    a.n_under = count_dev;
    a.batch = batch;
    a.n = (uint32_t)n;
    a.zj = (uint32_t)zj;
    a.ncomp = F.ncomp;
    a.ws = F.ws;
    a.group = F.group;
    const size_t head = (2 * (size_t)F.group + (F.group + 1) / 2 + 16) * sizeof(double);
    if (F.lane) {
        const size_t lds = head + (size_t)F.threads * F.ws * sizeof(double);
        const uint32_t grid = (uint32_t)std::min<size_t>((batch + F.group - 1) / F.group, 1u << 16);
        HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        hipLaunchKernelGGL(freedom_kernel<true>, dim3(grid), dim3(F.threads), lds, stream, a);
    } else {
        size_t lds = head + (size_t)F.ws * sizeof(double);
        uint32_t grid = (uint32_t)std::min<size_t>(batch, 1u << 16);
        if (lds > 128 * 1024) {  // workspace of the largest component does not fit LDS: global, bounded to 4 GiB
            lds = head;
            const size_t per = (size_t)F.ws * sizeof(double);
            grid = (uint32_t)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(batch, 1024), (4ull << 30) / per));
            if ((rc = F.gws.ensure((size_t)grid * F.ws)) != EZPZ_OK) return rc;
            a.gws = F.gws.p;
            if (F.ncomp == 1 && F.comp0.n >= 96) {
                // One big component: its pivoted QR as a chain of step launches over the whole device (freedom.hip.hpp),
                // `grid` systems side by side, then the ordinary kernel for rank / null space / participation.
                if ((rc = F.step_done.ensure(grid)) != EZPZ_OK) return rc;
                if ((rc = F.step_tau.ensure(grid)) != EZPZ_OK) return rc;
                HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const uint32_t m = F.comp0.m, nc = F.comp0.n, ndiag = std::min(m, nc);
                for (size_t base = 0; base < batch; base += grid) {
                    const uint32_t nb = (uint32_t)std::min<size_t>(grid, batch - base);
                    FreedomStepArgs sa{};
                    sa.gws = F.gws.p;
                    sa.jv = F.jv.p + base * zj;
                    sa.items = a.items;
                    sa.done = F.step_done.p;
                    sa.tau = F.step_tau.p;
                    sa.ws = F.ws;
                    sa.zj = (uint32_t)zj;
                    sa.m = m;
                    sa.n = nc;
                    sa.item0 = F.comp0.item0;
                    sa.item1 = F.comp0.item1;
                    const uint32_t bl_mn = (uint32_t)std::min<uint64_t>(((uint64_t)m * nc + 255) / 256, 4096);
                    const uint32_t bl_it = std::max<uint32_t>(1, std::min<uint32_t>((sa.item1 - sa.item0 + 255) / 256, 1024));
                    hipLaunchKernelGGL(fr_init_kernel, dim3(bl_mn, nb), dim3(256), 0, stream, sa);
                    hipLaunchKernelGGL(fr_scatter_kernel, dim3(bl_it, nb), dim3(256), 0, stream, sa);
                    hipLaunchKernelGGL(fr_norms_kernel, dim3((nc + 255) / 256, nb), dim3(256), 0, stream, sa);
                    for (uint32_t k = 0; k < ndiag; ++k) {
                        sa.k = k;
                        hipLaunchKernelGGL(fr_pivot_kernel, dim3(nb), dim3(256), 0, stream, sa);
                        if (nc - k - 1 > 0)
                            hipLaunchKernelGGL(fr_apply_kernel, dim3((nc - k - 1 + 63) / 64, nb), dim3(1024), 0, stream, sa);
                    }
                    FreedomArgs fa = a;
                    fa.jv = a.jv + base * zj;
                    fa.part = a.part + base * n;
                    fa.mask = a.mask + base * n;
                    fa.n_under = a.n_under ? a.n_under + base : nullptr;
                    fa.batch = nb;
                    fa.qr_done = 1;
                    hipLaunchKernelGGL(freedom_kernel<false>, dim3(nb), dim3(F.threads), lds, stream, fa);
                }
                HIP_TRY(hipGetLastError());
                return EZPZ_OK;
            }
        }
        HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
        hipLaunchKernelGGL(freedom_kernel<false>, dim3(grid), dim3(F.threads), lds, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return EZPZ_OK;
}

}  // namespace

extern "C" {

int ezpz_system_freedom_batch_device(EzpzSystem* sys, const double* x_dev, size_t batch, uint8_t* under_mask_dev,
                                     double* participation_dev, uint32_t* n_under_dev, void* stream) {
    if (!sys || (batch && (!x_dev || !under_mask_dev))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->mu);
    HIP_TRY(hipSetDevice(sys->device));
    return freedom_device(sys, x_dev, batch, under_mask_dev, participation_dev, n_under_dev, (hipStream_t)stream);
}

int ezpz_system_freedom_batch(EzpzSystem* sys, const double* x, size_t batch, uint8_t* under_mask,
                              double* participation) {
    if (!sys || (batch && (!x || !under_mask))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->mu);
    HIP_TRY(hipSetDevice(sys->device));
    auto& F = sys->freedom;
    const size_t n = sys->counts.n_vars;
    if (n == 0 || sys->counts.n_rows == 0) return EZPZ_ERR_EMPTY_SYSTEM;
    int rc;
    DevBuf<double> xd;
    if ((rc = xd.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.mask.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.part.ensure(batch * n)) != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(xd.p, x, batch * n * sizeof(double), hipMemcpyHostToDevice));
    if ((rc = freedom_device(sys, xd.p, batch, F.mask.p, F.part.p, nullptr, nullptr)) != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(under_mask, F.mask.p, batch * n, hipMemcpyDeviceToHost));
    if (participation) HIP_TRY(hipMemcpy(participation, F.part.p, batch * n * sizeof(double), hipMemcpyDeviceToHost));
    return EZPZ_OK;
}

}  // extern "C"
