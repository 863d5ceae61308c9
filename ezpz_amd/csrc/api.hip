// C ABI implementation (include/ezpz_amd.h): system lifetime (Model::new, reference
// ezpz/src/solver.rs:192-300, as a cached topology program), launch-shape selection and kernel dispatch for the LM
// solve (newton.rs:29-145), the evaluation-only kernel, and FreedomAnalysis (solver/find_dof.rs).  The host
// orchestration above it (solve, solve_inner, priority tiers, lint) is in solve.cpp.
// All numeric work runs in kernels on the GPU; there is no CPU solver in this library.
#include "system.hpp"

using namespace ezpz;

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
    EZPZ_ON_DEVICE(device);
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
        static const bool wave_all = [] {
            const char* e = std::getenv("EZPZ_JIT_WAVE");  // EZPZ_JIT_WAVE=all: every small system (measurements of the rule below)
            return e && std::strcmp(e, "all") == 0;
        }();
        const bool pays = wave_all || (n_vars >= 8 && non_linear >= 3);
        if (wave_enabled && !s->lane->wave_source.empty() &&
            (team_size == EZPZ_TEAM_LATENCY_WAVE || (team_size == EZPZ_TEAM_AUTO_LATENCY && pays)))
            s->wave_jit = comp_jit_create_source(s->lane->wave_source, "ezpz_jit_wave");
    }
    if (s->fronts) {
        HIP_TRY(hipMalloc(&s->dev_fronts, s->fronts->blob.size()));
        HIP_TRY(hipMemcpy(s->dev_fronts, s->fronts->blob.data(), s->fronts->blob.size(), hipMemcpyHostToDevice));
    }
    if (s->lanes) {
        HIP_TRY(hipMalloc((void**)&s->dev_lanes, s->lanes->blob.size() * 4));
        HIP_TRY(hipMemcpy(s->dev_lanes, s->lanes->blob.data(), s->lanes->blob.size() * 4, hipMemcpyHostToDevice));
    }
    *out = s.release();
    return EZPZ_OK;
}

}  // extern "C"

int ezpz::ensure_program(EzpzSystem* sys) {
    if (!sys->program_deferred.load(std::memory_order_acquire)) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->defer_mu);
    if (!sys->program_deferred.load()) return EZPZ_OK;
    Program P;
    std::vector<unsigned char> blob;
    int rc = analyze_into(sys->deferred_cs.data(), sys->deferred_cs.size(), sys->counts.n_vars, EZPZ_TEAM_AUTO_LATENCY, *sys, P, blob,
                          nullptr, nullptr, false, /*keep_comp=*/true);
    if (rc != EZPZ_OK) return rc;
    if (sys->device >= 0) {
        EZPZ_ON_DEVICE(sys->device);
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

// Diagnostic: a device buffer for in-kernel time stamps (the -DEZPZ_STAMPS builds of the list-walk and frontal kernels:
// tools/front_stamps.py; the run-time compiled kernel of a system on several workgroups in every build: tools/ladder_stamps.py).
void ezpz_debug_set_stamps(unsigned long long* dev_buf) { g_stamps = dev_buf; }

unsigned long long ezpz_debug_jit_compilations(void) { return comp_jit_compilations(); }


}  // extern "C"
