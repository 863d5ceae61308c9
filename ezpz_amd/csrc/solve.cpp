// Host orchestration the reference keeps above its numeric core, restated on the C ABI (api.hip, launch.hip, pipeline.cpp) (no device code
// here: everything numeric goes through ezpz_system_solve_batch / ezpz_system_freedom_batch) --
//   solve_inner                 reference ezpz/src/lib.rs:265-356
//   solve_with_priority_inner   reference ezpz/src/lib.rs:148-263 (solve, solve_analysis)
//   lint                        reference ezpz/src/warnings.rs:34-60
//   set_from_initial_values     reference ezpz/src/constraints.rs:146-193
// plus the per-process cache of analysed topologies and the batch form of solve().
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "call_trace.hpp"
#include "kinds.hpp"
#include "one_call.hpp"
#include "program.hpp"

using namespace ezpz;

thread_local CallTrace ezpz::t_call_trace;

extern "C" size_t ezpz_debug_call_trace(uint64_t* buf, size_t cap) {
    const size_t n = t_call_trace.n;
    t_call_trace.buf = buf;
    t_call_trace.cap = buf ? cap : 0;
    t_call_trace.n = 0;
    return n;
}

// ---- solve_inner / solve: host orchestration ---------------------------------------------------------------------
namespace {

// Small LRU of analysed topologies keyed by the request bytes, so that repeated solve() calls on one
// problem (what ezpz-cli's 100-run loop does, main.rs:96-98) skip the symbolic phase.
struct CacheEntry {
    uint64_t hash;
    std::vector<unsigned char> key;
    size_t n_vars;
    int device;
    bool one_solve;  // built for the latency of one solve (solve()) or for batch throughput (solve_batch())
    std::shared_ptr<EzpzSystem> sys;
};
std::mutex g_cache_mu;
std::list<CacheEntry> g_cache;
constexpr size_t kCacheMax = 16;

// The system comes back shared: solve() is callable from many threads at once (like the reference's), and an entry
// another thread evicts must outlive the solves still running on it.  Systems live on the calling thread's current
// HIP device.
int cached_system(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, bool one_solve,
                  std::shared_ptr<EzpzSystem>* out, int32_t* ec, int64_t* ev) {
    const uint64_t h = topology_hash(cs, n_cs, n_vars);
    const size_t bytes = n_cs * sizeof(EzpzConstraint);
    const int device = ezpz_current_device();
    if (device < 0) return EZPZ_ERR_NO_DEVICE;
    {
        std::lock_guard<std::mutex> lock(g_cache_mu);
        for (auto it = g_cache.begin(); it != g_cache.end(); ++it) {
            if (it->hash == h && it->n_vars == n_vars && it->device == device && it->one_solve == one_solve &&
                it->key.size() == bytes &&
                std::memcmp(it->key.data(), cs, bytes) == 0) {
                g_cache.splice(g_cache.begin(), g_cache, it);
                *out = g_cache.front().sys;
                return EZPZ_OK;
            }
        }
    }
    // the symbolic phase runs outside the lock; two threads racing on a new topology both build it, one entry wins
    EzpzSystem* raw = nullptr;
    int rc = ezpz_system_create(cs, n_cs, n_vars, device, one_solve ? EZPZ_TEAM_AUTO_LATENCY : 0, &raw, ec, ev);
    if (rc != EZPZ_OK) return rc;
    CacheEntry e;
    e.hash = h;
    e.key.assign(reinterpret_cast<const unsigned char*>(cs), reinterpret_cast<const unsigned char*>(cs) + bytes);
    e.n_vars = n_vars;
    e.device = device;
    e.one_solve = one_solve;
    e.sys = std::shared_ptr<EzpzSystem>(raw, [](EzpzSystem* p) { ezpz_system_destroy(p); });
    *out = e.sys;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    g_cache.push_front(std::move(e));
    while (g_cache.size() > kCacheMax) g_cache.pop_back();
    return EZPZ_OK;
}

struct WarnSink {
    EzpzWarning* buf;
    size_t cap;
    uint64_t count;
    void push(int32_t about, int32_t content) {
        if (buf && count < cap) {
            buf[count].about_constraint = about;
            buf[count].content = content;
        }
        ++count;
    }
};

double angle_to_degrees(uint8_t tag, double val) {  // datatypes.rs:58-64
    return tag == EZPZ_ANGLE_OTHER_DEG ? val : val * (180.0 / 3.14159265358979323846264338327950288);
}
bool nearly_eq(double a, double b) { return std::fabs(a - b) < 1e-4; }  // warnings.rs:85-87

void lint(const EzpzConstraint* cs, const uint64_t* orig_ids, size_t n, WarnSink& w) {  // warnings.rs:34-60
    for (size_t i = 0; i < n; ++i) {
        const EzpzConstraint& c = cs[i];
        if (c.kind != EZPZ_LINES_AT_ANGLE) continue;
        if (c.tag != EZPZ_ANGLE_OTHER_DEG && c.tag != EZPZ_ANGLE_OTHER_RAD) continue;
        double deg = angle_to_degrees(c.tag, c.param);
        int32_t id = (int32_t)(orig_ids ? orig_ids[i] : i);
        if (nearly_eq(deg, 0.0) || nearly_eq(deg, 360.0) || nearly_eq(deg, 180.0))
            w.push(id, EZPZ_WARN_SHOULD_BE_PARALLEL);
        else if (nearly_eq(deg, 90.0) || nearly_eq(deg, -90.0))
            w.push(id, EZPZ_WARN_SHOULD_BE_PERPENDICULAR);
    }
}

// constraints.rs:146-193
void set_from_initial_values(EzpzConstraint& c, const double* iv) {
    auto X = [&](int k) { return iv[c.ids[k]]; };
    if (c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE && c.tag == EZPZ_SIDE_UNDEFINED) {
        double ux = X(2) - X(0), uy = X(3) - X(1);
        double vx = X(4) - X(0), vy = X(5) - X(1);
        c.tag = (ux * vy - uy * vx >= 0.0) ? EZPZ_LINE_LEFT : EZPZ_LINE_RIGHT;
    } else if (c.kind == EZPZ_CIRCLE_TANGENT_TO_CIRCLE && c.tag == EZPZ_SIDE_UNDEFINED) {
        double dist = std::hypot(X(0) - X(3), X(1) - X(4));
        double a_r = X(2), b_r = X(5);
        double r_int = std::fabs(std::fabs(a_r - b_r) - dist);
        double r_ext = std::fabs(a_r + b_r - dist);
        c.tag = (r_int < r_ext) ? EZPZ_CIRCLE_INTERIOR : EZPZ_CIRCLE_EXTERIOR;
    }
}

}  // namespace

extern "C" {

void ezpz_multi_cache_clear(void);  // multi.cpp: the handles behind ezpz_system_solve_batch_multi

void ezpz_request_plans_clear(void);  // (below: the request plans of ezpz_solve)

void ezpz_cache_clear(void) {
    ezpz_multi_cache_clear();
    ezpz_request_plans_clear();
    std::lock_guard<std::mutex> lock(g_cache_mu);
    g_cache.clear();
}

int ezpz_resolve_sides(EzpzConstraint* cs, size_t n_cs, const double* values, size_t n_vars) {
    if ((n_cs && !cs) || (n_vars && !values)) return EZPZ_ERR_INVALID_ARGUMENT;
    const std::vector<double> iv(values, values + n_vars);
    for (size_t i = 0; i < n_cs; ++i) {
        EzpzConstraint& c = cs[i];
        const bool undefined_side = (c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE || c.kind == EZPZ_CIRCLE_TANGENT_TO_CIRCLE) &&
                                    c.tag == EZPZ_SIDE_UNDEFINED;
        if (!undefined_side) continue;
        for (int k = 0; k < kind_num_ids(c.kind); ++k)
            if (c.ids[k] >= n_vars) return EZPZ_ERR_MISSING_GUESS;
        set_from_initial_values(c, iv.data());
    }
    return EZPZ_OK;
}

}  // extern "C"

namespace {

int solve_inner_impl(const EzpzConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                     const double* guesses, size_t n_guesses, const EzpzConfig* cfg, double* x_out,
                     uint64_t* unsat_ids, EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out,
                     uint32_t* under_out, uint64_t* n_under_out) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    if (n_under_out) *n_under_out = 0;
    std::memset(out, 0, sizeof(*out));
    out->num_vars = n_guesses;
    uint64_t num_eqs = 0;
    for (size_t i = 0; i < n_cs; ++i) num_eqs += (uint64_t)residual_dim(cs[i].kind);
    out->num_eqs = num_eqs;
    WarnSink sink{warn_buf, warn_cap, 0};
    lint(cs, orig_ids, n_cs, sink);
    out->n_warnings = sink.count;
    EzpzConfig dcfg;
    if (!cfg) {
        ezpz_default_config(&dcfg);
        cfg = &dcfg;
    }
    // validate_variables (solver.rs:142-189): every id a constraint's rows mention must appear among the
    // guess ids.  Values are then addressed by id (Layout::index_of, solver.rs:107-109), so an id that is
    // present but >= n_guesses cannot be placed in the matrix (faer CreationError in the reference).
    bool dense = true;
    for (size_t i = 0; i < n_guesses && var_ids; ++i)
        if (var_ids[i] != i) dense = false;
    if (!dense) {
        uint32_t max_id = 0;
        for (size_t i = 0; i < n_guesses; ++i) max_id = std::max(max_id, var_ids[i]);
        std::vector<uint8_t> present((size_t)max_id + 1, 0);
        for (size_t i = 0; i < n_guesses; ++i) present[var_ids[i]] = 1;
        for (size_t i = 0; i < n_cs; ++i) {
            if (cs[i].kind >= EZPZ_NUM_KINDS) continue;
            const KindInfo& K = kKinds[cs[i].kind];
            for (int r = 0; r < K.n_rows; ++r)
                for (int e = 0; e < K.n_nz[r]; ++e) {
                    uint32_t v = cs[i].ids[K.nz[r][e]];
                    if (v > max_id || !present[v]) {
                        out->error = EZPZ_ERR_MISSING_GUESS;
                        out->err_constraint_id = (int32_t)(orig_ids ? orig_ids[i] : i);
                        out->err_variable = v;
                        return out->error;
                    }
                }
        }
    }
    std::shared_ptr<EzpzSystem> sys_ref;
    int32_t ec = -1;
    int64_t ev = -1;
    int rc = cached_system(cs, n_cs, n_guesses, true, &sys_ref, &ec, &ev);
    call_stamp(CALL_PLAN);
    EzpzSystem* sys = sys_ref.get();
    if (rc != EZPZ_OK) {
        if (rc == EZPZ_ERR_MISSING_GUESS && !dense) rc = EZPZ_ERR_MATRIX;  // id has a guess but no column
        out->error = rc;
        if (rc == EZPZ_ERR_MISSING_GUESS) {
            out->err_constraint_id = (int32_t)((orig_ids && ec >= 0) ? orig_ids[ec] : ec);
            out->err_variable = ev;
        }
        return rc;
    }
    if (num_eqs == 0 && cfg->max_iterations > 0) {  // newton.rs:54
        out->error = EZPZ_ERR_EMPTY_SYSTEM;
        return out->error;
    }
    // Every evaluation sweep may warn about every constraint: size the log so nothing is dropped.
    // (only the sixteen non-linear kinds have a degenerate guard)
    uint64_t n_guarded = 0;
    for (size_t i = 0; i < n_cs; ++i) n_guarded += kind_is_linear(cs[i].kind) ? 0 : 1;
    uint64_t want_log = n_guarded * (2 + 2 * std::min<uint64_t>(cfg->max_iterations, 1u << 20));
    uint32_t log_cap = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(want_log, 1), 1u << 22);
    std::unique_ptr<uint64_t[]> log_store(new uint64_t[log_cap]);  // uninitialised: only written entries are read
    uint64_t* log = log_store.get();
    std::vector<uint8_t> mask(std::max<size_t>(n_cs, 1));
    std::vector<double> x(std::max<size_t>(n_guesses, 1));
    EzpzStatus st{};
    call_stamp(CALL_SIDES);
    rc = ezpz_system_solve_batch(sys, guesses, 1, cfg, x.data(), &st, mask.data(), log, log_cap);
    if (rc != EZPZ_OK) {
        out->error = rc;
        return rc;
    }
    // Degenerate warnings in the reference's chronological order: sweep number, then constraint position;
    // about_constraint is the position inside this tier's slice (solver.rs:327,:343).
    uint32_t nlog = std::min<uint32_t>(st.n_warnings, log_cap);
    std::sort(log, log + nlog);
    for (uint32_t i = 0; i < nlog; ++i) sink.push((int32_t)(log[i] & 0xFFFFFFFFu), EZPZ_WARN_DEGENERATE);
    sink.count += st.n_warnings - nlog;
    out->n_warnings = sink.count;
    uint64_t n_unsat = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        if (mask[i]) {
            if (unsat_ids) unsat_ids[n_unsat] = orig_ids ? orig_ids[i] : i;
            ++n_unsat;
        }
    }
    out->n_unsatisfied = n_unsat;
    uint32_t lowest = 0;  // lib.rs:340-344
    for (size_t i = 0; i < n_cs; ++i) lowest = std::max(lowest, cs[i].priority);
    out->priority_solved = lowest;
    out->iterations = st.iterations;
    out->converged = (int32_t)st.converged;
    out->final_lambda = st.final_lambda;
    out->final_residual_inf = st.final_residual_inf;
    if (under_out) {  // lib.rs:328-338: A::analyze(model); an error fails the tier
        std::vector<uint8_t> free_mask(std::max<size_t>(n_guesses, 1));
        rc = ezpz_system_freedom_batch(sys, x.data(), 1, free_mask.data(), nullptr);
        if (rc != EZPZ_OK) {
            out->error = rc;
            return rc;
        }
        uint64_t k = 0;
        for (size_t v = 0; v < n_guesses; ++v)
            if (free_mask[v]) under_out[k++] = (uint32_t)v;
        *n_under_out = k;
    }
    if (x_out && n_guesses) std::memcpy(x_out, x.data(), n_guesses * sizeof(double));
    call_stamp(CALL_FINISHED);
    return EZPZ_OK;
}

int solve_impl(const EzpzConstraint* reqs_in, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
               size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf,
               size_t warn_cap, EzpzOutcome* out, uint32_t* under_out, uint64_t* n_under_out);

}  // namespace

extern "C" {

int ezpz_solve_inner(const EzpzConstraint* cs, const uint64_t* orig_ids, size_t n_cs, const uint32_t* var_ids,
                     const double* guesses, size_t n_guesses, const EzpzConfig* cfg, double* x_out,
                     uint64_t* unsat_ids, EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out) {
    return solve_inner_impl(cs, orig_ids, n_cs, var_ids, guesses, n_guesses, cfg, x_out, unsat_ids, warn_buf,
                            warn_cap, out, nullptr, nullptr);
}

int ezpz_solve(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
               size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf,
               size_t warn_cap, EzpzOutcome* out) {
    return solve_impl(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, x_out, unsat_ids, warn_buf, warn_cap, out,
                      nullptr, nullptr);
}

int ezpz_solve_analysis(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
                        size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids,
                        EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out, uint32_t* under_out,
                        uint64_t* n_under_out) {
    if (!under_out || !n_under_out) return EZPZ_ERR_INVALID_ARGUMENT;
    return solve_impl(reqs, n_reqs, var_ids, guesses, n_guesses, cfg, x_out, unsat_ids, warn_buf, warn_cap, out,
                      under_out, n_under_out);
}

}  // extern "C"

namespace {

// ---- the request plan: everything about a request list that does not depend on the guesses ----------------------------------
// The reference's solve() receives the whole request on every call (lib.rs:80-87) and the published figure times one
// call per iteration (main.rs:86-100).  What the host derives from the request alone -- priority tiers and their subsets
// (lib.rs:199-214), which requests have a side to infer, the lint warnings (warnings.rs:34-60), validate_variables'
// verdict (solver.rs:142-189), rows, the analysed topology of every tier -- is kept per request and found again by ONE
// comparison of the request bytes with the calling thread's previous plan (then a hashed look-up among the process's
// plans): a warm call scans the 112 KB of a 2000-constraint request once instead of nine times.
struct TierPlan {
    bool whole = false;             // the tier is the whole list in request order (ids = positions)
    std::vector<uint32_t> members;  // positions of its requests otherwise
    uint64_t num_eqs = 0, n_guarded = 0;
    uint32_t lowest = 0;            // lib.rs:340-344
    std::vector<EzpzWarning> lint;  // warnings.rs:34-60, about_constraint = position in the caller's list
    int validate_rc = EZPZ_OK;      // solver.rs:142-189 for ids that are not 0..n-1
    int32_t validate_constraint = -1;
    int64_t validate_variable = -1;
    struct Analysed {
        std::vector<uint8_t> tags;  // the inferred sides this topology was analysed for (order of RequestPlan::undefined)
        std::shared_ptr<EzpzSystem> sys;
        int rc = EZPZ_OK;
        int32_t ec = -1;
        int64_t ev = -1;
    };
    std::vector<Analysed> analysed;  // (under RequestPlan::mu)
};

struct RequestPlan {
    std::vector<unsigned char> key;  // the request bytes
    uint64_t hash = 0;
    size_t n_reqs = 0, n_guesses = 0;
    bool dense_ids = true;          // guess ids are 0..n-1 in order (or the caller passed none)
    std::vector<uint32_t> var_ids;  // otherwise
    size_t max_id = 0;
    int device = -1;
    uint64_t generation = 0;
    std::vector<uint32_t> undefined;  // positions of the requests whose side the guesses decide (constraints.rs:146-193)
    std::vector<TierPlan> tiers;      // ascending priority (lib.rs:199-203)
    std::mutex mu;
};

std::mutex g_plans_mu;
std::list<std::shared_ptr<RequestPlan>> g_plans;
std::atomic<uint64_t> g_plans_generation{1};
constexpr size_t kPlansMax = 16, kAnalysedMax = 8;
thread_local std::shared_ptr<RequestPlan> t_last_plan;

bool ids_are_dense(const uint32_t* var_ids, size_t n) {
    if (!var_ids) return true;
    for (size_t i = 0; i < n; ++i)
        if (var_ids[i] != i) return false;
    return true;
}

bool plan_matches(const RequestPlan& p, const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, size_t n_guesses,
                  int device) {
    if (p.n_reqs != n_reqs || p.n_guesses != n_guesses || p.device != device) return false;
    if (p.dense_ids ? !ids_are_dense(var_ids, n_guesses)
                    : (!var_ids || std::memcmp(p.var_ids.data(), var_ids, n_guesses * sizeof(uint32_t)) != 0))
        return false;
    return std::memcmp(p.key.data(), reqs, n_reqs * sizeof(EzpzConstraint)) == 0;
}

bool has_undefined_side(const EzpzConstraint& c) {
    return (c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE || c.kind == EZPZ_CIRCLE_TANGENT_TO_CIRCLE) && c.tag == EZPZ_SIDE_UNDEFINED;
}

std::shared_ptr<RequestPlan> build_plan(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, size_t n_guesses,
                                        int device, uint64_t hash) {
    auto plan = std::make_shared<RequestPlan>();
    RequestPlan& p = *plan;
    p.key.assign(reinterpret_cast<const unsigned char*>(reqs), reinterpret_cast<const unsigned char*>(reqs) + n_reqs * sizeof(EzpzConstraint));
    p.hash = hash;
    p.n_reqs = n_reqs;
    p.n_guesses = n_guesses;
    p.device = device;
    p.generation = g_plans_generation.load();
    p.dense_ids = ids_are_dense(var_ids, n_guesses);
    if (!p.dense_ids) p.var_ids.assign(var_ids, var_ids + n_guesses);
    for (size_t i = 0; i < n_guesses; ++i) p.max_id = std::max<size_t>(p.max_id, var_ids ? var_ids[i] : i);
    for (size_t i = 0; i < n_reqs; ++i) {
        const EzpzConstraint& c = reqs[i];
        if (!has_undefined_side(c)) continue;
        bool ok = n_guesses > 0;
        const int cnt = c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE ? 7 : 6;
        for (int k = 0; k < cnt; ++k)
            if (c.ids[k] > p.max_id) ok = false;  // the reference would panic on this index; the side stays Undefined
        if (ok) p.undefined.push_back((uint32_t)i);
    }
    std::vector<uint32_t> prios;  // distinct priorities, ascending (lib.rs:199-203)
    for (size_t i = 0; i < n_reqs; ++i) prios.push_back(reqs[i].priority);
    std::sort(prios.begin(), prios.end());
    prios.erase(std::unique(prios.begin(), prios.end()), prios.end());
    std::vector<uint8_t> present;
    if (!p.dense_ids) {
        present.assign(p.max_id + 1, 0);
        for (size_t i = 0; i < n_guesses; ++i) present[var_ids[i]] = 1;
    }
    for (uint32_t curr_max_priority : prios) {
        TierPlan t;
        t.whole = prios.size() == 1;
        std::vector<EzpzConstraint> subset;
        std::vector<uint64_t> ids;
        for (size_t i = 0; i < n_reqs; ++i) {
            if (reqs[i].priority > curr_max_priority) continue;
            if (!t.whole) t.members.push_back((uint32_t)i);
            const EzpzConstraint& c = reqs[i];
            t.num_eqs += (uint64_t)residual_dim(c.kind);
            t.n_guarded += kind_is_linear(c.kind) ? 0 : 1;
            t.lowest = std::max(t.lowest, c.priority);
            // validate_variables (solver.rs:142-189): every id a constraint's rows mention must appear among the guess ids.
            // Values are then addressed by id (Layout::index_of, solver.rs:107-109), so an id that is present but >=
            // n_guesses cannot be placed in the matrix (faer CreationError in the reference; see analysed_for).
            if (!p.dense_ids && t.validate_rc == EZPZ_OK && c.kind < EZPZ_NUM_KINDS) {
                const KindInfo& K = kKinds[c.kind];
                for (int r = 0; r < K.n_rows && t.validate_rc == EZPZ_OK; ++r)
                    for (int e = 0; e < K.n_nz[r]; ++e) {
                        const uint32_t v = c.ids[K.nz[r][e]];
                        if (v > p.max_id || !present[v]) {
                            t.validate_rc = EZPZ_ERR_MISSING_GUESS;
                            t.validate_constraint = (int32_t)i;
                            t.validate_variable = v;
                            break;
                        }
                    }
            }
        }
        WarnSink sink{nullptr, 0, 0};
        {  // lint over the tier's requests, in order, reported by position in the caller's list
            std::vector<EzpzWarning> buf(n_reqs + 1);
            sink.buf = buf.data();
            sink.cap = buf.size();
            if (t.whole) {
                lint(reqs, nullptr, n_reqs, sink);
            } else {
                for (uint32_t i : t.members) {
                    const uint64_t id = i;
                    lint(reqs + i, &id, 1, sink);
                }
            }
            t.lint.assign(buf.begin(), buf.begin() + (size_t)std::min<uint64_t>(sink.count, buf.size()));
        }
        p.tiers.push_back(std::move(t));
    }
    return plan;
}

// The plan of this request: the calling thread's previous one, one of the process's, or a new one.
std::shared_ptr<RequestPlan> find_plan(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, size_t n_guesses, int device) {
    const uint64_t gen = g_plans_generation.load(std::memory_order_relaxed);
    if (t_last_plan && t_last_plan->generation == gen && plan_matches(*t_last_plan, reqs, n_reqs, var_ids, n_guesses, device))
        return t_last_plan;
    const uint64_t h = topology_hash(reqs, n_reqs, n_guesses);
    {
        std::lock_guard<std::mutex> lock(g_plans_mu);
        for (auto it = g_plans.begin(); it != g_plans.end(); ++it) {
            if ((*it)->hash == h && plan_matches(**it, reqs, n_reqs, var_ids, n_guesses, device)) {
                g_plans.splice(g_plans.begin(), g_plans, it);
                return t_last_plan = g_plans.front();
            }
        }
    }
    std::shared_ptr<RequestPlan> plan = build_plan(reqs, n_reqs, var_ids, n_guesses, device, h);  // (outside the lock)
    call_stamp(COLD_PLAN_BUILT);
    std::lock_guard<std::mutex> lock(g_plans_mu);
    if (plan->generation == g_plans_generation.load()) {
        g_plans.push_front(plan);
        while (g_plans.size() > kPlansMax) g_plans.pop_back();
    }
    return t_last_plan = plan;
}

// The analysed topology of one tier for one choice of inferred sides (Model::new, solver.rs:192-300; cached).
int analysed_for(RequestPlan& p, TierPlan& t, const EzpzConstraint* reqs, const uint8_t* tags, std::shared_ptr<EzpzSystem>* out,
                 int32_t* ec, int64_t* ev) {
    const size_t nu = p.undefined.size();
    {
        std::lock_guard<std::mutex> lock(p.mu);
        for (TierPlan::Analysed& a : t.analysed)
            if (a.tags.size() == nu && (nu == 0 || std::memcmp(a.tags.data(), tags, nu) == 0)) {
                *out = a.sys;
                *ec = a.ec;
                *ev = a.ev;
                return a.rc;
            }
    }
    // the symbolic phase runs outside the lock; two threads racing on a new topology both build it, one entry wins
    std::vector<EzpzConstraint> tier;
    const EzpzConstraint* cs = reqs;
    size_t n_cs = p.n_reqs;
    if (!t.whole || nu) {
        if (t.whole) {
            tier.assign(reqs, reqs + p.n_reqs);
            for (size_t u = 0; u < nu; ++u) tier[p.undefined[u]].tag = tags[u];
        } else {
            std::vector<uint8_t> tag_of;
            if (nu) {
                tag_of.assign(p.n_reqs, 0xFF);
                for (size_t u = 0; u < nu; ++u) tag_of[p.undefined[u]] = tags[u];
            }
            for (uint32_t i : t.members) {
                tier.push_back(reqs[i]);
                if (nu && tag_of[i] != 0xFF) tier.back().tag = tag_of[i];
            }
        }
        cs = tier.data();
        n_cs = tier.size();
    }
    TierPlan::Analysed a;
    if (nu) a.tags.assign(tags, tags + nu);
    EzpzSystem* raw = nullptr;
    a.rc = ezpz_system_create(cs, n_cs, p.n_guesses, p.device, EZPZ_TEAM_AUTO_LATENCY, &raw, &a.ec, &a.ev);
    if (a.rc == EZPZ_OK) a.sys = std::shared_ptr<EzpzSystem>(raw, [](EzpzSystem* s) { ezpz_system_destroy(s); });
    if (a.rc == EZPZ_ERR_HIP || a.rc == EZPZ_ERR_NO_DEVICE) {  // not a property of the request: not remembered
        *ec = a.ec;
        *ev = a.ev;
        return a.rc;
    }
    std::lock_guard<std::mutex> lock(p.mu);
    if (t.analysed.size() >= kAnalysedMax) t.analysed.erase(t.analysed.begin());
    t.analysed.push_back(a);
    *out = a.sys;
    *ec = a.ec;
    *ev = a.ev;
    return a.rc;
}

// The calling thread's scratch for one call (grow-only: a warm call allocates nothing).
struct CallScratch {
    std::vector<double> initial_values, x_try;
    std::vector<uint8_t> tags, mask, free_mask;
    std::vector<uint64_t> log, unsat_try;
    std::vector<EzpzWarning> warn_try;
    std::vector<uint32_t> under_try;
};
thread_local CallScratch t_scratch;
template <class T>
T* grown(std::vector<T>& v, size_t n) {
    if (v.size() < n) v.resize(std::max(n, v.size() * 2));
    return v.data();
}

// solve_inner (lib.rs:265-356) of one tier of a planned request.  unsat_ids / warn_buf / x_out may be the caller's own.
int solve_tier(RequestPlan& p, TierPlan& t, const EzpzConstraint* reqs, const uint8_t* tags, const double* guesses,
               const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf, size_t warn_cap, EzpzOutcome* out,
               uint32_t* under_out, uint64_t* n_under_out) {
    std::memset(out, 0, sizeof(*out));
    if (n_under_out) *n_under_out = 0;
    out->num_vars = p.n_guesses;
    out->num_eqs = t.num_eqs;
    WarnSink sink{warn_buf, warn_cap, 0};
    for (const EzpzWarning& w : t.lint) sink.push(w.about_constraint, w.content);
    out->n_warnings = sink.count;
    if (t.validate_rc != EZPZ_OK) {
        out->error = t.validate_rc;
        out->err_constraint_id = t.validate_constraint;
        out->err_variable = t.validate_variable;
        return out->error;
    }
    std::shared_ptr<EzpzSystem> sys_ref;
    int32_t ec = -1;
    int64_t ev = -1;
    int rc = analysed_for(p, t, reqs, tags, &sys_ref, &ec, &ev);
    call_stamp(CALL_SIDES);
    if (rc != EZPZ_OK) {
        if (rc == EZPZ_ERR_MISSING_GUESS && !p.dense_ids) rc = EZPZ_ERR_MATRIX;  // id has a guess but no column
        out->error = rc;
        if (rc == EZPZ_ERR_MISSING_GUESS) {
            out->err_constraint_id = (int32_t)((!t.whole && ec >= 0) ? t.members[(size_t)ec] : ec);
            out->err_variable = ev;
        }
        return rc;
    }
    EzpzConfig dcfg;
    if (!cfg) {
        ezpz_default_config(&dcfg);
        cfg = &dcfg;
    }
    if (t.num_eqs == 0 && cfg->max_iterations > 0) {  // newton.rs:54
        out->error = EZPZ_ERR_EMPTY_SYSTEM;
        return out->error;
    }
    const size_t n_cs = t.whole ? p.n_reqs : t.members.size();
    // Every evaluation sweep may warn about every constraint with a degenerate guard (the sixteen non-linear kinds):
    // the log is sized so that nothing is dropped.
    const uint64_t want_log = t.n_guarded * (2 + 2 * std::min<uint64_t>(cfg->max_iterations, 1u << 20));
    const uint32_t log_cap = (uint32_t)std::min<uint64_t>(want_log, 1u << 22);
    uint64_t* log = log_cap ? grown(t_scratch.log, log_cap) : nullptr;
    uint8_t* mask = grown(t_scratch.mask, std::max<size_t>(n_cs, 1));
    EzpzStatus st{};
    rc = system_solve_one(sys_ref.get(), guesses, cfg, x_out, &st, mask, log, log_cap);
    if (rc != EZPZ_OK) {
        out->error = rc;
        return rc;
    }
    // Degenerate warnings in the reference's chronological order: sweep number, then constraint position;
    // about_constraint is the position inside this tier's slice (solver.rs:327,:343).
    if (st.n_warnings) {
        const uint32_t nlog = std::min<uint32_t>(st.n_warnings, log_cap);
        std::sort(log, log + nlog);
        for (uint32_t i = 0; i < nlog; ++i) sink.push((int32_t)(log[i] & 0xFFFFFFFFu), EZPZ_WARN_DEGENERATE);
        sink.count += st.n_warnings - nlog;
        out->n_warnings = sink.count;
    }
    uint64_t n_unsat = 0;
    if (st.n_unsatisfied) {
        for (size_t i = 0; i < n_cs; ++i)
            if (mask[i]) {
                if (unsat_ids) unsat_ids[n_unsat] = t.whole ? i : t.members[i];
                ++n_unsat;
            }
    }
    out->n_unsatisfied = n_unsat;
    out->priority_solved = t.lowest;
    out->iterations = st.iterations;
    out->converged = (int32_t)st.converged;
    out->final_lambda = st.final_lambda;
    out->final_residual_inf = st.final_residual_inf;
    if (under_out) {  // lib.rs:328-338: A::analyze(model); an error fails the tier
        uint8_t* free_mask = grown(t_scratch.free_mask, std::max<size_t>(p.n_guesses, 1));
        rc = ezpz_system_freedom_batch(sys_ref.get(), x_out, 1, free_mask, nullptr);
        if (rc != EZPZ_OK) {
            out->error = rc;
            return rc;
        }
        uint64_t k = 0;
        for (size_t v = 0; v < p.n_guesses; ++v)
            if (free_mask[v]) under_out[k++] = (uint32_t)v;
        *n_under_out = k;
    }
    call_stamp(CALL_FINISHED);
    return EZPZ_OK;
}

int solve_impl(const EzpzConstraint* reqs, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
               size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf,
               size_t warn_cap, EzpzOutcome* out, uint32_t* under_out, uint64_t* n_under_out) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    call_stamp(CALL_ENTER);
    if (n_under_out) *n_under_out = 0;  // A::no_constraints(), lib.rs:157,250
    if (n_reqs == 0) {  // lib.rs:155-170
        std::memset(out, 0, sizeof(*out));
        if (x_out && n_guesses) std::memcpy(x_out, guesses, n_guesses * sizeof(double));
        out->converged = 1;
        out->num_vars = n_guesses;
        return EZPZ_OK;
    }
    if (!reqs || (n_guesses && !guesses)) return EZPZ_ERR_INVALID_ARGUMENT;
    // (without a device the plan is still built: request errors are reported before the absence of a device)
    const int device = ezpz_current_device();
    const std::shared_ptr<RequestPlan> plan = find_plan(reqs, n_reqs, var_ids, n_guesses, device);
    RequestPlan& p = *plan;
    call_stamp(CALL_PLAN);
    // the sides the guesses decide: initial_values[id] = guess (lib.rs:172-180), set_from_initial_values (lib.rs:183-186)
    const uint8_t* tags = nullptr;
    if (!p.undefined.empty()) {
        const double* iv = guesses;
        if (!p.dense_ids) {
            double* v = grown(t_scratch.initial_values, p.max_id + 1);
            std::fill(v, v + p.max_id + 1, 0.0);
            for (size_t i = 0; i < n_guesses; ++i) v[var_ids[i]] = guesses[i];
            iv = v;
        }
        uint8_t* tg = grown(t_scratch.tags, p.undefined.size());
        for (size_t u = 0; u < p.undefined.size(); ++u) {
            EzpzConstraint c = reqs[p.undefined[u]];
            set_from_initial_values(c, iv);
            tg[u] = c.tag;
        }
        tags = tg;
    }
    if (p.tiers.size() == 1) {  // one tier (the common call): straight into the caller's buffers
        const int rc = solve_tier(p, p.tiers[0], reqs, tags, guesses, cfg, x_out ? x_out : grown(t_scratch.x_try, std::max<size_t>(n_guesses, 1)),
                                  unsat_ids, warn_buf, warn_cap, out, under_out, n_under_out);
        call_stamp(CALL_RETURN);
        return rc;
    }
    std::memset(out, 0, sizeof(*out));
    double* x_try = grown(t_scratch.x_try, std::max<size_t>(n_guesses, 1));
    uint64_t* unsat_try = grown(t_scratch.unsat_try, n_reqs + 1);
    EzpzWarning* warn_try = grown(t_scratch.warn_try, std::max<size_t>(warn_cap, 1));
    uint32_t* under_try = under_out ? grown(t_scratch.under_try, n_guesses + 1) : nullptr;
    uint64_t n_under_try = 0;
    bool have_res = false;
    int rc_final = EZPZ_OK;
    auto adopt = [&](const EzpzOutcome& o) {
        *out = o;
        if (warn_buf && warn_cap) {
            const size_t nw = (size_t)std::min<uint64_t>(o.n_warnings, warn_cap);
            std::memcpy(warn_buf, warn_try, nw * sizeof(EzpzWarning));
        }
    };
    for (TierPlan& t : p.tiers) {  // cumulative subsets, each from the original guesses (lib.rs:205-246)
        EzpzOutcome o;
        const int rc = solve_tier(p, t, reqs, tags, guesses, cfg, x_try, unsat_try, warn_try, warn_cap, &o, under_try, &n_under_try);
        if (rc == EZPZ_OK) {
            if (o.n_unsatisfied > 0 && have_res) break;  // lib.rs:232-234
            adopt(o);
            if (under_out) {
                std::memcpy(under_out, under_try, (size_t)n_under_try * sizeof(uint32_t));
                *n_under_out = n_under_try;
            }
            if (x_out && n_guesses) std::memcpy(x_out, x_try, n_guesses * sizeof(double));
            if (unsat_ids) std::memcpy(unsat_ids, unsat_try, (size_t)o.n_unsatisfied * sizeof(uint64_t));
            have_res = true;
            if (o.n_unsatisfied > 0) break;
        } else {
            if (!have_res) {  // lib.rs:239-244
                adopt(o);
                rc_final = rc;
            }
            break;
        }
    }
    call_stamp(CALL_RETURN);
    return rc_final;
}

}  // namespace

extern "C" void ezpz_request_plans_clear(void) {
    std::lock_guard<std::mutex> lock(g_plans_mu);
    g_plans_generation.fetch_add(1);  // (other threads' last plans go stale with it)
    g_plans.clear();
    t_last_plan.reset();
}

extern "C" {

int ezpz_solve_batch(const EzpzConstraint* reqs_in, size_t n_reqs, size_t n_vars, const double* x0, size_t batch,
                     const EzpzConfig* cfg, double* x_out, EzpzStatus* status, uint32_t* priority_solved,
                     uint8_t* unsat_mask, int32_t* err_constraint, int64_t* err_variable) {
    if ((batch && (!x_out || !status)) || (batch && n_vars && !x0)) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    if (n_reqs == 0) {  // lib.rs:155-170
        if (n_vars) std::memcpy(x_out, x0, batch * n_vars * sizeof(double));
        for (size_t b = 0; b < batch; ++b) {
            status[b] = EzpzStatus{};
            status[b].converged = 1;
            if (priority_solved) priority_solved[b] = 0;
        }
        return EZPZ_OK;
    }
    // ---- per-system side inference, systems grouped by the sides they infer ------------------------------------------
    std::vector<size_t> undefined;  // requests whose side is inferred
    for (size_t i = 0; i < n_reqs; ++i) {
        const EzpzConstraint& c = reqs_in[i];
        if ((c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE || c.kind == EZPZ_CIRCLE_TANGENT_TO_CIRCLE) &&
            c.tag == EZPZ_SIDE_UNDEFINED) {
            bool ok = true;
            const int cnt = c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE ? 7 : 6;
            for (int k = 0; k < cnt; ++k)
                if (c.ids[k] >= n_vars) ok = false;
            if (ok) undefined.push_back(i);
        }
    }
    std::map<std::vector<uint8_t>, std::vector<size_t>> groups;
    {
        std::vector<double> iv(n_vars);
        std::vector<uint8_t> key(undefined.size());
        if (undefined.empty()) {  // one group: every system, in order
            std::vector<size_t>& all = groups[key];
            all.resize(batch);
            for (size_t b = 0; b < batch; ++b) all[b] = b;
        } else {
            for (size_t b = 0; b < batch; ++b) {
                std::memcpy(iv.data(), x0 + b * n_vars, n_vars * sizeof(double));
                for (size_t u = 0; u < undefined.size(); ++u) {
                    EzpzConstraint c = reqs_in[undefined[u]];
                    set_from_initial_values(c, iv.data());
                    key[u] = c.tag;
                }
                groups[key].push_back(b);
            }
        }
    }
    std::vector<uint32_t> prios;
    for (size_t i = 0; i < n_reqs; ++i) prios.push_back(reqs_in[i].priority);
    std::sort(prios.begin(), prios.end());
    prios.erase(std::unique(prios.begin(), prios.end()), prios.end());

    std::vector<EzpzConstraint> reqs(reqs_in, reqs_in + n_reqs), subset;
    std::vector<size_t> subset_ids;
    std::vector<double> xin, xres;
    std::vector<EzpzStatus> stres;
    std::vector<uint8_t> maskres;
    for (auto& g : groups) {
        for (size_t u = 0; u < undefined.size(); ++u) reqs[undefined[u]].tag = g.first[u];
        std::vector<size_t> active = g.second;
        std::vector<char> have_res(batch, 0);
        bool first_tier = true;
        for (uint32_t curr_max_priority : prios) {
            if (active.empty()) break;
            subset.clear();
            subset_ids.clear();
            uint32_t lowest = 0;
            for (size_t i = 0; i < n_reqs; ++i)
                if (reqs[i].priority <= curr_max_priority) {
                    subset.push_back(reqs[i]);
                    subset_ids.push_back(i);
                    lowest = std::max(lowest, reqs[i].priority);
                }
            std::shared_ptr<EzpzSystem> sys_ref;
            int32_t ec = -1;
            int64_t ev = -1;
            int rc = cached_system(subset.data(), subset.size(), n_vars, batch == 1, &sys_ref, &ec, &ev);
            EzpzSystem* sys = sys_ref.get();
            if (rc != EZPZ_OK) {
                if (first_tier) {  // lib.rs:239-244: no earlier tier to fall back to
                    if (err_constraint) *err_constraint = ec >= 0 ? (int32_t)subset_ids[(size_t)ec] : -1;
                    if (err_variable) *err_variable = ev;
                    return rc;
                }
                break;  // every system of the group keeps its previous tier
            }
            const size_t na = active.size(), ns = subset.size();
            if (first_tier && na == batch) {
                // The whole batch in one group, first tier (the common call: no side to infer, one priority): the systems
                // are 0..batch-1 in order and every result is kept (lib.rs:232-234 only drops an unsatisfied *later*
                // tier), so the solve reads the caller's guesses and writes the caller's outputs -- no gather, no scatter.
                const bool direct_mask = unsat_mask && ns == n_reqs;  // subset_ids is the identity
                if (unsat_mask && !direct_mask) maskres.assign(na * std::max<size_t>(ns, 1), 0);
                rc = ezpz_system_solve_batch(sys, x0, na, cfg, x_out, status,
                                             unsat_mask ? (direct_mask ? unsat_mask : maskres.data()) : nullptr, nullptr, 0);
                if (rc != EZPZ_OK) return rc;
                if (priority_solved)
                    for (size_t b = 0; b < batch; ++b) priority_solved[b] = lowest;
                if (unsat_mask && !direct_mask) {
                    std::memset(unsat_mask, 0, batch * n_reqs);
                    for (size_t b = 0; b < batch; ++b)
                        for (size_t k = 0; k < ns; ++k) unsat_mask[b * n_reqs + subset_ids[k]] = maskres[b * ns + k];
                }
                std::fill(have_res.begin(), have_res.end(), 1);
                if (prios.size() > 1) {
                    std::vector<size_t> still;
                    for (size_t b = 0; b < batch; ++b)
                        if (status[b].n_unsatisfied == 0) still.push_back(b);
                    active.swap(still);
                } else {
                    active.clear();
                }
                first_tier = false;
                continue;
            }
            xin.resize(na * std::max<size_t>(n_vars, 1));
            xres.resize(xin.size());
            stres.resize(na);
            maskres.assign(na * std::max<size_t>(ns, 1), 0);
            for (size_t a = 0; a < na; ++a)
                std::memcpy(xin.data() + a * n_vars, x0 + active[a] * n_vars, n_vars * sizeof(double));
            rc = ezpz_system_solve_batch(sys, xin.data(), na, cfg, xres.data(), stres.data(), maskres.data(), nullptr, 0);
            if (rc != EZPZ_OK) return rc;
            std::vector<size_t> still;
            for (size_t a = 0; a < na; ++a) {
                const size_t b = active[a];
                const bool unsat = stres[a].n_unsatisfied > 0;
                if (unsat && have_res[b]) continue;  // lib.rs:232-234: keep the previous, satisfied tier
                std::memcpy(x_out + b * n_vars, xres.data() + a * n_vars, n_vars * sizeof(double));
                status[b] = stres[a];
                if (priority_solved) priority_solved[b] = lowest;
                if (unsat_mask) {
                    std::memset(unsat_mask + b * n_reqs, 0, n_reqs);
                    for (size_t k = 0; k < ns; ++k) unsat_mask[b * n_reqs + subset_ids[k]] = maskres[a * ns + k];
                }
                have_res[b] = 1;
                if (!unsat) still.push_back(b);
            }
            active.swap(still);
            first_tier = false;
        }
    }
    return EZPZ_OK;
}

}  // extern "C"
