// Host orchestration the reference keeps above its numeric core, restated on the C ABI of api.hip (no device code
// here: everything numeric goes through ezpz_system_solve_batch / ezpz_system_freedom_batch) --
//   solve_inner                 reference ezpz/src/lib.rs:265-356
//   solve_with_priority_inner   reference ezpz/src/lib.rs:148-263 (solve, solve_analysis)
//   lint                        reference ezpz/src/warnings.rs:34-60
//   set_from_initial_values     reference ezpz/src/constraints.rs:146-193
// plus the per-process cache of analysed topologies and the batch form of solve().
#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "kinds.hpp"
#include "program.hpp"

using namespace ezpz;

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
void set_from_initial_values(EzpzConstraint& c, const std::vector<double>& iv) {
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

void ezpz_cache_clear(void) {
    ezpz_multi_cache_clear();
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
        set_from_initial_values(c, iv);
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

int solve_impl(const EzpzConstraint* reqs_in, size_t n_reqs, const uint32_t* var_ids, const double* guesses,
               size_t n_guesses, const EzpzConfig* cfg, double* x_out, uint64_t* unsat_ids, EzpzWarning* warn_buf,
               size_t warn_cap, EzpzOutcome* out, uint32_t* under_out, uint64_t* n_under_out) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof(*out));
    if (n_under_out) *n_under_out = 0;  // A::no_constraints(), lib.rs:157,250
    if (n_reqs == 0) {  // lib.rs:155-170
        if (x_out && n_guesses) std::memcpy(x_out, guesses, n_guesses * sizeof(double));
        out->converged = 1;
        out->num_vars = n_guesses;
        return EZPZ_OK;
    }
    // initial_values[id] = guess (lib.rs:172-180), then side inference (lib.rs:183-186)
    size_t max_id = 0;
    for (size_t i = 0; i < n_guesses; ++i) max_id = std::max<size_t>(max_id, var_ids ? var_ids[i] : i);
    std::vector<double> initial_values(max_id + 1, 0.0);
    for (size_t i = 0; i < n_guesses; ++i) initial_values[var_ids ? var_ids[i] : i] = guesses[i];
    // (the request list is copied only if some side has to be filled in: 200 000 requests are 11 MB)
    std::vector<EzpzConstraint> resolved;
    const EzpzConstraint* reqs = reqs_in;
    for (size_t i = 0; i < n_reqs; ++i) {
        const EzpzConstraint& c = reqs_in[i];
        if ((c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE || c.kind == EZPZ_CIRCLE_TANGENT_TO_CIRCLE) &&
            c.tag == EZPZ_SIDE_UNDEFINED) {
            bool ok = n_guesses > 0;
            int cnt = c.kind == EZPZ_LINE_TANGENT_TO_CIRCLE ? 7 : 6;
            for (int k = 0; k < cnt; ++k)
                if (c.ids[k] > max_id) ok = false;  // the reference would panic on this index; leave Undefined
            if (ok) {
                if (resolved.empty()) {
                    resolved.assign(reqs_in, reqs_in + n_reqs);
                    reqs = resolved.data();
                }
                set_from_initial_values(resolved[i], initial_values);
            }
        }
    }
    // distinct priorities, ascending (lib.rs:199-203); one tier is the common case and needs no sort
    std::vector<uint32_t> prios;
    {
        bool one_tier = true;
        for (size_t i = 1; i < n_reqs && one_tier; ++i) one_tier = reqs[i].priority == reqs[0].priority;
        if (one_tier) {
            prios.push_back(reqs[0].priority);
        } else {
            for (size_t i = 0; i < n_reqs; ++i) prios.push_back(reqs[i].priority);
            std::sort(prios.begin(), prios.end());
            prios.erase(std::unique(prios.begin(), prios.end()), prios.end());
        }
    }
    const bool single_tier = prios.size() == 1;

    std::vector<EzpzConstraint> subset;
    std::vector<uint64_t> subset_ids;
    std::vector<double> x_try(std::max<size_t>(n_guesses, 1));
    std::vector<uint64_t> unsat_try(n_reqs + 1);
    std::vector<EzpzWarning> warn_try(std::max<size_t>(warn_cap, 1));
    std::vector<uint32_t> under_try(under_out ? n_guesses + 1 : 0);
    uint64_t n_under_try = 0;
    bool have_res = false;
    int rc_final = EZPZ_OK;
    auto adopt = [&](const EzpzOutcome& o) {
        *out = o;
        if (warn_buf && warn_cap) {
            size_t nw = (size_t)std::min<uint64_t>(o.n_warnings, warn_cap);
            std::memcpy(warn_buf, warn_try.data(), nw * sizeof(EzpzWarning));
        }
    };
    for (uint32_t curr_max_priority : prios) {
        const EzpzConstraint* tier = reqs;  // a single tier is the whole list, ids = positions
        const uint64_t* tier_ids = nullptr;
        size_t tier_n = n_reqs;
        if (!single_tier) {
            subset.clear();
            subset_ids.clear();
            for (size_t i = 0; i < n_reqs; ++i) {
                if (reqs[i].priority <= curr_max_priority) {
                    subset.push_back(reqs[i]);
                    subset_ids.push_back(i);
                }
            }
            tier = subset.data();
            tier_ids = subset_ids.data();
            tier_n = subset.size();
        }
        EzpzOutcome o;
        int rc = solve_inner_impl(tier, tier_ids, tier_n, var_ids, guesses, n_guesses, cfg, x_try.data(),
                                  unsat_try.data(), warn_try.data(), warn_cap, &o,
                                  under_out ? under_try.data() : nullptr, &n_under_try);
        if (rc == EZPZ_OK) {
            if (o.n_unsatisfied > 0 && have_res) break;  // lib.rs:232-234
            adopt(o);
            if (under_out) {
                std::memcpy(under_out, under_try.data(), (size_t)n_under_try * sizeof(uint32_t));
                *n_under_out = n_under_try;
            }
            if (x_out && n_guesses) std::memcpy(x_out, x_try.data(), n_guesses * sizeof(double));
            if (unsat_ids) std::memcpy(unsat_ids, unsat_try.data(), (size_t)o.n_unsatisfied * sizeof(uint64_t));
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
    return rc_final;
}

}  // namespace

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
                    set_from_initial_values(c, iv);
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
