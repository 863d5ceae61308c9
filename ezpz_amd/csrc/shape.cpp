// Launch-shape selection (include/ezpz_amd.h: ezpz_system_create, ezpz_analyze): the host symbolic phase's second half.
// build_program (program.cpp) turns one tier of constraints into index lists; analyze_into decides which kernel family
// runs them -- sub-wavefront teams, a wavefront-partitioned or barrier workgroup, a grid team, the record walk, the
// component-resident / lane-per-system / lanes-across-the-batch plans -- with which team size and which LDS plan, from
// the topology and the thresholds of policy.hpp.  Replaces nothing in the reference (faer picks simplicial or supernodal
// per pattern, solver.rs:289-300); no device code.
#include "system.hpp"

using namespace ezpz;

namespace {

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

}  // namespace

namespace ezpz {

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
int analyze_into(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint32_t team_size, EzpzSystem& s,
                 Program& P, std::vector<unsigned char>& blob, int32_t* err_constraint, int64_t* err_variable, bool may_defer,
                 bool keep_comp) {
    BuildError be;
    const bool no_fronts = team_size == EZPZ_TEAM_LATENCY_RECORDS;  // the automatic latency shape as it was before the fronts
    if (no_fronts) team_size = EZPZ_TEAM_AUTO_LATENCY;
    const bool latency_auto = team_size == EZPZ_TEAM_AUTO_LATENCY || team_size == EZPZ_TEAM_LATENCY_WAVE;
    const bool batch_auto = team_size == 0;
    static const bool comp_enabled0 = [] {
        const char* e = std::getenv("EZPZ_COMP");
        return !(e && e[0] == '0');
    }();
    if (may_defer && (team_size == EZPZ_TEAM_AUTO_LATENCY || team_size == EZPZ_TEAM_LATENCY_WAVE) && comp_enabled0 && !keep_comp) {
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
    auto front_options = [&]() {
        const EzpzLaunchPolicy& pol = s.lim.policy;
        FrontOptions fo;
        fo.wgs = 0;
        if (const char* e = std::getenv("EZPZ_FRONT_WGS")) fo.wgs = (uint32_t)std::atoi(e);
        fo.max_wgs = std::min<uint32_t>(pol.front_max_workgroups, (uint32_t)std::max(1, s.lim.cus));
        fo.vars_per_wg = pol.front_vars_per_workgroup;
        if (const char* e = std::getenv("EZPZ_FRONT_VARS_PER_WG")) fo.vars_per_wg = (uint32_t)std::atoi(e);
        fo.lds_bytes = s.lim.lds_bytes;
        return fo;
    };
    // One solve of a connected sketch with narrow fronts (at most 32 rows: no question of the record walk being cheaper): the frontal
    // plan is ALL a solve needs -- the kernel reads nothing of the program -- so the system returns with it alone, like a block
    // system with its component plan above, and the program (FreedomAnalysis, eval, the info) is analysed when somebody asks
    // (ensure_program).  A new topology's first solve is mostly its symbolic phase (300 variables: 1.2 ms against 0.18 ms of
    // kernel), and an edited sketch is a new topology: 300 / 2000 variables 1.16 -> 0.45, 6.4 -> 2.4 ms.
    if (may_defer && latency_auto && !no_fronts && !keep_comp && n_cs > 0 && n_vars >= s.lim.policy.front_min_vars_one_solve) {
        const char* fe = std::getenv("EZPZ_FRONTS");
        if (!(fe && std::atoi(fe) == 0)) {
            std::unique_ptr<FrontPlan> plan(new FrontPlan());
            const char* why = nullptr;
            if (front_plan_build(cs, n_cs, n_vars, front_options(), *plan, &why) && plan->n_components == 1 && plan->max_rows <= 32) {
                s.counts = ProgramCounts();
                s.counts.n_cons = plan->n_cons;
                s.counts.n_vars = plan->n_vars;
                s.counts.n_rows = plan->n_rows;
                s.counts.zj = plan->zj;
                s.unit_weights = plan->unit_weights;
                EzpzSystemInfo& info = s.info;
                std::memset(&info, 0, sizeof(info));
                info.n_constraints = plan->n_cons;
                info.n_vars = plan->n_vars;
                info.n_rows = plan->n_rows;
                info.nnz_j = plan->zj;
                info.n_components = 1;
                info.program_bytes = plan->blob.size();
                info.team_mode = 5;
                info.team_size = plan->threads;
                info.grid_workgroups = info.front_workgroups = plan->n_wgs;
                info.front_max_batch = 0xFFFFFFFFu;
                info.n_partitions = plan->n_fronts;
                info.n_levels = plan->n_levels;
                info.workspace_bytes = (uint64_t)plan->ws_doubles_max * 8;
                info.workspace_in_lds = 1;
                s.comp.reset();
                s.lane.reset();
                s.lanes.reset();
                s.fronts = std::move(plan);
                s.front_max_batch = ~0ull;
                s.deferred_cs.assign(cs, cs + n_cs);
                s.program_deferred.store(true);
                blob.clear();
                return EZPZ_OK;
            }
        }
    }
    const bool force_fronts = team_size == EZPZ_TEAM_FRONTS;  // the frontal shape whatever the size; the latency shape behind it
    if (force_fronts) team_size = EZPZ_TEAM_AUTO_LATENCY;
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
        // there).
        const int rec_small = (int)(for_latency ? s.lim.policy.rec_min_vars_one_solve : s.lim.policy.rec_min_vars_batch) - 1;
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
            static const bool rec_on2 = [] {
                const char* e = std::getenv("EZPZ_REC");
                return !(e && e[0] == '0');
            }();
            // (one solve of such a system too: 4 x 150 / 8 x 80 / 6 x 40 variables 790 -> 237, 727 -> 237, 224 -> 109 us)
            const bool few = rec_on2 && !team_size && !latency_phases && !lists_only && n_pieces >= 2 &&
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
    // ---- the frontal shape: one connected sketch as a tree of dense fronts (fronts.cpp), one solve from front_min_vars_one_solve
    //      variables on as many workgroups as its size asks for; the shapes above stay behind it (stragglers of the lanes, systems
    //      whose fronts would exceed a wavefront's rows).  A system created for batches carries the plan too and takes it for
    //      calls that would leave most of the device idle at one workgroup per system (EzpzSystem::front_max_batch); its
    //      EzpzSystemInfo keeps describing the shape of its large calls.  EZPZ_FRONTS=0: never; EZPZ_FRONT_WGS: workgroups per system.
    // (the rest of a deferred analysis keeps the plan the system has been solving on)
    const bool keep_fronts = keep_comp && s.fronts && s.front_max_batch == ~0ull;
    if (!keep_fronts) {
        s.fronts.reset();
        s.front_max_batch = 0;
    }
    bool fronts_tried = keep_fronts;
    auto build_fronts = [&]() {
        fronts_tried = true;
        if (keep_fronts) return;
        const char* fe = std::getenv("EZPZ_FRONTS");
        const int fronts_env = fe ? std::atoi(fe) : 1;
        const EzpzLaunchPolicy& pol = s.lim.policy;
        // (one solve: the automatic latency shape only -- EZPZ_TEAM_LATENCY_PHASES / _RECORDS ask for the older ones; batches: the
        // automatic shape)
        uint32_t min_vars = latency_auto ? pol.front_min_vars_one_solve : batch_auto ? pol.front_min_vars_batch : 0u;
        if (fronts_env == 0 || no_fronts) min_vars = 0;
        const bool connected = s.grid_wgs == 1 && P.c.n_parts == 1 && P.c.n_components >= 1 && P.c.n_components <= kRecMaxComponents;
        const bool want = force_fronts || (auto_shape && min_vars && n_vars >= min_vars && connected && !s.comp && !s.lane && !keep_comp);
        if (want && n_cs > 0) {
            const FrontOptions fo = front_options();
            std::unique_ptr<FrontPlan> plan(new FrontPlan());
            const char* why = nullptr;
            bool ok = front_plan_build(cs, n_cs, n_vars, fo, *plan, &why);
            // Wide fronts (a band of ten points, a comb: 40-63 rows, update matrices of a thousand entries and more) can cost more
            // than the record walk's rounds: one solve of a system the record walk holds in LDS is then left to it.  Both sides
            // estimated in cycles per LM iteration from what the symbolic phase has at this point -- fronts: 1.65 x the planner's
            // model (the model leaves out the sweeps, the assembly and the sums); record walk: 28 k + 1.75 k per level of its
            // elimination tree + 5 per entry of L -- fitted on the graph families of tests/gen.py (profiles/r05_graph_families.txt: of
            // 33 systems the fronts lost five by 12-25 %; with a margin of 5 % for the fronts the rule moves three of them to the
            // record walk -- the other two are within its error -- and none of the 28 winners).
            if (ok && !force_fronts && n_vars <= 1200 && s.grid_wgs == 1 && P.c.n_parts == 1) {
                const double fronts_est = 1.65 * plan->model_cycles;
                const double rec_est = 28000.0 + 1750.0 * (double)P.c.n_levels + 5.0 * (double)((uint64_t)P.c.zlo + P.c.n_vars);
                if (fronts_est > 1.05 * rec_est) {
                    ok = false;
                    why = "the record walk is estimated faster";
                }
            }
            if (ok) {
                if (for_latency || force_fronts) {
                    s.front_max_batch = ~0ull;
                } else {
                    // batches: as many systems as the device holds at once, times a round per `front_small_call_wgs_per_round`
                    // workgroups per system (policy.hpp)
                    const uint64_t cus = (uint64_t)std::max(1, s.lim.cus), G = plan->n_wgs;
                    const uint64_t rounds = std::max<uint64_t>(1, G / std::max<uint32_t>(1, pol.front_small_call_wgs_per_round));
                    s.front_max_batch = std::min<uint64_t>(std::max<uint64_t>(1, cus / G) * rounds, 0xFFFFFFFEull);
                }
                s.fronts = std::move(plan);
            } else if (debug_topic("front")) {
                std::fprintf(stderr, "front plan: not taken: %s\n", why ? why : "?");
            }
        }
    };
    // Where the fronts serve EVERY call of the system (one solve: the automatic latency shape, or asked for) and the system is one the
    // later plans would not take over (no component plan, not a sub-wavefront system), the plan is built now and the plans nobody would
    // run are skipped -- the record walk's rounds, the dense phases, the lanes across the batch: half of the symbolic phase of a
    // 2000-variable sketch (records 5.1 + dense phases 2.0 + lanes 3.8 of 21.8 ms on the build container's CPU).
    if ((latency_auto || force_fronts) && !s.comp && s.mode != MODE_SUB) build_fronts();
    const bool fronts_serve_all = s.fronts && s.front_max_batch == ~0ull;
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
    // 0.33 -> 0.61 M solves/s, 57 -> 86 k.
    const int saved_mode = s.mode;
    const uint32_t saved_team = s.team_size;
    const bool saved_lean = s.lean_lds;
    const bool rec_batch = rec_enabled && auto_shape && team_size == 0 && !for_latency && !want_sub &&
                           s.grid_wgs == 1 && P.c.n_parts == 1 && P.c.n_components >= 1 && P.c.n_components <= kRecMaxComponents && !P.c.dense;
    // Batches keep the Jacobian's values in global memory (SolveArgs::rec_jglobal) when the assembly can read them from packed
    // pairs (no list of more than twelve): a fifth of a system's LDS, one more workgroup per CU.
    constexpr bool jglobal_enabled = true;
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
        s.mode = MODE_WGB;
        s.team_size = t;
        s.lean_lds = true;  // (its lists stay in L2: the LDS is for as many systems as fit)
    }
    const bool rec_try = rec_enabled && !fronts_serve_all && auto_shape && ((for_latency && !latency_phases) || rec_batch) && s.mode == MODE_WGB &&
                         s.grid_wgs == 1 && P.c.n_parts == 1 && P.c.n_components >= 1 && P.c.n_components <= kRecMaxComponents && !P.c.dense;
    if (rec_try) {
        s.rec_jglobal = jglobal;
        s.rec_extra = (P.c.n_vars + 2 + 1) & ~1u;
#ifdef EZPZ_REC_TIMES
        s.rec_extra += 6 * 128;  // (diagnostic build: six cycle stamps per round of the second iteration's walk, behind the zero)
#endif
    }
    pack_and_shape(true);
    bool rec_wide = false;
    if (rec_try && !s.lds_ws) {
        // no room in the LDS with the walk's extra doubles: if the state fits without them the list walk keeps it there (with its dense
        // phases); a state that lives in global memory anyway walks records in the wide form
        constexpr bool wide_enabled = true;
        const uint32_t extra = s.rec_extra;
        s.rec_extra = 0;
        s.rec_jglobal = false;  // (J in global memory is for states that fit the LDS with it)
        pack_and_shape(true);
        // (batches: 4194 systems of 2000 variables 106 -> 114 k solves/s, 1677 of 5000 variables 6.3 -> 8.4 k; a round through
        // global memory is a store's acknowledgement, a rendezvous and a trip to L2)
        const uint32_t wide_one_solve_max = s.lim.policy.rec_wide_one_solve_max_vars;
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
    if (root_enabled && !fronts_serve_all && (auto_shape || lists_only) && s.grid_wgs == 1 && !s.rec &&
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
            if (debug_topic("dense"))
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
        if (debug_topic("rec"))
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
    if (auto_shape && lanes_enabled && !fronts_serve_all && !s.comp && !s.lane && s.grid_wgs == 1 && P.c.n_parts == 1 && n_vars > 20) {
        std::unique_ptr<BatchPlan> bp(new BatchPlan());
        if (batch_plan_build(cs, n_cs, n_vars, *bp)) s.lanes = std::move(bp);
    }
    // ---- the frontal shape (built ahead of the plans above where it serves every call, else here) -------------------------------------------
    if (!fronts_tried) build_fronts();
    if (s.fronts) {
        const FrontPlan& plan = *s.fronts;
        info.front_workgroups = plan.n_wgs;
        info.front_max_batch = (uint32_t)std::min<uint64_t>(s.front_max_batch, 0xFFFFFFFFull);
        if (s.front_max_batch == ~0ull) {
            info.team_mode = 5;
            info.team_size = plan.threads;
            info.grid_workgroups = plan.n_wgs;
            info.n_partitions = plan.n_fronts;
            info.n_levels = plan.n_levels;
            info.workspace_bytes = (uint64_t)plan.ws_doubles_max * 8;
            info.workspace_in_lds = 1;
            info.program_in_lds = 0;
        }
        info.program_bytes += plan.blob.size();
    }
    return EZPZ_OK;
}


}  // namespace ezpz
