// Symbolic phase of the FRONTAL launch shape (fronts.hpp, front_types.hpp): elimination order, supernodes, the tree of
// fronts, what each front assembles from the Jacobian, which workgroup owns which subtree, and the workspace carve-up.
// The reference's counterpart is faer's SymbolicLlt (ezpz/src/solver.rs:289-300: ordering, elimination tree, column counts,
// supernodal partition); nothing here is taken from it -- the structures are this kernel's own.  Host code, no device code.
#include "fronts.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <numeric>
#include <queue>

#include "kinds.hpp"
#include "policy.hpp"

namespace ezpz {

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
using UVec = std::vector<uint32_t>;

// What a front costs its wavefront, in cycles, from the stamps of front_kernel.hip.hpp on an MI355X (tools/front_stamps.py, K = 2 ...
// 16): the descriptor, the gather of the children's update matrices (a trip of 64 destinations ~ 300 cycles + ~500 of latency),
// the K dependent pivot steps (~150 cycles each: broadcast, 1 / sqrt by two Newton steps, scale) with the pivot block's K^2 / 2
// register updates (two v_readlane, a wait state, an fma each), and the Schur complement by trips of 64 entries (2 K operands
// each).  It decides which supernodes are merged and how subtrees are dealt to workgroups; only ratios matter.
double front_cost(uint32_t K, uint32_t S, bool children = true) {
    const uint32_t R = S - K, nU = (R + 1) * (R + 2) / 2;
    const double trips = (double)((nU + 63) / 64);
    return 180.0 + (children ? 800.0 : 0.0) + 700.0 + 40.0 * K + 17.0 * K * K + (R ? 500.0 + trips * (300.0 + 40.0 * K) : 0.0);
}
double front_bwd_cost(uint32_t K, uint32_t S) { return 900.0 + 60.0 * K + 15.0 * (S - K); }

// Nested dissection of one connected component by breadth-first level structures: the separator is the level that best
// balances the two sides by vertex count (thin levels preferred), sides are ordered first (recursively), the separator last;
// pieces of at most `leaf` vertices, or too compact to cut, are leaves in breadth-first order (a band stays a band).
// Minimum-degree ordering of one connected component (exact degrees on the explicit elimination graph, smallest (degree, id) first;
// lazily updated heap).  A random tree with chords, a comb: breadth-first levels there hold a third of the vertices and nested
// dissection by level structures makes fronts of hundreds of rows, while eliminating leaves first keeps them at the size of the
// cycles.  (The same rule as program.cpp's order_component, on this file's own graph.)
void min_degree(const UVec& verts, const std::vector<UVec>& adj, UVec& out) {
    const uint32_t k = (uint32_t)verts.size();
    static thread_local UVec local;
    uint32_t top = 0;
    for (uint32_t v : verts) top = std::max(top, v + 1);
    if (local.size() < top) local.resize(top);
    for (uint32_t i = 0; i < k; ++i) local[verts[i]] = i;
    std::vector<UVec> g(k);
    for (uint32_t i = 0; i < k; ++i) {
        for (uint32_t w : adj[verts[i]]) g[i].push_back(local[w]);
        std::sort(g[i].begin(), g[i].end());
    }
    std::vector<char> gone(k, 0);
    typedef std::pair<uint32_t, uint32_t> DegIdx;
    std::priority_queue<DegIdx, std::vector<DegIdx>, std::greater<DegIdx>> heap;
    for (uint32_t i = 0; i < k; ++i) heap.push(DegIdx((uint32_t)g[i].size(), i));
    UVec nb, merged;
    for (uint32_t step = 0; step < k; ++step) {
        uint32_t best = k;
        while (best == k) {
            const DegIdx t = heap.top();
            heap.pop();
            if (!gone[t.second] && g[t.second].size() == t.first) best = t.second;
        }
        gone[best] = 1;
        out.push_back(verts[best]);
        nb = g[best];
        for (uint32_t u : nb) {  // g[u] = (g[u] \ {best}) U (nb \ {u})
            merged.clear();
            std::set_union(g[u].begin(), g[u].end(), nb.begin(), nb.end(), std::back_inserter(merged));
            UVec& gu = g[u];
            gu.clear();
            for (uint32_t w : merged)
                if (w != u && w != best) gu.push_back(w);
            heap.push(DegIdx((uint32_t)gu.size(), u));
        }
        UVec().swap(g[best]);
    }
}

void dissect(const UVec& verts, const std::vector<UVec>& adj, uint32_t leaf, UVec& out) {
    const uint32_t n_all = (uint32_t)adj.size();
    static thread_local UVec region, level, mark;
    region.assign(n_all, NONE);
    level.assign(n_all, 0);
    mark.assign(n_all, 0);
    for (uint32_t v : verts) region[v] = 0;
    uint32_t next_region = 1, stamp = 0;
    struct Task {
        uint32_t region;
        bool emit;
        UVec members;
    };
    std::vector<Task> stack;
    stack.push_back(Task{0, false, verts});
    UVec queue;
    auto bfs = [&](uint32_t start, uint32_t reg) {
        ++stamp;
        queue.clear();
        queue.push_back(start);
        level[start] = 0;
        mark[start] = stamp;
        for (size_t h = 0; h < queue.size(); ++h) {
            const uint32_t v = queue[h];
            for (uint32_t w : adj[v])
                if (region[w] == reg && mark[w] != stamp) {
                    mark[w] = stamp;
                    level[w] = level[v] + 1;
                    queue.push_back(w);
                }
        }
    };
    while (!stack.empty()) {
        Task t = std::move(stack.back());
        stack.pop_back();
        if (t.emit) {
            out.insert(out.end(), t.members.begin(), t.members.end());
            continue;
        }
        for (uint32_t s0 : t.members) {
            if (region[s0] != t.region) continue;
            bfs(s0, t.region);
            bfs(queue.back(), t.region);  // pseudo-peripheral restart
            const uint32_t depth = level[queue.back()];
            if (queue.size() <= leaf || depth < 2) {
                for (uint32_t v : queue) {
                    out.push_back(v);
                    region[v] = NONE;
                }
                continue;
            }
            UVec width(depth + 1, 0);
            for (uint32_t v : queue) ++width[level[v]];
            // the cut: |left - right| + 2 x |separator|, levels 1 .. depth - 1
            uint32_t best = 1;
            uint64_t best_score = ~0ull;
            uint32_t below = width[0];
            const uint32_t total = (uint32_t)queue.size();
            for (uint32_t l = 1; l < depth; ++l) {
                const uint32_t above = total - below - width[l];
                const uint64_t score = (uint64_t)(below > above ? below - above : above - below) + 2ull * width[l];
                if (score < best_score) best_score = score, best = l;
                below += width[l];
            }
            Task sep{0, true, {}}, lo{next_region, false, {}}, hi{next_region + 1, false, {}};
            next_region += 2;
            for (uint32_t v : queue) {
                if (level[v] == best) {
                    sep.members.push_back(v);
                    region[v] = NONE;
                } else if (level[v] < best) {
                    lo.members.push_back(v);
                    region[v] = lo.region;
                } else {
                    hi.members.push_back(v);
                    region[v] = hi.region;
                }
            }
            stack.push_back(std::move(sep));
            stack.push_back(std::move(hi));
            stack.push_back(std::move(lo));
        }
    }
}

struct Blob {
    std::vector<unsigned char>& b;
    template <class T>
    uint32_t put(const std::vector<T>& v) {
        const size_t off = (b.size() + 15) & ~size_t(15);
        b.resize(off + std::max<size_t>(v.size() * sizeof(T), 16), 0);
        if (!v.empty()) std::memcpy(b.data() + off, v.data(), v.size() * sizeof(T));
        return (uint32_t)off;
    }
};

bool fail(const char** why, const char* msg) {
    if (why) *why = msg;
    return false;
}

}  // namespace

// The symbolic phase of the frontal shape, phase by phase (run(): one attempt with one set of options; front_plan_build retries with
// another ordering or smaller shares).  What a phase leaves for the later ones are the members.
struct FrontPlanner {
    struct ConInfo {
        uint32_t row0, jbase;
        uint8_t nslots, jloc[16];
    };
    struct Block {
        uint32_t c0, c1;  // columns [c0, c1)
    };
    struct Front {
        uint32_t c0, K, S;
        UVec below;  // rows beyond the pivots (positions, ascending)
        uint32_t parent = NONE, level = 0, wg = 0, local = 0;
        UVec kids;
        UVec cons;  // constraints assembled here
        double cost = 0.0, subtree = 0.0;
    };
    // the widest supernode: KMAX columns (measured: supernodes(), below); a front fits one wavefront: SMAX rows
    static constexpr uint32_t KMAX = 8, SMAX = kFrontMaxRows;

    const EzpzConstraint* const cs;
    const uint32_t n, C;
    const FrontOptions& opt;
    FrontPlan& out;
    const char** const why;
    const bool debug = debug_topic("front");

    // number_rows: rows and Jacobian slots as build_program numbers them
    std::vector<ConInfo> cinfo;
    UVec row_ptr, row_col, row_slot, row_con;
    std::vector<UVec> cvars;  // the variables a constraint's rows touch (sorted, unique)
    // build_graph ... postorder: the graph, the elimination order and its inverse, L's column structures, the elimination tree
    std::vector<UVec> adj;
    UVec order, pos;
    std::vector<UVec> cstruct;  // rows > j of column j of L, ascending
    UVec parent;
    // supernodes, build_fronts, home_fronts
    std::vector<Block> blocks;
    uint32_t F = 0;
    UVec block_of;
    std::vector<Front> fr;
    // deal_workgroups, workgroup_shares: who eliminates what, and the chunks that cross workgroups
    uint32_t G = 1;
    std::vector<UVec> wg_fronts, wg_ghost, wg_cons;  // (ghosts: positions)
    uint32_t n_chunks = 0, bad_chunk0 = 0, verdict_chunk = 0;
    UVec up_chunk, export_chunk;  // (exports: by position)
    // emit
    std::vector<FrontWg> wgs;
    Blob B;
    uint32_t waves = 1;
    double model_top = 0.0, model_sub = 0.0;
    size_t max_fronts_wg = 0;

    FrontPlanner(const EzpzConstraint* cs_, size_t n_cs, size_t n_vars, const FrontOptions& opt_, FrontPlan& out_, const char** why_)
        : cs(cs_), n((uint32_t)n_vars), C((uint32_t)n_cs), opt(opt_), out(out_), why(why_), B{out_.blob} {}

    bool number_rows();
    void build_graph();
    bool order_elimination();
    void symbolic();
    void column_structures();
    void postorder();
    void supernodes();
    bool build_fronts();
    bool home_fronts();
    void deal_workgroups();
    bool workgroup_shares();
    bool wavefront_schedules(uint32_t g, const UVec& fl, std::vector<uint16_t>& sched, UVec& sched_par, UVec& sched_nkids);
    bool emit_workgroup(uint32_t g);
    bool emit();
    bool run() {
        if (!number_rows()) return false;
        build_graph();
        if (!order_elimination()) return false;
        column_structures();
        postorder();
        supernodes();
        if (!build_fronts() || !home_fronts()) return false;
        deal_workgroups();
        return workgroup_shares() && emit();
    }
};

bool FrontPlanner::number_rows() {
    // ---- rows and Jacobian slots, numbered as build_program numbers them (constraint order; equal columns of a row share a
    //      slot: solver.rs:255-260, :418) ---------------------------------------------------------------------------------
    cinfo.assign(C, ConInfo{});
    row_ptr.assign(1, 0);
    cvars.assign(C, UVec());
    {
        uint32_t row_num = 0, jslot = 0;
        for (uint32_t i = 0; i < C; ++i) {
            if (cs[i].kind >= EZPZ_NUM_KINDS) return fail(why, "unknown kind");
            const KindInfo& K = kKinds[cs[i].kind];
            ConInfo& ci = cinfo[i];
            std::memset(&ci, 0, sizeof(ci));
            ci.row0 = row_num;
            ci.jbase = jslot;
            uint32_t local = 0;
            int e_global = 0;
            for (int r = 0; r < K.n_rows; ++r) {
                for (int e = 0; e < K.n_emit[r]; ++e, ++e_global) {
                    const uint32_t col = cs[i].ids[K.emit[r][e]];
                    if (col >= n) return fail(why, "variable id out of range");
                    int dup = -1;
                    for (int e2 = 0; e2 < e; ++e2)
                        if (cs[i].ids[K.emit[r][e2]] == col) {
                            dup = e2;
                            break;
                        }
                    if (dup >= 0) {
                        ci.jloc[e_global] = (uint8_t)((ci.jloc[e_global - e + dup] & 0x7F) | 0x80);
                    } else {
                        ci.jloc[e_global] = (uint8_t)local;
                        row_col.push_back(col);
                        row_slot.push_back(jslot + local);
                        ++local;
                    }
                    cvars[i].push_back(col);
                }
                for (int e = 0; e < K.n_nz[r]; ++e) {
                    const uint32_t col = cs[i].ids[K.nz[r][e]];
                    if (col >= n) return fail(why, "variable id out of range");
                    cvars[i].push_back(col);
                }
                ++row_num;
                row_ptr.push_back((uint32_t)row_col.size());
                row_con.push_back(i);
            }
            ci.nslots = (uint8_t)local;
            jslot += local;
            std::sort(cvars[i].begin(), cvars[i].end());
            cvars[i].erase(std::unique(cvars[i].begin(), cvars[i].end()), cvars[i].end());
            if (cs[i].weight != 1.0) out.unit_weights = false;
            if (!kind_is_linear(cs[i].kind)) out.linear_only = false;
        }
        out.n_rows = row_num;
        out.zj = jslot;
    }
    out.n_vars = n;
    out.n_cons = C;
    return true;
}

void FrontPlanner::build_graph() {
    // ---- the graph: every constraint a clique of all the variables it touches (one home front per constraint: whoever
    //      evaluates it owns all its rows) ---------------------------------------------------------------------------------
    adj.assign(n, UVec());
    for (uint32_t i = 0; i < C; ++i)
        for (uint32_t a : cvars[i])
            for (uint32_t b : cvars[i])
                if (a != b) adj[a].push_back(b);
    for (uint32_t v = 0; v < n; ++v) {
        std::sort(adj[v].begin(), adj[v].end());
        adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end());
    }
}

bool FrontPlanner::order_elimination() {
    // ---- elimination order: nested dissection per connected component -------------------------------------------------------
    order.clear();
    order.reserve(n);
    {
        std::vector<char> seen(n, 0);
        UVec verts, stack;
        constexpr uint32_t leaf = 10;  // (pieces of at most this many variables are not dissected further)
        out.n_components = 0;
        out.ordering = opt.ordering;
        for (uint32_t s = 0; s < n; ++s) {
            if (seen[s]) continue;
            ++out.n_components;
            verts.clear();
            stack.assign(1, s);
            seen[s] = 1;
            while (!stack.empty()) {
                const uint32_t v = stack.back();
                stack.pop_back();
                verts.push_back(v);
                for (uint32_t w : adj[v])
                    if (!seen[w]) seen[w] = 1, stack.push_back(w);
            }
            std::sort(verts.begin(), verts.end());
            if (verts.size() <= leaf)
                order.insert(order.end(), verts.begin(), verts.end());
            else if (opt.ordering == 1)
                min_degree(verts, adj, order);
            else
                dissect(verts, adj, leaf, order);
        }
        if (order.size() != n) return fail(why, "ordering lost a variable");
    }
    pos.assign(n, 0);
    for (uint32_t k = 0; k < n; ++k) pos[order[k]] = k;
    return true;
}

void FrontPlanner::symbolic() {
    std::vector<UVec> kids(n);
    for (uint32_t j = 0; j < n; ++j) {
        UVec& s = cstruct[j];
        s.clear();
        for (uint32_t w : adj[order[j]])
            if (pos[w] > j) s.push_back(pos[w]);
        for (uint32_t c : kids[j])
            for (uint32_t i : cstruct[c])
                if (i != j) s.push_back(i);
        std::sort(s.begin(), s.end());
        s.erase(std::unique(s.begin(), s.end()), s.end());
        parent[j] = s.empty() ? NONE : s[0];
        if (parent[j] != NONE) kids[parent[j]].push_back(j);
    }
}

void FrontPlanner::column_structures() {
    // ---- column structures and the elimination tree (positions) -------------------------------------------------------------
    cstruct.assign(n, UVec());
    parent.assign(n, NONE);
    symbolic();
}

void FrontPlanner::postorder() {
    // ---- postorder (the tallest child last, next to its parent), then the same again in the new numbering -------------------
    {
        std::vector<UVec> kids(n);
        UVec height(n, 1);
        for (uint32_t j = 0; j < n; ++j)
            if (parent[j] != NONE) {
                kids[parent[j]].push_back(j);
                height[parent[j]] = std::max(height[parent[j]], height[j] + 1);
            }
        UVec post;
        post.reserve(n);
        std::vector<std::pair<uint32_t, uint32_t>> st;  // (node, next child)
        for (uint32_t r = 0; r < n; ++r) {
            if (parent[r] != NONE) continue;
            st.push_back({r, 0});
            while (!st.empty()) {
                auto& [v, k] = st.back();
                if (k == 0) std::stable_sort(kids[v].begin(), kids[v].end(), [&](uint32_t a, uint32_t b) { return height[a] < height[b]; });
                if (k < kids[v].size()) {
                    const uint32_t c = kids[v][k++];
                    st.push_back({c, 0});
                } else {
                    post.push_back(v);
                    st.pop_back();
                }
            }
        }
        UVec order2(n);
        for (uint32_t k = 0; k < n; ++k) order2[k] = order[post[k]];
        order.swap(order2);
        for (uint32_t k = 0; k < n; ++k) pos[order[k]] = k;
        symbolic();
    }
}

void FrontPlanner::supernodes() {
    // ---- supernodes: fundamental ones first, then a child is merged into the parent it immediately precedes while the merged
    //      front is cheaper than the two (front_cost: a front's fixed cost is most of a small front) ----------------------------
    // the widest supernode: 8 columns (measured, one solve of 800 / 2000 / 5000 variables with supernodes of at most 6 / 8 / 16
    // columns: 2.09 / 2.08 / 2.35 ms, 0.52 / 0.52 / 0.62 ms, 3.24 / 3.16 / 4.06 ms: the pivot block's register updates grow with K^2;
    // the kernel itself takes up to kFrontMaxPivots; up to 12 or 16 columns where the parent has ONE child -- a chain, no parallelism
    // to lose -- with the wavefronts' schedules: 500 / 800 / 2000 / 5000 variables 343 -> 359, 1798 -> 1750, 450 -> 462, 2649 -> 2743 us)
    UVec nkids(n, 0);
    for (uint32_t j = 0; j < n; ++j)
        if (parent[j] != NONE) ++nkids[parent[j]];
    blocks.clear();
    for (uint32_t j = 0; j < n; ++j) {
        const bool chain = j > 0 && parent[j - 1] == j && nkids[j] == 1 && cstruct[j - 1].size() == cstruct[j].size() + 1 &&
                           !blocks.empty() && j - blocks.back().c0 < KMAX;
        if (chain)
            blocks.back().c1 = j + 1;
        else
            blocks.push_back(Block{j, j + 1});
    }
    {
        constexpr bool relax = true;
        // rows of a block = its columns + the structure of its last column (a block's columns chain: each column's structure
        // is the next one's plus itself) -- until blocks are merged: then the structure of the merged block is that of the
        // parent, its rows the child's columns + the parent's rows
        std::vector<Block> merged;
        std::vector<uint32_t> srows;  // S of each merged block
        for (const Block& b : blocks) {
            Block cur = b;
            uint32_t S = (cur.c1 - cur.c0) + (uint32_t)cstruct[cur.c1 - 1].size();
            while (relax && !merged.empty()) {
                const Block& c = merged.back();
                if (c.c1 != cur.c0 || parent[c.c1 - 1] == NONE || parent[c.c1 - 1] < cur.c0 || parent[c.c1 - 1] >= cur.c1) break;
                const uint32_t Kc = c.c1 - c.c0, Sc = srows.back(), K2 = Kc + (cur.c1 - cur.c0), S2 = Kc + S;
                if (K2 > KMAX || S2 > SMAX) break;
                const double apart = front_cost(Kc, Sc) + front_bwd_cost(Kc, Sc) + front_cost(cur.c1 - cur.c0, S) + front_bwd_cost(cur.c1 - cur.c0, S);
                const double joined = front_cost(K2, S2) + front_bwd_cost(K2, S2);
                if (joined > apart) break;
                cur.c0 = c.c0;
                S = S2;
                merged.pop_back();
                srows.pop_back();
            }
            merged.push_back(cur);
            srows.push_back(S);
        }
        blocks.swap(merged);
    }
    F = (uint32_t)blocks.size();
    block_of.assign(n, 0);
    for (uint32_t f = 0; f < F; ++f)
        for (uint32_t j = blocks[f].c0; j < blocks[f].c1; ++j) block_of[j] = f;
}

bool FrontPlanner::build_fronts() {
    // ---- the fronts: rows, parents, children, levels --------------------------------------------------------------------------
    fr.assign(F, Front());
    for (uint32_t f = 0; f < F; ++f) {
        Front& t = fr[f];
        t.c0 = blocks[f].c0;
        t.K = blocks[f].c1 - blocks[f].c0;
        for (uint32_t j = blocks[f].c0; j < blocks[f].c1; ++j)
            for (uint32_t i : cstruct[j])
                if (i >= blocks[f].c1) t.below.push_back(i);
        std::sort(t.below.begin(), t.below.end());
        t.below.erase(std::unique(t.below.begin(), t.below.end()), t.below.end());
        t.S = t.K + (uint32_t)t.below.size();
        if (t.S > SMAX) return fail(why, "a front has more than 63 rows");
        t.parent = t.below.empty() ? NONE : block_of[t.below[0]];
        t.cost = front_cost(t.K, t.S) + front_bwd_cost(t.K, t.S);
        out.max_rows = std::max(out.max_rows, t.S);
        out.max_pivots = std::max(out.max_pivots, t.K);
    }
    for (uint32_t f = 0; f < F; ++f) {
        if (fr[f].parent == NONE) continue;
        Front& p = fr[fr[f].parent];
        p.kids.push_back(f);
        p.level = std::max(p.level, fr[f].level + 1);  // (children precede parents)
        // every row of the child beyond its pivots is a row of the parent
        for (uint32_t i : fr[f].below) {
            const bool in_p = (i >= p.c0 && i < p.c0 + p.K) || std::binary_search(p.below.begin(), p.below.end(), i);
            if (!in_p) return fail(why, "internal: a child's row is missing from its parent");
        }
    }
    for (uint32_t f = 0; f < F; ++f) {
        fr[f].subtree += fr[f].cost;
        if (fr[f].parent != NONE) fr[fr[f].parent].subtree += fr[f].subtree;
    }
    return true;
}

bool FrontPlanner::home_fronts() {
    // ---- home fronts of the constraints: where their earliest variable is eliminated ---------------------------------------------
    for (uint32_t i = 0; i < C; ++i) {
        uint32_t first = NONE;
        for (uint32_t v : cvars[i]) first = std::min(first, pos[v]);
        if (first == NONE) return fail(why, "a constraint without variables");
        Front& h = fr[block_of[first]];
        for (uint32_t v : cvars[i]) {
            const uint32_t p = pos[v];
            const bool in_h = (p >= h.c0 && p < h.c0 + h.K) || std::binary_search(h.below.begin(), h.below.end(), p);
            if (!in_h) return fail(why, "internal: a constraint's variable is missing from its home front");
        }
        h.cons.push_back(i);
    }
    return true;
}

void FrontPlanner::deal_workgroups() {
    // ---- workgroups: the top of the tree on workgroup 0, whole subtrees dealt to the others ------------------------------------
    G = opt.wgs;
    double total_cost = 0.0;
    for (uint32_t f = 0; f < F; ++f) total_cost += fr[f].cost;
    if (G == 0) {
        // (one solve, 160 / 240 / 320 variables per workgroup: 800 variables 2.06 / 2.25 / 2.29 ms, 2000: 0.51 / 0.52 / 0.56, 5000: 3.14 /
        // 2.84 / 3.07, 10 000: 1.10 / 0.90 / 0.95 -- the top of the tree, on workgroup 0, grows with the number of subtrees)
        uint32_t per_wg = std::max(16u, opt.vars_per_wg);
        if (n > 16 * per_wg) per_wg += per_wg / 2;
        G = n <= 2 * per_wg ? 1u : std::min<uint32_t>(opt.max_wgs, (n + per_wg - 1) / per_wg + 1);
    }
    G = std::max(1u, std::min<uint32_t>(G, std::min<uint32_t>(opt.max_wgs, kFrontMaxWgs)));
    if (G > 1) {
        // open subtrees, heaviest first; a subtree's root moves to the top (workgroup 0) and its children open while the
        // heaviest open subtree is more than its fair share of what is not yet on top
        using Item = std::pair<double, uint32_t>;
        std::priority_queue<Item> open;
        for (uint32_t f = 0; f < F; ++f)
            if (fr[f].parent == NONE) open.push(Item{fr[f].subtree, f});
        std::vector<char> on_top(F, 0);
        double top_cost = 0.0;
        uint32_t top_vars = 0;
        constexpr double share = 0.75;
        while (!open.empty()) {
            const Item it = open.top();
            const double rest = total_cost - top_cost;
            if (open.size() >= G - 1 && it.first <= share * rest / (G - 1)) break;
            if (fr[it.second].kids.empty()) break;  // a leaf front cannot be split
            open.pop();
            on_top[it.second] = 1;
            top_cost += fr[it.second].cost;
            top_vars += fr[it.second].K;
            for (uint32_t c : fr[it.second].kids) open.push(Item{fr[c].subtree, c});
        }
        // longest-processing-time first into G - 1 bins
        std::vector<Item> subs;
        while (!open.empty()) subs.push_back(open.top()), open.pop();
        std::vector<double> load(G, 0.0);
        std::vector<uint32_t> owner(F, 0);
        UVec root_wg(F, 0);
        for (const Item& it : subs) {
            uint32_t best = 1;
            for (uint32_t g = 2; g < G; ++g)
                if (load[g] < load[best]) best = g;
            load[best] += it.first;
            root_wg[it.second] = best;
        }
        // propagate: a front not on top belongs to its subtree root's workgroup (parents come later: walk down from the roots)
        for (uint32_t f = F; f-- > 0;) {
            if (on_top[f]) {
                fr[f].wg = 0;
            } else if (fr[f].parent == NONE || on_top[fr[f].parent]) {
                fr[f].wg = root_wg[f];
            } else {
                fr[f].wg = fr[fr[f].parent].wg;
            }
        }
        // workgroups that received nothing: renumber densely (workgroup 0 stays the top even when it is empty of subtrees)
        std::vector<char> used(G, 0);
        used[0] = 1;
        for (uint32_t f = 0; f < F; ++f) used[fr[f].wg] = 1;
        UVec renum(G, 0);
        uint32_t g2 = 0;
        for (uint32_t g = 0; g < G; ++g)
            if (used[g]) renum[g] = g2++;
        for (uint32_t f = 0; f < F; ++f) fr[f].wg = renum[fr[f].wg];
        G = g2;
        bool any_top = false;
        for (uint32_t f = 0; f < F; ++f) any_top = any_top || fr[f].wg == 0;
        if (!any_top || G < 2) {  // nothing to split: one workgroup
            for (uint32_t f = 0; f < F; ++f) fr[f].wg = 0;
            G = 1;
        }
        (void)top_vars;
    }
    out.n_wgs = G;
}

bool FrontPlanner::workgroup_shares() {
    // ---- per workgroup: fronts by (level within the workgroup, cost), local variables, constraints ------------------------------
    // Levels are per workgroup: a front's level = 1 + the highest level of its children IN THE SAME workgroup (children in other
    // workgroups arrive as chunks and are waited for).
    for (uint32_t f = 0; f < F; ++f) {
        fr[f].level = 0;
    }
    for (uint32_t f = 0; f < F; ++f)
        if (fr[f].parent != NONE && fr[fr[f].parent].wg == fr[f].wg)
            fr[fr[f].parent].level = std::max(fr[fr[f].parent].level, fr[f].level + 1);
    wg_fronts.assign(G, UVec());
    for (uint32_t f = 0; f < F; ++f) wg_fronts[fr[f].wg].push_back(f);
    // exported variables: pivots of workgroup 0 that another workgroup sees as a ghost (decided below); chunk layout of a system's
    // scratch: [update matrices of fronts with a remote parent][one flag per workgroup: a pivot failed][steps of the exported
    // variables][the verdict on the factorisation]
    n_chunks = 0;
    up_chunk.assign(F, NONE);
    for (uint32_t f = 0; f < F; ++f)
        if (fr[f].parent != NONE && fr[fr[f].parent].wg != fr[f].wg) {
            const uint32_t R = fr[f].S - fr[f].K;
            up_chunk[f] = n_chunks;
            n_chunks += (R + 1) * (R + 2) / 2;
        }
    bad_chunk0 = n_chunks;
    if (G > 1) n_chunks += G;
    export_chunk.assign(n, NONE);
    // ghosts first (they decide the exports)
    wg_ghost.assign(G, UVec());
    wg_cons.assign(G, UVec());
    for (uint32_t g = 0; g < G; ++g) {
        UVec& gh = wg_ghost[g];
        for (uint32_t f : wg_fronts[g]) {
            for (uint32_t i : fr[f].below)
                if (fr[block_of[i]].wg != g) gh.push_back(i);
            for (uint32_t c : fr[f].cons) wg_cons[g].push_back(c);
        }
        std::sort(gh.begin(), gh.end());
        gh.erase(std::unique(gh.begin(), gh.end()), gh.end());
        for (uint32_t i : gh) {
            if (fr[block_of[i]].wg != 0) return fail(why, "internal: a ghost that is not eliminated on workgroup 0");
            if (export_chunk[i] == NONE) export_chunk[i] = 0;  // marked; numbered below
        }
    }
    for (uint32_t j = 0; j < n; ++j)
        if (export_chunk[j] != NONE) export_chunk[j] = n_chunks++;
    verdict_chunk = n_chunks;
    if (G > 1) ++n_chunks;
    out.n_chunks = n_chunks;
    return true;
}

// ---- the wavefronts' schedules: list scheduling of the workgroup's fronts on `waves` wavefronts by the cost model -- forward:
//      a front is ready when its children in this workgroup are done (children elsewhere arrive as chunks, whenever), the
//      ready front with the longest way up to the top goes to the wavefront that is free first; backward: ready when the
//      parent is done, the longest way down first.  A wavefront runs its list in order and waits only for what its next
//      front needs (front_kernel.hip.hpp): the per-level barriers cost the 300-variable sketch 11 rounds of fronts where
//      the tree's critical path is 6.  (Deadlock-free: a front's dependencies start earlier in the simulated time than it
//      does, on whatever wavefront, and every list is in simulated start order.)  Measured and not kept: ONE list in the
//      order of the simulated starts from which a free wavefront takes the next front (a counter in LDS) -- it does not
//      depend on the cost model's accuracy, but its in-order takes block more than they balance: one solve of 150 / 300 /
//      2000 variables 96 -> 103, 179 -> 189, 450 -> 463 us (10 000: 924 -> 887).
bool FrontPlanner::wavefront_schedules(const uint32_t g, const UVec& fl, std::vector<uint16_t>& sched, UVec& sched_par, UVec& sched_nkids) {
    const uint32_t nf = (uint32_t)fl.size();
    UVec loc_of(F, NONE);
    for (uint32_t k = 0; k < nf; ++k) loc_of[fl[k]] = k;
    UVec par(nf, NONE), nkids(nf, 0);
    std::vector<UVec> kid_list(nf);
    for (uint32_t k = 0; k < nf; ++k) {
        const uint32_t p = fr[fl[k]].parent;
        if (p != NONE && fr[p].wg == g) {
            par[k] = loc_of[p];
            ++nkids[par[k]];
            kid_list[par[k]].push_back(k);
        }
    }
    std::vector<double> cf(nf), cb(nf), up(nf, 0.0), down(nf, 0.0);
    for (uint32_t k = 0; k < nf; ++k) {
        cf[k] = front_cost(fr[fl[k]].K, fr[fl[k]].S, !fr[fl[k]].kids.empty());
        cb[k] = front_bwd_cost(fr[fl[k]].K, fr[fl[k]].S);
    }
    // (fl is sorted by level: parents behind their children)
    for (uint32_t k = nf; k-- > 0;) up[k] = cf[k] + (par[k] != NONE ? up[par[k]] : 0.0);
    for (uint32_t k = 0; k < nf; ++k) {
        down[k] += cb[k];
        if (par[k] != NONE) down[par[k]] = std::max(down[par[k]], down[k]);
    }
    // (down[k] so far = cb[k] + the longest way below: recompute top-down as a priority = longest way down including itself)
    auto run = [&](bool forward, std::vector<UVec>& lists) -> double {
        lists.assign(waves, UVec());
        std::vector<double> free_at(waves, 0.0), done_at(nf, 0.0);
        UVec waiting(nf, 0);
        std::vector<uint32_t> ready;
        for (uint32_t k = 0; k < nf; ++k) {
            waiting[k] = forward ? nkids[k] : (par[k] != NONE ? 1u : 0u);
            if (!waiting[k]) ready.push_back(k);
        }
        std::vector<double> ready_at(nf, 0.0);
        uint32_t left = nf;
        double makespan = 0.0;
        while (left) {
            // the wavefront that is free first takes, of the fronts ready by then (or the one ready soonest), the one with
            // the highest priority
            uint32_t w = 0;
            for (uint32_t i = 1; i < waves; ++i)
                if (free_at[i] < free_at[w]) w = i;
            double soonest = 1e300;
            for (uint32_t k : ready) soonest = std::min(soonest, ready_at[k]);
            const double now = std::max(free_at[w], soonest);
            size_t best = ready.size();
            for (size_t i = 0; i < ready.size(); ++i) {
                const uint32_t k = ready[i];
                if (ready_at[k] > now) continue;
                const double pr = forward ? up[k] : down[k];
                if (best == ready.size() || pr > (forward ? up[ready[best]] : down[ready[best]])) best = i;
            }
            const uint32_t k = ready[best];
            ready.erase(ready.begin() + (long)best);
            const double end = now + (forward ? cf[k] : cb[k]) + 60.0;
            free_at[w] = end;
            done_at[k] = end;
            makespan = std::max(makespan, end);
            lists[w].push_back(k);
            --left;
            if (forward) {
                if (par[k] != NONE) {
                    ready_at[par[k]] = std::max(ready_at[par[k]], end);
                    if (--waiting[par[k]] == 0) ready.push_back(par[k]);
                }
            } else {
                for (uint32_t c : kid_list[k]) {
                    ready_at[c] = end;
                    waiting[c] = 0;
                    ready.push_back(c);
                }
            }
        }
        return makespan;
    };
    std::vector<UVec> fwd, bwd;
    const double span_f = run(true, fwd), span_b = run(false, bwd);
    sched.assign(2 * (waves + 1), 0);
    for (int pass = 0; pass < 2; ++pass) {
        const std::vector<UVec>& L = pass ? bwd : fwd;
        for (uint32_t w = 0; w < waves; ++w) {
            sched[(size_t)pass * (waves + 1) + w] = (uint16_t)sched.size();
            for (uint32_t k : L[w]) sched.push_back((uint16_t)k);
        }
        sched[(size_t)pass * (waves + 1) + waves] = (uint16_t)sched.size();
    }
    if (sched.size() >= 65535) return fail(why, "a workgroup's schedule does not fit 16-bit offsets");
    if (sched.size() & 1) sched.push_back(0);
    for (uint32_t k = 0; k < nf; ++k) {
        sched_par.push_back(par[k]);
        sched_nkids.push_back(nkids[k]);
    }
    // the model: the schedules' makespans instead of the levels' busiest wavefronts
    if (g == 0)
        model_top = span_f + span_b;
    else
        model_sub = std::max(model_sub, span_f + span_b);
    return true;
}

// One workgroup's share of the plan: its fronts by level, the wavefronts' schedules, tables, assembly streams, workspace layout.
bool FrontPlanner::emit_workgroup(const uint32_t g) {
    FrontWg& W = wgs[g];
    std::memset(&W, 0, sizeof(W));
    UVec& fl = wg_fronts[g];
    UVec sched_par, sched_nkids;
    std::stable_sort(fl.begin(), fl.end(), [&](uint32_t a, uint32_t b) {
        if (fr[a].level != fr[b].level) return fr[a].level < fr[b].level;
        return fr[a].cost > fr[b].cost;
    });
    uint32_t nlev = 0;
    for (uint32_t f : fl) nlev = std::max(nlev, fr[f].level + 1);
    UVec level_ptr(nlev + 1, 0);
    for (uint32_t f : fl) ++level_ptr[fr[f].level + 1];
    for (uint32_t l = 0; l < nlev; ++l) level_ptr[l + 1] += level_ptr[l];
    // model: per level the busiest wavefront
    {
        double up = 0.0;
        for (uint32_t l = 0; l < nlev; ++l) {
            std::vector<double> w(waves, 0.0);
            for (uint32_t k = level_ptr[l]; k < level_ptr[l + 1]; ++k) w[(k - level_ptr[l]) % waves] += fr[fl[k]].cost;
            up += *std::max_element(w.begin(), w.end()) + 150.0;
        }
        if (g == 0)
            model_top = up;  // (workgroup 0 runs after the others)
        else
            model_sub = std::max(model_sub, up);
    }
    // (the wavefronts' schedules: wavefront_schedules, above)
    std::vector<uint16_t> sched;
    if (!wavefront_schedules(g, fl, sched, sched_par, sched_nkids)) return false;
    // local variables: own pivots in front order, ghosts behind
    UVec local_of(n, NONE);  // position -> local index
    UVec var_glob;
    for (uint32_t k = 0; k < fl.size(); ++k) {
        Front& t = fr[fl[k]];
        t.local = k;
        for (uint32_t j = t.c0; j < t.c0 + t.K; ++j) {
            local_of[j] = (uint32_t)var_glob.size();
            var_glob.push_back(order[j]);
        }
    }
    W.n_own = (uint32_t)var_glob.size();
    std::vector<FrontGhost> ghosts;
    for (uint32_t i : wg_ghost[g]) {
        local_of[i] = (uint32_t)var_glob.size();
        ghosts.push_back(FrontGhost{local_of[i], export_chunk[i]});
        var_glob.push_back(order[i]);
    }
    W.n_ghost = (uint32_t)ghosts.size();
    W.n_loc = (uint32_t)var_glob.size();
    if (W.n_loc >= 65535) return fail(why, "a workgroup's variables do not fit 16-bit indices");
    // constraints: by kind (the sweeps' lanes then mostly run the same evaluator), rows and slots renumbered
    UVec& cl = wg_cons[g];
    std::stable_sort(cl.begin(), cl.end(), [&](uint32_t a, uint32_t b) { return cs[a].kind < cs[b].kind; });
    std::vector<DevCon> dcons(cl.size());
    UVec lrow0(C, NONE), ljbase(C, NONE);
    uint32_t lrows = 0, lslots = 0;
    for (uint32_t k = 0; k < cl.size(); ++k) {
        const uint32_t i = cl[k];
        const KindInfo& K = kKinds[cs[i].kind];
        DevCon& d = dcons[k];
        std::memset(&d, 0, sizeof(d));
        for (int e = 0; e < 8; ++e) {
            const uint32_t v = e < K.n_ids ? cs[i].ids[e] : NONE;
            // (ids a kind lists but neither differentiates nor reads -- a circle's centre for CircleRadius -- may lie outside
            // this workgroup's variables: any valid index serves)
            d.ids[e] = (v != NONE && v < n && local_of[pos[v]] != NONE) ? local_of[pos[v]] : 0u;
        }
        d.param = cs[i].param;
        d.weight = cs[i].weight;
        d.row0 = lrows;
        d.jbase = lslots;
        d.pos = i;
        d.kind = (uint8_t)cs[i].kind;
        d.tag = cs[i].tag;
        d.nrows = K.n_rows;
        d.nslots = cinfo[i].nslots;
        std::memcpy(d.jloc, cinfo[i].jloc, 16);
        lrow0[i] = lrows;
        ljbase[i] = lslots;
        lrows += K.n_rows;
        lslots += cinfo[i].nslots;
    }
    W.n_cons = (uint32_t)cl.size();
    W.n_rows = lrows;
    W.zj = lslots;
    // which entry of J a slot is (FrontWg::o_slotmap)
    std::vector<uint16_t> slotmap(lslots, 0);
    for (uint32_t k = 0; k < cl.size(); ++k) {
        const uint32_t i = cl[k];
        for (int r = 0; r < kKinds[cs[i].kind].n_rows; ++r) {
            const uint32_t grow = cinfo[i].row0 + (uint32_t)r;
            for (uint32_t q = row_ptr[grow]; q < row_ptr[grow + 1]; ++q) {
                const uint32_t lv = local_of[pos[row_col[q]]];
                if (lv == NONE || lv >= 0x8000u) return fail(why, "internal: a Jacobian column without a local variable");
                slotmap[row_slot[q] - cinfo[i].jbase + ljbase[i]] = (uint16_t)(lv | ((uint32_t)r << 15));
            }
        }
    }
    if (lslots >= 65535 || lrows >= 65535) return fail(why, "a workgroup's Jacobian does not fit 16-bit indices");
    // ---- workspace carve-up ------------------------------------------------------------------------------------------------
    uint32_t off = 0;
    auto take = [&](uint32_t doubles) {
        const uint32_t o = off;
        off += (doubles + 1) & ~1u;
        return o;
    };
    W.l_x = take(W.n_loc);
    W.l_d = take(W.n_loc);
    W.l_r = take(lrows + 1);   // (+ the zero row padding operands read)
    W.l_rn = take(lrows + 1);
    W.l_jv = take(lslots + 1);  // (+ the zero slot)
    W.l_panels = off;
    (void)take(2);  // (the first double of the region stays zero: the padding source of the source streams)
    std::vector<FrontDesc> descs(fl.size());
    std::vector<FrontChild> children;
    std::vector<uint16_t> rows;
    UVec exports;
    std::vector<uint8_t> maps;
    UVec stream_words;
    for (uint32_t k = 0; k < fl.size(); ++k) {
        const Front& t = fr[fl[k]];
        FrontDesc& d = descs[k];
        std::memset(&d, 0, sizeof(d));
        d.K = (uint16_t)t.K;
        d.S = (uint16_t)t.S;
        d.n_kids_local = (uint16_t)sched_nkids[k];
        d.parent_local = sched_par[k];
        d.panel = take((t.S + 1) * t.K);
        out.panel_doubles += (uint64_t)(t.S + 1) * t.K;
    }
    W.l_upool = off;
    // update matrices: every front keeps its own for the whole solve (they are small: all of them together about a third of
    // the panels), so that the assembly stream can fill them all at once
    for (uint32_t k = 0; k < fl.size(); ++k) {
        const Front& t = fr[fl[k]];
        const uint32_t R = t.S - t.K;
        if (R == 0) continue;
        const uint32_t len = (R + 1) * (R + 2) / 2;
        descs[k].upd = take(len);
        out.update_doubles += len;
    }
    if (off - W.l_panels >= 65536) return fail(why, "a workgroup's fronts do not fit 16-bit offsets");
    W.ws_doubles = off;
    struct FlatEntry {
        uint32_t hdr;
        UVec ops;
    };
    std::vector<FlatEntry> flat;
    // ---- per front: rows, children + maps, exports, the source stream ----------------------------------------------------------------
    for (uint32_t k = 0; k < fl.size(); ++k) {
        const Front& t = fr[fl[k]];
        FrontDesc& d = descs[k];
        d.rows = (uint32_t)rows.size();
        UVec frow;  // positions of the front's rows
        for (uint32_t j = t.c0; j < t.c0 + t.K; ++j) frow.push_back(j);
        frow.insert(frow.end(), t.below.begin(), t.below.end());
        for (uint32_t p : frow) {
            if (local_of[p] == NONE) return fail(why, "internal: a front row without a local variable");
            rows.push_back((uint16_t)local_of[p]);
        }
        auto row_in_front = [&](uint32_t p) -> uint32_t {  // position -> row of this front
            if (p >= t.c0 && p < t.c0 + t.K) return p - t.c0;
            const auto it = std::lower_bound(t.below.begin(), t.below.end(), p);
            return (it != t.below.end() && *it == p) ? t.K + (uint32_t)(it - t.below.begin()) : NONE;
        };
        // children in other workgroups: their update matrices arrive as chunks and are added through row maps; this workgroup's
        // own children are sources of the assembly stream below
        d.child0 = (uint32_t)children.size();
        for (uint32_t c : t.kids) {
            const Front& ch = fr[c];
            if (ch.wg == g) continue;
            FrontChild fc;
            std::memset(&fc, 0, sizeof(fc));
            fc.flags = FRONT_CHILD_REMOTE;
            fc.rows = (uint16_t)(ch.S - ch.K + 1);
            fc.map = (uint32_t)maps.size();
            for (uint32_t i : ch.below) {
                const uint32_t r = row_in_front(i);
                if (r == NONE) return fail(why, "internal: extend-add map");
                maps.push_back((uint8_t)r);
            }
            maps.push_back((uint8_t)t.S);  // the right-hand side's row
            fc.upd = up_chunk[c];
            ++W.n_remote_children;
            children.push_back(fc);
            ++d.n_child;
        }
        if (t.parent != NONE && fr[t.parent].wg != g) {
            d.flags |= FRONT_REMOTE_PARENT;
            d.up_chunk = up_chunk[fl[k]];
        }
        bool exp_any = false;
        for (uint32_t j = t.c0; j < t.c0 + t.K; ++j) exp_any = exp_any || export_chunk[j] != NONE;
        if (exp_any) {
            d.flags |= FRONT_EXPORTS;
            d.exp0 = (uint32_t)exports.size();
            for (uint32_t j = t.c0; j < t.c0 + t.K; ++j) exports.push_back(export_chunk[j]);
        }
        // ---- what the front's elements receive: operand pairs of this workgroup's constraints (-> the workgroup's assembly
        //      stream) and elements of its local children's update matrices (-> the front's source stream) -------------------------
        struct Entry {
            uint32_t dest;  // doubles from l_panels
            uint32_t flags;
            UVec ops, srcs;
        };
        std::vector<Entry> entries;
        {
            const uint32_t S1 = t.S + 1;
            std::vector<int32_t> at((size_t)S1 * S1, -1);
            auto entry = [&](uint32_t i, uint32_t j, uint32_t flags) -> Entry& {  // i >= j; i == S: right-hand side
                int32_t& e = at[(size_t)i * S1 + j];
                if (e < 0) {
                    e = (int32_t)entries.size();
                    uint32_t dest;
                    if (j < t.K) {
                        dest = d.panel - W.l_panels + j * S1 + i;  // panel, column-major
                    } else {
                        const uint32_t a = i - t.K, b = j - t.K;
                        dest = d.upd - W.l_panels + a * (a + 1) / 2 + b;
                    }
                    entries.push_back(Entry{dest, flags, {}, {}});
                }
                return entries[(size_t)e];
            };
            // the diagonal of every pivot exists even without a constraint (lambda)
            for (uint32_t j = 0; j < t.K; ++j) entry(j, j, FASM_DIAG);
            for (uint32_t c : t.cons) {
                const KindInfo& K = kKinds[cs[c].kind];
                for (int r = 0; r < K.n_rows; ++r) {
                    const uint32_t grow = cinfo[c].row0 + (uint32_t)r;
                    const uint32_t lrow = lrow0[c] + (uint32_t)r;
                    for (uint32_t qa = row_ptr[grow]; qa < row_ptr[grow + 1]; ++qa) {
                        const uint32_t ra = row_in_front(pos[row_col[qa]]);
                        const uint32_t sa = row_slot[qa] - cinfo[c].jbase + ljbase[c];
                        if (ra == NONE) return fail(why, "internal: assembly row");
                        entry(t.S, ra, FASM_RHS).ops.push_back(sa | (lrow << 16));
                        for (uint32_t qb = row_ptr[grow]; qb < row_ptr[grow + 1]; ++qb) {
                            const uint32_t rb = row_in_front(pos[row_col[qb]]);
                            if (rb == NONE || rb > ra) continue;
                            if (rb == ra && qb != qa) continue;  // (one slot per column and row: cannot happen)
                            const uint32_t sb = row_slot[qb] - cinfo[c].jbase + ljbase[c];
                            entry(ra, rb, ra == rb && ra < t.K ? FASM_DIAG : 0u).ops.push_back(sa | (sb << 16));
                        }
                    }
                }
            }
            // extend-add: every element of a local child's update matrix is a source of the element its rows map to
            for (uint32_t c : t.kids) {
                const Front& ch = fr[c];
                if (ch.wg != g) continue;
                const uint32_t Rc = ch.S - ch.K;
                const uint32_t base = descs[ch.local].upd - W.l_panels;
                UVec prow(Rc + 1);
                for (uint32_t a = 0; a < Rc; ++a) {
                    prow[a] = row_in_front(ch.below[a]);
                    if (prow[a] == NONE) return fail(why, "internal: extend-add map");
                }
                prow[Rc] = t.S;
                for (uint32_t a = 0; a <= Rc; ++a)
                    for (uint32_t b = 0; b <= a; ++b) {
                        if (a == Rc && b == Rc) continue;  // (the right-hand side's row against itself is nobody's)
                        if (prow[a] < prow[b]) return fail(why, "internal: extend-add order");
                        entry(prow[a], prow[b], 0u).srcs.push_back(base + a * (a + 1) / 2 + b);
                    }
            }
        }
        // the front's source stream: the entries that have sources, widest first
        {
            std::vector<const Entry*> se;
            for (const Entry& en : entries)
                if (!en.srcs.empty()) se.push_back(&en);
            std::stable_sort(se.begin(), se.end(), [](const Entry* a, const Entry* b) { return a->srcs.size() > b->srcs.size(); });
            if (se.size() > 65535) return fail(why, "a front's source stream is too long");
            d.src_n = (uint16_t)se.size();
            d.src_off = (uint32_t)stream_words.size();
            const uint32_t trips = ((uint32_t)se.size() + 63) / 64;
            uint32_t tv[4] = {0, 0, 0, 0};
            for (uint32_t e = 0; e < se.size(); ++e) tv[std::min(e / 64, 3u)] = std::max<uint32_t>(tv[std::min(e / 64, 3u)], (uint32_t)(se[e]->srcs.size() + 1) / 2);
            for (int q = 0; q < 4; ++q) {
                if (tv[q] > 255) return fail(why, "an element with more than 510 sources");
                d.src_v[q] = (uint8_t)tv[q];
            }
            for (uint32_t tr = 0; tr < trips; ++tr) {
                const uint32_t v = tv[std::min(tr, 3u)];
                const size_t base = stream_words.size();
                stream_words.resize(base + 64 * (1 + (size_t)v), 0);
                for (uint32_t l = 0; l < 64; ++l) {
                    const uint32_t e = tr * 64 + l;
                    if (e >= se.size()) {
                        stream_words[base + l] = FASM_NOP;
                        continue;  // (source words stay 0: the zero at l_panels)
                    }
                    stream_words[base + l] = se[e]->dest;
                    for (uint32_t q = 0; q < v; ++q) {
                        const UVec& sr = se[e]->srcs;
                        const uint32_t s0 = 2 * q < sr.size() ? sr[2 * q] : 0u, s1 = 2 * q + 1 < sr.size() ? sr[2 * q + 1] : 0u;
                        stream_words[base + 64 * (1 + q) + l] = s0 | (s1 << 16);
                    }
                }
            }
        }
        // ... and its share of the workgroup's assembly stream
        for (Entry& en : entries)
            if (!en.ops.empty() || (en.flags & FASM_DIAG)) flat.push_back(FlatEntry{en.dest | en.flags, std::move(en.ops)});
    }
    // ---- the workgroup's assembly stream: entries by operand count (most first), trips of 64 -----------------------------------------
    {
        std::stable_sort(flat.begin(), flat.end(), [](const FlatEntry& a, const FlatEntry& b) { return a.ops.size() > b.ops.size(); });
        W.asm_word0 = (uint32_t)stream_words.size();
        W.asm_trips = ((uint32_t)flat.size() + 63) / 64;
        // (first the trips' word offsets from the streams' start: a wavefront takes every n-th trip)
        stream_words.resize(stream_words.size() + ((W.asm_trips + 3) & ~3u), 0);
        for (uint32_t tr = 0; tr < W.asm_trips; ++tr) {
            const uint32_t w = (uint32_t)flat[(size_t)tr * 64].ops.size();
            if (w > 255) return fail(why, "an element with more than 255 operand pairs");
            const size_t base = stream_words.size();
            stream_words[W.asm_word0 + tr] = (uint32_t)base;
            stream_words.resize(base + 64 * (1 + (size_t)w), 0);
            for (uint32_t l = 0; l < 64; ++l) {
                const size_t e = (size_t)tr * 64 + l;
                if (e >= flat.size()) {
                    stream_words[base + l] = FASM_NOP | (w << 24);
                    for (uint32_t q = 0; q < w; ++q) stream_words[base + 64 * (1 + q) + l] = lslots | (lslots << 16);
                    continue;
                }
                const FlatEntry& en = flat[e];
                stream_words[base + l] = en.hdr | (w << 24);
                const uint32_t padw = (en.hdr & FASM_RHS) ? (lslots | (lrows << 16)) : (lslots | (lslots << 16));
                for (uint32_t q = 0; q < w; ++q) stream_words[base + 64 * (1 + q) + l] = q < en.ops.size() ? en.ops[q] : padw;
            }
        }
    }
    // ---- serialise ----------------------------------------------------------------------------------------------------------------
    W.n_fronts = (uint32_t)fl.size();
    W.n_levels = nlev;
    W.o_var_glob = B.put(var_glob);
    W.o_cons = B.put(dcons);
    // the staged tables, one block
    {
        std::vector<unsigned char> tab;
        Blob T{tab};
        T.put(descs);
        W.t_level_ptr = T.put(level_ptr);
        W.t_children = T.put(children);
        W.t_rows = T.put(rows);
        W.t_exports = T.put(exports);
        W.t_maps = T.put(maps);
        W.t_stream = T.put(stream_words);
        W.t_sched = T.put(sched);
        tab.resize((tab.size() + 15) & ~size_t(15), 0);
        // the constraint table rides along where the LDS has room to spare (the sweeps then read their 80-byte records from
        // LDS instead of L2: two sweeps per iteration)
        W.t_cons = 0xFFFFFFFFu;
        const size_t extras = 2 * 4 * 16 * 8 + 2080 * 2 + 64 + 8 * fl.size();
        if (!dcons.empty() && (tab.size() + dcons.size() * sizeof(DevCon) + (size_t)W.ws_doubles * 8 + extras) * 5 <= opt.lds_bytes * 4) {
            W.t_cons = T.put(dcons);
            tab.resize((tab.size() + 15) & ~size_t(15), 0);
        }
        W.o_tables = B.put(tab);
        W.tab_bytes = (uint32_t)tab.size();
    }
    W.o_ghosts = B.put(ghosts);
    W.o_slotmap = B.put(slotmap);
    out.n_fronts += W.n_fronts;
    out.n_levels = std::max(out.n_levels, nlev);
    out.ws_doubles_max = std::max(out.ws_doubles_max, W.ws_doubles);
    out.tab_bytes_max = std::max(out.tab_bytes_max, W.tab_bytes);
    max_fronts_wg = std::max<size_t>(max_fronts_wg, fl.size());
    if (debug)
        std::fprintf(stderr, "front plan: wg %u: %u fronts in %u levels, %u own + %u ghost variables, %u constraints, workspace %u doubles, tables %u B, streams %zu words\n",
                     g, W.n_fronts, nlev, W.n_own, W.n_ghost, W.n_cons, W.ws_doubles, W.tab_bytes, stream_words.size());
    return true;
}

bool FrontPlanner::emit() {
    // ---- emit -----------------------------------------------------------------------------------------------------------------
    wgs.assign(G, FrontWg{});
    out.blob.assign(((size_t)G * sizeof(FrontWg) + 15) & ~size_t(15), 0);
    waves = std::max(1u, opt.threads / 64);
    model_top = model_sub = 0.0;
    max_fronts_wg = 0;
    for (uint32_t g = 0; g < G; ++g)
        if (!emit_workgroup(g)) return false;
    out.bad_chunk0 = bad_chunk0;
    out.verdict_chunk = verdict_chunk;
    std::memcpy(out.blob.data(), wgs.data(), (size_t)G * sizeof(FrontWg));
    out.threads = opt.threads;
    const double model = model_top + model_sub;
    out.model_cycles = model;
    // dynamic LDS: [tables][workspace][reduction scratch: 2 x 4 x 16 doubles][tri table 2080 x u16][small ints]
    // ... [per front: children signed in, substituted-back stamp: 2 x u32]
    out.lds_bytes = ((size_t)out.tab_bytes_max + 15) / 16 * 16 + (size_t)out.ws_doubles_max * 8 + 2 * 4 * 16 * 8 + 2080 * 2 + 64 + 8 * max_fronts_wg;
    if (out.lds_bytes > opt.lds_bytes) return fail(why, "a workgroup's share does not fit the LDS");
    if (debug)
        std::fprintf(stderr, "front plan: %u variables, %u fronts (largest %u x %u), %u levels, %u workgroups, LDS %zu B, model %.0f cycles, chunks %u\n",
                     n, out.n_fronts, out.max_rows, out.max_pivots, out.n_levels, G, out.lds_bytes, model, out.n_chunks);
    return true;
}

static bool plan_once(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, const FrontOptions& opt, FrontPlan& out, const char** why) {
    out = FrontPlan();
    if (n_cs == 0 || n_vars == 0 || n_cs > 0x3FFFFFFFu || n_vars > 0x3FFFFFFFu) return fail(why, "empty or oversized system");
    return FrontPlanner(cs, n_cs, n_vars, opt, out, why).run();
}

}  // namespace ezpz

namespace ezpz {
bool front_plan_build(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, const FrontOptions& opt, FrontPlan& out,
                      const char** why) {
    const char* w = nullptr;
    // nested dissection first (balanced trees); where its separators make a front of more than a wavefront's rows, minimum degree
    for (uint32_t ordering = opt.ordering; ordering <= 1; ++ordering) {
        FrontOptions o = opt;
        o.ordering = ordering;
        for (;;) {
            if (plan_once(cs, n_cs, n_vars, o, out, &w)) return true;
            // the planner's own choice of workgroups: a share that does not fit one CU's LDS asks for more of them
            if (opt.wgs != 0 || !w || !std::strstr(w, "LDS") || out.n_wgs >= std::min<uint32_t>(o.max_wgs, kFrontMaxWgs) || o.vars_per_wg <= 24) break;
            o.vars_per_wg = o.vars_per_wg * 3 / 4;
        }
        if (!w || !std::strstr(w, "63 rows")) break;
    }
    if (why) *why = w;
    return false;
}
}  // namespace ezpz

extern "C" long ezpz_debug_front_plan(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint32_t wgs, uint32_t max_wgs,
                                      uint64_t lds_bytes, unsigned char* buf, size_t cap, uint64_t* info) {
    if (!cs && n_cs) return EZPZ_ERR_INVALID_ARGUMENT;
    ezpz::FrontOptions opt;
    opt.wgs = wgs;
    if (max_wgs) opt.max_wgs = max_wgs;
    if (lds_bytes) opt.lds_bytes = (size_t)lds_bytes;
    ezpz::FrontPlan plan;
    const char* why = nullptr;
    if (!ezpz::front_plan_build(cs, n_cs, n_vars, opt, plan, &why)) {
        if (ezpz::debug_topic("front")) std::fprintf(stderr, "front plan: not applicable: %s\n", why ? why : "?");
        return 0;
    }
    if (info) {
        const uint64_t v[16] = {plan.n_wgs, plan.n_chunks, plan.bad_chunk0, plan.verdict_chunk, plan.lds_bytes, plan.n_fronts,
                                plan.n_levels, plan.max_rows, plan.max_pivots, plan.threads, (uint64_t)plan.model_cycles,
                                plan.panel_doubles, plan.update_doubles, plan.ordering, plan.n_components, 0};
        std::memcpy(info, v, sizeof(v));
    }
    if (buf && cap) std::memcpy(buf, plan.blob.data(), std::min(cap, plan.blob.size()));
    return (long)plan.blob.size();
}
