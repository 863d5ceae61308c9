// Host symbolic phase (see program.hpp).  Plain C++17, no device code.
#include "program.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory_resource>
#include <numeric>
#include <queue>

#include "kinds.hpp"

namespace ezpz {

int residual_dim(uint16_t kind) { return kind < EZPZ_NUM_KINDS ? kKinds[kind].n_rows : 1; }
int kind_num_ids(uint16_t kind) { return kind < EZPZ_NUM_KINDS ? kKinds[kind].n_ids : 0; }

uint64_t topology_hash(const EzpzConstraint* cs, size_t n_cs, size_t n_vars) {
    // 64-bit multiply-xorshift over the request 8 bytes at a time (params and weights live in the program too, so
    // they are part of the key).  sizeof(EzpzConstraint) == 56 is a multiple of 8.
    // Four independent chains (one multiply latency each per 32 bytes instead of per 8): the hash of the 2000-constraint
    // request was a third of a warm solve() call.
    uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ (uint64_t)n_vars, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull,
                     0x27D4EB2F165667C5ull};
    const size_t words = n_cs * sizeof(EzpzConstraint) / 8;
    const unsigned char* p = reinterpret_cast<const unsigned char*>(cs);
    auto mix = [](uint64_t a, uint64_t w) {
        a = (a ^ w) * 0xFF51AFD7ED558CCDull;
        return a ^ (a >> 29);
    };
    size_t i = 0;
    for (; i + 4 <= words; i += 4) {
        uint64_t w[4];
        std::memcpy(w, p + 8 * i, 32);
        for (int k = 0; k < 4; ++k) h[k] = mix(h[k], w[k]);
    }
    for (; i < words; ++i) {
        uint64_t w;
        std::memcpy(&w, p + 8 * i, 8);
        h[0] = mix(h[0], w);
    }
    return mix(mix(mix(h[0], h[1]), h[2]), h[3]);
}

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// Minimum-degree ordering of one connected component of the JtJ graph (exact degrees, explicit
// elimination graph).  Components in constraint sketches are small; beyond `kMinDegLimit` vertices we
// keep the natural order instead.
constexpr size_t kMinDegLimit = 16384;

// Stable counting sort: perm[i] = the element that comes i-th when ordered by key (< n_keys), ties in `input` order.
// `input` is the current order (a permutation or identity given as nullptr).
std::vector<uint32_t> counting_order(const std::vector<uint32_t>& key, uint32_t n_keys, const std::vector<uint32_t>* input) {
    const size_t n = key.size();
    std::vector<uint32_t> start((size_t)n_keys + 1, 0), perm(n);
    for (size_t i = 0; i < n; ++i) ++start[key[i] + 1];
    for (uint32_t k = 0; k < n_keys; ++k) start[k + 1] += start[k];
    for (size_t i = 0; i < n; ++i) {
        const uint32_t e = input ? (*input)[i] : (uint32_t)i;
        perm[start[key[e]]++] = e;
    }
    return perm;
}

// The symbolic phase builds hundreds of thousands of short index lists (one per variable, several times over) and
// throws them all away at the end: they live in one monotonic arena per build_program call instead of the heap.
using IVec = std::pmr::vector<uint32_t>;
using IVecs = std::pmr::vector<IVec>;

void order_component(const std::vector<uint32_t>& verts, const IVecs& adj, std::vector<uint32_t>& local_id,
                     std::vector<uint32_t>& out_order, std::pmr::memory_resource* pool) {
    const size_t k = verts.size();
    if (k <= 2 || k > kMinDegLimit) {
        for (uint32_t v : verts) out_order.push_back(v);
        return;
    }
    for (size_t i = 0; i < k; ++i) local_id[verts[i]] = (uint32_t)i;
    IVecs g(k, pool);
    for (size_t i = 0; i < k; ++i) {
        for (uint32_t w : adj[verts[i]]) g[i].push_back(local_id[w]);
        std::sort(g[i].begin(), g[i].end());
    }
    std::pmr::vector<char> gone(k, 0, pool);
    IVec merged(pool), nb(pool);
    // smallest (degree, index) first; entries whose degree is out of date are skipped when they surface
    typedef std::pair<uint32_t, uint32_t> DegIdx;
    std::priority_queue<DegIdx, std::vector<DegIdx>, std::greater<DegIdx>> heap;
    const bool scan = k <= 64;  // the typical sketch component: a linear scan beats the heap
    if (!scan)
        for (size_t i = 0; i < k; ++i) heap.push(DegIdx((uint32_t)g[i].size(), (uint32_t)i));
    for (size_t step = 0; step < k; ++step) {
        size_t best = k;
        if (scan) {
            size_t best_deg = (size_t)-1;
            for (size_t i = 0; i < k; ++i) {
                if (!gone[i] && g[i].size() < best_deg) {
                    best_deg = g[i].size();
                    best = i;
                }
            }
        }
        while (best == k) {
            const DegIdx top = heap.top();
            heap.pop();
            if (!gone[top.second] && g[top.second].size() == top.first) best = top.second;
        }
        gone[best] = 1;
        out_order.push_back(verts[best]);
        nb.assign(g[best].begin(), g[best].end());
        for (uint32_t u : nb) {
            // g[u] = (g[u] \ {best}) U (nb \ {u})
            merged.clear();
            std::set_union(g[u].begin(), g[u].end(), nb.begin(), nb.end(), std::back_inserter(merged));
            IVec& gu = g[u];
            gu.clear();
            for (uint32_t w : merged)
                if (w != u && w != best) gu.push_back(w);
            if (!scan) heap.push(DegIdx((uint32_t)gu.size(), u));
        }
        g[best].clear();
    }
}

// Strictly-lower entries of the Cholesky factor of one component under the elimination order `ord`, and the height of
// its elimination tree = the number of levels the level-scheduled factorisation runs one after the other
// (symbolic elimination with an elimination tree; O(nnz(L))).
struct OrderCost {
    uint64_t fill;
    uint32_t height;
};
OrderCost component_cost(const std::vector<uint32_t>& ord, const IVecs& adj, std::vector<uint32_t>& local_id,
                         std::pmr::memory_resource* pool) {
    const size_t k = ord.size();
    for (size_t i = 0; i < k; ++i) local_id[ord[i]] = (uint32_t)i;
    IVec parent(k, NONE, pool), flag(k, NONE, pool), depth(k, 0u, pool);
    uint64_t fill = 0;
    for (uint32_t p = 0; p < k; ++p) {
        flag[p] = p;
        for (uint32_t w : adj[ord[p]]) {
            uint32_t i = local_id[w];
            if (i >= p) continue;
            // etree update (Liu) and row-pattern count in one walk
            for (uint32_t j = i; flag[j] != p; j = parent[j]) {
                flag[j] = p;
                ++fill;
                if (parent[j] == NONE) {
                    parent[j] = p;
                    break;
                }
            }
        }
    }
    uint32_t height = 0;
    for (uint32_t j = 0; j < k; ++j) {  // children come before parents
        if (parent[j] != NONE) depth[parent[j]] = std::max(depth[parent[j]], depth[j] + 1);
        height = std::max(height, depth[j] + 1);
    }
    return OrderCost{fill, height};
}

// Nested dissection (George's automatic scheme): breadth-first level structure from a pseudo-peripheral vertex, the
// middle level is the separator, both sides are ordered first (recursively), the separator last.  A polyline or any
// other band-like sketch gets an elimination tree of height O(band * log n) instead of n: what the level-scheduled
// factorisation needs, since it pays one synchronisation per level.
void nested_dissection(const std::vector<uint32_t>& verts, const IVecs& adj, std::vector<uint32_t>& local_id,
                       std::vector<uint32_t>& out_order, std::pmr::memory_resource* pool) {
    const uint32_t k = (uint32_t)verts.size();
    for (uint32_t i = 0; i < k; ++i) local_id[verts[i]] = i;
    IVec region(k, 0u, pool);  // which open region a vertex belongs to (0 = the whole component)
    IVec level(k, 0u, pool), queue(pool);
    std::pmr::vector<char> seen(k, 0, pool);
    struct Task {
        uint32_t region;
        bool emit;     // second visit: `members` is the region's separator and goes out after both sides
        IVec members;  // first visit: the region's vertices
    };
    std::pmr::vector<Task> stack(pool);
    {
        Task all{0, false, IVec(pool)};
        all.members.resize(k);
        std::iota(all.members.begin(), all.members.end(), 0u);
        stack.push_back(std::move(all));
    }
    uint32_t next_region = 1;
    auto bfs = [&](uint32_t start, uint32_t reg) {  // level structure of start's connected piece of `reg`, in `queue`
        queue.clear();
        queue.push_back(start);
        level[start] = 0;
        seen[start] = 1;
        for (size_t h = 0; h < queue.size(); ++h) {
            const uint32_t v = queue[h];
            for (uint32_t wv : adj[verts[v]]) {
                const uint32_t w = local_id[wv];
                if (region[w] == reg && !seen[w]) {
                    seen[w] = 1;
                    level[w] = level[v] + 1;
                    queue.push_back(w);
                }
            }
        }
        for (uint32_t v : queue) seen[v] = 0;
    };
    while (!stack.empty()) {
        Task t = std::move(stack.back());
        stack.pop_back();
        if (t.emit) {
            for (uint32_t v : t.members) out_order.push_back(verts[v]);
            continue;
        }
        for (uint32_t s0 : t.members) {  // every connected piece of this region
            if (region[s0] != t.region) continue;  // already handed to a leaf / separator / sub-region
            bfs(s0, t.region);
            bfs(queue.back(), t.region);  // pseudo-peripheral: restart from the farthest vertex
            const uint32_t depth = level[queue.back()];
            // Small (<= 4 vertices) or compact (level structure of depth < 2) piece: a leaf, in request order.  Leaves are
            // chains -- a leaf of 16 vertices of a band-like sketch is 10-13 elimination levels at the bottom of the
            // tree -- so they are kept tiny (measured with leaves of <= 16 / depth < 4 before: 300 variables 32 -> 27
            // levels, one solve 295 -> 257 us, batches +12-15 %; 2000 variables 2.52 -> 2.22 ms; no more fill).
            if (queue.size() <= 4 || depth < 2) {
                IVec piece(queue.begin(), queue.end(), pool);
                std::sort(piece.begin(), piece.end());
                for (uint32_t v : piece) {
                    out_order.push_back(verts[v]);
                    region[v] = NONE;
                }
                continue;
            }
            const uint32_t mid = depth / 2;
            Task sep{0, true, IVec(pool)}, left{next_region, false, IVec(pool)}, right{next_region + 1, false, IVec(pool)};
            next_region += 2;
            for (uint32_t v : queue) {
                if (level[v] == mid) {
                    sep.members.push_back(v);
                    region[v] = NONE;
                } else if (level[v] < mid) {
                    left.members.push_back(v);
                    region[v] = left.region;
                } else {
                    right.members.push_back(v);
                    region[v] = right.region;
                }
            }
            std::sort(sep.members.begin(), sep.members.end());
            stack.push_back(std::move(sep));  // LIFO: comes out after both sides
            stack.push_back(std::move(right));
            stack.push_back(std::move(left));
        }
    }
}

}  // namespace

bool build_program(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, Program& P, BuildError& err,
                   uint32_t want_parts, bool dense, bool serial) {
    P = Program();
    P.c.dense = dense ? 1u : 0u;
    if (dense) want_parts = 1;
    if (n_cs > 0x7FFFFFF0u || n_vars > 0x7FFFFFF0u) {
        err.code = EZPZ_ERR_TOO_LARGE;
        err.message = "more than 2^31 constraints or variables";
        return false;
    }
    // ---- validate (solver.rs:142-189) + row numbering (solver.rs:226-253) --------------------------
    uint64_t m64 = 0;
    for (size_t i = 0; i < n_cs; ++i) {
        const EzpzConstraint& c = cs[i];
        if (c.kind >= EZPZ_NUM_KINDS) {
            err.code = EZPZ_ERR_INVALID_ARGUMENT;
            err.constraint = (int32_t)i;
            err.message = "unknown constraint kind";
            return false;
        }
        const KindInfo& K = kKinds[c.kind];
        for (int r = 0; r < K.n_rows; ++r) {
            for (int e = 0; e < K.n_nz[r]; ++e) {
                uint32_t v = c.ids[K.nz[r][e]];
                if (v >= n_vars) {
                    err.code = EZPZ_ERR_MISSING_GUESS;
                    err.constraint = (int32_t)i;
                    err.variable = v;
                    return false;
                }
            }
        }
        m64 += K.n_rows;
    }
    if (m64 > 0x7FFFFFF0u) {
        err.code = EZPZ_ERR_TOO_LARGE;
        return false;
    }
    const uint32_t n = (uint32_t)n_vars, C = (uint32_t)n_cs, m = (uint32_t)m64;
    P.c.n_cons = C;
    P.c.n_vars = n;
    P.c.n_rows = m;

    // ---- constraint table + Jacobian slots -----------------------------------------------------------
    // Slots are constraint-contiguous; within a row equal column ids share one slot (the reference
    // deduplicates (row,col) pairs and accumulates partials into the shared cell, solver.rs:255-260,:418).
    std::vector<DevCon> cons(C);
    // column view of J, built in row order so b = -Jt r sums rows ascending
    std::pmr::monotonic_buffer_resource arena(1u << 20);
    std::pmr::memory_resource* pool = &arena;
    IVecs colj(n, pool);  // (jslot,row) flattened
    // row view: cols and slots per row (unique)
    std::vector<uint32_t> row_ptr(m + 1, 0), row_col, row_slot;
    row_col.reserve((size_t)m * 4);
    row_slot.reserve((size_t)m * 4);
    uint32_t row_num = 0, jslot = 0;
    for (uint32_t i = 0; i < C; ++i) {
        const EzpzConstraint& c = cs[i];
        const KindInfo& K = kKinds[c.kind];
        DevCon& d = cons[i];
        std::memset(&d, 0, sizeof(d));
        std::memcpy(d.ids, c.ids, sizeof(d.ids));
        for (int k = K.n_ids; k < 8; ++k) d.ids[k] = 0;  // never dereference garbage
        d.param = c.param;
        d.weight = c.weight;
        d.row0 = row_num;
        d.jbase = jslot;
        d.pos = i;
        d.kind = (uint8_t)c.kind;
        d.tag = c.tag;
        d.nrows = K.n_rows;
        uint32_t local = 0;
        int e_global = 0;
        for (int r = 0; r < K.n_rows; ++r) {
            uint32_t first_local_of_row = local;
            for (int e = 0; e < K.n_emit[r]; ++e, ++e_global) {
                uint32_t col = c.ids[K.emit[r][e]];
                // search earlier entries of this row for the same column
                int dup = -1;
                for (int e2 = 0; e2 < e; ++e2)
                    if (c.ids[K.emit[r][e2]] == col) {
                        dup = e2;
                        break;
                    }
                if (dup >= 0) {
                    d.jloc[e_global] = (uint8_t)((d.jloc[e_global - e + dup] & 0x7F) | 0x80);
                } else {
                    d.jloc[e_global] = (uint8_t)local;
                    row_col.push_back(col);
                    row_slot.push_back(jslot + local);
                    colj[col].push_back(jslot + local);
                    colj[col].push_back(row_num);
                    ++local;
                }
            }
            (void)first_local_of_row;
            ++row_num;
            row_ptr[row_num] = (uint32_t)row_col.size();
        }
        d.nslots = (uint8_t)local;
        jslot += local;
    }
    P.c.zj = jslot;

    // ---- JtJ graph, connected components ----------------------------------------------------------------
    IVecs adj(n, pool);
    for (uint32_t r = 0; r < m; ++r) {
        for (uint32_t a = row_ptr[r]; a < row_ptr[r + 1]; ++a)
            for (uint32_t b = row_ptr[r]; b < row_ptr[r + 1]; ++b)
                if (a != b) adj[row_col[a]].push_back(row_col[b]);
    }
    uint64_t za = n;
    for (uint32_t v = 0; v < n; ++v) {
        std::sort(adj[v].begin(), adj[v].end());
        adj[v].erase(std::unique(adj[v].begin(), adj[v].end()), adj[v].end());
        for (uint32_t w : adj[v])
            if (w < v) ++za;
    }
    P.c.za = (uint32_t)za;
    // The dense layout pays when JtJ is mostly full to begin with (square, parallelogram, circle_tangent: 100 %); a
    // sketch of mostly Fixed / axis-aligned constraints (tiny: 1 of 6 entries, arc_radius: 11 of 28) keeps its lists.
    if (dense && (za - n) * 5 < (uint64_t)n * (n - 1) / 2 * 3) {
        dense = false;
        P.c.dense = 0;
    }
    // Elimination order, per connected component: minimum degree, unless the request order already gives a
    // factor that is no denser -- then the variables are eliminated in id order, which makes the factorisation
    // operation-for-operation the textbook left-looking Cholesky of the matrix as the caller numbered it.
    // A "component" is what one wavefront / workgroup can own end to end.  The rows of a two-row constraint
    // (PointsCoincident: an x row and a y row) may touch disjoint sets of variables, i.e. different blocks of JtJ, but
    // the constraint is evaluated as a whole by whoever owns it: both blocks must land in the same component, or the
    // owner of the second row's variables would read residuals and Jacobian entries another wavefront is still
    // writing.  `link` joins the first variable of every row of a constraint for the traversal only (the
    // factorisation keeps seeing the true, finer block structure: a merged component is just a disconnected graph).
    IVecs link(n, pool);
    for (uint32_t i = 0; i < C; ++i) {
        const KindInfo& K = kKinds[cs[i].kind];
        for (int r = 1; r < K.n_rows; ++r) {
            const uint32_t a0 = cs[i].ids[K.nz[0][0]], b0 = cs[i].ids[K.nz[r][0]];
            if (a0 != b0) {
                link[a0].push_back(b0);
                link[b0].push_back(a0);
            }
        }
    }
    std::vector<uint32_t> comp(n, NONE);
    std::vector<uint32_t> order;  // position -> var
    order.reserve(n);
    {
        std::vector<uint32_t> stack, verts, local_id(n, 0), cand, nd;
        uint32_t ncomp = 0;
        for (uint32_t s = 0; s < n; ++s) {
            if (comp[s] != NONE) continue;
            verts.clear();
            stack.clear();
            stack.push_back(s);
            comp[s] = ncomp;
            while (!stack.empty()) {
                uint32_t v = stack.back();
                stack.pop_back();
                verts.push_back(v);
                for (uint32_t w : adj[v])
                    if (comp[w] == NONE) {
                        comp[w] = ncomp;
                        stack.push_back(w);
                    }
                for (uint32_t w : link[v])
                    if (comp[w] == NONE) {
                        comp[w] = ncomp;
                        stack.push_back(w);
                    }
            }
            std::sort(verts.begin(), verts.end());
            // request order, minimum degree, nested dissection: the cheapest by entries of L plus what its levels cost
            // (a level is one or two synchronisations, worth about 64 entries); ties go to the earlier candidate, so
            // the request order -- the textbook left-looking Cholesky of the matrix as the caller numbered it -- wins
            // whenever nothing is gained by leaving it
            auto score = [&](const std::vector<uint32_t>& o) {
                const OrderCost c = component_cost(o, adj, local_id, pool);
                return c.fill + (serial ? 0ull : 64ull * c.height);  // (one lane per system: levels cost nothing)
            };
            uint64_t best = score(verts);
            const std::vector<uint32_t>* pick = &verts;
            cand.clear();
            order_component(verts, adj, local_id, cand, pool);
            if (cand != verts) {
                const uint64_t sc = score(cand);
                if (sc < best) {
                    best = sc;
                    pick = &cand;
                }
            }
            nd.clear();
            if (verts.size() > 24) {  // (from 25 vertices: a 50-variable sketch walks 12 levels + phases in request order, 5 dissected)
                nested_dissection(verts, adj, local_id, nd, pool);
                if (nd.size() == verts.size() && score(nd) < best) pick = &nd;
            }
            order.insert(order.end(), pick->begin(), pick->end());
            ++ncomp;
        }
        P.c.n_components = ncomp;
    }
    std::vector<uint32_t> pos(n);
    for (uint32_t k = 0; k < n; ++k) pos[order[k]] = k;

    // ---- symbolic Cholesky in elimination order: etree, row patterns, levels ------------------------------
    IVecs upper(n, pool);  // upper[k] = positions i<k adjacent to k
    for (uint32_t k = 0; k < n; ++k) {
        for (uint32_t w : adj[order[k]])
            if (pos[w] < k) upper[k].push_back(pos[w]);
        std::sort(upper[k].begin(), upper[k].end());
    }
    std::vector<uint32_t> parent(n, NONE), ancestor(n, NONE);
    for (uint32_t k = 0; k < n; ++k) {
        for (uint32_t i0 : upper[k]) {
            uint32_t i = i0;
            while (i != NONE && i < k) {
                uint32_t inext = ancestor[i];
                ancestor[i] = k;
                if (inext == NONE) parent[i] = k;
                i = inext;
            }
        }
    }
    IVecs rowpat(n, pool);   // columns j<k with L(k,j) != 0, ascending
    IVecs colrows(n, pool);  // rows k>j with L(k,j) != 0, ascending
    {
        std::vector<uint32_t> flag(n, NONE);
        uint64_t zlo = 0;
        for (uint32_t k = 0; k < n; ++k) {
            flag[k] = k;
            for (uint32_t i0 : upper[k]) {
                for (uint32_t i = i0; flag[i] != k; i = parent[i]) {
                    rowpat[k].push_back(i);
                    flag[i] = k;
                }
            }
            std::sort(rowpat[k].begin(), rowpat[k].end());
            zlo += rowpat[k].size();
            for (uint32_t j : rowpat[k]) colrows[j].push_back(k);
            if (zlo > 0x3FFFFFFFull) {
                err.code = EZPZ_ERR_TOO_LARGE;
                err.message = "Cholesky factor has more than 2^30 entries";
                return false;
            }
        }
        P.c.zlo = (uint32_t)zlo;
    }
    std::vector<uint32_t> level(n, 0);
    for (uint32_t j = 0; j < n; ++j)
        if (parent[j] != NONE) level[parent[j]] = std::max(level[parent[j]], level[j] + 1);
    if (dense) {  // every strictly-lower entry, every column its own level
        uint64_t zlo = 0;
        for (uint32_t k = 0; k < n; ++k) {
            rowpat[k].clear();
            colrows[k].clear();
            for (uint32_t j = 0; j < k; ++j) rowpat[k].push_back(j);
            for (uint32_t i = k + 1; i < n; ++i) colrows[k].push_back(i);
            level[k] = k;
            zlo += k;
        }
        P.c.zlo = (uint32_t)zlo;
    }

    // ---- partitions: balanced unions of components, one per wavefront (longest-processing-time first) -------
    const uint32_t ncomp = P.c.n_components;
    std::vector<uint32_t> part_of_comp(ncomp, 0);
    uint32_t n_parts = 1;
    if (want_parts > 1 && ncomp >= want_parts) {
        std::vector<uint64_t> w(ncomp, 0);
        for (uint32_t v = 0; v < n; ++v) w[comp[v]] += 1 + rowpat[pos[v]].size();
        for (uint32_t i = 0; i < C; ++i) w[comp[cs[i].ids[kKinds[cs[i].kind].nz[0][0]]]] += 4;
        std::vector<uint32_t> by_weight(ncomp);
        std::iota(by_weight.begin(), by_weight.end(), 0u);
        std::stable_sort(by_weight.begin(), by_weight.end(), [&](uint32_t a, uint32_t b) { return w[a] > w[b]; });
        std::vector<uint64_t> load(want_parts, 0);
        uint64_t total = 0;
        // least-loaded partition first (ties: lowest index), as a heap: grid teams ask for thousands of partitions
        using Slot = std::pair<uint64_t, uint32_t>;
        std::priority_queue<Slot, std::vector<Slot>, std::greater<Slot>> heap;
        for (uint32_t p = 0; p < want_parts; ++p) heap.push(Slot{0, p});
        for (uint32_t c : by_weight) {
            const uint32_t best = heap.top().second;
            heap.pop();
            part_of_comp[c] = best;
            load[best] += w[c];
            total += w[c];
            heap.push(Slot{load[best], best});
        }
        uint64_t worst = *std::max_element(load.begin(), load.end());
        if (worst * want_parts <= total + total / 3 + 64) {
            n_parts = want_parts;
            // Same loads, better locality: among components of equal weight it does not matter which ones a partition
            // gets, so hand every partition a contiguous run of them (components are numbered in the order of their
            // first variable).  The 500 equal blocks of the 2000 x 2000 system then sit in 8 contiguous 2 KB pieces of
            // the guess vector instead of being dealt round-robin: a wavefront's x0 loads / x stores touch a fifth of
            // the cache lines.
            std::vector<uint32_t> idx(ncomp);
            std::iota(idx.begin(), idx.end(), 0u);
            std::stable_sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return w[a] < w[b]; });
            std::vector<uint32_t> cnt(want_parts);
            for (size_t i0 = 0; i0 < idx.size();) {
                size_t i1 = i0;
                while (i1 < idx.size() && w[idx[i1]] == w[idx[i0]]) ++i1;
                std::fill(cnt.begin(), cnt.end(), 0u);
                for (size_t i = i0; i < i1; ++i) cnt[part_of_comp[idx[i]]]++;
                size_t i = i0;  // idx[i0..i1) ascends by component number (stable sort)
                for (uint32_t p = 0; p < want_parts; ++p)
                    for (uint32_t k = 0; k < cnt[p]; ++k) part_of_comp[idx[i++]] = p;
                i0 = i1;
            }
        } else {
            std::fill(part_of_comp.begin(), part_of_comp.end(), 0u);  // one component dominates: keep one partition
        }
    }
    P.c.n_parts = n_parts;
    auto part_of_pos = [&](uint32_t k) { return part_of_comp[comp[order[k]]]; };
    // columns grouped by (partition, level)
    // stable order by (partition, level), ties in elimination order: two counting passes (level, then partition)
    std::vector<uint32_t> colorder;
    {
        std::vector<uint32_t> pkey(n);
        uint32_t max_level = 0;
        for (uint32_t k = 0; k < n; ++k) {
            pkey[k] = part_of_pos(k);
            max_level = std::max(max_level, level[k]);
        }
        const std::vector<uint32_t> by_level = counting_order(level, max_level + 1, nullptr);
        colorder = counting_order(pkey, n_parts, &by_level);
    }
    P.parts.assign(n_parts, PartDesc{0, 0, 0, 0});
    P.lvl_cptr.clear();
    P.lvl_sptr.clear();
    std::vector<uint32_t> lvl_cols(n);
    std::vector<uint32_t> col_slot0(n, 0);  // first offdiag slot of column (position space)
    uint32_t nlev = 0;
    {
        uint32_t slot = 0, idx = 0;
        for (uint32_t p = 0; p < n_parts; ++p) {
            P.parts[p].lvl0 = (uint32_t)P.lvl_cptr.size();
            uint32_t lv = 0;
            while (idx < n && part_of_pos(colorder[idx]) == p) {
                P.lvl_cptr.push_back(idx);
                P.lvl_sptr.push_back(slot);
                while (idx < n && part_of_pos(colorder[idx]) == p && level[colorder[idx]] == lv) {
                    uint32_t j = colorder[idx];
                    lvl_cols[idx] = order[j];
                    col_slot0[j] = slot;
                    slot += (uint32_t)colrows[j].size();
                    ++idx;
                }
                ++lv;
            }
            P.parts[p].nlev = lv;
            P.lvl_cptr.push_back(idx);  // closing entry of this partition's slice
            P.lvl_sptr.push_back(slot);
            nlev = std::max(nlev, lv);
        }
    }
    P.c.n_levels = nlev;
    const uint32_t zlo = P.c.zlo;
    // slot lookup per row: rowslot[k][t] = slot of (k, rowpat[k][t])
    IVecs rowslot(n, pool);
    for (uint32_t k = 0; k < n; ++k) rowslot[k].resize(rowpat[k].size());
    P.l_col.assign(zlo, 0);
    {
        std::vector<uint32_t> fill(n, 0);  // how many entries of rowpat[k] have been assigned
        // colrows[j] ascending in k and rowpat[k] ascending in j: entry (k,j) index within rowpat[k] found by search
        for (uint32_t j = 0; j < n; ++j) {
            uint32_t s = col_slot0[j];
            for (uint32_t k : colrows[j]) {
                auto it = std::lower_bound(rowpat[k].begin(), rowpat[k].end(), j);
                rowslot[k][(size_t)(it - rowpat[k].begin())] = s;
                P.l_col[s] = order[j];
                ++s;
            }
        }
        (void)fill;
    }
    auto find_slot = [&](uint32_t k, uint32_t j) -> uint32_t {  // k > j positions
        auto it = std::lower_bound(rowpat[k].begin(), rowpat[k].end(), j);
        if (it == rowpat[k].end() || *it != j) return NONE;
        return rowslot[k][(size_t)(it - rowpat[k].begin())];
    };

    // ---- column view of J -------------------------------------------------------------------------------------
    P.colj_ptr.assign(n + 1, 0);
    for (uint32_t v = 0; v < n; ++v) P.colj_ptr[v + 1] = P.colj_ptr[v] + (uint32_t)(colj[v].size() / 2);
    P.colj_items.reserve((size_t)P.c.zj * 2);
    for (uint32_t v = 0; v < n; ++v) P.colj_items.insert(P.colj_items.end(), colj[v].begin(), colj[v].end());

    // ---- J-slot pairs for the strict lower part of JtJ, keyed by L slot ----------------------------------------------
    {
        std::vector<uint32_t> cnt(zlo + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) {
                P.apair_ptr.assign(zlo + 1, 0);
                for (uint32_t s = 0; s < zlo; ++s) P.apair_ptr[s + 1] = P.apair_ptr[s] + cnt[s];
                P.c.n_apairs = P.apair_ptr[zlo];
                P.apairs.assign((size_t)P.c.n_apairs * 2, 0);
                std::fill(cnt.begin(), cnt.end(), 0);
            }
            for (uint32_t r = 0; r < m; ++r) {
                for (uint32_t a = row_ptr[r]; a < row_ptr[r + 1]; ++a) {
                    for (uint32_t b = row_ptr[r]; b < row_ptr[r + 1]; ++b) {
                        uint32_t pa = pos[row_col[a]], pb = pos[row_col[b]];
                        if (pa <= pb) continue;  // (row pa, col pb), pa > pb
                        uint32_t s = find_slot(pa, pb);
                        if (s == NONE) {
                            err.code = EZPZ_ERR_INVALID_ARGUMENT;
                            err.message = "internal: JtJ entry outside L pattern";
                            return false;
                        }
                        if (pass == 1) {
                            size_t o = ((size_t)P.apair_ptr[s] + cnt[s]) * 2;
                            P.apairs[o] = row_slot[a];
                            P.apairs[o + 1] = row_slot[b];
                        }
                        ++cnt[s];
                    }
                }
            }
        }
    }

    // ---- Cholesky pair lists: L(i,j) -= sum_k L(i,k) L(j,k), k < j ---------------------------------------------------
    if (dense) {
        P.lpair_ptr.assign(zlo + 1, 0);
        P.lpairs.clear();
        P.c.n_lpairs = 0;
    } else {
        uint64_t total = 0;
        P.lpair_ptr.assign(zlo + 1, 0);
        // count
        for (uint32_t j = 0; j < n; ++j) {
            uint32_t s = col_slot0[j];
            for (uint32_t k : colrows[j]) {
                // |{c in rowpat[k], c<j} ^ rowpat[j]|
                const auto& A = rowpat[k];
                const auto& B = rowpat[j];
                size_t ia = 0, ib = 0, cntp = 0;
                while (ia < A.size() && ib < B.size() && A[ia] < j) {
                    if (A[ia] == B[ib]) {
                        ++cntp;
                        ++ia;
                        ++ib;
                    } else if (A[ia] < B[ib])
                        ++ia;
                    else
                        ++ib;
                }
                P.lpair_ptr[s + 1] = (uint32_t)cntp;
                total += cntp;
                ++s;
            }
        }
        if (total > 0x7FFFFFFFull) {
            err.code = EZPZ_ERR_TOO_LARGE;
            err.message = "sparse Cholesky needs more than 2^31 multiply-adds per factorisation";
            return false;
        }
        for (uint32_t s = 0; s < zlo; ++s) P.lpair_ptr[s + 1] += P.lpair_ptr[s];
        P.c.n_lpairs = total;
        P.lpairs.assign((size_t)total * 2, 0);
        for (uint32_t j = 0; j < n; ++j) {
            uint32_t s = col_slot0[j];
            for (uint32_t k : colrows[j]) {
                const auto& A = rowpat[k];
                const auto& B = rowpat[j];
                size_t ia = 0, ib = 0;
                size_t o = (size_t)P.lpair_ptr[s] * 2;
                while (ia < A.size() && ib < B.size() && A[ia] < j) {
                    if (A[ia] == B[ib]) {
                        P.lpairs[o++] = rowslot[k][ia];
                        P.lpairs[o++] = rowslot[j][ib];
                        ++ia;
                        ++ib;
                    } else if (A[ia] < B[ib])
                        ++ia;
                    else
                        ++ib;
                }
                ++s;
            }
        }
    }

    // ---- rows / columns of L for the triangular solves (indexed by variable id) -----------------------------------------
    P.fwd_ptr.assign(n + 1, 0);
    P.bwd_ptr.assign(n + 1, 0);
    for (uint32_t v = 0; v < n; ++v) {
        P.fwd_ptr[v + 1] = P.fwd_ptr[v] + (uint32_t)rowpat[pos[v]].size();
        P.bwd_ptr[v + 1] = P.bwd_ptr[v] + (uint32_t)colrows[pos[v]].size();
    }
    if (dense) {  // the kernel does not walk these
        std::fill(P.fwd_ptr.begin(), P.fwd_ptr.end(), 0u);
        std::fill(P.bwd_ptr.begin(), P.bwd_ptr.end(), 0u);
    }
    P.fwd_items.assign(dense ? 0 : (size_t)zlo * 2, 0);
    P.bwd_items.assign(dense ? 0 : (size_t)zlo * 2, 0);
    for (uint32_t v = 0; v < n && !dense; ++v) {
        uint32_t k = pos[v];
        size_t o = (size_t)P.fwd_ptr[v] * 2;
        for (size_t t = 0; t < rowpat[k].size(); ++t) {
            P.fwd_items[o++] = rowslot[k][t];
            P.fwd_items[o++] = order[rowpat[k][t]];
        }
        o = (size_t)P.bwd_ptr[v] * 2;
        uint32_t s = col_slot0[k];
        for (uint32_t rk : colrows[k]) {
            P.bwd_items[o++] = s++;
            P.bwd_items[o++] = order[rk];
        }
    }

    // ---- kind-sort the constraint table (wave-uniform evaluator branches) -------------------------------------------------
    auto part_of_con = [&](const DevCon& d) { return part_of_comp[comp[d.ids[kKinds[d.kind].nz[0][0]]]]; };
    {
        // stable order by (partition, kind): one counting pass, then every 80-byte record moves once
        std::vector<uint32_t> key(C);
        for (uint32_t i = 0; i < C; ++i) key[i] = part_of_con(cons[i]) * (uint32_t)EZPZ_NUM_KINDS + cons[i].kind;
        const std::vector<uint32_t> perm = counting_order(key, n_parts * (uint32_t)EZPZ_NUM_KINDS, nullptr);
        std::vector<DevCon> sorted(C);
        for (uint32_t i = 0; i < C; ++i) sorted[i] = cons[perm[i]];
        cons.swap(sorted);
    }
    {
        uint32_t i = 0;
        for (uint32_t p = 0; p < n_parts; ++p) {
            P.parts[p].con0 = i;
            while (i < C && part_of_con(cons[i]) == p) ++i;
            P.parts[p].con1 = i;
        }
    }

    // ---- internal renumbering ---------------------------------------------------------------------------------------
    // Every state array of the kernel is addressed through these lists only, so the numbering is ours to choose:
    // variables in schedule order (lvl_cols becomes the identity and is dropped), rows and Jacobian slots in
    // constraint-table order.  Lane l of a phase then touches word l of its array: no index load, no LDS bank
    // conflict.  List orders are unchanged, so every floating-point sum is formed in the same order as before.
    P.var_of = lvl_cols;
    std::vector<uint32_t> ivar(n);
    for (uint32_t k = 0; k < n; ++k) ivar[P.var_of[k]] = k;
    std::vector<uint32_t> new_row(m), new_slot(P.c.zj);
    P.row_of.assign(m, 0);
    {
        uint32_t run_row = 0, run_slot = 0;
        for (DevCon& d : cons) {
            const uint32_t r0 = run_row, s0 = run_slot;
            for (uint32_t r = 0; r < d.nrows; ++r) {
                new_row[d.row0 + r] = run_row;
                P.row_of[run_row] = d.row0 + r;
                ++run_row;
            }
            for (uint32_t t = 0; t < d.nslots; ++t) new_slot[d.jbase + t] = run_slot++;
            d.row0 = r0;
            d.jbase = s0;
            const KindInfo& K = kKinds[d.kind];
            for (int k = 0; k < K.n_ids; ++k) {
                // ids a kind never dereferences (not in `nonzeroes`) may be out of range: leave them at 0
                d.ids[k] = d.ids[k] < n ? ivar[d.ids[k]] : 0;
            }
        }
    }
    P.slot_row.assign(P.c.zj, 0);
    P.slot_col.assign(P.c.zj, 0);
    {
        std::vector<uint32_t> ptr(n + 1, 0), items;
        items.reserve(P.colj_items.size());
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t v = P.var_of[k];
            for (uint32_t q = P.colj_ptr[v]; q < P.colj_ptr[v + 1]; ++q) {
                const uint32_t s_new = new_slot[P.colj_items[2 * q]], row_old = P.colj_items[2 * q + 1];
                items.push_back(s_new);
                items.push_back(new_row[row_old]);
                P.slot_row[s_new] = row_old;
                P.slot_col[s_new] = v;
            }
            ptr[k + 1] = (uint32_t)(items.size() / 2);
        }
        P.colj_ptr.swap(ptr);
        P.colj_items.swap(items);
    }
    for (uint32_t& s2 : P.apairs) s2 = new_slot[s2];
    for (uint32_t& v : P.l_col) v = ivar[v];
    auto renumber_rows_of_l = [&](std::vector<uint32_t>& ptr_in, std::vector<uint32_t>& items_in) {
        std::vector<uint32_t> ptr(n + 1, 0), items;
        items.reserve(items_in.size());
        for (uint32_t k = 0; k < n; ++k) {
            const uint32_t v = P.var_of[k];
            for (uint32_t q = ptr_in[v]; q < ptr_in[v + 1]; ++q) {
                items.push_back(items_in[2 * q]);
                items.push_back(ivar[items_in[2 * q + 1]]);
            }
            ptr[k + 1] = (uint32_t)(items.size() / 2);
        }
        ptr_in.swap(ptr);
        items_in.swap(items);
    };
    renumber_rows_of_l(P.fwd_ptr, P.fwd_items);
    renumber_rows_of_l(P.bwd_ptr, P.bwd_items);
    P.cons = std::move(cons);
    return true;
}

}  // namespace ezpz
