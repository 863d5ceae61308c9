// The topology program as device data: the index lists of a Program (program.hpp) packed into one blob with 16- or
// 32-bit indices (pack_program), a grid team's per-workgroup slices, the record walk's rounds (build_records), the groups of
// lanes that share a level's lists, and the dense phases at the top of a connected sketch's elimination tree.  Host code; the
// kernels that read these layouts are lm_kernel.hip.hpp's.
#include "system.hpp"

using namespace ezpz;

namespace ezpz {

size_t pack_program(const Program& P, bool idx16, bool pack_table, std::vector<unsigned char>& blob, ProgramView& v) {
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
bool pack_grid_slices(EzpzSystem& s, const Program& P, uint32_t G, uint32_t W) {
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
// (analyze_into: up to this many components walk records as one partition; from 128 the component-resident shape may take the system)
extern const uint32_t kRecMaxComponents = launch_policy_for(256).rec_max_components;
// `wide`: the workspace lives in global memory -- 32-bit addresses counted from its start (lds_base = 0), chunk 0 = target,
// diagonal, destination, lane flags, then up to four chunks of two (a, b) pairs; no packed assembly.
// `jglobal` (LDS form): the Jacobian's values live in global memory (SolveArgs::rec_jglobal): no room for them in the workspace, and
// the packed assembly's J operands are plain slot numbers (padding: slot zJ, a zero behind the values).
bool build_records(const Program& P, uint32_t T, uint32_t lds_base, bool wide, bool jglobal, RecPlan& out) {
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
void choose_level_groups(Program& P, const EzpzSystem& s) {
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
bool make_dense_phases(Program& P, uint32_t n_waves, size_t lds_room_bytes) {
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
    constexpr uint32_t max_phases = 4;
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
    if (debug_topic("dense")) {
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


}  // namespace ezpz
