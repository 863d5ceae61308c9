// FreedomAnalysis on the device (reference ezpz/src/solver/find_dof.rs:14-103, analysis.rs:24-77): the host side of
// freedom.hip.hpp's kernels -- the per-component program (build_freedom), the LANE / TEAM / WIDE launch paths and the two
// entry points of the C ABI.
#include "system.hpp"

#include "freedom.hip.hpp"

using namespace ezpz;

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
    F.big = 0;
    for (size_t c = 1; c < comps.size(); ++c) {
        auto words = [](const FreedomComp& k) { return (uint64_t)k.m * k.n + 2ull * k.n * k.n + 2ull * k.n; };
        if (words(comps[c]) > words(comps[F.big])) F.big = (uint32_t)c;
    }
    if (!comps.empty()) F.comp0 = comps[F.big];
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
// `done_flag` (optional): a word of mapped host memory; when the call fits ONE launch of one workgroup -- a small system's
// analysis: values gathered, Jacobian evaluated and analysed by the same kernel -- the kernel stores `done_seq` there at its end and
// *flagged says so (the caller polls the word instead of the stream: a stream query costs more than this kernel runs).
int freedom_device(EzpzSystem* sys, const double* x_dev, size_t batch, uint8_t* mask_dev, double* part_dev,
                   uint32_t* count_dev, hipStream_t stream, unsigned long long* done_flag = nullptr, unsigned long long done_seq = 0,
                   bool* flagged = nullptr, bool chain_route = false) {
    auto& F = sys->freedom;
    // (a kernel of the calling thread's one-call path that waits on the device for its next request holds hipStreamPerThread: work
    // put there would queue behind it until its lease runs out, so it is told to leave -- unless this call has a stream of its own)
    if (!done_flag) release_thread_kernel(sys->device);
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
    // Small calls whose components fit the LDS layouts (LANE, or TEAM with its workspace in LDS) are ONE launch: the kernel gathers
    // and evaluates itself (FreedomArgs::x_caller): solve_nonsquare_analysis 37.6 -> 23.8 us per call.
    constexpr bool fuse_enabled = true;
    const size_t head0 = (2 * (size_t)F.group + (F.group + 1) / 2 + 16) * sizeof(double);
    const bool in_lds = F.lane || head0 + (size_t)F.ws * sizeof(double) <= 128 * 1024;
    const bool fused = fuse_enabled && in_lds && batch <= 64 && (uint64_t)batch * sys->counts.n_cons <= 4096;
    if (flagged) *flagged = false;
    if (!fused) {
        const uint32_t* var_of = reinterpret_cast<const uint32_t*>(sys->view.base + sys->view.o_var_of);
        const uint64_t total = (uint64_t)batch * n;
        hipLaunchKernelGGL(gather_values_kernel, dim3((uint32_t)std::min<uint64_t>((total + 255) / 256, 65536)), dim3(256), 0,
                           stream, x_dev, var_of, F.x_int.p, (uint32_t)n, total);
        launch_eval(sys, F.x_int.p, batch, nullptr, F.jv.p, nullptr, (uint32_t)std::min<size_t>(batch, 8192), stream);
    }
    FreedomArgs a{};
    if (fused) {
        a.x_caller = x_dev;
        a.x_int = F.x_int.p;
        a.jv_out = F.jv.p;
        a.prog = sys->view;
    }
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
        static std::atomic<size_t> raised_lane[16];  // (the attribute belongs to the kernel: raised once per device and size)
        if (raised_lane[sys->device & 15].load(std::memory_order_relaxed) < lds) {
            HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised_lane[sys->device & 15].store(lds, std::memory_order_relaxed);
        }
        if (fused && grid == 1 && done_flag) {
            a.done_flag = done_flag;
            a.done_seq = done_seq;
            if (flagged) *flagged = true;
        }
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
            if (F.comp0.n >= 96) {
                // A big component (the system's largest; the others, if any, follow on the ordinary kernel): its pivoted QR as a chain of step launches over the whole device (freedom.hip.hpp),
                // `grid` systems side by side, then the ordinary kernel for rank / null space / participation.
                if ((rc = F.step_done.ensure(grid)) != EZPZ_OK) return rc;
                if ((rc = F.step_tau.ensure(grid)) != EZPZ_OK) return rc;
                HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<false>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                const uint32_t m = F.comp0.m, nc = F.comp0.n, ndiag = std::min(m, nc);
                // The route of the QR, decided before the initialisation (which zeroes the resident route's chunk area):
                //   * the matrix resident in the workgroups' registers for the whole factorisation (fr_qrc_kernel: up to 2048 rows,
                //     8 columns per workgroup, its reflector slots inside the null-space and Q blocks), as many systems side by side
                //     as the device holds that way (one at 1400 variables, two at 800) -- 7.9 / 15 / 23 ms per round at 800 / 1400 /
                //     2000 variables;
                //   * the chain of one launch pair per step: 24 / 65 / 126 ms, and a dozen systems side by side cost it little more
                //     -- so a call of more systems than a few resident rounds hold goes there;
                //   * one cooperative launch streaming the trailing matrix (fr_qr_kernel): 16 / 39 / 76 ms for ONE system, slower
                //     than the chain from two (its workgroups are dealt among the systems) -- one system the resident route
                //     cannot take.
                // EZPZ_FREEDOM_CHAIN=2 / 1 (read per call: tests switch it): not the first / only the last.
                const char* const chain_env = std::getenv("EZPZ_FREEDOM_CHAIN");
                // (chain_route: the caller's second attempt after a resident launch timed out -- not every workgroup of a cooperative
                // launch becomes resident beside another process's or stream's kernels; the chain's launches wait for nobody)
                const bool chain_only = chain_route || (chain_env && chain_env[0] == '1'), no_resident = chain_env && chain_env[0] == '2';
                int coop = 0;
                (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, sys->device);
                uint64_t cap_resident = 0, cap_streaming = 0;
                if (coop && !chain_only) {
                    int per_cu = 0;
                    if (!no_resident && m <= kQcRows && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fr_qrc_kernel, 1024, 0) == hipSuccess && per_cu > 0)
                        cap_resident = (uint64_t)sys->lim.cus * (uint64_t)per_cu;
                    per_cu = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fr_qr_kernel, 1024, 0) == hipSuccess && per_cu > 0)
                        cap_streaming = (uint64_t)sys->lim.cus * (uint64_t)per_cu;
                    (void)hipGetLastError();
                }
                // (workgroups x columns per workgroup of the resident route for `nb` systems side by side; 0 columns: does not fit)
                auto resident_shape = [&](uint32_t nb, uint32_t& cper_out, uint32_t& G_out, uint32_t& chunk_doubles) {
                    cper_out = G_out = chunk_doubles = 0;
                    if (!cap_resident || !nb) return;
                    const uint32_t g_max = (uint32_t)std::min<uint64_t>(kQcMaxWgs, cap_resident / nb);
                    const uint64_t m_slot = ((uint64_t)m + 3) & ~3ull, room = 2ull * nc * nc;  // doubles of the two blocks
                    for (uint32_t cper = g_max ? std::max(1u, (nc + g_max - 1) / g_max) : kQcCols + 1; cper <= kQcCols; ++cper) {
                        const uint32_t G = (nc + cper - 1) / cper;
                        const uint64_t need = kQcSmallDoubles + 4ull * G * m_slot;  // two sets of slots, two doubles a chunk
                        if (need <= room && need < 0xFFFFFFFFull) {
                            cper_out = cper, G_out = G, chunk_doubles = (uint32_t)need;
                            return;
                        }
                    }
                };
                uint32_t resident_side_by_side = 0;
                for (uint32_t nb = 1, cper, G, chunk; nb <= grid; ++nb) {
                    resident_shape(nb, cper, G, chunk);
                    if (!cper) break;
                    resident_side_by_side = nb;
                }
                // (a resident round is 10-12 us per Householder step, a step of the chain 29 / 46 / 63 us at 800 / 1400 / 2000 columns)
                const uint32_t resident_rounds_max = nc >= 1800 ? 5 : nc >= 1200 ? 4 : 2;
                const bool by_resident_rounds = resident_side_by_side && (batch + resident_side_by_side - 1) / resident_side_by_side <= resident_rounds_max;
                if (by_resident_rounds) grid = std::min(grid, resident_side_by_side);
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
                    uint32_t res_cper = 0, res_G = 0;
                    sa.k = 0;
                    if (by_resident_rounds) resident_shape(nb, res_cper, res_G, sa.k);
                    hipLaunchKernelGGL(fr_init_kernel, dim3(bl_mn, nb), dim3(256), 0, stream, sa);
                    hipLaunchKernelGGL(fr_scatter_kernel, dim3(bl_it, nb), dim3(256), 0, stream, sa);
                    hipLaunchKernelGGL(fr_norms_kernel, dim3((nc + 255) / 256, nb), dim3(256), 0, stream, sa);
                    bool cooperative = false, resident = false;
                    if (res_cper) {
                        uint32_t nd = ndiag, cper = res_cper;
                        void* params[] = {&sa, &nd, &cper};
                        if (hip_debug()) std::fprintf(stderr, "[ezpz hip] resident QR: m %u n %u, %u workgroups x %u columns, %u systems\n", m, nc, res_G, cper, nb);
                        const hipError_t ce = hipLaunchCooperativeKernel((const void*)fr_qrc_kernel, dim3(res_G, nb), dim3(1024), params, 0, stream);
                        cooperative = resident = ce == hipSuccess;
                        if (!cooperative && hip_debug()) std::fprintf(stderr, "[ezpz hip] resident QR launch -> %s\n", hipGetErrorString(ce));
                        (void)hipGetLastError();
                    }
                    if (!cooperative && cap_streaming && batch == 1) {
                        const uint32_t G = (uint32_t)std::min<uint64_t>(std::max<uint32_t>(1, (nc + kQrCols - 1) / kQrCols), cap_streaming);
                        uint32_t nd = ndiag;
                        void* params[] = {&sa, &nd};
                        const hipError_t ce = hipLaunchCooperativeKernel((const void*)fr_qr_kernel, dim3(G, nb), dim3(1024), params, 0, stream);
                        cooperative = ce == hipSuccess;
                        if (!cooperative && hip_debug())
                            std::fprintf(stderr, "[ezpz hip] cooperative QR launch (%u workgroups of %llu) -> %s\n", G, (unsigned long long)cap_streaming, hipGetErrorString(ce));
                        (void)hipGetLastError();
                    }
                    for (uint32_t k = 0; k < ndiag && !cooperative; ++k) {
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
                    fa.qr_done = F.big + 1;
                    fa.qr_timed_out = resident ? F.step_done.p : nullptr;  // (the chain uses these words for something else)
                    hipLaunchKernelGGL(freedom_kernel<false>, dim3(nb), dim3(F.threads), lds, stream, fa);
                }
                HIP_TRY(hipGetLastError());
                return EZPZ_OK;
            }
        }
        static std::atomic<size_t> raised_team[16];
        if (raised_team[sys->device & 15].load(std::memory_order_relaxed) < lds) {
            HIP_TRY(hipFuncSetAttribute((const void*)freedom_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            raised_team[sys->device & 15].store(lds, std::memory_order_relaxed);
        }
        if (fused && grid == 1 && done_flag && !a.gws) {
            a.done_flag = done_flag;
            a.done_seq = done_seq;
            if (flagged) *flagged = true;
        }
        hipLaunchKernelGGL(freedom_kernel<false>, dim3(grid), dim3(F.threads), lds, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return EZPZ_OK;
}

// ---- FreedomAnalysis of a system the FRONTS serve, without the pivoted QR --------------------------------------------------------------
// The reference factorises the dense Jacobian with column pivoting and reads the null space off R (find_dof.rs:31-103): for one
// connected sketch that is the unblocked QR of a 300 ... 2000-column matrix over the whole device, 2.6 ... 19 ms, fourteen to forty
// times the solve it follows.  What the analysis reports is basis-independent -- a variable's participation is the diagonal entry of
// the orthogonal projector P onto null(J) (the squared row norm of ANY orthonormal basis) -- and the frontal factorisation gives P
// applied to a vector for the price of one linear solve: lambda (JtJ + lambda I)^-1 w -> P w as lambda -> 0 (front_kernel.hip.hpp,
// FrontArgs::probe_m; lambda = 1e-11 x J's largest squared entry; what rounding adds to the answer lies along the small eigenvectors,
// i.e. INSIDE the null space).  m pseudo-random vectors w_j (entries uniform in [-1, 1)) give Y = P W (n x m), whose range is null(J) as soon as m exceeds
// its dimension: the eigenvalues of the m x m matrix Yt Y split into a cluster of order m / 3 (one per degree of freedom) and values of
// order m (lambda / sigma^2)^2 for everything else, and participation_i = sum over the cluster of (y_i . q_t)^2 / e_t.  A fully
// constrained sketch -- the common case -- is Y = 0 after the first eight solves.  Otherwise the directions of Y that stand out are
// refined by subspace iteration with the same operator (V <- orthonormalised Op V): its Ritz values on span(V) are lambda /
// (sigma^2 + lambda) themselves -- 1 for a null vector, below one half for anything whose singular value exceeds 3e-6 x J's
// largest entry, which is dropped -- and what is left of other directions in the kept vectors shrinks by that factor per round.
// Eight or more candidates, a Ritz value that stays between 0.5 and 0.999 after six rounds (a singular value of J within 1e-7 ...
// 3e-6 of its largest entry: where the reference's rank decision, 1e-8 on R's diagonal, and lambda's may disagree), a failed
// pivot, more than 64 systems, a system without a frontal plan: the pivoted QR.
namespace {

// Eigenvalues (ascending is not needed) and eigenvectors of a small symmetric matrix by cyclic Jacobi rotations: a[m][m] -> its
// diagonal holds the eigenvalues, q[m][m] the eigenvectors as columns.
void jacobi_eigen(std::vector<double>& a, std::vector<double>& q, uint32_t m) {
    q.assign((size_t)m * m, 0.0);
    for (uint32_t i = 0; i < m; ++i) q[(size_t)i * m + i] = 1.0;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (uint32_t i = 0; i < m; ++i)
            for (uint32_t j = 0; j < m; ++j) (i == j ? diag : off) += a[(size_t)i * m + j] * a[(size_t)i * m + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (uint32_t p = 0; p + 1 < m; ++p)
            for (uint32_t r = p + 1; r < m; ++r) {
                const double apr = a[(size_t)p * m + r];
                if (apr == 0.0) continue;
                const double theta = (a[(size_t)r * m + r] - a[(size_t)p * m + p]) / (2.0 * apr);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (uint32_t k = 0; k < m; ++k) {
                    const double akp = a[(size_t)k * m + p], akr = a[(size_t)k * m + r];
                    a[(size_t)k * m + p] = c * akp - sn * akr;
                    a[(size_t)k * m + r] = sn * akp + c * akr;
                }
                for (uint32_t k = 0; k < m; ++k) {
                    const double apk = a[(size_t)p * m + k], ark = a[(size_t)r * m + k];
                    a[(size_t)p * m + k] = c * apk - sn * ark;
                    a[(size_t)r * m + k] = sn * apk + c * ark;
                }
                for (uint32_t k = 0; k < m; ++k) {
                    const double qkp = q[(size_t)k * m + p], qkr = q[(size_t)k * m + r];
                    q[(size_t)k * m + p] = c * qkp - sn * qkr;
                    q[(size_t)k * m + r] = sn * qkp + c * qkr;
                }
            }
    }
}

// How the calls of freedom_by_probes ended, per process (ezpz_debug_freedom_exits; tests/test_gpu_freedom_fuzz.py logs them):
// [0] systems decided fully constrained by the first eight probes, [1] decided with null vectors found, [2] calls that took the
// second opinion; calls handed to the pivoted QR because: [3] an answer was not finite, [4] five or more candidates, [5] a direction
// undecided at both lambdas, [6] an unsettled direction at the end, [7] not applicable (switched off, no frontal plan, > 64 systems).
std::atomic<unsigned long long> g_probe_exits[8];

// 0: done (mask / participation written); 1: not decided here -- the caller runs the pivoted QR; negative: an error.
int freedom_by_probes(EzpzSystem* sys, const double* x, size_t batch, uint8_t* under_mask, double* participation) {
    const char* const env = std::getenv("EZPZ_FREEDOM_PROBES");  // =0: the pivoted QR for every system (read per call: tests switch it)
    const bool enabled = !(env && env[0] == '0');
    if (!enabled || !sys->fronts || !sys->dev_fronts || batch > 64) {
        g_probe_exits[7].fetch_add(1);
        return 1;
    }
    auto give_up = [&](const char* why2, int reason) {
        if (hip_debug()) std::fprintf(stderr, "[ezpz hip] null-space probes: %s -> the pivoted QR\n", why2);
        g_probe_exits[reason].fetch_add(1);
        return 1;
    };
    auto& F = sys->freedom;
    const size_t n = sys->counts.n_vars;
    constexpr uint32_t m = 8;
    int rc;
    // (the copies and probe launches below run on the null stream, which waits for hipStreamPerThread: a kernel of the calling
    // thread's one-call path that sits there waiting for its next request is told to leave first)
    release_thread_kernel(sys->device);
    if ((rc = F.x_in.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.probe.ensure(batch * m * n)) != EZPZ_OK) return rc;
    if ((rc = F.probe_w.ensure(batch * m * n)) != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(F.x_in.p, x, batch * n * sizeof(double), hipMemcpyHostToDevice));
    // ---- eight random probes: Y = Op W, Op = lambda (JtJ + lambda I)^-1 (symmetric; eigenvalue 1 on null(J), lambda / sigma^2 off it)
    if ((rc = front_launch_probe(*sys, F.x_in.p, batch, F.probe.p, m, nullptr)) != EZPZ_OK) return rc;
    std::vector<double> Y(batch * m * n), V(batch * m * n, 0.0), G, Q, H;
    HIP_TRY(hipMemcpy(Y.data(), F.probe.p, Y.size() * sizeof(double), hipMemcpyDeviceToHost));
    auto gram = [&](const double* A, const double* B2, uint32_t k, std::vector<double>& out) {  // out[j][t] = A_j . B_t
        out.assign((size_t)k * k, 0.0);
        for (uint32_t j2 = 0; j2 < k; ++j2)
            for (uint32_t t = 0; t < k; ++t) {
                double sum = 0.0;
                for (size_t i2 = 0; i2 < n; ++i2) sum += A[j2 * n + i2] * B2[t * n + i2];
                out[(size_t)j2 * k + t] = sum;
            }
    };
    // per system: the subspace that may hold null vectors -- the eigenvectors of Yt Y that stand out of the (lambda / sigma^2)^2 floor
    std::vector<uint32_t> kdim(batch, 0);
    bool any = false;
    for (size_t b = 0; b < batch; ++b) {
        const double* Yb = Y.data() + b * m * n;
        gram(Yb, Yb, m, G);
        for (double g : G)
            if (!(std::fabs(g) < 1e300)) return give_up("a probe's answer is not finite (a pivot failed)", 3);
        for (uint32_t a2 = 0; a2 < m; ++a2)
            for (uint32_t b2 = a2 + 1; b2 < m; ++b2) G[(size_t)b2 * m + a2] = G[(size_t)a2 * m + b2] = 0.5 * (G[(size_t)a2 * m + b2] + G[(size_t)b2 * m + a2]);
        jacobi_eigen(G, Q, m);
        double* Vb = V.data() + b * m * n;
        uint32_t k = 0;
        for (uint32_t t = 0; t < m; ++t) {
            const double e = G[(size_t)t * m + t];
            // (a null vector's eigenvalue is a sum of eight squares of variance 1/3: of order m / 3; a direction answered by a tenth of
            // itself or less is not one)
            if (!(e > 0.005 * m)) continue;
            const double inv = 1.0 / std::sqrt(e);
            for (size_t i2 = 0; i2 < n; ++i2) {
                double d = 0.0;
                for (uint32_t j2 = 0; j2 < m; ++j2) d += Q[(size_t)j2 * m + t] * Yb[j2 * n + i2];
                Vb[k * n + i2] = d * inv;
            }
            ++k;
        }
        if (k >= 5) return give_up("five or more candidate directions from eight probes", 4);
        kdim[b] = k;
        any = any || k;
    }
    // ---- subspace iteration on the candidates: V <- orthonormalised Op V, until every Ritz value of Op on span(V) is 1 (a null
    //      vector) or below 1e-4 (not one).  The Ritz values ARE lambda / (sigma^2 + lambda): no threshold on magnitudes of J.
    std::vector<std::vector<double>> ritz(batch);
    for (int round = 0; any && round < 6; ++round) {
        // (as many probes as the system with the most candidates has; rows of `mk` vectors per system)
        uint32_t mk = 0;
        for (size_t b = 0; b < batch; ++b) mk = std::max(mk, kdim[b]);
        any = mk != 0;
        if (!any) break;
        for (size_t b = 0; b < batch; ++b)
            HIP_TRY(hipMemcpy(F.probe_w.p + b * mk * n, V.data() + b * m * n, (size_t)mk * n * sizeof(double), hipMemcpyHostToDevice));
        if ((rc = front_launch_probe(*sys, F.x_in.p, batch, F.probe.p, mk, nullptr, F.probe_w.p)) != EZPZ_OK) return rc;
        for (size_t b = 0; b < batch; ++b)
            HIP_TRY(hipMemcpy(Y.data() + b * m * n, F.probe.p + b * mk * n, (size_t)mk * n * sizeof(double), hipMemcpyDeviceToHost));
        bool settled = true;
        for (size_t b = 0; b < batch; ++b) {
            const uint32_t k = kdim[b];
            if (!k) continue;
            const double* Zb = Y.data() + b * m * n;
            double* Vb = V.data() + b * m * n;
            gram(Vb, Zb, k, H);  // Vt Op V
            for (double g : H)
                if (!(std::fabs(g) < 1e300)) return give_up("a refinement's answer is not finite", 3);
            for (uint32_t a2 = 0; a2 < k; ++a2)
                for (uint32_t b2 = a2 + 1; b2 < k; ++b2) H[(size_t)b2 * k + a2] = H[(size_t)a2 * k + b2] = 0.5 * (H[(size_t)a2 * k + b2] + H[(size_t)b2 * k + a2]);
            jacobi_eigen(H, Q, k);
            ritz[b].assign(k, 0.0);
            // the Ritz vectors' images, orthonormalised (modified Gram-Schmidt, largest Ritz value first), become the next V
            std::vector<uint32_t> order(k);
            for (uint32_t t = 0; t < k; ++t) order[t] = t;
            std::sort(order.begin(), order.end(), [&](uint32_t a2, uint32_t b2) { return H[(size_t)a2 * k + a2] > H[(size_t)b2 * k + b2]; });
            std::vector<double> next((size_t)k * n, 0.0);
            uint32_t kept = 0;
            for (uint32_t o = 0; o < k; ++o) {
                const uint32_t t = order[o];
                const double f = H[(size_t)t * k + t];
                double* u = next.data() + (size_t)kept * n;
                for (size_t i2 = 0; i2 < n; ++i2) {
                    double d = 0.0;
                    for (uint32_t j2 = 0; j2 < k; ++j2) d += Q[(size_t)j2 * k + t] * Zb[j2 * n + i2];
                    u[i2] = d;
                }
                for (uint32_t q2 = 0; q2 < kept; ++q2) {
                    const double* w2 = next.data() + (size_t)q2 * n;
                    double dot = 0.0;
                    for (size_t i2 = 0; i2 < n; ++i2) dot += u[i2] * w2[i2];
                    for (size_t i2 = 0; i2 < n; ++i2) u[i2] -= dot * w2[i2];
                }
                double nrm = 0.0;
                for (size_t i2 = 0; i2 < n; ++i2) nrm += u[i2] * u[i2];
                // (a null vector is answered by itself -- f = 1 up to the rounding of J w, ~1e-5 at this lambda; a direction answered
                // by less than half of itself has sigma^2 > lambda: not one, and out of the candidates)
                if (!(f >= 0.5) || !(nrm > 1e-28)) continue;
                const double inv = 1.0 / std::sqrt(nrm);
                for (size_t i2 = 0; i2 < n; ++i2) u[i2] *= inv;
                ritz[b][kept] = f;
                ++kept;
            }
            ritz[b].resize(kept);
            std::fill(Vb, Vb + (size_t)m * n, 0.0);
            std::copy(next.begin(), next.begin() + (size_t)kept * n, Vb);
            kdim[b] = kept;
            for (double f : ritz[b])
                if (f < 0.999) settled = false;
        }
        // (three rounds at least: what is not a null vector leaves the kept ones at lambda / sigma^2 per round)
        if (settled && round >= 2) break;
        if (round == 5) {
            g_probe_exits[2].fetch_add(1);
            // A Ritz value stays between 0.5 and 0.999: a singular value of J within 1e-7 ... 3e-6 of its largest entry, which the
            // reference's rank decision (1e-8 on R's diagonal) counts as non-zero.  A second opinion at lambda_p / 1000: what it
            // answers by less than half of itself there is NOT a null vector and leaves; the null vectors' own answers are noisier
            // (the rounding of J w over a smaller lambda: 1e-2), so they only have to stay above 0.9.  Anything between, or a failed
            // pivot at that lambda: the pivoted QR.
            uint32_t mk2 = 0;
            for (size_t b = 0; b < batch; ++b) mk2 = std::max(mk2, kdim[b]);
            for (size_t b = 0; b < batch; ++b)
                HIP_TRY(hipMemcpy(F.probe_w.p + b * mk2 * n, V.data() + b * m * n, (size_t)mk2 * n * sizeof(double), hipMemcpyHostToDevice));
            if ((rc = front_launch_probe(*sys, F.x_in.p, batch, F.probe.p, mk2, nullptr, F.probe_w.p, 1e-14)) != EZPZ_OK) return rc;
            for (size_t b = 0; b < batch; ++b)
                HIP_TRY(hipMemcpy(Y.data() + b * m * n, F.probe.p + b * mk2 * n, (size_t)mk2 * n * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t b = 0; b < batch; ++b) {
                const uint32_t k = kdim[b];
                const double* Zb = Y.data() + b * m * n;
                double* Vb = V.data() + b * m * n;
                uint32_t kept = 0;
                for (uint32_t t = 0; t < k; ++t) {
                    double f2 = 0.0;
                    for (size_t i2 = 0; i2 < n; ++i2) f2 += Vb[t * n + i2] * Zb[t * n + i2];
                    if (!(std::fabs(f2) < 1e300)) return give_up("a pivot failed at the second opinion's lambda", 3);
                    if (ritz[b][t] >= 0.999 || f2 > 0.9) {  // a null vector (kept as it is)
                        if (kept != t) std::copy(Vb + (size_t)t * n, Vb + (size_t)(t + 1) * n, Vb + (size_t)kept * n);
                        ritz[b][kept] = 1.0;
                        ++kept;
                    } else if (f2 >= 0.5) {
                        return give_up("a direction answered by 0.5 ... 0.9 of itself at both lambdas (a singular value of J within 3e-9 ... 1e-7 of its largest entry)", 5);
                    }
                }
                ritz[b].resize(kept);
                kdim[b] = kept;
            }
            break;
        }
    }
    // ---- what the probes call a null vector must be one by the REFERENCE's measure -------------------------------------------------
    // The probes tell sigma = 0 from sigma > 3e-6 x J's largest entry; a direction held by a WEAK row -- a singular value within
    // ~1e-10 ... 3e-6 of the largest -- is answered like a null vector (the operator's own rounding is 1e-5), while the reference
    // counts it as rank whenever its pivot exceeds 1e-8 of the largest (find_dof.rs:36-49): tests/test_gpu_freedom_fuzz.py found
    // every such disagreement at weights 3e-8 ... 1e-6.  ||J v|| is exact to ~1e-16 x ||J||: a kept direction with ||J v|| above
    // 1e-10 of J's largest column norm (the reference's scale: its first pivot) hands the system to the pivoted QR, which makes the
    // reference's decision the reference's way; below, every pivot of that direction is a hundred times under its threshold.
    {
        bool any_kept = false;
        for (size_t b = 0; b < batch; ++b) any_kept = any_kept || kdim[b] != 0;
        if (any_kept) {
            if ((rc = ensure_program(sys)) != EZPZ_OK) return rc;
            const size_t zj = sys->counts.zj, mrows = sys->counts.n_rows;
            std::vector<double> rr(batch * std::max<size_t>(mrows, 1)), jv(batch * std::max<size_t>(zj, 1)), y(mrows), col2(n);
            if ((rc = eval_batch_locked(sys, x, batch, rr.data(), jv.data(), nullptr)) != EZPZ_OK) return rc;
            for (size_t b = 0; b < batch; ++b) {
                if (!kdim[b]) continue;
                const double* jb = jv.data() + b * zj;
                std::fill(col2.begin(), col2.end(), 0.0);
                for (size_t s2 = 0; s2 < zj; ++s2) col2[sys->host_slot_col[s2]] += jb[s2] * jb[s2];
                double scale2 = 0.0;
                for (double c : col2) scale2 = std::max(scale2, c);
                for (uint32_t t = 0; t < kdim[b]; ++t) {
                    const double* v = V.data() + b * m * n + (size_t)t * n;
                    std::fill(y.begin(), y.end(), 0.0);
                    for (size_t s2 = 0; s2 < zj; ++s2) y[sys->host_slot_row[s2]] += jb[s2] * v[sys->host_slot_col[s2]];
                    double jv2 = 0.0;
                    for (double e : y) jv2 += e * e;
                    if (!(jv2 <= 1e-20 * scale2))
                        return give_up("a kept direction is held by a singular value above 1e-10 of J's largest column norm: the reference's rank rule decides", 5);
                }
            }
        }
    }
    // ---- participation = squared row norms of the null vectors (find_dof.rs:90-103) ---------------------------------------------------
    std::vector<double> proj(n);
    for (size_t b = 0; b < batch; ++b) {
        const double* Vb = V.data() + b * m * n;
        std::fill(proj.begin(), proj.end(), 0.0);
        uint32_t nullity = 0;
        for (uint32_t t = 0; t < kdim[b]; ++t) {
            if (!(ritz[b][t] >= 0.999)) return give_up("an unsettled direction", 6);
            ++nullity;
            for (size_t i2 = 0; i2 < n; ++i2) proj[i2] += Vb[t * n + i2] * Vb[t * n + i2];
        }
        double most_part = 0.0;
        for (size_t i2 = 0; i2 < n; ++i2) most_part = std::max(most_part, proj[i2]);
        const double var_tol = kFreedomVarTol * most_part, squared_tol = var_tol * var_tol;
        uint8_t* mk = under_mask + b * n;
        double* pt = participation ? participation + b * n : nullptr;
        for (size_t i2 = 0; i2 < n; ++i2) {
            mk[i2] = nullity && proj[i2] > squared_tol ? 1 : 0;
            if (pt) pt[i2] = proj[i2];
        }
        g_probe_exits[nullity ? 1 : 0].fetch_add(1);
    }
    return 0;
}

}  // namespace

extern "C" void ezpz_debug_freedom_exits(unsigned long long* out8) {
    for (int i = 0; i < 8; ++i) out8[i] = g_probe_exits[i].load();
}

// A system whose resident QR timed out (FreedomArgs::qr_timed_out) carries kFreedomPoisonedMask in every byte of its mask.
bool mask_poisoned(const uint8_t* mask, size_t batch, size_t n) {
    for (size_t b = 0; b < batch; ++b)
        if (mask[b * n] == kFreedomPoisonedMask) return true;
    return false;
}

}  // namespace

extern "C" {

int ezpz_system_freedom_batch_device(EzpzSystem* sys, const double* x_dev, size_t batch, uint8_t* under_mask_dev,
                                     double* participation_dev, uint32_t* n_under_dev, void* stream) {
    if (!sys || (batch && (!x_dev || !under_mask_dev))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->mu);
    EZPZ_ON_DEVICE(sys->device);
    return freedom_device(sys, x_dev, batch, under_mask_dev, participation_dev, n_under_dev, (hipStream_t)stream);
}

int ezpz_system_freedom_batch(EzpzSystem* sys, const double* x, size_t batch, uint8_t* under_mask,
                              double* participation) {
    if (!sys || (batch && (!x || !under_mask))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    std::lock_guard<std::mutex> lock(sys->mu);
    EZPZ_ON_DEVICE(sys->device);
    auto& F = sys->freedom;
    const size_t n = sys->counts.n_vars;
    if (n == 0 || sys->counts.n_rows == 0) return EZPZ_ERR_EMPTY_SYSTEM;
    int rc;
    if ((rc = freedom_by_probes(sys, x, batch, under_mask, participation)) <= 0) return rc;  // (a system on the fronts: no QR)
    const size_t x_bytes = batch * n * sizeof(double), mask_bytes = (batch * n + 15) & ~size_t(15), part_bytes = participation ? x_bytes : 0;
    if (x_bytes + mask_bytes + part_bytes <= sys->lim.policy.zero_copy_max_bytes) {
        // Small call (solve_analysis of one sketch): no copies and no allocation.  The kernels read the values from, and
        // write the mask (and the participation) to, the calling thread's pinned buffer mapped into the device address
        // space, on the thread's own stream; one poll of that stream at the end.  (The analysis of a 4-variable system
        // was 270 us through a per-call hipMalloc, three blocking copies on the null stream and a hipFree.)
        static thread_local PinnedBuf t_buf[16];
        PinnedBuf& pinned = t_buf[sys->device & 15];
        // a stream of the thread's own for these calls: the analysis then runs BESIDE a resident one-call kernel of the same thread
        // (solve_analysis = solve, then this: the solve's kernel keeps waiting for the next solve, no relaunch per call)
        static thread_local hipStream_t t_stream[16] = {};
        hipStream_t& fstream = t_stream[sys->device & 15];
        if (!fstream) HIP_TRY(hipStreamCreateWithFlags(&fstream, hipStreamNonBlocking));
        if ((rc = pinned.ensure(x_bytes + mask_bytes + part_bytes + 16)) != EZPZ_OK) return rc;
        unsigned char* h = pinned.p;
        std::memcpy(h, x, x_bytes);
        uint8_t* hmask = h + x_bytes;
        double* hpart = participation ? reinterpret_cast<double*>(h + x_bytes + mask_bytes) : nullptr;
        // (the completion word behind the buffers: the one-launch form stores the call's sequence number there at its end)
        unsigned long long* const flag = reinterpret_cast<unsigned long long*>(h + ((x_bytes + mask_bytes + part_bytes + 7) & ~size_t(7)));
        static thread_local unsigned long long t_seq = 0;
        const unsigned long long seq = ++t_seq;
        bool flagged = false;
        if ((rc = freedom_device(sys, reinterpret_cast<const double*>(h), batch, hmask, hpart, nullptr, fstream, flag, seq, &flagged)) != EZPZ_OK) return rc;
        hipError_t q = hipSuccess;
        int spins = 0;
        if (flagged) {
            // ~6 us launch-to-flag against ~13 us launch-to-hipStreamQuery (tools/launch_floor.hip); a kernel that never signals
            // (a fault) is caught by the stream after a while
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq)
                if ((++spins & 0xFFFF) == 0 && hipStreamQuery(fstream) != hipErrorNotReady) break;
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) q = hipStreamSynchronize(fstream);
        } else
        while ((q = hipStreamQuery(fstream)) == hipErrorNotReady)
            if (++spins > 4000) {  // (long analyses -- a large component -- block instead of burning a core)
                q = hipStreamSynchronize(fstream);
                break;
            }
        if (q != hipSuccess) {
            (void)hipGetLastError();
            return EZPZ_ERR_HIP;
        }
        if (mask_poisoned(hmask, batch, n)) {
            // the resident QR gave up waiting for workgroups that never became resident (a co-tenant): once more by the chain of launches
            if (hip_debug()) std::fprintf(stderr, "[ezpz hip] the resident QR timed out -> the chain of launches\n");
            if ((rc = freedom_device(sys, reinterpret_cast<const double*>(h), batch, hmask, hpart, nullptr, fstream, nullptr, 0, nullptr, true)) != EZPZ_OK) return rc;
            if (hipStreamSynchronize(fstream) != hipSuccess) {
                (void)hipGetLastError();
                return EZPZ_ERR_HIP;
            }
        }
        std::memcpy(under_mask, hmask, batch * n);
        if (participation) std::memcpy(participation, hpart, x_bytes);
        return mask_poisoned(under_mask, batch, n) ? EZPZ_ERR_HIP : EZPZ_OK;
    }
    if ((rc = F.x_in.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.mask.ensure(batch * n)) != EZPZ_OK) return rc;
    if ((rc = F.part.ensure(batch * n)) != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(F.x_in.p, x, batch * n * sizeof(double), hipMemcpyHostToDevice));
    if ((rc = freedom_device(sys, F.x_in.p, batch, F.mask.p, F.part.p, nullptr, nullptr)) != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(under_mask, F.mask.p, batch * n, hipMemcpyDeviceToHost));
    if (mask_poisoned(under_mask, batch, n)) {  // (as above: the resident QR timed out -> the chain of launches)
        if (hip_debug()) std::fprintf(stderr, "[ezpz hip] the resident QR timed out -> the chain of launches\n");
        if ((rc = freedom_device(sys, F.x_in.p, batch, F.mask.p, F.part.p, nullptr, nullptr, nullptr, 0, nullptr, true)) != EZPZ_OK) return rc;
        HIP_TRY(hipMemcpy(under_mask, F.mask.p, batch * n, hipMemcpyDeviceToHost));
    }
    if (participation) HIP_TRY(hipMemcpy(participation, F.part.p, batch * n * sizeof(double), hipMemcpyDeviceToHost));
    if (mask_poisoned(under_mask, batch, n)) return EZPZ_ERR_HIP;
    return EZPZ_OK;
}

}  // extern "C"

