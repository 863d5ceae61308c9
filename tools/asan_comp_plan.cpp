// g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iezpz_amd/csrc -Iinclude tools/asan_comp_plan.cpp ezpz_amd/csrc/comp_program.cpp ezpz_amd/csrc/program.cpp -o /tmp/asan_comp && /tmp/asan_comp
// round 6: comp_plan_build (comp_program.cpp: classes, slots, the source of the run-time compiled kernels, the per-wavefront ranges of
// the rows the kernels that do not wait for verdicts move as whole lines) under ASan + UBSan on random BLOCK systems: blocks of
// 1-6 variables drawn from a few random topologies of linear and non-linear kinds, 1 ... 60 000 blocks, shuffled or in order.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "comp_program.hpp"
using namespace ezpz;
int main() {
    std::mt19937_64 rng(777);
    size_t tried = 0, built = 0, with_source = 0, source_bytes = 0;
    const uint16_t linear_kinds[] = {0, 1, 2, 3, 4, 5, 7, 8};  // Fixed ... Midpoint are not all of them: any kind id below 25 is legal input
    for (int trial = 0; trial < 400; ++trial) {
        const int n_topo = 1 + (int)(rng() % 3);
        struct Topo { int nv; std::vector<EzpzConstraint> cs; };
        std::vector<Topo> topo(n_topo);
        for (auto& t : topo) {
            t.nv = 1 + (int)(rng() % 6);
            const int nc = 1 + (int)(rng() % 6);
            const bool linear = rng() % 3 != 0;
            for (int i = 0; i < nc; ++i) {
                EzpzConstraint c;
                std::memset(&c, 0, sizeof(c));
                c.kind = linear ? linear_kinds[rng() % 8] : (uint16_t)(rng() % 25);
                for (int k = 0; k < 8; ++k) c.ids[k] = (uint32_t)(rng() % t.nv);
                c.param = 1.0 + (double)(rng() % 5);
                c.weight = (rng() % 5) ? 1.0 : 0.25;
                c.tag = (uint8_t)(rng() % 3);
                t.cs.push_back(c);
            }
        }
        const size_t blocks = trial % 7 == 0 ? 20000 + rng() % 40000 : 1 + rng() % 3000;
        std::vector<EzpzConstraint> cs;
        size_t n_vars = 0;
        std::vector<uint32_t> base;
        for (size_t b = 0; b < blocks; ++b) {
            const Topo& t = topo[rng() % n_topo];
            for (EzpzConstraint c : t.cs) {
                for (int k = 0; k < 8; ++k) c.ids[k] += (uint32_t)n_vars;
                cs.push_back(c);
            }
            n_vars += t.nv;
        }
        if (trial % 3 == 0) std::shuffle(cs.begin(), cs.end(), rng);
        CompLimits lim;
        if (trial % 5 == 0) lim.lds_bytes = 64 * 1024;
        CompPlan plan;
        ++tried;
        if (!comp_plan_build(cs.data(), cs.size(), n_vars, lim, plan)) continue;
        ++built;
        if (!plan.jit_source.empty()) {
            ++with_source;
            source_bytes += plan.jit_source.size();
        }
    }
    std::printf("block systems %zu, component plans %zu, with a run-time compiled kernel's source %zu (%zu bytes of source)\n", tried, built, with_source, source_bytes);
    return 0;
}
