// g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iezpz_amd/csrc -Iinclude tools/asan_lane_plan.cpp ezpz_amd/csrc/comp_program.cpp ezpz_amd/csrc/program.cpp -o /tmp/asan_lane && /tmp/asan_lane
// round 4: "lane plans 1399, with a wave source 1235, with the elimination across lanes 1235", no reports.
// lane_plan_build (comp_program.cpp: lane class, wave tables, the elimination across lanes) under ASan + UBSan on random small systems
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "comp_program.hpp"
using namespace ezpz;
int main() {
    std::mt19937_64 rng(4242);
    size_t built = 0, wave = 0, tail = 0;
    for (int trial = 0; trial < 1500; ++trial) {
        const size_t n_vars = 2 + rng() % 19;
        std::vector<EzpzConstraint> cs;
        const int ncs = 1 + (int)(rng() % 24);
        for (int i = 0; i < ncs; ++i) {
            EzpzConstraint c;
            std::memset(&c, 0, sizeof(c));
            c.kind = (uint16_t)(rng() % 25);
            for (int k = 0; k < 8; ++k) c.ids[k] = (uint32_t)(rng() % n_vars);
            c.param = 1.0 + (double)(rng() % 7);
            c.weight = (rng() % 4) ? 1.0 : 0.5;
            c.tag = (uint8_t)(rng() % 3);
            cs.push_back(c);
        }
        LanePlan plan;
        if (!lane_plan_build(cs.data(), cs.size(), n_vars, plan)) continue;
        ++built;
        wave += !plan.wave_source.empty();
        tail += plan.wave_source.find("HAS_TAIL_WAVE = true") != std::string::npos;
    }
    std::printf("lane plans %zu, with a wave source %zu, with the elimination across lanes %zu\n", built, wave, tail);
    return 0;
}
