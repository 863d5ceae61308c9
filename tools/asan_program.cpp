// Host symbolic phase under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the pool):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iezpz_amd/csrc -Iinclude tools/asan_program.cpp ezpz_amd/csrc/program.cpp -o /tmp/asan_program && /tmp/asan_program
// 600 random systems (all-kinds soups, hubs declared first/last up to 3000 points, chains up to 6000 points, block systems)
// through build_program with 1 and 8 partitions (and the dense layout for <= 8 variables): "built 1200 failed 0", no reports.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "program.hpp"
using namespace ezpz;
int main() {
    std::mt19937_64 rng(12345);
    size_t built = 0, failed = 0;
    for (int trial = 0; trial < 600; ++trial) {
        const int shape = trial % 4;
        size_t n_vars = 0;
        std::vector<EzpzConstraint> cs;
        auto add = [&](uint16_t kind, std::initializer_list<uint32_t> ids, double param) {
            EzpzConstraint c;
            std::memset(&c, 0, sizeof(c));
            c.kind = kind;
            int k = 0;
            for (uint32_t v : ids) c.ids[k++] = v;
            c.param = param;
            c.weight = 1.0;
            cs.push_back(c);
        };
        if (shape == 0) {  // random soup of all kinds
            n_vars = 4 + rng() % 60;
            const int nc = 1 + rng() % 40;
            for (int i = 0; i < nc; ++i) {
                EzpzConstraint c;
                std::memset(&c, 0, sizeof(c));
                c.kind = (uint16_t)(rng() % EZPZ_NUM_KINDS);
                for (int k = 0; k < 8; ++k) c.ids[k] = (uint32_t)(rng() % n_vars);
                c.param = 1.0; c.weight = 1.0; c.tag = 1;
                cs.push_back(c);
            }
        } else if (shape == 1) {  // hub, declared first or last
            const uint32_t npts = 10 + rng() % 3000;
            const bool last = rng() & 1;
            n_vars = 2 * (npts + 1);
            const uint32_t hub = last ? npts : 0;
            add(EZPZ_FIXED, {2 * hub}, 1.0); add(EZPZ_FIXED, {2 * hub + 1}, 2.0);
            for (uint32_t i = 0; i < npts; ++i) {
                const uint32_t p = last ? i : i + 1;
                add(EZPZ_DISTANCE, {2 * p, 2 * p + 1, 2 * hub, 2 * hub + 1}, 3.0);
                add(EZPZ_HORIZONTAL_DISTANCE, {2 * p, 2 * p + 1, 2 * hub, 2 * hub + 1}, 1.0);
            }
        } else if (shape == 2) {  // chain
            const uint32_t npts = 2 + rng() % 6000;
            n_vars = 2 * npts;
            add(EZPZ_FIXED, {0}, 0.0); add(EZPZ_FIXED, {1}, 0.0);
            for (uint32_t k = 1; k < npts; ++k) {
                add(EZPZ_DISTANCE, {2 * (k - 1), 2 * k - 1, 2 * k, 2 * k + 1}, 1.0);
                if (rng() % 3) add(EZPZ_HORIZONTAL, {2 * (k - 1), 2 * k - 1, 2 * k, 2 * k + 1}, 0.0);
                if (k > 3 && rng() % 4 == 0) { uint32_t b = k - 2 - rng() % 2; add(EZPZ_DISTANCE, {2 * b, 2 * b + 1, 2 * k, 2 * k + 1}, 2.0); }
            }
        } else {  // blocks
            const uint32_t nb = 1 + rng() % 900;
            n_vars = 4 * nb;
            for (uint32_t b = 0; b < nb; ++b) {
                add(EZPZ_VERTICAL, {4 * b, 4 * b + 1, 4 * b + 2, 4 * b + 3}, 0.0);
                add(EZPZ_FIXED, {4 * b}, 1.0); add(EZPZ_FIXED, {4 * b + 1}, 0.0); add(EZPZ_FIXED, {4 * b + 3}, 4.0);
            }
        }
        for (uint32_t want : {1u, 8u}) {
            Program P; BuildError e;
            if (build_program(cs.data(), cs.size(), n_vars, P, e, want, false)) ++built; else ++failed;
        }
        if (n_vars <= 8) { Program P; BuildError e; build_program(cs.data(), cs.size(), n_vars, P, e, 1, true); }
    }
    std::printf("built %zu failed %zu\n", built, failed);
    return 0;
}
