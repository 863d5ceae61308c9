// Exactness of the shared-reciprocal division of the class-specialised kernels (ezpz_amd/csrc/jit_kernel.hip.hpp:
// recip_of / div_by) against the compiler's own correctly rounded `n / D`, bit for bit, over operands that sweep the
// whole binary64 range: wherever the guard `ok` stays true the two must agree exactly (the kernels repeat the solve
// with plain divisions where it does not).  Test infrastructure: built and run by tests/test_gpu_div.py.
// Prints: <pairs> <ok pairs> <mismatches among ok pairs> <ok pairs with zero numerator> <not-ok pairs>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "jit_kernel.hip.hpp"

__device__ __forceinline__ uint64_t mix(uint64_t z) {  // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ double make(uint64_t sign, uint64_t biased_exp, uint64_t mant) {
    return __builtin_bit_cast(double, (sign << 63) | (biased_exp << 52) | (mant & 0xFFFFFFFFFFFFFull));
}

// counts[0] pairs, [1] ok, [2] mismatches, [3] ok with zero numerator, [4] not ok
__global__ void sweep(unsigned long long* counts, uint64_t seed, int mode) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long pairs = 0, oks = 0, bad = 0, zeros = 0, notok = 0;
    for (int rep = 0; rep < 64; ++rep) {
        const uint64_t h0 = mix(seed + t * 131 + rep), h1 = mix(h0), h2 = mix(h1), h3 = mix(h2);
        double n, D;
        // denominators: what a pivot's square root can be -- any positive exponent in and around the guard's range,
        // random or special mantissas (powers of two, all ones)
        const uint64_t de = 1023 - 48 + (h0 % 97);
        const uint64_t dm = (h0 >> 8) % 5 == 0 ? 0 : (h0 >> 8) % 5 == 1 ? 0xFFFFFFFFFFFFFull : h1;
        D = make(0, de, dm);
        if (mode == 0) {  // numerators over every exponent, denormals, zeros, infinities, NaN
            const uint64_t ne = h2 % 2048;
            const uint64_t nm = (h2 >> 12) % 7 == 0 ? 0 : (h2 >> 12) % 7 == 1 ? 0xFFFFFFFFFFFFFull : (h2 >> 12) % 7 == 2 ? 1 : h3;
            n = make(h3 >> 63, ne, nm);
        } else if (mode == 1) {  // the everyday range: |n| in [2^-60, 2^60], quotients that land near rounding boundaries
            n = make(h3 >> 63, 1023 - 60 + (h2 % 121), h3);
            if ((h2 >> 20) % 3 == 0) n = D * make(h3 >> 63, 1023 - 30 + (h2 % 61), (h3 >> 7) & 0xFFFFFull);  // n = D * (short mantissa): exact quotients
            if ((h2 >> 20) % 3 == 1) n = D * make(h3 >> 63, 1023 - 30 + (h2 % 61), h3 | 1) ;                    // products rounded once
        } else {  // the edges of the guard's range
            const uint64_t edge[6] = {1023 - 902, 1023 - 900, 1023 - 898, 1023 + 598, 1023 + 600, 1023 + 602};
            n = make(h3 >> 63, edge[h2 % 6] + (h2 >> 8) % 2, (h2 >> 16) % 3 == 0 ? 0 : h3);
        }
        bool ok = true;
        const double y = ezpz::jit::recip_of(D, ok);
        const double q = ezpz::jit::div_by(n, D, y, ok);
        const double want = n / D;
        ++pairs;
        if (ok) {
            ++oks;
            if (n == 0.0) ++zeros;
            const bool same = __builtin_bit_cast(uint64_t, q) == __builtin_bit_cast(uint64_t, want) || (q != q && want != want);
            if (!same) ++bad;
        } else {
            ++notok;
        }
    }
    atomicAdd(&counts[0], pairs);
    atomicAdd(&counts[1], oks);
    atomicAdd(&counts[2], bad);
    atomicAdd(&counts[3], zeros);
    atomicAdd(&counts[4], notok);
}

int main() {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 5 * sizeof(unsigned long long)) != hipSuccess) return 2;
    for (int mode = 0; mode < 3; ++mode) {
        (void)hipMemset(d, 0, 5 * sizeof(unsigned long long));
        for (int pass = 0; pass < 8; ++pass) sweep<<<4096, 256>>>(d, 0x657A707Aull * (pass + 1) + mode, mode);
        unsigned long long h[5];
        if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 3;
        std::printf("%d %llu %llu %llu %llu %llu\n", mode, h[0], h[1], h[2], h[3], h[4]);
    }
    (void)hipFree(d);
    return 0;
}
