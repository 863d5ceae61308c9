// Stress test of what the kernels that span workgroups rest on (ezpz_amd/csrc/jit_kernel.hip.hpp: grid_store / grid_peek, the ring of
// solve_kernel_grid_fast; grid_ops.hip.hpp; front_kernel.hip.hpp): a 16-byte chunk (value, sequence number, flags) written by ONE
// device-coherent 128-bit store (`global_store_dwordx4 ... sc0 sc1`) is observed WHOLE by a 128-bit load of the same kind on
// another compute unit -- never the new sequence number beside an old value.  The architecture does not promise it in so many
// words; this sweeps it.  Test infrastructure: built and run by tests/test_gpu_chunks.py.
//
//   chunk_stress pairs <pairs> <round trips>     ping-pong between workgroups 2p and 2p + 1, eight lanes each on the eight
//                                                chunks of one 128-byte line (as a workgroup publishes its partials): every
//                                                chunk a lane looks at -- the awaited one and every stale one on the way --
//                                                must be consistent with ITS OWN sequence number.
//                                                prints: <exchanges> <chunks looked at> <torn> <timeouts>
//   chunk_stress occupy <workgroups> <lds bytes> <milliseconds>
//                                                a co-tenant: workgroups that hold their places (and `lds bytes` of LDS each) for
//                                                that long; prints "occupying" once they are resident, "done" at the end.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned int chunk_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t z) {
    z = (z ^ 61u) ^ (z >> 16);
    z *= 9u;
    z ^= z >> 4;
    z *= 0x27d4eb2du;
    return z ^ (z >> 15);
}
// what a chunk with sequence number q on line `line`, place `lane` must hold
__device__ __forceinline__ chunk_t make_chunk(uint32_t q, uint32_t line, uint32_t lane) {
    chunk_t c;
    c.x = mix32(q * 2654435761u + line * 97u + lane);
    c.y = ~c.x ^ (q << 7);
    c.z = q;
    c.w = mix32(c.x + 0x9E3779B9u);
    return c;
}
__device__ __forceinline__ bool consistent(const chunk_t& c, uint32_t line, uint32_t lane) {
    if (c.z == 0) return c.x == 0 && c.y == 0 && c.w == 0;  // the zeroed scratch
    const chunk_t want = make_chunk(c.z, line, lane);
    return c.x == want.x && c.y == want.y && c.w == want.w;
}
__device__ __forceinline__ void put(chunk_t* p, chunk_t c) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(c) : "memory"); }
__device__ __forceinline__ chunk_t peek(const chunk_t* p) {
    chunk_t c;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(c) : "v"(p) : "memory");
    return c;
}

// lines[2 p] is written by workgroup 2 p and read by 2 p + 1, lines[2 p + 1] the other way round
__global__ void pingpong(chunk_t* lines, unsigned long long* counts, uint32_t trips) {
    const uint32_t pair = blockIdx.x >> 1, side = blockIdx.x & 1u, lane = threadIdx.x;
    if (lane >= 8) return;
    chunk_t* mine = lines + (size_t)(2 * pair + side) * 8 + lane;
    const chunk_t* theirs = lines + (size_t)(2 * pair + (side ^ 1u)) * 8 + lane;
    const uint32_t my_line = 2 * pair + side, their_line = 2 * pair + (side ^ 1u);
    unsigned long long looked = 0, torn = 0, timeouts = 0;
    for (uint32_t q = 1; q <= trips; ++q) {
        if (side == 0) put(mine, make_chunk(q, my_line, lane));
        for (uint32_t spins = 0;; ++spins) {
            const chunk_t c = peek(theirs);
            ++looked;
            if (!consistent(c, their_line, lane)) ++torn;
            if (c.z == q) break;
            if (spins > (1u << 22)) {
                ++timeouts;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (side == 1) put(mine, make_chunk(q, my_line, lane));
    }
    atomicAdd(&counts[0], (unsigned long long)trips);
    atomicAdd(&counts[1], looked);
    atomicAdd(&counts[2], torn);
    atomicAdd(&counts[3], timeouts);
}

__global__ void occupy(unsigned int* arrived, unsigned long long ticks) {
    extern __shared__ unsigned char held[];
    if (threadIdx.x == 0) {
        held[0] = 1;
        atomicAdd(arrived, 1u);
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    }
    __syncthreads();
}

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 2;                                                              \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    if (argc >= 4 && !std::strcmp(argv[1], "pairs")) {
        const unsigned pairs = (unsigned)std::atoi(argv[2]), trips = (unsigned)std::atoi(argv[3]);
        chunk_t* lines;
        unsigned long long* counts;
        CHECK(hipMalloc(&lines, (size_t)pairs * 2 * 8 * sizeof(chunk_t)));
        CHECK(hipMemset(lines, 0, (size_t)pairs * 2 * 8 * sizeof(chunk_t)));
        CHECK(hipMalloc(&counts, 4 * sizeof(unsigned long long)));
        CHECK(hipMemset(counts, 0, 4 * sizeof(unsigned long long)));
        hipLaunchKernelGGL(pingpong, dim3(2 * pairs), dim3(64), 0, 0, lines, counts, trips);
        CHECK(hipDeviceSynchronize());
        unsigned long long h[4];
        CHECK(hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost));
        std::printf("%llu %llu %llu %llu\n", h[0], h[1], h[2], h[3]);
        return 0;
    }
    if (argc >= 5 && !std::strcmp(argv[1], "occupy")) {
        const unsigned wgs = (unsigned)std::atoi(argv[2]), lds = (unsigned)std::atoi(argv[3]);
        const double ms = std::atof(argv[4]);
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        unsigned int* arrived;
        CHECK(hipHostMalloc((void**)&arrived, sizeof(unsigned int), hipHostMallocMapped));
        *arrived = 0;
        unsigned int* arrived_dev;
        CHECK(hipHostGetDevicePointer((void**)&arrived_dev, arrived, 0));
        hipLaunchKernelGGL(occupy, dim3(wgs), dim3(64), lds, 0, arrived_dev, (unsigned long long)(ms * 1e5));  // 100 MHz
        CHECK(hipGetLastError());
        while (*(volatile unsigned int*)arrived < wgs) {
        }
        std::printf("occupying\n");
        std::fflush(stdout);
        CHECK(hipDeviceSynchronize());
        std::printf("done\n");
        return 0;
    }
    std::fprintf(stderr, "usage: chunk_stress pairs <pairs> <round trips> | occupy <workgroups> <lds bytes> <milliseconds>\n");
    return 1;
}
