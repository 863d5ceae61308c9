// Measurement, not product code: what one host -> device -> host round trip costs on this box, by completion method.
// The one-call path of ezpz_solve (solve.cpp / request.cpp) is designed against these numbers (DESIGN.md "One solve() call").
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_floor tools/launch_floor.hip && /tmp/launch_floor [bar]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                 \
    do {                                                                      \
        hipError_t e_ = (x);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e_));             \
            std::exit(1);                                                     \
        }                                                                     \
    } while (0)

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); }

__global__ void k_empty() {}

__global__ void k_flag(volatile uint64_t* flag, uint64_t seq) {
    if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint64_t*>(flag), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// reads `n` doubles from `src` (host-mapped or device), sums, writes them back to dst (host-mapped), then the flag
__global__ void k_copy_flag(const double* src, double* dst, uint32_t n, uint64_t* flag, uint64_t seq) {
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] + 1.0;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// resident kernel: waits for the host's request word, answers, leaves after `idle_ticks` of the 100 MHz clock without a
// request (or when the request word is ~0)
__global__ void k_resident(uint64_t* req, uint64_t* rsp, const double* src, double* dst, uint32_t n, uint64_t idle_ticks) {
    __shared__ uint64_t s_seq;
    uint64_t seen = 0;
    uint64_t last = wall_clock64();
    for (;;) {
        if (threadIdx.x == 0) {
            uint64_t v;
            for (;;) {
                v = __hip_atomic_load(req, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v != seen) break;
                if (wall_clock64() - last > idle_ticks) {
                    v = ~0ull;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            s_seq = v;
        }
        __syncthreads();
        const uint64_t v = s_seq;
        __syncthreads();
        if (v == ~0ull) break;
        seen = v;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i] + 1.0;
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(rsp, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = wall_clock64();
    }
    if (threadIdx.x == 0) __hip_atomic_store(rsp, ~0ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static void spin_until(volatile uint64_t* p, uint64_t want) {
    while (*p != want) __builtin_ia32_pause();
}

int main(int argc, char** argv) {
    const bool try_bar = argc > 1 && std::strcmp(argv[1], "bar") == 0;
    CK(hipSetDevice(0));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    uint64_t* flag;
    CK(hipHostMalloc((void**)&flag, 4096, hipHostMallocMapped));
    std::memset(flag, 0, 4096);
    const int N = 3000;
    // 1. launch + hipStreamSynchronize
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, 1, 64, 0, st);
    CK(hipStreamSynchronize(st));
    {
        auto t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_empty, 1, 64, 0, st);
            (void)hipStreamSynchronize(st);
        }
        std::printf("empty kernel, launch + hipStreamSynchronize       : %6.2f us\n", us_since(t0) / N);
    }
    {
        auto t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_empty, 1, 64, 0, st);
            while (hipStreamQuery(st) == hipErrorNotReady) {}
        }
        std::printf("empty kernel, launch + hipStreamQuery spin         : %6.2f us\n", us_since(t0) / N);
    }
    {
        double call = 0;
        auto t0 = clk::now();
        for (int i = 0; i < N; ++i) {
            auto t1 = clk::now();
            hipLaunchKernelGGL(k_flag, 1, 64, 0, st, flag, (uint64_t)(i + 1));
            call += us_since(t1);
            spin_until(flag, (uint64_t)(i + 1));
        }
        std::printf("flag kernel, launch + spin on host-mapped word     : %6.2f us (the launch call itself %5.2f us)\n", us_since(t0) / N, call / N);
        CK(hipStreamSynchronize(st));
    }
    // the same with the stream never drained by the host: does the queue's backlog of completion signals matter?
    for (uint32_t n : {0u, 8u, 2048u, 16384u}) {
        double *hsrc, *hdst, *dsrc;
        CK(hipHostMalloc((void**)&hsrc, 16384 * 8, hipHostMallocMapped));
        CK(hipHostMalloc((void**)&hdst, 16384 * 8, hipHostMallocMapped));
        CK(hipMalloc((void**)&dsrc, 16384 * 8));
        CK(hipMemset(dsrc, 0, 16384 * 8));
        std::memset(hsrc, 0, 16384 * 8);
        *flag = 0;
        for (int dev = 0; dev < 2; ++dev) {
            auto t0 = clk::now();
            for (int i = 0; i < N; ++i) {
                hipLaunchKernelGGL(k_copy_flag, 1, 256, 0, st, dev ? dsrc : hsrc, hdst, n, flag, (uint64_t)(i + 1));
                spin_until(flag, (uint64_t)(i + 1));
            }
            std::printf("copy kernel, %5u doubles from %-11s -> host, flag: %6.2f us\n", n, dev ? "device" : "host-mapped", us_since(t0) / N);
            CK(hipStreamSynchronize(st));
            *flag = 0;
        }
        // resident kernel
        uint64_t* req = flag + 64;
        uint64_t* rsp = flag + 128;
        *req = 0;
        *rsp = 0;
        hipLaunchKernelGGL(k_resident, 1, 256, 0, st, req, rsp, hsrc, hdst, n, (uint64_t)(100 * 1000 * 20) /* 20 ms */);
        {
            const int M = 20000;
            auto t0 = clk::now();
            for (int i = 0; i < M; ++i) {
                __atomic_store_n(req, (uint64_t)(i + 1), __ATOMIC_RELEASE);
                spin_until(rsp, (uint64_t)(i + 1));
            }
            std::printf("resident kernel, %5u doubles host-mapped both ways   : %6.2f us per round trip\n", n, us_since(t0) / M);
            __atomic_store_n(req, ~0ull, __ATOMIC_RELEASE);
            CK(hipStreamSynchronize(st));
        }
        if (try_bar) {
            // host stores straight into device memory (large BAR): does the mapping exist, and what does it cost?
            double* fg = nullptr;
            if (hipExtMallocWithFlags((void**)&fg, 16384 * 8, hipDeviceMallocFinegrained) == hipSuccess) {
                hipPointerAttribute_t a;
                CK(hipPointerGetAttributes(&a, fg));
                std::printf("fine-grained device memory %p, host pointer %p\n", (void*)fg, a.hostPointer);
                std::fflush(stdout);
                auto t0 = clk::now();
                for (int i = 0; i < 1000; ++i) {
                    for (uint32_t k = 0; k < (n ? n : 8); ++k) fg[k] = (double)i;
                    __atomic_thread_fence(__ATOMIC_SEQ_CST);
                }
                std::printf("host stores of %5u doubles into device memory       : %6.2f us\n", n ? n : 8, us_since(t0) / 1000);
                *flag = 0;
                t0 = clk::now();
                for (int i = 0; i < N; ++i) {
                    for (uint32_t k = 0; k < n; ++k) fg[k] = (double)i;
                    __atomic_thread_fence(__ATOMIC_SEQ_CST);
                    hipLaunchKernelGGL(k_copy_flag, 1, 256, 0, st, fg, hdst, n, flag, (uint64_t)(i + 1));
                    spin_until(flag, (uint64_t)(i + 1));
                }
                std::printf("host stores into device memory + copy kernel + flag   : %6.2f us (result %g)\n", us_since(t0) / N, hdst[0]);
                CK(hipStreamSynchronize(st));
                (void)hipFree(fg);
            } else {
                std::printf("hipExtMallocWithFlags(finegrained) failed\n");
            }
        }
        (void)hipHostFree(hsrc);
        (void)hipHostFree(hdst);
        (void)hipFree(dsrc);
    }
    return 0;
}
