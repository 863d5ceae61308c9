// Diagnostic (tools/stress_noise.py): a co-running kernel that keeps some wavefronts and a share of the memory system busy on a stream
// of its own, to shift the timing of kernels whose workgroups exchange values inside a launch (chunks, rendezvous): their results are
// deterministic by design, so a result that changes beside the noise is a race.  Ends when the host says so, or after ~20 s.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libnoise.so tools/noise.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void __launch_bounds__(64) noise_kernel(double* buf, size_t words, const volatile unsigned int* stop, unsigned int stride) {
    const unsigned long long t0 = wall_clock64();  // 100 MHz
    size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) * 8191u % words;
    double acc = 0.0;
    for (unsigned int round = 0;; ++round) {
        for (int k = 0; k < 256; ++k) {
            acc += buf[i];
            buf[i] = acc * 0.5;
            i = (i + stride) % words;
        }
        if ((round & 15u) == 0 && (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) || wall_clock64() - t0 > 2000000000ull)) break;
    }
    if (acc == 1.2345) buf[0] = acc;
}

static hipStream_t g_stream = nullptr;
static double* g_buf = nullptr;
static unsigned int* g_stop = nullptr;

extern "C" int noise_start(int workgroups) {
    const size_t words = (size_t)32 << 20;  // 256 MB
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    if (!g_buf && hipMalloc(&g_buf, words * 8) != hipSuccess) return -2;
    if (!g_stop && hipHostMalloc(&g_stop, 4, hipHostMallocMapped) != hipSuccess) return -3;
    *g_stop = 0;
    hipLaunchKernelGGL(noise_kernel, dim3(workgroups), dim3(64), 0, g_stream, g_buf, words, g_stop, 1048583u);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

extern "C" int noise_stop() {
    if (!g_stop) return 0;
    __atomic_store_n(g_stop, 1u, __ATOMIC_RELEASE);
    return hipStreamSynchronize(g_stream) == hipSuccess ? 0 : -1;
}
