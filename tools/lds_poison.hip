// Diagnostic (tools/stress_lds_poison.py): fills the LDS of every CU with a pattern -- AMD devices do not clear the LDS between kernels,
// so the next kernel's workgroups find it -- to show whether any result depends on LDS a kernel reads before writing.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/liblds_poison.so tools/lds_poison.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void __launch_bounds__(1024) poison_kernel(unsigned long long pattern, unsigned int words, unsigned long long* sink) {
    extern __shared__ unsigned long long lds[];
    for (unsigned int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    // (keep the stores: read one word back)
    if (threadIdx.x == 0 && lds[(blockIdx.x * 2654435761u) % words] != pattern) sink[0] = 1;
    // stay a while so that the dispatcher has to spread the workgroups over every CU
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 20000) {}
}

extern "C" int lds_poison(unsigned long long pattern) {
    static unsigned long long* sink = nullptr;
    if (!sink && hipMalloc(&sink, 8) != hipSuccess) return -1;
    int dev = 0, cus = 0, lds = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -2;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev);
    int bytes = 160 * 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        bytes = lds;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(cus * 4), dim3(1024), bytes, nullptr, pattern, (unsigned int)(bytes / 8), sink);
    if (hipGetLastError() != hipSuccess) return -3;
    return hipDeviceSynchronize() == hipSuccess ? bytes : -4;
}

// The control: how much of the LDS the NEXT kernel's workgroups find holding `pattern` (words equal / words looked at, over a launch
// of one 64 KB workgroup per CU x 4).
__global__ void __launch_bounds__(256) peek_kernel(unsigned long long pattern, unsigned int words, unsigned long long* counts) {
    extern __shared__ unsigned long long lds[];
    unsigned long long hit = 0;
    for (unsigned int i = threadIdx.x; i < words; i += blockDim.x) hit += lds[i] == pattern;
    atomicAdd(&counts[0], hit);
    atomicAdd(&counts[1], (unsigned long long)((words - threadIdx.x + blockDim.x - 1) / blockDim.x));
}

extern "C" double lds_peek(unsigned long long pattern) {
    unsigned long long* counts = nullptr;
    if (hipMalloc(&counts, 16) != hipSuccess) return -1.0;
    (void)hipMemset(counts, 0, 16);
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int bytes = 64 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peek_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(peek_kernel, dim3(cus * 4), dim3(256), bytes, nullptr, pattern, (unsigned int)(bytes / 8), counts);
    unsigned long long h[2] = {0, 0};
    (void)hipMemcpy(h, counts, 16, hipMemcpyDeviceToHost);
    (void)hipFree(counts);
    return h[1] ? (double)h[0] / (double)h[1] : -2.0;
}
