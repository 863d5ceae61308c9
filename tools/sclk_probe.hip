// Measurement: the shader clock a lone wavefront runs at.  One wavefront executes a dependent fma chain; clock64() counts
// shader cycles, wall_clock64() a constant 100 MHz: their ratio is the clock the chain ran at -- (a) in a kernel launched on
// an idle device, (b) while a second kernel keeps every CU busy on another stream.
//   hipcc --offload-arch=gfx950 -O2 -o tools/sclk_probe.bin tools/sclk_probe.hip && tools/sclk_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(double* out, unsigned long long* t, int n) {
    double x = out[0];
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) t[0] = c1 - c0, t[1] = w1 - w0;
}
__global__ void busy(double* out, int n) {
    double x = out[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    double *a, *b;
    unsigned long long *t, h[2];
    hipMalloc(&a, 64 * 8); hipMalloc(&b, 1024 * 1024 * 8); hipMalloc(&t, 16);
    hipMemset(a, 0, 64 * 8); hipMemset(b, 0, 1024 * 1024 * 8);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    for (int n : {2000, 20000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, a, t, n);
        hipStreamSynchronize(s1);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        std::printf("idle device,  %6d dependent fmas: %.2f shader cycles each, %.1f ns each, shader clock %.0f MHz\n", n, (double)h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0) * 1000.0);
    }
    hipLaunchKernelGGL(busy, dim3(4096), dim3(256), 0, s2, b, 4000000);
    for (int n : {2000, 20000, 200000}) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, a, t, n);
        hipStreamSynchronize(s1);
        hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        std::printf("busy device,  %6d dependent fmas: %.2f shader cycles each, %.1f ns each, shader clock %.0f MHz\n", n, (double)h[0] / n, h[1] * 10.0 / n, h[0] / (h[1] * 10.0) * 1000.0);
    }
    hipDeviceSynchronize();
    return 0;
}
