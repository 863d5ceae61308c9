// Measurement: how to move a batch's guesses in and its results out over the host link AT THE SAME TIME.
// The pipelined host-to-host path of ezpz_system_solve_batch (registered caller buffers) is designed against these numbers.
//   hipcc --offload-arch=gfx950 -O2 -o tools/pcie_duplex.bin tools/pcie_duplex.hip && tools/pcie_duplex.bin
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                     \
    do {                                                          \
        hipError_t e_ = (x);                                      \
        if (e_ != hipSuccess) {                                   \
            std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); \
            std::exit(1);                                         \
        }                                                         \
    } while (0)
using clk = std::chrono::steady_clock;
static double secs(clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// grid-stride 16-byte copy; nontemporal on the device side so the staging buffers do not sweep L2
__global__ void __launch_bounds__(256) copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int main() {
    CK(hipSetDevice(0));
    const size_t N = 256u << 20;  // per direction (16 384 systems x 16 KB)
    void *h_in, *h_out, *d_a, *d_b;
    // the caller's buffers: ordinary memory, registered (what ezpz_host_register does)
    h_in = std::aligned_alloc(4096, N);
    h_out = std::aligned_alloc(4096, N);
    std::memset(h_in, 1, N);
    std::memset(h_out, 2, N);
    CK(hipHostRegister(h_in, N, hipHostRegisterPortable));
    CK(hipHostRegister(h_out, N, hipHostRegisterPortable));
    CK(hipMalloc(&d_a, N));
    CK(hipMalloc(&d_b, N));
    CK(hipMemset(d_a, 3, N));
    CK(hipMemset(d_b, 4, N));
    hipStream_t s[8];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto sync_all = [&] {
        for (auto& x : s) CK(hipStreamSynchronize(x));
    };
    auto run = [&](const char* name, auto&& f) {
        f();
        sync_all();
        double best = 1e9;
        for (int r = 0; r < 5; ++r) {
            auto t0 = clk::now();
            f();
            sync_all();
            best = std::min(best, secs(t0));
        }
        std::printf("%-86s %6.1f GB/s each way\n", name, N / best / 1e9);
    };
    const int grids[] = {64, 256, 1024};
    run("SDMA in alone (hipMemcpyAsync H2D, one piece)", [&] { CK(hipMemcpyAsync(d_a, h_in, N, hipMemcpyHostToDevice, s[0])); });
    run("SDMA out alone", [&] { CK(hipMemcpyAsync(h_out, d_b, N, hipMemcpyDeviceToHost, s[0])); });
    for (int g : grids) {
        char nm[128];
        std::snprintf(nm, sizeof nm, "kernel in alone, %d workgroups", g);
        run(nm, [&] { hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[0], (const u32x4*)h_in, (u32x4*)d_a, N / 16); });
        std::snprintf(nm, sizeof nm, "kernel out alone, %d workgroups", g);
        run(nm, [&] { hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[0], (const u32x4*)d_b, (u32x4*)h_out, N / 16); });
    }
    for (size_t piece : {size_t(2) << 20, size_t(8) << 20, size_t(32) << 20}) {
        char nm[160];
        std::snprintf(nm, sizeof nm, "SDMA in + SDMA out, pieces of %zu MB on 2 streams", piece >> 20);
        run(nm, [&] {
            for (size_t o = 0; o < N; o += piece) {
                CK(hipMemcpyAsync((char*)d_a + o, (char*)h_in + o, piece, hipMemcpyHostToDevice, s[0]));
                CK(hipMemcpyAsync((char*)h_out + o, (char*)d_b + o, piece, hipMemcpyDeviceToHost, s[1]));
            }
        });
        std::snprintf(nm, sizeof nm, "SDMA in + SDMA out, pieces of %zu MB on 3 + 3 streams", piece >> 20);
        run(nm, [&] {
            int k = 0;
            for (size_t o = 0; o < N; o += piece, ++k) {
                CK(hipMemcpyAsync((char*)d_a + o, (char*)h_in + o, piece, hipMemcpyHostToDevice, s[k % 3]));
                CK(hipMemcpyAsync((char*)h_out + o, (char*)d_b + o, piece, hipMemcpyDeviceToHost, s[3 + k % 3]));
            }
        });
        for (int g : {64, 256}) {
            std::snprintf(nm, sizeof nm, "kernel in (%d workgroups) + SDMA out, pieces of %zu MB", g, piece >> 20);
            run(nm, [&] {
                for (size_t o = 0; o < N; o += piece) {
                    hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[0], (const u32x4*)((char*)h_in + o), (u32x4*)((char*)d_a + o), piece / 16);
                    CK(hipMemcpyAsync((char*)h_out + o, (char*)d_b + o, piece, hipMemcpyDeviceToHost, s[1]));
                }
            });
            std::snprintf(nm, sizeof nm, "SDMA in + kernel out (%d workgroups), pieces of %zu MB", g, piece >> 20);
            run(nm, [&] {
                for (size_t o = 0; o < N; o += piece) {
                    CK(hipMemcpyAsync((char*)d_a + o, (char*)h_in + o, piece, hipMemcpyHostToDevice, s[0]));
                    hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[1], (const u32x4*)((char*)d_b + o), (u32x4*)((char*)h_out + o), piece / 16);
                }
            });
            std::snprintf(nm, sizeof nm, "kernel in + kernel out (%d workgroups each), pieces of %zu MB", g, piece >> 20);
            run(nm, [&] {
                for (size_t o = 0; o < N; o += piece) {
                    hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[0], (const u32x4*)((char*)h_in + o), (u32x4*)((char*)d_a + o), piece / 16);
                    hipLaunchKernelGGL(copy_kernel, g, 256, 0, s[1], (const u32x4*)((char*)d_b + o), (u32x4*)((char*)h_out + o), piece / 16);
                }
            });
        }
    }
    return 0;
}
