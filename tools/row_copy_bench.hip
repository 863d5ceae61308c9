// What bounds the kernels that stream a block system's rows (jit_kernel.hip.hpp: fast_wave): the same bytes moved with the same
// occupancy by (A) their access pattern -- a wavefront reads and writes its 4 KB of a row as eight 8-byte accesses per lane at strides
// of 32 / 16 bytes (a 128-byte line is shared by three or four instructions), the next row's loads in flight while this one's are
// used -- and (B) full lines: four 16-byte accesses per lane, consecutive lanes on consecutive bytes.  Rows of 16 KB (2000 doubles
// rounded up to 2048), 65 536 of them, 1024 persistent workgroups of 4 wavefronts.  usage: row_copy_bench [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v2 __attribute__((ext_vector_type(2)));
typedef int v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_at(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
}
template <int MODE>
__global__ void __launch_bounds__(256, 4) copy_rows(const double* in, double* out, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row_doubles = 2048;
    // pattern A: per lane two "class 0" slots (elements 4 l, 4 l + 2 of a 128-line half) and four "class 1" slots (2 l + 1 style)
    int off[8];
    for (int s = 0; s < 2; ++s) {
        const int line = wave * 128 + s * 64 + lane;
        off[2 * s] = (4 * line) * 8;
        off[2 * s + 1] = (4 * line + 2) * 8;
    }
    for (int s = 0; s < 4; ++s) {
        const int inst = wave * 256 + s * 64 + lane;  // instance i <-> variable 2 i + 1
        off[4 + s] = (2 * inst + 1) * 8;
    }
    double x[8], xn[8];
    int row = blockIdx.x;
    if (row < rows) {
        const __amdgpu_buffer_rsrc_t r = row_at(in + (size_t)row * row_doubles);
        if (MODE == 0)
            for (int k = 0; k < 8; ++k) xn[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off[k], 0, 0));
        else
            for (int k = 0; k < 4; ++k) {
                v4 t = __builtin_amdgcn_raw_buffer_load_b128(r, wave * 4096 + k * 1024 + lane * 16, 0, 0);
                xn[2 * k] = __builtin_bit_cast(double, v2{t.x, t.y});
                xn[2 * k + 1] = __builtin_bit_cast(double, v2{t.z, t.w});
            }
    }
    for (; row < rows; row += gridDim.x) {
        for (int k = 0; k < 8; ++k) x[k] = xn[k];
        const int nxt = row + gridDim.x;
        if (nxt < rows) {
            const __amdgpu_buffer_rsrc_t r = row_at(in + (size_t)nxt * row_doubles);
            if (MODE == 0)
                for (int k = 0; k < 8; ++k) xn[k] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, off[k], 0, 0));
            else
                for (int k = 0; k < 4; ++k) {
                    v4 t = __builtin_amdgcn_raw_buffer_load_b128(r, wave * 4096 + k * 1024 + lane * 16, 0, 0);
                    xn[2 * k] = __builtin_bit_cast(double, v2{t.x, t.y});
                    xn[2 * k + 1] = __builtin_bit_cast(double, v2{t.z, t.w});
                }
        }
        // ~600 dependent-ish flops of "work" per wavefront and row, like the solve
        for (int it = 0; it < 20; ++it)
            for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], 1.0000001, 1e-9);
        const __amdgpu_buffer_rsrc_t w = row_at(out + (size_t)row * row_doubles);
        if (MODE == 0)
            for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2, x[k]), w, off[k], 0, 0);
        else
            for (int k = 0; k < 4; ++k) {
                const v2 a = __builtin_bit_cast(v2, x[2 * k]), b = __builtin_bit_cast(v2, x[2 * k + 1]);
                __builtin_amdgcn_raw_buffer_store_b128(v4{a.x, a.y, b.x, b.y}, w, wave * 4096 + k * 1024 + lane * 16, 0, 0);
            }
    }
}
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 65536;
    double *in, *out;
    hipMalloc(&in, (size_t)rows * 2048 * 8);
    hipMalloc(&out, (size_t)rows * 2048 * 8);
    hipMemset(in, 0, (size_t)rows * 2048 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs : {768, 1024, 2048}) {
            float best = 1e9;
            for (int rep = 0; rep < 12; ++rep) {
                hipEventRecord(e0);
                if (mode == 0)
                    hipLaunchKernelGGL(copy_rows<0>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                else
                    hipLaunchKernelGGL(copy_rows<1>, dim3(wgs), dim3(256), 0, 0, in, out, rows);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("%s accesses, %4d workgroups: %.3f ms per %d rows = %.1f M rows/s, %.2f TB/s read + written\n", mode ? "full-line (16 B per lane, consecutive)" : "strided 8 B (the kernels' pattern)",
                   wgs, best, rows, rows / best / 1e3, 2.0 * rows * 16384 / best / 1e9);
        }
    return 0;
}
