// Measurement: what one fp64 vector instruction costs a lone wavefront on gfx950 -- dependent chain vs independent chains,
// fma vs mul+add, one active lane vs 64 (the one-lane-per-system kernels run ~8 cycles per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS, bool FMA>
__global__ void k(double* out, unsigned long long* t, double a, double b, int lanes) {
    if ((int)threadIdx.x >= lanes) return;
    double v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = a + c + threadIdx.x;
    unsigned long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 512 / CHAINS; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (FMA) v[c] = __builtin_fma(v[c], b, a);
            else { double m = v[c] * b; asm volatile("" : "+v"(m)); v[c] = m + a; }
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += v[c];
    asm volatile("" ::"v"(s));
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) { t[0] = t1 - t0; }
    out[threadIdx.x] = s;
}
__global__ void kdiv(double* out, unsigned long long* t, double a, double b, int lanes) {
    if ((int)threadIdx.x >= lanes) return;
    double v = a + threadIdx.x;
    unsigned long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) v = v / b + a;
    asm volatile("" ::"v"(v));
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    out[threadIdx.x] = v;
}
__global__ void ksqrt(double* out, unsigned long long* t, double a, double b, int lanes) {
    if ((int)threadIdx.x >= lanes) return;
    double v = a + threadIdx.x;
    unsigned long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < 64; ++i) v = sqrt(v) + a;
    asm volatile("" ::"v"(v));
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    out[threadIdx.x] = v;
}
template <class K>
void run(const char* name, K kern, int n_ops, int lanes) {
    double* out; unsigned long long* t; unsigned long long h = 0;
    hipMalloc(&out, 64 * 8); hipMalloc(&t, 8);
    for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(kern, 1, 64, 0, 0, out, t, 1.0000001, 0.9999999, lanes); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); }
    printf("%-44s lanes %2d: %6llu cycles, %5.2f per operation\n", name, lanes, h, (double)h / n_ops);
    hipFree(out); hipFree(t);
}
int main() {
    for (int lanes : {1, 64}) {
        run("fma, 1 dependent chain (512)", k<1, true>, 512, lanes);
        run("fma, 2 chains", k<2, true>, 512, lanes);
        run("fma, 4 chains", k<4, true>, 512, lanes);
        run("fma, 8 chains", k<8, true>, 512, lanes);
        run("mul+add, 1 chain (512 pairs)", k<1, false>, 512, lanes);
        run("mul+add, 4 chains", k<4, false>, 512, lanes);
        run("mul+add, 8 chains", k<8, false>, 512, lanes);
        run("division + add, dependent (64)", kdiv, 64, lanes);
        run("sqrt + add, dependent (64)", ksqrt, 64, lanes);
    }
}
