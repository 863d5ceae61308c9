// A/B microbenchmarks for the two design questions of BASELINE.json's north_star ("hard parts"):
//   (a) constraint evaluation: one LANE per constraint inside a system (a system's value vector is one contiguous
//       row; what eval_kernel / the list-walk sweeps do) against one WAVEFRONT per constraint with its 64 lanes on 64
//       systems of the batch and the value vectors stored variable-major (x[var][system]: every read of a variable is
//       one coalesced 512-byte row) -- "one-wavefront-per-constraint evaluation with coalesced variable-vector reads";
//   (b) the dense normal-equations block J^T J of a small system (two_rectangles-sized: 16 x 16 from 16 rows) on the
//       fp64 matrix pipe (v_mfma_f64_16x16x4_f64, one wavefront per system, 4 MFMAs) against plain FMAs on the vector
//       pipe (one lane per entry of the lower triangle, 16-term dot products in row order like the solver's gather).
// Standalone: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I ezpz_amd/csrc tools/ab_microbench.hip -o ab_microbench
// Prints one line per kernel with its time; tools/ab_microbench.py runs it under rocprofv3 and files the summaries.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "constraint_eval.hip.hpp"

using namespace ezpz;

#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));         \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

// ---- (a) ------------------------------------------------------------------------------------------------------
// A: one workgroup per system, lane c evaluates constraint c, c + 256, ... ; x is [system][var].
__global__ void __launch_bounds__(256) eval_lane_per_constraint(const DevCon* cons, int n_cons, const double* x, int n_vars, double* r,
                                                                 double* jv, int n_rows, int zj, int batch) {
    for (int sys = blockIdx.x; sys < batch; sys += gridDim.x) {
        const double* xs = x + (size_t)sys * n_vars;
        for (int ci = threadIdx.x; ci < n_cons; ci += blockDim.x) {
            const DevCon c = cons[ci];
            double r0, r1;
            dev::con_residual<false>(c, xs, r0, r1);
            r[(size_t)sys * n_rows + c.row0] = c.weight * r0;
            if (c.nrows > 1) r[(size_t)sys * n_rows + c.row0 + 1] = c.weight * r1;
            dev::JacWriter<double*> w;
            w.jv = jv + (size_t)sys * zj;
            w.jbase = c.jbase;
            const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
            w.loc[0] = loc[0], w.loc[1] = loc[1], w.loc[2] = loc[2], w.loc[3] = loc[3];
            w.weight = c.weight;
            dev::con_jacobian<false>(c, xs, w);
        }
    }
}
// B: one wavefront per constraint, lane l on system 64 g + l; x, r, jv are variable-major: [index][system].
struct SoA {
    const double* p;
    size_t stride;  // systems
    __device__ double operator[](uint32_t e) const { return p[(size_t)e * stride]; }
};
struct SoAW {
    double* p;
    size_t stride;
    struct Ref {
        double* q;
        __device__ void operator=(double v) const { *q = v; }
        __device__ operator double() const { return *q; }
    };
    __device__ Ref operator[](uint32_t e) const { return Ref{p + (size_t)e * stride}; }
};
__global__ void __launch_bounds__(256) eval_wave_per_constraint(const DevCon* cons, int n_cons, const double* x, double* r, double* jv,
                                                                 int batch) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int groups = batch / 64;
    const int total = n_cons * groups;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int item = wave; item < total; item += nwaves) {
        const int ci = item % n_cons, g = item / n_cons;  // consecutive wavefronts: consecutive constraints of one group of systems
        const DevCon c = cons[ci];                         // wave-uniform: the kind switch is a scalar branch
        const size_t sys = (size_t)g * 64 + lane;
        const SoA xs{x + sys, (size_t)batch};
        double r0, r1;
        dev::con_residual<false>(c, xs, r0, r1);
        r[(size_t)c.row0 * batch + sys] = c.weight * r0;
        if (c.nrows > 1) r[(size_t)(c.row0 + 1) * batch + sys] = c.weight * r1;
        dev::JacWriter<SoAW> w;
        w.jv = SoAW{jv + sys, (size_t)batch};
        w.jbase = c.jbase;
        const uint32_t* loc = reinterpret_cast<const uint32_t*>(c.jloc);
        w.loc[0] = loc[0], w.loc[1] = loc[1], w.loc[2] = loc[2], w.loc[3] = loc[3];
        w.weight = c.weight;
        dev::con_jacobian<false>(c, xs, w);
    }
}

// ---- (b) ------------------------------------------------------------------------------------------------------
// J: [system][16 rows][16 vars] dense fp64, A = J^T J (16 x 16), one wavefront per system.  Each kernel loads its J
// once and forms the product REPS times (the solver holds J on chip: the question is the arithmetic, not HBM).
constexpr int REPS = 32;
// vector pipe: J staged in LDS, one lane per entry (i >= j) of the lower triangle (136 entries: 3 rounds of 64 lanes),
// 16-term dot products in row order -- the shape of the solver's gather over its pair lists.
__global__ void __launch_bounds__(256) jtj_valu(const double* J, double* A, int batch) {
    __shared__ double lds[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int sys = wave; sys < batch; sys += nwaves) {
        const double* Js = J + (size_t)sys * 256;
        for (int e = lane; e < 256; e += 64) lds[w][e] = Js[e];
        __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < 136; e += 64) {
            int i = 0, acc = 0;
            while (acc + i + 1 <= e) acc += ++i;  // row i of the lower triangle holds entries acc .. acc + i
            const int j = e - acc;
            double s = 0.0;
            for (int rep = 0; rep < REPS; ++rep) {
                double t = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) t += lds[w][r * 16 + i] * lds[w][r * 16 + j];
                s += t;
                __builtin_amdgcn_wave_barrier();
            }
            A[(size_t)sys * 256 + i * 16 + j] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
// matrix pipe: D (16 x 16) = sum over 4 row chunks of A_k (16 x 4) * B_k (4 x 16) with A_k = J_k^T, B_k = J_k, i.e. both
// operands of v_mfma_f64_16x16x4_f64 are the same register: a = A[i = lane % 16][k = lane / 16] = J[4 kc + lane / 16][lane % 16]
// = B[k = lane / 16][j = lane % 16].  The four result registers are written out raw ([lane][v]); the host finds the
// (i, j) they belong to.
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) jtj_mfma(const double* J, double* A, int batch) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int sys = wave; sys < batch; sys += nwaves) {
        const double* Js = J + (size_t)sys * 256;
        double v[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) v[kc] = Js[(4 * kc + lane / 16) * 16 + (lane % 16)];
        double4_t s = {0.0, 0.0, 0.0, 0.0};
        for (int rep = 0; rep < REPS; ++rep) {
            double4_t d = {0.0, 0.0, 0.0, 0.0};
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));  // a fresh product every time (no hoisting)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) d = __builtin_amdgcn_mfma_f64_16x16x4f64(v[kc], v[kc], d, 0, 0, 0);
            s += d;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) A[(size_t)sys * 256 + lane * 4 + q] = s[q];
    }
}

template <class F>
static double time_ms(F&& launch, int reps) {
    launch();
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    // ---- (a): the kind-sorted constraint table of gen_big_problem.py 500 (2000 constraints, linear kinds) and a table
    // of 2000 non-linear constraints (distance / points-at-angle / line-tangent-to-circle on a chain of points)
    const int batch = 4096;
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<DevCon> cons;
        int n_vars = 2000, row = 0, slot = 0;
        auto add = [&](int kind, std::vector<uint32_t> ids, double param, int nrows, int nemit, int tag = 0) {
            DevCon c{};
            for (size_t i = 0; i < ids.size(); ++i) c.ids[i] = ids[i];
            c.param = param, c.weight = 1.0, c.row0 = row, c.jbase = slot, c.kind = (uint8_t)kind, c.tag = (uint8_t)tag, c.nrows = (uint8_t)nrows;
            for (int e = 0; e < 16; ++e) c.jloc[e] = (uint8_t)e;
            c.nslots = (uint8_t)nemit;
            row += nrows, slot += nemit;
            cons.push_back(c);
        };
        if (variant == 0) {  // kind-sorted like the kernel's table: 500 Vertical, then 1500 Fixed
            for (uint32_t l = 0; l < 500; ++l) add(EZPZ_VERTICAL, {4 * l, 4 * l + 1, 4 * l + 2, 4 * l + 3}, 0.0, 1, 2);
            for (uint32_t l = 0; l < 500; ++l) add(EZPZ_FIXED, {4 * l}, (double)l, 1, 1), add(EZPZ_FIXED, {4 * l + 1}, 0.0, 1, 1),
                                               add(EZPZ_FIXED, {4 * l + 3}, 4.0, 1, 1);
        } else {
            for (uint32_t p = 0; p + 3 < 1000 && cons.size() < 700; ++p) add(EZPZ_DISTANCE, {2 * p, 2 * p + 1, 2 * p + 2, 2 * p + 3}, 1.5, 1, 4);
            for (uint32_t p = 0; p + 5 < 1000 && cons.size() < 1400; ++p)
                add(EZPZ_POINTS_AT_ANGLE, {2 * p, 2 * p + 1, 2 * p + 2, 2 * p + 3, 2 * p + 4, 2 * p + 5}, 30.0, 2, 12, EZPZ_ANGLE_OTHER_DEG);
            for (uint32_t p = 0; p + 6 < 1000 && cons.size() < 2000; ++p)
                add(EZPZ_LINE_TANGENT_TO_CIRCLE, {2 * p, 2 * p + 1, 2 * p + 2, 2 * p + 3, 2 * p + 4, 2 * p + 5, 2 * p + 6}, 0.0, 1, 7, EZPZ_LINE_LEFT);
        }
        const int n_cons = (int)cons.size(), n_rows = row, zj = slot;
        std::vector<double> x((size_t)batch * n_vars);
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::fmod(i * 0.6180339887, 7.0) - 3.0;
        DevCon* d_cons;
        double *d_x, *d_r, *d_j;
        CHECK(hipMalloc(&d_cons, cons.size() * sizeof(DevCon)));
        CHECK(hipMalloc(&d_x, x.size() * 8));
        CHECK(hipMalloc(&d_r, (size_t)batch * n_rows * 8));
        CHECK(hipMalloc(&d_j, (size_t)batch * zj * 8));
        CHECK(hipMemcpy(d_cons, cons.data(), cons.size() * sizeof(DevCon), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(d_x, x.data(), x.size() * 8, hipMemcpyHostToDevice));
        const double bytes = (double)batch * (8.0 * n_vars + 8.0 * n_rows + 8.0 * zj);  // x in, r and J values out
        const double ta = time_ms([&] { hipLaunchKernelGGL(eval_lane_per_constraint, dim3(batch), dim3(256), 0, 0, d_cons, n_cons, d_x, n_vars, d_r, d_j, n_rows, zj, batch); }, 20);
        const double tb = time_ms([&] { hipLaunchKernelGGL(eval_wave_per_constraint, dim3(8192), dim3(256), 0, 0, d_cons, n_cons, d_x, d_r, d_j, batch); }, 20);
        std::printf("(a) %s table, %d constraints x %d systems: lane-per-constraint %.3f ms (%.0f GB/s), wavefront-per-constraint over 64 systems %.3f ms (%.0f GB/s)\n",
                    variant ? "non-linear" : "massive_parallel_system", n_cons, batch, ta, bytes / ta / 1e6, tb, bytes / tb / 1e6);
        CHECK(hipFree(d_cons));
        CHECK(hipFree(d_x));
        CHECK(hipFree(d_r));
        CHECK(hipFree(d_j));
    }
    // ---- (b)
    {
        const int nb = 1 << 18;
        std::vector<double> J((size_t)nb * 256);
        for (size_t i = 0; i < J.size(); ++i) J[i] = std::fmod(i * 0.7548776662, 2.0) - 1.0;
        double *d_J, *d_A, *d_B;
        CHECK(hipMalloc(&d_J, J.size() * 8));
        CHECK(hipMalloc(&d_A, J.size() * 8));
        CHECK(hipMalloc(&d_B, J.size() * 8));
        CHECK(hipMemset(d_A, 0, J.size() * 8));
        CHECK(hipMemset(d_B, 0, J.size() * 8));
        CHECK(hipMemcpy(d_J, J.data(), J.size() * 8, hipMemcpyHostToDevice));
        const double tv = time_ms([&] { hipLaunchKernelGGL(jtj_valu, dim3(8192), dim3(256), 0, 0, d_J, d_A, nb); }, 20);
        const double tm = time_ms([&] { hipLaunchKernelGGL(jtj_mfma, dim3(8192), dim3(256), 0, 0, d_J, d_B, nb); }, 20);
        std::vector<double> A(J.size()), B(J.size());
        CHECK(hipMemcpy(A.data(), d_A, A.size() * 8, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(B.data(), d_B, B.size() * 8, hipMemcpyDeviceToHost));
        // which (i, j) does result register q of lane l hold?  two candidates; the one that reproduces the row-order sums wins
        double best = 1e300;
        int best_map = -1;
        size_t differ = 0, compared = 0;
        for (int map = 0; map < 2; ++map) {
            double maxdiff = 0.0;
            size_t nd = 0, nc = 0;
            for (int sidx = 0; sidx < 4096; ++sidx)
                for (int l = 0; l < 64; ++l)
                    for (int q = 0; q < 4; ++q) {
                        const int i = map == 0 ? 4 * (l / 16) + q : (l / 16) + 4 * q, j = l % 16;
                        if (j > i) continue;
                        const double a = A[(size_t)sidx * 256 + i * 16 + j], b = B[(size_t)sidx * 256 + l * 4 + q];
                        maxdiff = std::fmax(maxdiff, std::fabs(a - b) / std::fmax(1.0, std::fabs(a)));
                        nd += a != b;
                        ++nc;
                    }
            if (maxdiff < best) best = maxdiff, best_map = map, differ = nd, compared = nc;
        }
        const double flops = (double)nb * REPS * 16 * 16 * 16 * 2;
        std::printf("(b) J^T J of 16 x 16 blocks held on chip, %d systems x %d products: vector FMAs from LDS (lower triangle, row order) %.3f ms "
                    "(%.2f TFLOP/s of useful lower-triangle work), v_mfma_f64_16x16x4_f64 (full matrix) %.3f ms (%.2f TFLOP/s); result layout "
                    "i = %s, j = lane %% 16; entries whose bits differ from the row-order sum: %zu of %zu (max rel diff %.3g)\n",
                    nb, REPS, tv, flops * (136.0 / 256.0) / tv / 1e9, tm, flops / tm / 1e9,
                    best_map == 0 ? "4 (lane / 16) + q" : "lane / 16 + 4 q", differ, compared, best);
    }
    return 0;
}
