// One batch of systems of DIFFERENT topologies (include/ezpz_amd.h: ezpz_mixed_*, ezpz_system_solve_batch_mixed).
//
// The reference's callers loop over arbitrary systems, one solve() after the other (ezpz-cli/src/main.rs:96-98,
// ezpz-wasm/src/lib.rs:96); SURVEY.md 8b's last row asks for that loop as ONE call.  Systems are independent, so the batch
// is regrouped by topology: the rows of topology t -- scattered through the caller's ragged batch -- are gathered into one
// contiguous [count_t][n_t] block by a row-copy kernel, solved by that topology's own kernels
// (ezpz_system_solve_batch_device), and scattered back, every topology on its own stream, all of them forked from and
// joined to the caller's stream by events.  A topology whose systems are one contiguous run of the batch is solved in
// place.  Gather and scatter are two more passes over x (16 bytes per variable, HBM-bound): ~3 % of the 1 M mixed batch.
// No numeric work here beyond moving rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "system.hpp"  // (lane_indexed_launch: a topology's systems solved in place)

namespace {

// Rows of `n` doubles between a ragged batch (row j at ragged + offset[j]) and a contiguous block: one wavefront per
// row, lanes along the row (a row of 8 values is one 64-byte request; longer rows are coalesced 512 bytes at a time).
// GATHER: block <- ragged; else ragged <- block.  Statuses ride along on the scatter.
template <bool GATHER>
__global__ void __launch_bounds__(256) rows_kernel(double* ragged, const double* ragged_in, double* block, const uint64_t* offset,
                                                   uint32_t n, uint64_t count, const EzpzStatus* st_block, EzpzStatus* st_out,
                                                   const uint32_t* sys_of) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    // short rows: 64 / rows_per lanes each, several rows per wavefront
    const uint32_t per = n <= 8 ? 8u : n <= 16 ? 16u : n <= 32 ? 32u : 64u;
    const uint32_t rows_per = 64u / per, sub = lane / per, l = lane % per;
    for (uint64_t j0 = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rows_per; j0 < count; j0 += waves * rows_per) {
        const uint64_t j = j0 + sub;
        if (j >= count) continue;
        const uint64_t off = offset[j];
        for (uint32_t k = l; k < n; k += per) {
            if (GATHER)
                block[j * n + k] = ragged_in[off + k];
            else
                ragged[off + k] = block[j * n + k];
        }
        if (!GATHER && l == 0) st_out[sys_of[j]] = st_block[j];
    }
}

struct Group {
    EzpzSystem* sys = nullptr;
    uint32_t n = 0;
    uint64_t count = 0;
    bool contiguous = false;  // its systems are one run of the batch: solved in place
    uint64_t first = 0;       // ... starting at this system
    uint64_t first_off = 0;   // ... whose row starts here
    std::vector<uint32_t> sys_of;    // host copies (host entry point: gather / scatter on the CPU)
    std::vector<uint64_t> offset;
    uint32_t* d_sys_of = nullptr;
    uint64_t* d_offset = nullptr;
    double* d_block = nullptr;
    EzpzStatus* d_status = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
};

}  // namespace

struct EzpzMixedBatch {
    int device = -1;
    size_t batch = 0;
    uint64_t total = 0;  // doubles of x0 / x_out
    std::vector<uint64_t> x_offset;  // batch + 1
    std::vector<Group> groups;       // topologies that have systems in this batch
    hipEvent_t fork = nullptr;
    std::mutex mu;  // one solve at a time (the staging blocks are the handle's)
    // host entry point: device copies of the whole ragged batch
    double* d_x = nullptr;
    EzpzStatus* d_st = nullptr;
    ~EzpzMixedBatch() {
        // (freed with the batch's device current; the caller's own is put back)
        ezpz::DeviceGuard on_device(device >= 0 ? device : 0);
        for (Group& g : groups) {
            if (g.stream) (void)hipStreamSynchronize(g.stream);
            for (void* p : {(void*)g.d_sys_of, (void*)g.d_offset, (void*)g.d_block, (void*)g.d_status})
                if (p) (void)hipFree(p);
            if (g.stream) (void)hipStreamDestroy(g.stream);
            if (g.done) (void)hipEventDestroy(g.done);
        }
        if (fork) (void)hipEventDestroy(fork);
        if (d_x) (void)hipFree(d_x);
        if (d_st) (void)hipFree(d_st);
    }
};

extern "C" {

int ezpz_mixed_create(EzpzSystem* const* handles, size_t n_handles, const uint32_t* topology_of_system, size_t batch,
                      EzpzMixedBatch** out) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if ((batch && (!handles || !n_handles || !topology_of_system)) || batch >= (1ull << 32)) return EZPZ_ERR_INVALID_ARGUMENT;
    // the batch lives where its topologies live: one device for all of them (its streams, events and staging blocks are created
    // there, whatever device the calling thread is on)
    int device = -1;
    for (size_t t = 0; t < n_handles; ++t) {
        if (!handles[t] || handles[t]->device < 0) return EZPZ_ERR_INVALID_ARGUMENT;
        if (device < 0) device = handles[t]->device;
        if (handles[t]->device != device) return EZPZ_ERR_INVALID_ARGUMENT;
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) {
            (void)hipGetLastError();
            return EZPZ_ERR_NO_DEVICE;
        }
    }
    EZPZ_ON_DEVICE(device);
    std::vector<EzpzSystemInfo> info(n_handles);
    for (size_t t = 0; t < n_handles; ++t) {
        if (!handles[t]) return EZPZ_ERR_INVALID_ARGUMENT;
        const int rc = ezpz_system_info(handles[t], &info[t]);
        if (rc != EZPZ_OK) return rc;
    }
    std::unique_ptr<EzpzMixedBatch> m(new EzpzMixedBatch);
    m->device = device;
    m->batch = batch;
    m->x_offset.resize(batch + 1);
    std::vector<uint64_t> count(n_handles, 0);
    uint64_t off = 0;
    for (size_t b = 0; b < batch; ++b) {
        const uint32_t t = topology_of_system[b];
        if (t >= n_handles) return EZPZ_ERR_INVALID_ARGUMENT;
        m->x_offset[b] = off;
        off += info[t].n_vars;
        ++count[t];
    }
    m->x_offset[batch] = off;
    m->total = off;
    std::vector<int> group_of(n_handles, -1);
    for (size_t t = 0; t < n_handles; ++t) {
        if (!count[t]) continue;
        group_of[t] = (int)m->groups.size();
        m->groups.emplace_back();
        Group& g = m->groups.back();
        g.sys = handles[t];
        g.n = (uint32_t)info[t].n_vars;
        g.sys_of.reserve(count[t]);
        g.offset.reserve(count[t]);
    }
    for (size_t b = 0; b < batch; ++b) {
        Group& g = m->groups[(size_t)group_of[topology_of_system[b]]];
        g.sys_of.push_back((uint32_t)b);
        g.offset.push_back(m->x_offset[b]);
    }
    HIP_TRY(hipEventCreateWithFlags(&m->fork, hipEventDisableTiming));
    for (Group& g : m->groups) {
        g.count = g.sys_of.size();
        g.first = g.sys_of.front();
        g.first_off = g.offset.front();
        g.contiguous = (uint64_t)g.sys_of.back() - g.sys_of.front() + 1 == g.count;
        HIP_TRY(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
        // (a scattered group of systems WITHOUT variables still has statuses to scatter: the same staging, an empty block)
        if (g.contiguous) continue;
        HIP_TRY(hipMalloc((void**)&g.d_sys_of, g.count * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&g.d_offset, g.count * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void**)&g.d_block, std::max<size_t>(g.count * g.n, 1) * sizeof(double)));
        HIP_TRY(hipMalloc((void**)&g.d_status, g.count * sizeof(EzpzStatus)));
        HIP_TRY(hipMemcpy(g.d_sys_of, g.sys_of.data(), g.count * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(g.d_offset, g.offset.data(), g.count * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    *out = m.release();
    return EZPZ_OK;
}

void ezpz_mixed_destroy(EzpzMixedBatch* m) { delete m; }

size_t ezpz_mixed_total_values(const EzpzMixedBatch* m) { return m ? (size_t)m->total : 0; }

void ezpz_mixed_offsets(const EzpzMixedBatch* m, uint64_t* x_offset) {
    if (m && x_offset) std::memcpy(x_offset, m->x_offset.data(), m->x_offset.size() * sizeof(uint64_t));
}

int ezpz_mixed_solve_device(EzpzMixedBatch* m, const double* x0_dev, const EzpzConfig* cfg, double* x_out_dev,
                            EzpzStatus* status_dev, void* stream_) {
    if (!m) return EZPZ_ERR_INVALID_ARGUMENT;
    if (m->batch == 0) return EZPZ_OK;
    if (!status_dev || (m->total && (!x0_dev || !x_out_dev))) return EZPZ_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(m->mu);
    EZPZ_ON_DEVICE(m->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    HIP_TRY(hipEventRecord(m->fork, stream));
    int rc = EZPZ_OK;
    for (Group& g : m->groups) {
        HIP_TRY(hipStreamWaitEvent(g.stream, m->fork, 0));
        if (g.contiguous) {
            const int r = ezpz_system_solve_batch_device(g.sys, x0_dev + g.first_off, g.count, cfg, x_out_dev + g.first_off,
                                                         status_dev + g.first, nullptr, nullptr, 0, g.stream);
            if (rc == EZPZ_OK) rc = r;
        } else if (lane_indexed_launch(g.sys, x0_dev, g.d_offset, g.d_sys_of, g.count, cfg, x_out_dev, status_dev, g.stream) == EZPZ_OK) {
            // a small system on its lane-per-system kernel: every lane reads and writes its system's row of the caller's
            // ragged buffers directly (1 M mixed fixtures: 2.7-3.2 -> G solves/s without the two passes over the batch)
        } else {
            const unsigned rows_per = g.n <= 8 ? 8u : g.n <= 16 ? 4u : g.n <= 32 ? 2u : 1u;
            const unsigned grid = (unsigned)std::min<uint64_t>((g.count + 4 * rows_per - 1) / (4 * rows_per), 4096);
            hipLaunchKernelGGL(rows_kernel<true>, dim3(grid), dim3(256), 0, g.stream, nullptr, x0_dev, g.d_block, g.d_offset, g.n,
                               g.count, nullptr, nullptr, nullptr);
            const int r = ezpz_system_solve_batch_device(g.sys, g.d_block, g.count, cfg, g.d_block, g.d_status, nullptr, nullptr, 0,
                                                         g.stream);
            if (rc == EZPZ_OK) rc = r;
            hipLaunchKernelGGL(rows_kernel<false>, dim3(grid), dim3(256), 0, g.stream, x_out_dev, nullptr, g.d_block, g.d_offset, g.n,
                               g.count, g.d_status, status_dev, g.d_sys_of);
            if (hipGetLastError() != hipSuccess && rc == EZPZ_OK) rc = EZPZ_ERR_HIP;
        }
        // (joined whatever happened: the caller's stream must not run ahead of what was enqueued)
        HIP_TRY(hipEventRecord(g.done, g.stream));
        HIP_TRY(hipStreamWaitEvent(stream, g.done, 0));
    }
    return rc;
}

int ezpz_mixed_solve(EzpzMixedBatch* m, const double* x0, const EzpzConfig* cfg, double* x_out, EzpzStatus* status) {
    if (!m) return EZPZ_ERR_INVALID_ARGUMENT;
    if (m->batch == 0) return EZPZ_OK;
    if (!status || (m->total && (!x0 || !x_out))) return EZPZ_ERR_INVALID_ARGUMENT;
    EZPZ_ON_DEVICE(m->device);  // (the copies below run on the batch's device too)
    {
        std::lock_guard<std::mutex> lock(m->mu);
        if (!m->d_x) HIP_TRY(hipMalloc((void**)&m->d_x, std::max<uint64_t>(m->total, 1) * sizeof(double)));
        if (!m->d_st) HIP_TRY(hipMalloc((void**)&m->d_st, m->batch * sizeof(EzpzStatus)));
    }
    HIP_TRY(hipMemcpy(m->d_x, x0, m->total * sizeof(double), hipMemcpyHostToDevice));
    const int rc = ezpz_mixed_solve_device(m, m->d_x, cfg, m->d_x, m->d_st, nullptr);
    HIP_TRY(hipStreamSynchronize(nullptr));
    if (rc != EZPZ_OK) return rc;
    HIP_TRY(hipMemcpy(x_out, m->d_x, m->total * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status, m->d_st, m->batch * sizeof(EzpzStatus), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < m->batch; ++b)
        if (status[b].iterations == EZPZ_ITERATIONS_TEAM_TIMEOUT) return EZPZ_ERR_HIP;
    return EZPZ_OK;
}

int ezpz_system_solve_batch_mixed(EzpzSystem* const* handles, size_t n_handles, const uint32_t* topology_of_system,
                                  const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status) {
    if (batch == 0) return EZPZ_OK;
    EzpzMixedBatch* m = nullptr;
    int rc = ezpz_mixed_create(handles, n_handles, topology_of_system, batch, &m);
    if (rc != EZPZ_OK) return rc;
    rc = ezpz_mixed_solve(m, x0, cfg, x_out, status);
    ezpz_mixed_destroy(m);
    return rc;
}

}  // extern "C"
