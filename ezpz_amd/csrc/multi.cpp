// One batch over several devices of a node (include/ezpz_amd.h: ezpz_multi_*, ezpz_system_solve_batch_multi).
//
// The reference solves one system per call on one host thread (its caller is a host loop, ezpz-cli/src/main.rs:96-98);
// independent systems have no exchange step, so a batch shards contiguously over the devices and needs
// no collective and no peer copy: device d gets systems [d * ceil(B / G), ...) and moves them over ITS OWN host link,
// straight between the caller's buffers and that device.  One worker thread per device owns that device's analysed
// topology (an EzpzSystem) and runs the ordinary single-device entry point on its shard -- pageable buffers in 16 MB
// pieces, registered buffers through the three-stream pipeline -- so G devices move G shards at once.  Everything here
// is host code on the C ABI (api.hip, launch.hip, pipeline.cpp); there is no numeric work in this file.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <list>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ezpz_amd.h"
#include "program.hpp"

namespace {

struct Job {
    const double* x0 = nullptr;
    size_t batch = 0;
    const EzpzConfig* cfg = nullptr;
    double* x_out = nullptr;
    EzpzStatus* status = nullptr;
    uint8_t* unsat_mask = nullptr;
    int specialize = -1;  // >= 0: ezpz_system_specialize(sys, specialize) instead of a solve
    std::function<int(EzpzSystem*)> fn;  // anything else that must run on the device's own thread (mixed batches)
};

// The worker of a device runs on the CPUs of the device's NUMA node (an 8-GPU node has two sockets: pageable copies are
// staged by the calling thread, and a pipeline's enqueue calls ring the device's doorbells -- both across the socket link
// for half the devices otherwise).  Best effort: /sys/bus/pci/devices/<bdf>/numa_node and that node's cpulist; a missing
// file, node -1 (one socket) or a failing sched_setaffinity leave the thread where the scheduler puts it.
void pin_to_device_node(int device) {
    static const bool enabled = [] {
        const char* e = std::getenv("EZPZ_MULTI_PIN");  // EZPZ_MULTI_PIN=0: no affinity (A/B runs)
        return !(e && e[0] == '0');
    }();
    if (!enabled) return;
    char bdf[32] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), device) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    for (char* c = bdf; *c; ++c)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    char path[128];
    std::snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE* f = std::fopen(path, "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return;
    std::snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE* f = std::fopen(path, "r");
    if (!f) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    int a = 0, b = 0, any = 0;
    for (;;) {  // "0-31,64-95"
        if (std::fscanf(f, "%d", &a) != 1) break;
        b = a;
        int ch = std::fgetc(f);
        if (ch == '-') {
            if (std::fscanf(f, "%d", &b) != 1) break;
            ch = std::fgetc(f);
        }
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET(c, &set), any = 1;
        if (ch != ',') break;
    }
    std::fclose(f);
    if (any) (void)sched_setaffinity(0, sizeof(set), &set);
}

// A device's worker: its thread keeps the device current, owns the EzpzSystem and runs one job at a time.
struct Worker {
    int device = 0;
    EzpzSystem* sys = nullptr;
    std::thread thread;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, done = false, quit = false;
    Job job;
    int rc = EZPZ_OK;

    void run() {
        (void)hipSetDevice(device);
        pin_to_device_node(device);
        std::unique_lock<std::mutex> lock(mu);
        for (;;) {
            cv.wait(lock, [&] { return has_job || quit; });
            if (quit) {
                // the system goes where its device is current: the thread that destroys the handle keeps its own device
                if (sys) ezpz_system_destroy(sys);
                sys = nullptr;
                return;
            }
            const Job j = job;
            lock.unlock();
            int r;
            if (j.fn)
                r = j.fn(sys);
            else if (j.specialize >= 0)
                r = ezpz_system_specialize(sys, j.specialize);
            else
                r = ezpz_system_solve_batch(sys, j.x0, j.batch, j.cfg, j.x_out, j.status, j.unsat_mask, nullptr, 0);
            lock.lock();
            rc = r;
            has_job = false;
            done = true;
            cv.notify_all();
        }
    }
    void post(const Job& j) {
        std::lock_guard<std::mutex> lock(mu);
        job = j;
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lock(mu);
        cv.wait(lock, [&] { return done; });
        done = false;
        return rc;
    }
};

}  // namespace

struct EzpzMultiSystem {
    size_t n_vars = 0, n_cs = 0;
    std::vector<std::unique_ptr<Worker>> workers;
    std::mutex call_mu;  // one batch call at a time per handle
    ~EzpzMultiSystem() {
        for (auto& w : workers) {
            {
                std::lock_guard<std::mutex> lock(w->mu);
                w->quit = true;
                w->cv.notify_all();
            }
            if (w->thread.joinable()) w->thread.join();
            if (w->sys) ezpz_system_destroy(w->sys);  // (a worker whose thread never started: ezpz_system_destroy restores the caller's device)
        }
    }
};

namespace {

// Small LRU behind ezpz_system_solve_batch_multi, keyed by the request bytes and the device mask.
struct MultiEntry {
    uint64_t hash;
    std::vector<unsigned char> key;
    size_t n_vars;
    uint64_t mask;
    std::shared_ptr<EzpzMultiSystem> multi;
};
std::mutex g_multi_mu;
std::list<MultiEntry> g_multi;
constexpr size_t kMultiMax = 4;

}  // namespace

extern "C" {

int ezpz_multi_create(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint64_t device_mask, uint32_t team_size,
                      EzpzMultiSystem** out, int32_t* err_constraint, int64_t* err_variable) {
    if (!out) return EZPZ_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    const int ndev = ezpz_device_count();
    if (ndev < 1) return EZPZ_ERR_NO_DEVICE;
    // EZPZ_MULTI_OVERSUBSCRIBE=1 (tests on a one-GPU box): bit d of the mask means worker d on device d % devices, so the
    // sharded path -- several workers, several EzpzSystems, shard offsets -- runs for real on one device
    static const bool oversubscribe = [] {
        const char* e = std::getenv("EZPZ_MULTI_OVERSUBSCRIBE");
        return e && std::atoi(e) != 0;
    }();
    if (device_mask == 0) device_mask = ndev >= 64 ? ~0ull : (1ull << ndev) - 1;
    std::vector<int> devices;
    for (int d = 0; d < 64; ++d)
        if (device_mask >> d & 1) {
            if (d >= ndev && !oversubscribe) return EZPZ_ERR_INVALID_ARGUMENT;
            devices.push_back(d % ndev);
        }
    std::unique_ptr<EzpzMultiSystem> m(new EzpzMultiSystem);
    m->n_vars = n_vars;
    m->n_cs = n_cs;
    // the symbolic phase once per device, in parallel (each on its own thread: hipSetDevice is per thread)
    std::vector<int> rcs(devices.size(), EZPZ_OK);
    std::vector<int32_t> ecs(devices.size(), -1);
    std::vector<int64_t> evs(devices.size(), -1);
    for (size_t i = 0; i < devices.size(); ++i) {
        m->workers.emplace_back(new Worker);
        m->workers.back()->device = devices[i];
    }
    {
        std::vector<std::thread> builders;
        for (size_t i = 0; i < devices.size(); ++i)
            builders.emplace_back([&, i] {
                rcs[i] = ezpz_system_create(cs, n_cs, n_vars, devices[i], team_size, &m->workers[i]->sys, &ecs[i], &evs[i]);
            });
        for (auto& t : builders) t.join();
    }
    for (size_t i = 0; i < devices.size(); ++i)
        if (rcs[i] != EZPZ_OK) {
            if (err_constraint) *err_constraint = ecs[i];
            if (err_variable) *err_variable = evs[i];
            return rcs[i];
        }
    for (auto& w : m->workers) w->thread = std::thread([p = w.get()] { p->run(); });
    *out = m.release();
    return EZPZ_OK;
}

void ezpz_multi_destroy(EzpzMultiSystem* m) { delete m; }

int ezpz_multi_device_count(const EzpzMultiSystem* m) { return m ? (int)m->workers.size() : 0; }

int ezpz_multi_device(const EzpzMultiSystem* m, int index) {
    return m && index >= 0 && (size_t)index < m->workers.size() ? m->workers[(size_t)index]->device : -1;
}

void ezpz_multi_shard(const EzpzMultiSystem* m, size_t batch, int index, size_t* first, size_t* count) {
    const size_t G = m ? m->workers.size() : 0;
    size_t a = 0, b = 0;
    if (G && index >= 0 && (size_t)index < G) {
        const size_t per = (batch + G - 1) / G;
        a = std::min(batch, per * (size_t)index);
        b = std::min(batch, per * ((size_t)index + 1));
    }
    if (first) *first = a;
    if (count) *count = b - a;
}

int ezpz_multi_specialize(EzpzMultiSystem* m, int wait) {
    if (!m) return EZPZ_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> call(m->call_mu);
    Job j;
    j.specialize = wait != 0;
    for (auto& w : m->workers) w->post(j);
    int least = 2;
    for (auto& w : m->workers) least = std::min(least, w->wait());
    return least;
}

int ezpz_multi_solve_batch(EzpzMultiSystem* m, const double* x0, size_t batch, const EzpzConfig* cfg, double* x_out,
                           EzpzStatus* status, uint8_t* unsat_mask) {
    if (!m || (batch && (!x_out || !status || (m->n_vars && !x0)))) return EZPZ_ERR_INVALID_ARGUMENT;
    if (batch == 0) return EZPZ_OK;
    std::lock_guard<std::mutex> call(m->call_mu);
    const size_t G = m->workers.size(), n = m->n_vars, C = m->n_cs;
    std::vector<size_t> posted;
    for (size_t g = 0; g < G; ++g) {
        size_t first, count;
        ezpz_multi_shard(m, batch, (int)g, &first, &count);
        if (!count) continue;
        Job j;
        j.x0 = x0 ? x0 + first * n : nullptr;
        j.batch = count;
        j.cfg = cfg;
        j.x_out = x_out + first * n;
        j.status = status + first;
        j.unsat_mask = unsat_mask ? unsat_mask + first * C : nullptr;
        m->workers[g]->post(j);
        posted.push_back(g);
    }
    int rc = EZPZ_OK;
    for (size_t g : posted) {  // every shard is waited for, whatever the others returned: the caller's buffers are in use until then
        const int r = m->workers[g]->wait();
        if (rc == EZPZ_OK) rc = r;
    }
    return rc;
}

int ezpz_system_solve_batch_multi(const EzpzConstraint* cs, size_t n_cs, size_t n_vars, uint64_t device_mask, const double* x0,
                                  size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status) {
    if (!cs && n_cs) return EZPZ_ERR_INVALID_ARGUMENT;
    const uint64_t h = ezpz::topology_hash(cs, n_cs, n_vars);
    const size_t bytes = n_cs * sizeof(EzpzConstraint);
    std::shared_ptr<EzpzMultiSystem> multi;
    {
        std::lock_guard<std::mutex> lock(g_multi_mu);
        for (auto it = g_multi.begin(); it != g_multi.end(); ++it)
            if (it->hash == h && it->n_vars == n_vars && it->mask == device_mask && it->key.size() == bytes &&
                std::memcmp(it->key.data(), cs, bytes) == 0) {
                g_multi.splice(g_multi.begin(), g_multi, it);
                multi = g_multi.front().multi;
                break;
            }
    }
    if (!multi) {
        EzpzMultiSystem* raw = nullptr;
        const int rc = ezpz_multi_create(cs, n_cs, n_vars, device_mask, 0, &raw, nullptr, nullptr);
        if (rc != EZPZ_OK) return rc;
        multi.reset(raw, [](EzpzMultiSystem* p) { ezpz_multi_destroy(p); });
        MultiEntry e;
        e.hash = h;
        e.key.assign(reinterpret_cast<const unsigned char*>(cs), reinterpret_cast<const unsigned char*>(cs) + bytes);
        e.n_vars = n_vars;
        e.mask = device_mask;
        e.multi = multi;
        std::lock_guard<std::mutex> lock(g_multi_mu);
        g_multi.push_front(std::move(e));
        while (g_multi.size() > kMultiMax) g_multi.pop_back();
    }
    return ezpz_multi_solve_batch(multi.get(), x0, batch, cfg, x_out, status, nullptr);
}

int ezpz_multi_solve_batch_mixed(EzpzMultiSystem* const* multis, size_t n_multis, const uint32_t* topology_of_system, const double* x0,
                                 size_t batch, const EzpzConfig* cfg, double* x_out, EzpzStatus* status) {
    if (batch == 0) return EZPZ_OK;
    if (!multis || !n_multis || !topology_of_system || !status) return EZPZ_ERR_INVALID_ARGUMENT;
    for (size_t t = 0; t < n_multis; ++t) {
        if (!multis[t] || multis[t]->workers.size() != multis[0]->workers.size()) return EZPZ_ERR_INVALID_ARGUMENT;
        for (size_t g = 0; g < multis[0]->workers.size(); ++g)
            if (multis[t]->workers[g]->device != multis[0]->workers[g]->device) return EZPZ_ERR_INVALID_ARGUMENT;
    }
    // every handle serves one batch call at a time; locked in address order (two mixed calls over the same handles in
    // another order must not deadlock)
    std::vector<EzpzMultiSystem*> order(multis, multis + n_multis);
    std::sort(order.begin(), order.end());
    order.erase(std::unique(order.begin(), order.end()), order.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    for (EzpzMultiSystem* m : order) locks.emplace_back(m->call_mu);
    // row offsets of the ragged batch (system b: n_vars of its topology)
    std::vector<uint64_t> off(batch + 1);
    uint64_t total = 0;
    for (size_t b = 0; b < batch; ++b) {
        if (topology_of_system[b] >= n_multis) return EZPZ_ERR_INVALID_ARGUMENT;
        off[b] = total;
        total += multis[topology_of_system[b]]->n_vars;
    }
    off[batch] = total;
    if (total && (!x0 || !x_out)) return EZPZ_ERR_INVALID_ARGUMENT;
    EzpzMultiSystem* lead = multis[0];
    const size_t G = lead->workers.size();
    std::vector<size_t> posted;
    for (size_t g = 0; g < G; ++g) {
        size_t first, count;
        ezpz_multi_shard(lead, batch, (int)g, &first, &count);
        if (!count) continue;
        Job j;
        j.fn = [=](EzpzSystem*) {  // on device g's worker thread: this device's systems of every topology, this shard of the batch
            std::vector<EzpzSystem*> handles(n_multis);
            for (size_t t = 0; t < n_multis; ++t) handles[t] = multis[t]->workers[g]->sys;
            return ezpz_system_solve_batch_mixed(handles.data(), n_multis, topology_of_system + first, x0 ? x0 + off[first] : nullptr, count,
                                                 cfg, x_out ? x_out + off[first] : nullptr, status + first);
        };
        lead->workers[g]->post(j);
        posted.push_back(g);
    }
    int rc = EZPZ_OK;
    for (size_t g : posted) {
        const int r = lead->workers[g]->wait();
        if (rc == EZPZ_OK) rc = r;
    }
    return rc;
}

void ezpz_multi_cache_clear(void) {
    std::lock_guard<std::mutex> lock(g_multi_mu);
    g_multi.clear();
}

}  // extern "C"
