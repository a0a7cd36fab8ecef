// group.cpp -- several GPUs behind ONE caller thread: the crf_group part of the C ABI (include/corrfield.h).
//
// The reference is one process whose calculators are invoked from the render thread with a caller-owned host buffer
// (src/Volume/VolumeData.cpp:1214-1226, 1469-1472).  A renderer that links libcorrfield can therefore not use the
// one-process-per-GPU driver (correrender_amd/distributed.py); this is the same z-slab decomposition inside the library:
//   * the global grid is cut into z-slabs, one per device (the split of distributed.py:slab_bounds -- with x-fastest
//     volumes a slab of every member is one contiguous range and the result slabs concatenate in the caller's buffer);
//   * every device has its own crf_context (members resident in ITS HBM, its own streams) and a persistent worker
//     thread bound to it, so the launches of the N devices are issued concurrently, not one device after the other;
//   * per evaluation there is ONE exchange: the device whose slab holds the reference point gathers the cs reference
//     values and they are broadcast -- RCCL (ncclBroadcast over xGMI, one persistent single-process communicator from
//     ncclCommInitAll, one rank per worker thread) when the ordinals are distinct, or a stream-ordered peer copy when a
//     device ordinal repeats (rehearsal of an N-slab group on fewer GPUs: RCCL refuses two ranks on one device) or
//     when CRF_GROUP_EXCHANGE=peer asks for it;
//   * each device then runs the ranged evaluation of api.cpp (crf::compute_to_host) straight into its part of the
//     caller's buffer: the D2H copies of the N devices run concurrently over their own PCIe links.
// librccl is loaded on first use (dlopen): a single-device process never needs it.
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "crf_context.h"

namespace {

// ---- the few RCCL entry points the exchange needs (rccl.h: ncclCommInitAll :236, ncclBroadcast :591) -------------
using ncclComm_t = void*;
constexpr int kNcclFloat32 = 7;  // ncclFloat32, rccl.h:466
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
    bool load() {
        if (handle) return true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // the copy the process already has (e.g. torch's)
        if (!handle)
            for (const char* n : names)
                if ((handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (!handle) {
            error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "");
            return false;
        }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(handle, "ncclBroadcast"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !Broadcast) {
            error = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclBroadcast";
            return false;
        }
        return true;
    }
};

std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

thread_local std::string g_group_create_error;

// One persistent thread per device: run(job) hands the same job to every worker and returns when all are done.
// Hand-off in both directions first SPINS for a short while (evaluations of an interactive session follow each other
// within milliseconds, and a futex wake-up out of an idle state costs 50-150 us each way -- measured 0.28 ms per
// evaluation with plain condition variables) and only then sleeps on a condition variable.
class Workers {
public:
    explicit Workers(int n) : n_(n), status_(size_t(n), 0) {
        for (int r = 0; r < n; r++) threads_.emplace_back([this, r] { loop(r); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }
    // returns the first non-zero status (by rank)
    int run(const std::function<int(int)>& job) {
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &job;
            remaining_.store(n_, std::memory_order_relaxed);
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        if (!spin_until([this] { return remaining_.load(std::memory_order_acquire) == 0; })) {
            std::unique_lock<std::mutex> lk(m_);
            done_cv_.wait(lk, [this] { return remaining_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
        for (int s : status_)
            if (s) return s;
        return 0;
    }
    // rendezvous of all workers inside a job (every worker must call it the same number of times)
    void barrier() {
        std::unique_lock<std::mutex> lk(bm_);
        const unsigned long gen = barrier_gen_;
        if (++arrived_ == n_) {
            arrived_ = 0;
            barrier_gen_++;
            bcv_.notify_all();
        } else {
            bcv_.wait(lk, [&] { return barrier_gen_ != gen; });
        }
    }

private:
    template <class Pred>
    static bool spin_until(Pred done, double seconds = 300e-6) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0;; i++) {
            if (done()) return true;
            if ((i & 63) == 63 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds)
                return false;
            std::this_thread::yield();
        }
    }
    void loop(int r) {
        unsigned long seen = 0;
        for (;;) {
            if (!spin_until([&] { return generation_.load(std::memory_order_acquire) != seen; })) {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen; });
            }
            const std::function<int(int)>* job;
            {
                std::lock_guard<std::mutex> lk(m_);  // pairs with run(): job_ is published under the same lock
                seen = generation_.load(std::memory_order_acquire);
                if (stop_) return;
                job = job_;
            }
            const int s = (*job)(r);
            status_[size_t(r)] = s;
            if (remaining_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> lk(m_);
                done_cv_.notify_all();
            }
        }
    }
    int n_;
    std::vector<std::thread> threads_;
    std::vector<int> status_;
    std::mutex m_, bm_;
    std::condition_variable cv_, done_cv_, bcv_;
    const std::function<int(int)>* job_ = nullptr;
    std::atomic<unsigned long> generation_{0};
    std::atomic<int> remaining_{0};
    unsigned long barrier_gen_ = 0;
    int arrived_ = 0;
    bool stop_ = false;
};

}  // namespace

struct crf_group {
    int n = 0;
    std::vector<int> ordinals;
    std::vector<crf_context*> ctx;
    std::vector<float*> d_refvec;        // per device: the cs reference values of the current evaluation
    std::vector<hipEvent_t> ref_ready;   // per device: recorded by the owner after its gather (peer-copy exchange)
    std::vector<ncclComm_t> comms;       // RCCL communicators (empty: peer-copy exchange)
    Rccl rccl;
    std::string exchange = "none";
    std::unique_ptr<Workers> workers;
    int xs = 0, ys = 0, zs = 0, cs = 0;
    std::vector<int> z_begin, z_count;
    std::string err;
};

namespace {

int gfail(crf_group* g, int code, const std::string& msg) {
    if (g) g->err = msg;
    return code;
}

// first failing rank's status and message -> the group's error
int collect(crf_group* g, int status, const char* where) {
    if (status == 0) return CRF_OK;
    for (int r = 0; r < g->n; r++) {
        const char* m = crf_last_error(g->ctx[size_t(r)]);
        if (m && *m) return gfail(g, status, fmt("%s (device slot %d): %s", where, r, m));
    }
    return gfail(g, status, fmt("%s failed with status %d", where, status));
}

void slab(int zs, int n, int r, int* z0, int* zn) {  // the split of distributed.py:slab_bounds
    const int base = zs / n, rem = zs % n;
    *z0 = r * base + std::min(r, rem);
    *zn = base + (r < rem ? 1 : 0);
}

void release_buffers(crf_group* g) {
    for (int r = 0; r < g->n; r++) {
        if (g->d_refvec[size_t(r)]) {
            (void)hipSetDevice(g->ordinals[size_t(r)]);
            (void)hipFree(g->d_refvec[size_t(r)]);
            g->d_refvec[size_t(r)] = nullptr;
        }
    }
}

}  // namespace

extern "C" {

const char* crf_group_last_error(const crf_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int crf_group_size(const crf_group* g) { return g ? g->n : 0; }

crf_context* crf_group_context(crf_group* g, int slot) {
    return (g && slot >= 0 && slot < g->n) ? g->ctx[size_t(slot)] : nullptr;
}

const char* crf_group_exchange(const crf_group* g) { return g ? g->exchange.c_str() : ""; }

void crf_group_destroy(crf_group* g) {
    if (!g) return;
    g->workers.reset();
    for (ncclComm_t c : g->comms)
        if (c && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(c);
    release_buffers(g);
    for (int r = 0; r < g->n; r++) {
        if (g->ref_ready[size_t(r)]) {
            (void)hipSetDevice(g->ordinals[size_t(r)]);
            (void)hipEventDestroy(g->ref_ready[size_t(r)]);
        }
        crf_destroy(g->ctx[size_t(r)]);
    }
    delete g;
}

int crf_group_create(const int* device_ordinals, int num_devices, crf_group** out_group) {
    if (!out_group) return CRF_ERR_ARGUMENT;
    *out_group = nullptr;
    if (!device_ordinals || num_devices < 1 || num_devices > 64) {
        g_group_create_error = "crf_group_create: need 1..64 device ordinals";
        return CRF_ERR_ARGUMENT;
    }
    auto* g = new crf_group();
    g->n = num_devices;
    g->ordinals.assign(device_ordinals, device_ordinals + num_devices);
    g->ctx.assign(size_t(num_devices), nullptr);
    g->d_refvec.assign(size_t(num_devices), nullptr);
    g->ref_ready.assign(size_t(num_devices), nullptr);
    g->z_begin.assign(size_t(num_devices), 0);
    g->z_count.assign(size_t(num_devices), 0);
    for (int r = 0; r < num_devices; r++) {
        const int rc = crf_create(device_ordinals[r], &g->ctx[size_t(r)]);
        if (rc != CRF_OK) {
            g_group_create_error = fmt("crf_group_create: device ordinal %d: %s", device_ordinals[r], crf_last_error(nullptr));
            g->n = r;  // destroy what exists
            g->ordinals.resize(size_t(r));
            crf_group_destroy(g);
            return rc;
        }
        (void)hipSetDevice(device_ordinals[r]);
        if (hipEventCreateWithFlags(&g->ref_ready[size_t(r)], hipEventDisableTiming) != hipSuccess) {
            g_group_create_error = "crf_group_create: hipEventCreate failed";
            g->n = r + 1;
            crf_group_destroy(g);
            return CRF_ERR_DEVICE;
        }
    }
    // the exchange: RCCL over xGMI when every slot has its own device
    const bool distinct = std::set<int>(g->ordinals.begin(), g->ordinals.end()).size() == size_t(num_devices);
    const char* forced = getenv("CRF_GROUP_EXCHANGE");  // "peer" | "rccl"
    const bool want_peer = forced && strcmp(forced, "peer") == 0;
    if (num_devices == 1 && !(forced && strcmp(forced, "rccl") == 0)) {
        g->exchange = "none (one device)";
    } else if (distinct && !want_peer) {
        if (!g->rccl.load()) {
            g_group_create_error = "crf_group_create: " + g->rccl.error;
            crf_group_destroy(g);
            return CRF_ERR_DEVICE;
        }
        g->comms.assign(size_t(num_devices), nullptr);
        const int rc = g->rccl.CommInitAll(g->comms.data(), num_devices, g->ordinals.data());
        if (rc != 0) {
            g_group_create_error = fmt("crf_group_create: ncclCommInitAll failed: %s",
                                       g->rccl.GetErrorString ? g->rccl.GetErrorString(rc) : "?");
            g->comms.clear();
            crf_group_destroy(g);
            return CRF_ERR_DEVICE;
        }
        g->exchange = "rccl (ncclBroadcast, single-process communicator)";
    } else {
        // peer copies: let every device read its peers' memory directly where the fabric allows it (errors ignored:
        // hipMemcpyPeerAsync falls back to staging)
        for (int a = 0; a < num_devices; a++) {
            (void)hipSetDevice(g->ordinals[size_t(a)]);
            for (int b = 0; b < num_devices; b++) {
                if (g->ordinals[size_t(a)] == g->ordinals[size_t(b)]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g->ordinals[size_t(a)], g->ordinals[size_t(b)]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(g->ordinals[size_t(b)], 0);
            }
        }
        (void)hipGetLastError();
        g->exchange = distinct ? "peer copy (forced)" : "peer copy (a device ordinal repeats: rehearsal)";
    }
    g->workers = std::make_unique<Workers>(num_devices);
    *out_group = g;
    return CRF_OK;
}

int crf_group_set_grid(crf_group* g, int xs, int ys, int zs, int cs) {
    if (!g) return CRF_ERR_ARGUMENT;
    if (xs <= 0 || ys <= 0 || zs <= 0 || cs <= 0)
        return gfail(g, CRF_ERR_ARGUMENT, fmt("invalid grid %dx%dx%d with %d members", xs, ys, zs, cs));
    if (zs < g->n)
        return gfail(g, CRF_ERR_ARGUMENT, fmt("%d devices cannot share a grid of %d z-slices: every device needs one", g->n, zs));
    release_buffers(g);
    g->xs = xs;
    g->ys = ys;
    g->zs = zs;
    g->cs = cs;
    for (int r = 0; r < g->n; r++) slab(zs, g->n, r, &g->z_begin[size_t(r)], &g->z_count[size_t(r)]);
    const int status = g->workers->run([&](int r) -> int {
        if (int rc = crf_set_grid(g->ctx[size_t(r)], xs, ys, g->z_count[size_t(r)], cs)) return rc;
        return hipMalloc(reinterpret_cast<void**>(&g->d_refvec[size_t(r)]), sizeof(float) * size_t(cs)) == hipSuccess
                   ? CRF_OK
                   : CRF_ERR_DEVICE;
    });
    return collect(g, status, "crf_group_set_grid");
}

int crf_group_slab(const crf_group* g, int slot, int* z_begin, int* z_count) {
    if (!g || slot < 0 || slot >= g->n || g->cs <= 0) return CRF_ERR_ARGUMENT;
    if (z_begin) *z_begin = g->z_begin[size_t(slot)];
    if (z_count) *z_count = g->z_count[size_t(slot)];
    return CRF_OK;
}

static int upload_slabs(crf_group* g, const float* const* host_members, bool secondary) {
    if (!g || !host_members) return gfail(g, CRF_ERR_ARGUMENT, "null argument");
    if (g->cs <= 0) return gfail(g, CRF_ERR_STATE, "crf_group_set_grid has not been called");
    for (int c = 0; c < g->cs; c++)
        if (!host_members[c]) return gfail(g, CRF_ERR_ARGUMENT, fmt("member %d is a null pointer", c));
    const size_t slice = size_t(g->xs) * size_t(g->ys);
    const int status = g->workers->run([&](int r) -> int {
        std::vector<const float*> slabs(size_t(g->cs));
        for (int c = 0; c < g->cs; c++) slabs[size_t(c)] = host_members[c] + slice * size_t(g->z_begin[size_t(r)]);
        return secondary ? crf_upload_secondary_members(g->ctx[size_t(r)], slabs.data())
                         : crf_upload_members(g->ctx[size_t(r)], slabs.data());
    });
    return collect(g, status, secondary ? "crf_group_upload_secondary_members" : "crf_group_upload_members");
}

int crf_group_upload_members(crf_group* g, const float* const* host_members) { return upload_slabs(g, host_members, false); }

int crf_group_upload_secondary_members(crf_group* g, const float* const* host_members) {
    return upload_slabs(g, host_members, true);
}

static int group_minmax(crf_group* g, bool secondary, float* out_min, float* out_max) {
    if (!g || !out_min || !out_max) return gfail(g, CRF_ERR_ARGUMENT, "null argument");
    std::vector<float> mn(size_t(g->n)), mx(size_t(g->n));
    const int status = g->workers->run([&](int r) -> int {
        return secondary ? crf_secondary_member_minmax(g->ctx[size_t(r)], &mn[size_t(r)], &mx[size_t(r)])
                         : crf_member_minmax(g->ctx[size_t(r)], &mn[size_t(r)], &mx[size_t(r)]);
    });
    if (int rc = collect(g, status, "crf_group_member_minmax")) return rc;
    *out_min = *std::min_element(mn.begin(), mn.end());
    *out_max = *std::max_element(mx.begin(), mx.end());
    return CRF_OK;
}

int crf_group_member_minmax(crf_group* g, float* out_min, float* out_max) { return group_minmax(g, false, out_min, out_max); }

int crf_group_secondary_member_minmax(crf_group* g, float* out_min, float* out_max) {
    return group_minmax(g, true, out_min, out_max);
}

int crf_group_set_kraskov_noise(crf_group* g, const double* ref_noise, const double* query_noise) {
    if (!g) return CRF_ERR_ARGUMENT;
    const int status = g->workers->run([&](int r) -> int { return crf_set_kraskov_noise(g->ctx[size_t(r)], ref_noise, query_noise); });
    return collect(g, status, "crf_group_set_kraskov_noise");
}

int crf_group_set_profiling(crf_group* g, int enabled) {
    if (!g) return CRF_ERR_ARGUMENT;
    for (crf_context* c : g->ctx) crf_set_profiling(c, enabled);
    return CRF_OK;
}

int crf_group_take_kernel_time(crf_group* g, double* out_ms_max, int* out_launches) {
    if (!g || !out_ms_max || !out_launches) return gfail(g, CRF_ERR_ARGUMENT, "null argument");
    double worst = 0.0;
    int launches = 0;
    for (crf_context* c : g->ctx) {
        double ms = 0.0;
        int n = 0;
        if (int rc = crf_take_kernel_time(c, &ms, &n)) return gfail(g, rc, crf_last_error(c));
        worst = std::max(worst, ms);
        launches = std::max(launches, n);
    }
    *out_ms_max = worst;
    *out_launches = launches;
    return CRF_OK;
}

// One evaluation over the whole grid: exchange of the reference vector, then every device evaluates its slab -- into its
// part of the caller's HOST buffer (host_out != null: calculateCpu(t, e, buffer)) or into the caller's per-device DEVICE
// buffers (device_outs[slot] receives the slab of that slot: xs*ys*z_count floats, resident for a device-side consumer).
static int group_compute(crf_group* g, const crf_params* p, float* host_out, void* const* device_outs) {
    if (!g || !p || (!host_out && !device_outs)) return gfail(g, CRF_ERR_ARGUMENT, "null argument");
    if (g->cs <= 0) return gfail(g, CRF_ERR_STATE, "crf_group_set_grid has not been called");
    if (p->prepared_slot != 0) return gfail(g, CRF_ERR_ARGUMENT, "prepared slots are per context, not per group");
    if (device_outs)
        for (int r = 0; r < g->n; r++)
            if (!device_outs[r]) return gfail(g, CRF_ERR_ARGUMENT, fmt("device output of slot %d is a null pointer", r));
    const bool symmetric = (p->flags & CRF_FLAG_SYMMETRIC) != 0;
    const bool host_vector = p->reference_values != nullptr;
    const bool needs_exchange = !symmetric && !host_vector;
    int owner = -1, local_z = 0;
    if (needs_exchange) {
        if (p->ref_x < 0 || p->ref_y < 0 || p->ref_z < 0 || p->ref_x >= g->xs || p->ref_y >= g->ys || p->ref_z >= g->zs)
            return gfail(g, CRF_ERR_ARGUMENT, fmt("reference point (%d,%d,%d) outside the grid %dx%dx%d", p->ref_x, p->ref_y,
                                                  p->ref_z, g->xs, g->ys, g->zs));
        for (int r = 0; r < g->n; r++)
            if (p->ref_z >= g->z_begin[size_t(r)] && p->ref_z < g->z_begin[size_t(r)] + g->z_count[size_t(r)]) {
                owner = r;
                local_z = p->ref_z - g->z_begin[size_t(r)];
            }
    }
    const size_t slice = size_t(g->xs) * size_t(g->ys);
    const bool from_secondary = (p->flags & CRF_FLAG_REFERENCE_FROM_SECONDARY) != 0;
    const bool use_rccl = !g->comms.empty();
    const bool single = g->n == 1 && !use_rccl;
    const char* trace_env = getenv("CRF_GROUP_TRACE");  // development: host-side phase times of slot 0 on stderr
    const bool trace = trace_env && *trace_env == '1';
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(); };
    const int status = g->workers->run([&](int r) -> int {
        const double t_start = since();
        crf_context* c = g->ctx[size_t(r)];
        if (hipSetDevice(g->ordinals[size_t(r)]) != hipSuccess) return CRF_ERR_DEVICE;
        crf_params local = *p;
        local.flags &= ~CRF_FLAG_REFERENCE_FROM_SECONDARY;  // the reference vector arrives as a device array
        const void* dref = nullptr;
        int rc = CRF_OK;
        if (needs_exchange) {
            float* mine = g->d_refvec[size_t(r)];
            if (r == owner) rc = crf::gather_reference_to(c, from_secondary, p->ref_x, p->ref_y, local_z, mine, c->stream);
            if (use_rccl) {
                // every worker calls its rank's broadcast even after a local error: the collective must be matched
                const int nrc = g->rccl.Broadcast(mine, mine, size_t(g->cs), kNcclFloat32, owner, g->comms[size_t(r)], c->stream);
                if (nrc != 0 && rc == CRF_OK) {
                    c->err = fmt("ncclBroadcast failed: %s", g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "?");
                    rc = CRF_ERR_DEVICE;
                }
            } else if (!single) {
                if (r == owner && rc == CRF_OK && hipEventRecord(g->ref_ready[size_t(r)], c->stream) != hipSuccess)
                    rc = CRF_ERR_DEVICE;
                g->workers->barrier();  // the owner's event is recorded: the others may wait on it
                if (r != owner) {
                    if (hipStreamWaitEvent(c->stream, g->ref_ready[size_t(owner)], 0) != hipSuccess ||
                        hipMemcpyPeerAsync(mine, g->ordinals[size_t(r)], g->d_refvec[size_t(owner)],
                                           g->ordinals[size_t(owner)], sizeof(float) * size_t(g->cs), c->stream) != hipSuccess) {
                        c->err = "peer copy of the reference vector failed";
                        if (rc == CRF_OK) rc = CRF_ERR_DEVICE;
                    }
                }
            }
            dref = mine;
            if (rc != CRF_OK) return rc;
        }
        const double t_exchanged = since();
        int rc2;
        if (host_out) {
            rc2 = crf::compute_to_host(c, &local, dref, host_out + slice * size_t(g->z_begin[size_t(r)]));
        } else {
            rc2 = crf_compute_device(c, &local, dref, device_outs[r], nullptr);
            if (rc2 == CRF_OK && hipStreamSynchronize(c->stream) != hipSuccess) {
                c->err = "hipStreamSynchronize failed after the evaluation";
                rc2 = CRF_ERR_DEVICE;
            }
        }
        if (trace && r == 0)
            fprintf(stderr, "crf_group_compute slot 0: job started %.0f us after the call, exchange issued by %.0f us, done %.0f us\n",
                    t_start, t_exchanged, since());
        return rc2;
    });
    if (trace) fprintf(stderr, "crf_group_compute: returned to the caller after %.0f us\n", since());
    return collect(g, status, host_out ? "crf_group_compute" : "crf_group_compute_device");
}

int crf_group_compute(crf_group* g, const crf_params* p, float* host_out) {
    if (!host_out) return gfail(g, CRF_ERR_ARGUMENT, "null output");
    return group_compute(g, p, host_out, nullptr);
}

int crf_group_compute_device(crf_group* g, const crf_params* p, void* const* device_outs) {
    if (!device_outs) return gfail(g, CRF_ERR_ARGUMENT, "null output table");
    return group_compute(g, p, nullptr, device_outs);
}

}  // extern "C"
