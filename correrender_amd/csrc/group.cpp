// group.cpp -- several GPUs behind ONE caller thread: the crf_group part of the C ABI (include/corrfield.h).
//
// The reference is one process whose calculators are invoked from the render thread with a caller-owned host buffer
// (src/Volume/VolumeData.cpp:1214-1226, 1469-1472).  A renderer that links libcorrfield can therefore not use the
// one-process-per-GPU driver (correrender_amd/distributed.py); this is the same z-slab decomposition inside the library:
//   * the global grid is cut into z-slabs, one per device (the split of distributed.py:slab_bounds -- with x-fastest
//     volumes a slab of every member is one contiguous range and the result slabs concatenate in the caller's buffer);
//   * every device has its own crf_context (members resident in ITS HBM, its own streams) and a persistent worker
//     thread bound to it, so the launches of the N devices are issued concurrently, not one device after the other;
//   * per evaluation there is ONE exchange: the cs values of the reference point live in the slab of one device (the
//     owner).  Default, whenever every pair of devices has peer access (the xGMI-connected MI355X of a node; trivially
//     when an ordinal repeats -- the rehearsal of an N-slab group on fewer GPUs): a DIRECT READ -- every device's
//     reference-side preparation kernel reads the cs values straight out of the owner's member volumes -- no collective,
//     no copy, no event, no rendezvous between the workers.  CRF_GROUP_EXCHANGE=rccl (and the default without peer
//     access): the owner gathers and ncclBroadcast distributes (one persistent single-process communicator from
//     ncclCommInitAll, one rank per worker thread; RCCL refuses two ranks on one device).  CRF_GROUP_EXCHANGE=copy: the
//     staged form (owner gathers, the others hipMemcpyPeerAsync);
//   * crf_group_compute_batch[_device] evaluates MANY reference points per hand-off: the reference vectors of up to 32
//     points travel in one collective (owners fill their rows, one ncclAllReduce(sum)) or are read directly, the
//     reference-side preparations of a block run first into prepared slots and its per-voxel kernels follow back to
//     back; the workers synchronise once per call;
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
#include "crf_pool.h"

namespace {

// ---- the few RCCL entry points the exchange needs (rccl.h: ncclCommInitAll :236, ncclBroadcast :591) -------------
using ncclComm_t = void*;
constexpr int kNcclFloat32 = 7;  // ncclFloat32, rccl.h:466
constexpr int kNcclSum = 0;      // ncclSum, rccl.h (ncclRedOp_t)
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;  // rccl.h: ncclAllReduce
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
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !Broadcast || !AllReduce) {
            error = "librccl lacks ncclCommInitAll / ncclCommDestroy / ncclBroadcast / ncclAllReduce";
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

}  // namespace

// per device slot: the events that order a batch's preparation stream against its main stream
struct SlotEvents {
    hipEvent_t prep_done[2] = {nullptr, nullptr}, block_done[2] = {nullptr, nullptr}, rows_ready = nullptr;
};

struct crf_group {
    int n = 0;
    std::vector<SlotEvents> events;
    std::vector<int> ordinals;
    std::vector<crf_context*> ctx;
    std::vector<float*> d_refvec;        // per device: the cs reference values of the current evaluation
    std::vector<hipEvent_t> ref_ready;   // per device: recorded by the owner after its gather (peer-copy exchange)
    std::vector<ncclComm_t> comms;       // RCCL communicators (empty: peer-copy exchange)
    Rccl rccl;
    std::string exchange = "none";
    std::unique_ptr<crf::SpinPool> workers;
    bool direct = false;                 // peer exchange by direct reads of the owner's members (no copy, no host rendezvous)
    std::vector<float*> d_rows;          // per device: kBatchRows x cs reference rows of a batch (RCCL / staged exchange), lazily
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
        if (g->d_refvec[size_t(r)] || g->d_rows[size_t(r)]) {
            (void)hipSetDevice(g->ordinals[size_t(r)]);
            if (g->d_refvec[size_t(r)]) (void)hipFree(g->d_refvec[size_t(r)]);
            if (g->d_rows[size_t(r)]) (void)hipFree(g->d_rows[size_t(r)]);
            g->d_refvec[size_t(r)] = nullptr;
            g->d_rows[size_t(r)] = nullptr;
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
        (void)hipSetDevice(g->ordinals[size_t(r)]);
        if (g->ref_ready[size_t(r)]) (void)hipEventDestroy(g->ref_ready[size_t(r)]);
        if (size_t(r) < g->events.size()) {
            SlotEvents& ev = g->events[size_t(r)];
            for (hipEvent_t e : {ev.prep_done[0], ev.prep_done[1], ev.block_done[0], ev.block_done[1], ev.rows_ready})
                if (e) (void)hipEventDestroy(e);
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
    g->d_rows.assign(size_t(num_devices), nullptr);
    g->events.assign(size_t(num_devices), SlotEvents());
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
        // the devices of a group share the host: bound each context's copier threads (host-output evaluations)
        g->ctx[size_t(r)]->copy_threads_cap = std::max(2, 16 / num_devices);
        (void)hipSetDevice(device_ordinals[r]);
        SlotEvents& ev = g->events[size_t(r)];
        bool events_ok = true;
        for (hipEvent_t* e : {&ev.prep_done[0], &ev.prep_done[1], &ev.block_done[0], &ev.block_done[1], &ev.rows_ready})
            events_ok = events_ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        if (!events_ok || hipEventCreateWithFlags(&g->ref_ready[size_t(r)], hipEventDisableTiming) != hipSuccess) {
            g_group_create_error = "crf_group_create: hipEventCreate failed";
            g->n = r + 1;
            crf_group_destroy(g);
            return CRF_ERR_DEVICE;
        }
    }
    // The exchange.  CRF_GROUP_EXCHANGE = "peer" | "copy" | "rccl" forces a form; by default the reference vector is READ
    // DIRECTLY out of the owner's member volumes when every pair of devices has peer access (all MI355X of a node do, over
    // xGMI; trivially true when an ordinal repeats) -- no collective, no copy, no rendezvous between the workers --, else
    // RCCL (ncclBroadcast / ncclAllReduce on a single-process communicator) when the ordinals are distinct, else staged
    // peer copies.
    const bool distinct = std::set<int>(g->ordinals.begin(), g->ordinals.end()).size() == size_t(num_devices);
    const char* forced = getenv("CRF_GROUP_EXCHANGE");
    const bool want_rccl = forced && strcmp(forced, "rccl") == 0;
    const bool want_copy = forced && strcmp(forced, "copy") == 0;
    const bool want_peer = forced && strcmp(forced, "peer") == 0;
    bool all_peers = true;
    if (num_devices > 1 && !want_rccl) {
        for (int a = 0; a < num_devices; a++) {
            (void)hipSetDevice(g->ordinals[size_t(a)]);
            for (int b = 0; b < num_devices; b++) {
                if (g->ordinals[size_t(a)] == g->ordinals[size_t(b)]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, g->ordinals[size_t(a)], g->ordinals[size_t(b)]) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(g->ordinals[size_t(b)], 0);
                    if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) all_peers = false;
                } else {
                    all_peers = false;
                }
            }
        }
        (void)hipGetLastError();
    }
    const char* why = forced ? "CRF_GROUP_EXCHANGE" : distinct ? "every pair of devices has peer access" : "a device ordinal repeats: rehearsal";
    if (num_devices == 1 && !want_rccl) {
        g->exchange = "none (one device)";
    } else if (!want_rccl && !want_copy && all_peers) {
        g->direct = true;
        g->exchange = fmt("peer read (direct gather from the owner's members; %s)", why);
    } else if (want_rccl || (distinct && !want_copy && !want_peer)) {
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
        g->exchange = fmt("rccl (ncclBroadcast / ncclAllReduce, single-process communicator; %s)",
                          forced ? "CRF_GROUP_EXCHANGE" : "no peer access between some pair of devices");
    } else {
        g->exchange = fmt("peer copy (staged: owner gathers, the others copy; %s)",
                          forced ? "CRF_GROUP_EXCHANGE" : "no peer access between some pair of devices");
    }
    // one persistent worker per device slot, bound to its device once
    g->workers = std::make_unique<crf::SpinPool>(num_devices, [g](int r) { (void)hipSetDevice(g->ordinals[size_t(r)]); });
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

}  // extern "C"

namespace {

constexpr int kBatchRows = crf::kMaxGatherRows;  // reference vectors exchanged per collective of a batch

// Where the reference vector of one evaluation comes from, resolved once on the caller thread.
struct RefPlan {
    bool exchange = false;    // the reference point's values have to travel (not symmetric, no host vector)
    int owner = -1;           // slot whose slab holds the reference point
    int local_z = 0;          // its z inside that slab
    crf::RefOverride direct;  // direct exchange: the owner's member table + the voxel inside the owner's slab
};

int plan_reference(crf_group* g, const crf_params* p, RefPlan* plan) {
    if (p->prepared_slot != 0) return gfail(g, CRF_ERR_ARGUMENT, "prepared slots are per context, not per group");
    const bool symmetric = (p->flags & CRF_FLAG_SYMMETRIC) != 0;
    plan->exchange = !symmetric && p->reference_values == nullptr;
    if (!plan->exchange) return CRF_OK;
    if (p->ref_x < 0 || p->ref_y < 0 || p->ref_z < 0 || p->ref_x >= g->xs || p->ref_y >= g->ys || p->ref_z >= g->zs)
        return gfail(g, CRF_ERR_ARGUMENT, fmt("reference point (%d,%d,%d) outside the grid %dx%dx%d", p->ref_x, p->ref_y,
                                              p->ref_z, g->xs, g->ys, g->zs));
    for (int r = 0; r < g->n; r++)
        if (p->ref_z >= g->z_begin[size_t(r)] && p->ref_z < g->z_begin[size_t(r)] + g->z_count[size_t(r)]) {
            plan->owner = r;
            plan->local_z = p->ref_z - g->z_begin[size_t(r)];
        }
    if (g->direct) {
        const bool from_secondary = (p->flags & CRF_FLAG_REFERENCE_FROM_SECONDARY) != 0;
        if (int rc = crf::reference_override(g->ctx[size_t(plan->owner)], from_secondary, p->ref_x, p->ref_y, plan->local_z,
                                             &plan->direct))
            return gfail(g, rc, crf_last_error(g->ctx[size_t(plan->owner)]));
    }
    return CRF_OK;
}

// Staged / RCCL exchange of ONE reference vector on slot r's stream; leaves it in g->d_refvec[r].  Every worker calls
// its rank's collective even after a local error: the collective must be matched.
int exchange_one(crf_group* g, int r, const crf_params* p, const RefPlan& plan) {
    crf_context* c = g->ctx[size_t(r)];
    float* mine = g->d_refvec[size_t(r)];
    const bool from_secondary = (p->flags & CRF_FLAG_REFERENCE_FROM_SECONDARY) != 0;
    int rc = CRF_OK;
    if (r == plan.owner) rc = crf::gather_reference_to(c, from_secondary, p->ref_x, p->ref_y, plan.local_z, mine, c->stream);
    if (!g->comms.empty()) {
        const int nrc = g->rccl.Broadcast(mine, mine, size_t(g->cs), kNcclFloat32, plan.owner, g->comms[size_t(r)], c->stream);
        if (nrc != 0 && rc == CRF_OK) {
            c->err = fmt("ncclBroadcast failed: %s", g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "?");
            rc = CRF_ERR_DEVICE;
        }
        return rc;
    }
    if (r == plan.owner && rc == CRF_OK && hipEventRecord(g->ref_ready[size_t(r)], c->stream) != hipSuccess) rc = CRF_ERR_DEVICE;
    g->workers->barrier();  // the owner's event is recorded: the others may wait on it
    if (r != plan.owner) {
        // hipMemcpyPeerAsync is NOT reliably ordered behind kernels launched earlier on the same stream (measured with two
        // slots on one device: in a batch the copy of evaluation i + 1 overtook the preparation kernel of evaluation i,
        // which then read the next vector -- tools/repro_group_batch.py): drain the stream before the copy is issued.
        if (hipStreamSynchronize(c->stream) != hipSuccess && rc == CRF_OK) rc = CRF_ERR_DEVICE;
        if (hipStreamWaitEvent(c->stream, g->ref_ready[size_t(plan.owner)], 0) != hipSuccess ||
            hipMemcpyPeerAsync(mine, g->ordinals[size_t(r)], g->d_refvec[size_t(plan.owner)], g->ordinals[size_t(plan.owner)],
                               sizeof(float) * size_t(g->cs), c->stream) != hipSuccess) {
            c->err = "peer copy of the reference vector failed";
            if (rc == CRF_OK) rc = CRF_ERR_DEVICE;
        }
    }
    // The owner must not overwrite its vector (the next evaluation's gather, possibly within the same batch job) before
    // the others' copies have EXECUTED, not just been enqueued: every copier waits for its copy before the rendezvous.
    // (The staged form is the fallback for devices without peer access; the direct form has no such step.)
    if (r != plan.owner && hipStreamSynchronize(c->stream) != hipSuccess && rc == CRF_OK) rc = CRF_ERR_DEVICE;
    g->workers->barrier();
    return rc;
}

// One evaluation over the whole grid: exchange of the reference vector, then every device evaluates its slab -- into its
// part of the caller's HOST buffer (host_out != null: calculateCpu(t, e, buffer)) or into the caller's per-device DEVICE
// buffers (device_outs[slot] receives the slab of that slot: xs*ys*z_count floats, resident for a device-side consumer).
// `count` evaluations are handed to the workers in ONE job: for count > 1 (crf_group_compute_batch*) the reference
// vectors of up to kBatchRows evaluations travel in one collective, the reference-side preparations of a block run
// first (prepared slots) and its per-voxel kernels follow back to back; the workers synchronise once, at the end.
int group_compute(crf_group* g, const crf_params* params, int count, float* const* host_outs, void* const* device_outs) {
    if (!g || !params || count < 1 || (!host_outs && !device_outs)) return gfail(g, CRF_ERR_ARGUMENT, "null argument");
    if (g->cs <= 0) return gfail(g, CRF_ERR_STATE, "crf_group_set_grid has not been called");
    for (int i = 0; i < count; i++) {
        if (host_outs && !host_outs[i]) return gfail(g, CRF_ERR_ARGUMENT, fmt("host output %d is a null pointer", i));
        if (device_outs)
            for (int r = 0; r < g->n; r++)
                if (!device_outs[size_t(i) * size_t(g->n) + size_t(r)])
                    return gfail(g, CRF_ERR_ARGUMENT, fmt("device output of evaluation %d, slot %d is a null pointer", i, r));
    }
    std::vector<RefPlan> plans(static_cast<size_t>(count));
    for (int i = 0; i < count; i++)
        if (int rc = plan_reference(g, &params[i], &plans[size_t(i)])) return rc;
    const size_t slice = size_t(g->xs) * size_t(g->ys);
    const bool use_rccl = !g->comms.empty();
    const bool single = g->n == 1 && !use_rccl;  // one slab: global == local coordinates, the context does it all itself
    const bool batched = count > 1;
    const char* trace_env = getenv("CRF_GROUP_TRACE");  // development: host-side phase times of slot 0 on stderr
    const bool trace = trace_env && *trace_env == '1';
    const char* sync_env = getenv("CRF_GROUP_SYNC");
    const bool stream_sync = sync_env && strcmp(sync_env, "stream") == 0;
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(); };

    const int status = g->workers->run([&](int r) -> int {
        const double t_start = since();
        crf_context* c = g->ctx[size_t(r)];
        int rc = CRF_OK;
        auto note = [&](int e) {
            if (e != CRF_OK && rc == CRF_OK) rc = e;
        };
        // what slot r hands to its context for evaluation i: parameters, a device vector or a direct-read override
        auto local_params = [&](int i) {
            crf_params local = params[i];
            if (!single) local.flags &= ~CRF_FLAG_REFERENCE_FROM_SECONDARY;  // the vector arrives / is read remotely
            return local;
        };
        if (batched && (use_rccl || (!g->direct && !single)) && !g->d_rows[size_t(r)])
            if (hipMalloc(reinterpret_cast<void**>(&g->d_rows[size_t(r)]), sizeof(float) * size_t(kBatchRows) * size_t(g->cs)) !=
                hipSuccess) {
                c->err = "hipMalloc of the batch's reference rows failed";
                note(CRF_ERR_DEVICE);  // keep going: the collectives below must still be matched (they will fail alike)
            }
        for (int b0 = 0; b0 < count; b0 += kBatchRows) {
            const int bn = std::min(kBatchRows, count - b0);
            // ---- 1. exchange of the block's reference vectors ------------------------------------------------------
            std::vector<const void*> dref(static_cast<size_t>(bn), nullptr);
            std::vector<const crf::RefOverride*> ov(static_cast<size_t>(bn), nullptr);
            if (!single) {
                if (g->direct) {
                    for (int j = 0; j < bn; j++)
                        if (plans[size_t(b0 + j)].exchange) ov[size_t(j)] = &plans[size_t(b0 + j)].direct;
                } else if (!batched) {
                    if (plans[0].exchange) {
                        note(exchange_one(g, r, &params[0], plans[0]));
                        dref[0] = g->d_refvec[size_t(r)];
                    }
                } else if (use_rccl) {
                    // owners fill their rows (zeros elsewhere), one all-reduce(sum) gives every device every row
                    float* rows = g->d_rows[size_t(r)];
                    int32_t xyz[3 * kBatchRows];
                    bool any = false, secondary = false;
                    for (int j = 0; j < bn; j++) {
                        const RefPlan& pl = plans[size_t(b0 + j)];
                        const crf_params& p = params[b0 + j];
                        const bool mine = pl.exchange && pl.owner == r;
                        xyz[3 * j] = p.ref_x;
                        xyz[3 * j + 1] = p.ref_y;
                        xyz[3 * j + 2] = mine ? pl.local_z : -1;
                        any = any || pl.exchange;
                        secondary = secondary || (pl.exchange && (p.flags & CRF_FLAG_REFERENCE_FROM_SECONDARY));
                    }
                    if (any && rows) {
                        if (secondary) {  // rows from the secondary field: one gather per such row, the rest zero-filled first
                            if (hipMemsetAsync(rows, 0, sizeof(float) * size_t(bn) * size_t(g->cs), c->stream) != hipSuccess)
                                note(CRF_ERR_DEVICE);
                            for (int j = 0; j < bn; j++) {
                                const RefPlan& pl = plans[size_t(b0 + j)];
                                if (!pl.exchange || pl.owner != r) continue;
                                const bool sec = (params[b0 + j].flags & CRF_FLAG_REFERENCE_FROM_SECONDARY) != 0;
                                note(crf::gather_reference_to(c, sec, params[b0 + j].ref_x, params[b0 + j].ref_y, pl.local_z,
                                                              rows + size_t(j) * size_t(g->cs), c->stream));
                            }
                        } else {
                            note(crf_gather_reference_rows_device(c, xyz, bn, rows, nullptr));
                        }
                        const int nrc = g->rccl.AllReduce(rows, rows, size_t(bn) * size_t(g->cs), kNcclFloat32, kNcclSum,
                                                          g->comms[size_t(r)], c->stream);
                        if (nrc != 0 && rc == CRF_OK) {
                            c->err = fmt("ncclAllReduce failed: %s", g->rccl.GetErrorString ? g->rccl.GetErrorString(nrc) : "?");
                            rc = CRF_ERR_DEVICE;
                        }
                        for (int j = 0; j < bn; j++)
                            if (plans[size_t(b0 + j)].exchange) dref[size_t(j)] = rows + size_t(j) * size_t(g->cs);
                    }
                }
            }
            const double t_exchanged = since();
            // ---- 2. evaluation -------------------------------------------------------------------------------------
            if (!batched) {
                crf_params local = local_params(0);
                if (rc == CRF_OK) {
                    if (host_outs) {
                        note(crf::compute_to_host(c, &local, dref[0], host_outs[0] + slice * size_t(g->z_begin[size_t(r)]), ov[0]));
                    } else {
                        note(crf::compute_device_ex(c, &local, dref[0], device_outs[r], nullptr, ov[0]));
                    }
                }
            } else if (!g->direct && !single && !use_rccl) {
                // staged peer copies (no direct access between some pair of devices): one exchange per evaluation
                for (int j = 0; j < bn; j++) {
                    const int i = b0 + j;
                    crf_params local = local_params(i);
                    const void* vec = nullptr;
                    if (plans[size_t(i)].exchange) {
                        note(exchange_one(g, r, &params[i], plans[size_t(i)]));
                        vec = g->d_refvec[size_t(r)];
                    }
                    if (rc != CRF_OK) continue;
                    if (host_outs) note(crf::compute_to_host(c, &local, vec, host_outs[i] + slice * size_t(g->z_begin[size_t(r)]), nullptr));
                    else note(crf::compute_device_ex(c, &local, vec, device_outs[size_t(i) * size_t(g->n) + size_t(r)], nullptr, nullptr));
                }
            } else if (host_outs) {
                for (int j = 0; j < bn && rc == CRF_OK; j++) {
                    const int i = b0 + j;
                    crf_params local = local_params(i);
                    note(crf::compute_to_host(c, &local, dref[size_t(j)], host_outs[i] + slice * size_t(g->z_begin[size_t(r)]),
                                              ov[size_t(j)]));
                }
            } else if (rc == CRF_OK) {
                // The reference-side preparations of the block (tiny kernels) run on the context's SECOND stream, one block
                // ahead of the per-voxel kernels, which follow back to back on the main stream: the main stream carries
                // nothing but per-voxel kernels.  Slots alternate between the two halves of the prepared-slot table;
                // events order a half's re-use after the kernels that read it.
                const int parity = (b0 / kBatchRows) % 2;
                const int slot0 = parity * kBatchRows;
                SlotEvents& ev = g->events[size_t(r)];
                hipStream_t aux = nullptr;
                note(crf::second_stream(c, &aux));
                bool ok = rc == CRF_OK;
                if (ok && b0 >= 2 * kBatchRows) ok = hipStreamWaitEvent(aux, ev.block_done[parity], 0) == hipSuccess;
                bool has_rows = false;
                for (int j = 0; j < bn; j++) has_rows = has_rows || dref[size_t(j)] != nullptr;
                if (ok && has_rows)  // rows exchanged on the main stream (RCCL): the preparations read them
                    ok = hipEventRecord(ev.rows_ready, c->stream) == hipSuccess && hipStreamWaitEvent(aux, ev.rows_ready, 0) == hipSuccess;
                for (int j = 0; j < bn && ok && rc == CRF_OK; j++) {
                    crf_params local = local_params(b0 + j);
                    if (local.flags & CRF_FLAG_SYMMETRIC) continue;  // no reference side
                    note(crf::prepare_device_ex(c, &local, dref[size_t(j)], slot0 + j, aux, ov[size_t(j)]));
                }
                if (ok) ok = hipEventRecord(ev.prep_done[parity], aux) == hipSuccess && hipStreamWaitEvent(c->stream, ev.prep_done[parity], 0) == hipSuccess;
                if (!ok) {
                    c->err = "ordering the preparation stream of a batch failed";
                    note(CRF_ERR_DEVICE);
                }
                for (int j = 0; j < bn && rc == CRF_OK; j++) {
                    const int i = b0 + j;
                    crf_params local = local_params(i);
                    if (!(local.flags & CRF_FLAG_SYMMETRIC)) local.prepared_slot = slot0 + j + 1;
                    note(crf_compute_device(c, &local, nullptr, device_outs[size_t(i) * size_t(g->n) + size_t(r)], nullptr));
                }
                if (rc == CRF_OK && hipEventRecord(ev.block_done[parity], c->stream) != hipSuccess) note(CRF_ERR_DEVICE);
            }
            if (trace && r == 0)
                fprintf(stderr, "crf_group slot 0: job started %.0f us after the call, exchange of block %d issued by %.0f us, "
                                "evaluations issued by %.0f us\n", t_start, b0 / kBatchRows, t_exchanged, since());
        }
        // the host-output path returns synchronised; device-resident results: one synchronisation per job, by polling
        // the slot's event (CRF_GROUP_SYNC=stream: hipStreamSynchronize)
        if (!host_outs) {
            const bool ok = stream_sync ? hipStreamSynchronize(c->stream) == hipSuccess
                                        : (hipEventRecord(g->ref_ready[size_t(r)], c->stream) == hipSuccess &&
                                           crf::spin_on_event(g->ref_ready[size_t(r)]) == hipSuccess);
            if (!ok) {
                c->err = "synchronisation failed after the evaluation";
                note(CRF_ERR_DEVICE);
            }
        }
        return rc;
    });
    if (trace) fprintf(stderr, "crf_group: returned to the caller after %.0f us\n", since());
    return collect(g, status, host_outs ? "crf_group_compute" : "crf_group_compute_device");
}

}  // namespace

extern "C" {

int crf_group_compute(crf_group* g, const crf_params* p, float* host_out) {
    if (!host_out) return gfail(g, CRF_ERR_ARGUMENT, "null output");
    return group_compute(g, p, 1, &host_out, nullptr);
}

int crf_group_compute_device(crf_group* g, const crf_params* p, void* const* device_outs) {
    if (!device_outs) return gfail(g, CRF_ERR_ARGUMENT, "null output table");
    return group_compute(g, p, 1, nullptr, device_outs);
}

int crf_group_compute_batch(crf_group* g, const crf_params* params, int count, float* const* host_outs) {
    if (!host_outs) return gfail(g, CRF_ERR_ARGUMENT, "null output table");
    if (count == 0) return CRF_OK;
    return group_compute(g, params, count, host_outs, nullptr);
}

int crf_group_compute_batch_device(crf_group* g, const crf_params* params, int count, void* const* device_outs) {
    if (!device_outs) return gfail(g, CRF_ERR_ARGUMENT, "null output table");
    if (count == 0) return CRF_OK;
    return group_compute(g, params, count, nullptr, device_outs);
}

}  // extern "C"
