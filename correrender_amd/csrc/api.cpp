// api.cpp -- the C ABI of libcorrfield.so (include/corrfield.h) over the gfx950 kernels.
//
// Host-side responsibilities, mirroring what CorrelationCalculator::calculateCpu does around its hot loop
// (reference: src/Calculators/CorrelationCalculator.cpp:781-866): hold the member volumes (resident in HBM), obtain
// the reference vector, pick the estimator, launch, hand back xs*ys*zs floats.  No CPU fallback exists: without a
// gfx950 device every entry point fails.
#include "../../include/corrfield.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <functional>
#include <immintrin.h>
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "crf_context.h"
#include "crf_internal.h"

namespace {
thread_local std::string g_create_error;

std::string fmt(const char* f, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}
}  // namespace


namespace {

int fail(crf_context* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define CRF_HIP(ctx, call)                                                                                     \
    do {                                                                                                       \
        hipError_t e_ = (call);                                                                                \
        if (e_ != hipSuccess)                                                                                  \
            return fail(ctx, CRF_ERR_DEVICE, fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                                                 __LINE__));                                                   \
    } while (0)

int bind_device(crf_context* c) {
    CRF_HIP(c, hipSetDevice(c->device));
    return CRF_OK;
}

void release_members(crf_context* c) {
    if (c->owned_block) (void)hipFree(c->owned_block);
    c->owned_block = nullptr;
    c->members.clear();
    c->minmax_valid = false;
}

void release_secondary(crf_context* c) {
    if (c->sec_owned_block) (void)hipFree(c->sec_owned_block);
    c->sec_owned_block = nullptr;
    c->sec_members.clear();
    c->sec_minmax_valid = false;
}

int alignment_vpt(const void* p) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return (a & 15u) == 0 ? 4 : ((a & 7u) == 0 ? 2 : 1);
}

int install_member_table(crf_context* c) {
    int vpt = 4;
    for (const float* p : c->members) vpt = std::min(vpt, alignment_vpt(p));
    c->max_vpt = vpt;
    CRF_HIP(c, hipMemcpyAsync(c->d_member_table, c->members.data(), sizeof(float*) * size_t(c->cs),
                              hipMemcpyHostToDevice, c->stream));
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    c->minmax_valid = false;
    c->host_chunks = 0;  // the per-range pointer tables of the host-output path describe the old members
    c->windows = 0;      // ... and so do the window tables of a >= 4 GiB grid
    return CRF_OK;
}

// Host-built fp64 tables that depend on the member count only.
//   psi(n) at integers: the reference calls boost::math::digamma on positive integers only
//     (MutualInformation.cpp:235,237,438,439,503,504) = -gamma + H_{n-1}; accumulated in long double.
//   T[c] = p ln p, p = c/cs: the value computeMutualInformationBinned adds for a bin holding c of cs samples
//     (MutualInformation.cpp:120-140), evaluated with the host libm log like the reference.
//   noise: the 1e-10 tie-breaking jitter of the Kraskov estimator (MutualInformation.cpp:409-420).  sgl's
//     XorshiftRandomGenerator is not available (un-vendored, unpinned); this is the repo's documented stream:
//     Marsaglia xorshift32 (13,17,5), seeds 617406168 / 864730169, u = (state >> 8) * 2^-24.
std::vector<double> build_tables(int cs) {
    std::vector<double> t(size_t(4 * cs + 2));
    const long double gamma = 0.577215664901532860606512090082402431L;
    long double h = 0.0L;
    t[0] = std::numeric_limits<double>::quiet_NaN();
    for (int n = 1; n <= cs; n++) {
        t[size_t(n)] = double(h - gamma);
        h += 1.0L / (long double)n;
    }
    double* T = t.data() + (cs + 1);
    T[0] = 0.0;
    for (int c = 1; c <= cs; c++) {
        const double p = double(c) / double(cs);
        T[c] = p * std::log(p);
    }
    const uint32_t seeds[2] = {617406168u, 864730169u};
    for (int w = 0; w < 2; w++) {
        uint32_t s = seeds[w];
        double* dst = t.data() + 2 * (cs + 1) + w * cs;
        for (int e = 0; e < cs; e++) {
            s ^= s << 13;
            s ^= s >> 17;
            s ^= s << 5;
            dst[e] = double(float(s >> 8) * (1.0f / 16777216.0f)) * 1e-10;
        }
    }
    return t;
}

// psi(n) = -gamma + H_{n-1} at a positive integer, the same long-double accumulation as build_tables()
double psi_int(int n) {
    long double h = 0.0L;
    for (int i = 1; i < n; i++) h += 1.0L / (long double)i;
    return double(h - 0.577215664901532860606512090082402431L);
}
// the k-dependent constant of the KSG estimators: psi(k) (KSG-1, MutualInformation.cpp:438) or psi(k) - 1/k (KSG-2, :503)
double kraskov_c_term(int k, int estimator) {
    double c = psi_int(k);
    if (estimator != 1) c -= 1.0 / double(k);
    return c;
}

int check_ready(crf_context* c) {
    if (!c) return CRF_ERR_ARGUMENT;
    if (c->cs <= 0 || c->num_voxels == 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    if (int(c->members.size()) != c->cs) return fail(c, CRF_ERR_STATE, "no member volumes uploaded or bound");
    return CRF_OK;
}

hipEvent_t take_event(crf_context* c) {
    if (!c->ev_free.empty()) {
        hipEvent_t e = c->ev_free.back();
        c->ev_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

}  // namespace

extern "C" {

int crf_abi_version(void) { return 5; }

const char* crf_last_error(const crf_context* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int crf_create(int device_ordinal, crf_context** out_ctx) {
    if (!out_ctx) return CRF_ERR_ARGUMENT;
    *out_ctx = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = fmt("no HIP device available (%s); libcorrfield has no CPU fallback",
                             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return CRF_ERR_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= count) {
        g_create_error = fmt("device ordinal %d out of range [0,%d)", device_ordinal, count);
        return CRF_ERR_ARGUMENT;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_ordinal);
    if (e != hipSuccess) {
        g_create_error = fmt("hipGetDeviceProperties: %s", hipGetErrorString(e));
        return CRF_ERR_DEVICE;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = fmt("device %d is %s; libcorrfield is built for gfx950 (MI355X) only", device_ordinal,
                             prop.gcnArchName);
        return CRF_ERR_DEVICE;
    }
    auto* c = new crf_context();
    c->device = device_ordinal;
    auto bail = [&](const char* what, hipError_t err) {
        g_create_error = fmt("%s: %s", what, hipGetErrorString(err));
        crf_destroy(c);
        return CRF_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    if ((e = hipMalloc(&c->d_prep, crf::kPrepBytes)) != hipSuccess) return bail("hipMalloc(prep)", e);
    if ((e = hipMalloc(&c->d_minmax, 2 * sizeof(uint32_t))) != hipSuccess) return bail("hipMalloc(minmax)", e);
    *out_ctx = c;
    return CRF_OK;
}

void crf_destroy(crf_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    release_members(c);
    release_secondary(c);
    if (c->d_member_table) (void)hipFree(c->d_member_table);
    if (c->d_sec_table) (void)hipFree(c->d_sec_table);
    if (c->d_ref) (void)hipFree(c->d_ref);
    if (c->d_prep) (void)hipFree(c->d_prep);
    if (c->d_prep_slots) (void)hipFree(c->d_prep_slots);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_todo) (void)hipFree(c->d_todo);
    if (c->d_workspace) (void)hipFree(c->d_workspace);
    if (c->d_requests) (void)hipFree(c->d_requests);
    if (c->d_request_out) (void)hipFree(c->d_request_out);
    if (c->d_minmax) (void)hipFree(c->d_minmax);
    if (c->d_chunk_tables) (void)hipFree(c->d_chunk_tables);
    c->copy_pool.reset();
    if (c->h_staging) (void)hipHostFree(c->h_staging);
    if (c->prep_done) (void)hipEventDestroy(c->prep_done);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    for (hipEvent_t e : c->chunk_done)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->chunk_copied)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (auto& p : c->ev_pending) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : c->ev_free) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int crf_set_grid(crf_context* c, int xs, int ys, int zs, int cs) {
    if (!c) return CRF_ERR_ARGUMENT;
    if (xs <= 0 || ys <= 0 || zs <= 0 || cs <= 0)
        return fail(c, CRF_ERR_ARGUMENT, fmt("invalid grid %dx%dx%d with %d members", xs, ys, zs, cs));
    const size_t n = size_t(xs) * size_t(ys) * size_t(zs);
    if (int r = bind_device(c)) return r;
    CRF_HIP(c, hipDeviceSynchronize());  // evaluations the caller left in flight on its own streams still read the scratch
    release_members(c);
    release_secondary(c);
    if (c->d_member_table) (void)hipFree(c->d_member_table);
    if (c->d_sec_table) (void)hipFree(c->d_sec_table);
    c->d_sec_table = nullptr;
    if (c->d_ref) (void)hipFree(c->d_ref);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_todo) (void)hipFree(c->d_todo);
    if (c->d_workspace) (void)hipFree(c->d_workspace);
    if (c->d_chunk_tables) (void)hipFree(c->d_chunk_tables);
    if (c->h_staging) (void)hipHostFree(c->h_staging);
    c->h_staging = nullptr;
    c->d_staging = nullptr;
    c->d_chunk_tables = nullptr;
    c->host_chunks = 0;
    c->d_todo = nullptr;
    c->d_workspace = nullptr;
    c->workspace_bytes = 0;
    c->d_tables = nullptr;
    c->d_member_table = nullptr;
    c->d_ref = nullptr;
    c->d_out = nullptr;
    c->xs = xs;
    c->ys = ys;
    c->zs = zs;
    c->cs = cs;
    c->num_voxels = n;
    c->alloc_voxels = n;
    // The kernels address a member with 32-bit byte offsets, and their out-of-range sentinel offset (crf_device.h
    // kOutOfRangeOffset = 0xFFFFFFF0) must lie beyond the end of every member: a member volume of 4 GiB or more (1024^3 is
    // exactly 4 GiB; the reference has no limit) is evaluated in WINDOWS of kWindowVoxels voxels, one launch each, through
    // member-pointer tables advanced by the window's first voxel (ensure_windows below).
    c->windowed = n * sizeof(float) >= size_t(0xFFFFFFF0u);
    if (c->d_window_tables) (void)hipFree(c->d_window_tables);
    c->d_window_tables = nullptr;
    c->windows = 0;
    CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_member_table), sizeof(float*) * size_t(cs)));
    CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_ref), sizeof(float) * size_t(cs)));
    const std::vector<double> tables = build_tables(cs);
    CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_tables), tables.size() * sizeof(double)));
    CRF_HIP(c, hipMemcpy(c->d_tables, tables.data(), tables.size() * sizeof(double), hipMemcpyHostToDevice));
    return CRF_OK;
}

int crf_set_kraskov_noise(crf_context* c, const double* ref_noise, const double* query_noise) {
    if (!c) return CRF_ERR_ARGUMENT;
    if (c->cs <= 0 || !c->d_tables) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    if ((ref_noise == nullptr) != (query_noise == nullptr))
        return fail(c, CRF_ERR_ARGUMENT, "crf_set_kraskov_noise: give both tables, or NULL for both (default stream)");
    if (int r = bind_device(c)) return r;
    const size_t cs = size_t(c->cs);
    std::vector<double> both(2 * cs);
    if (ref_noise) {
        for (size_t e = 0; e < cs; e++) {
            // the estimator's contract: a jitter far below the data's resolution, never negative
            if (!(ref_noise[e] >= 0.0 && ref_noise[e] < 1e-9) || !(query_noise[e] >= 0.0 && query_noise[e] < 1e-9))
                return fail(c, CRF_ERR_ARGUMENT, fmt("crf_set_kraskov_noise: entry %zu outside [0, 1e-9)", e));
            both[e] = ref_noise[e];
            both[cs + e] = query_noise[e];
        }
    } else {
        const std::vector<double> t = build_tables(c->cs);
        std::copy(t.begin() + 2 * (c->cs + 1), t.end(), both.begin());
    }
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    CRF_HIP(c, hipMemcpy(c->d_tables + 2 * (cs + 1), both.data(), both.size() * sizeof(double), hipMemcpyHostToDevice));
    return CRF_OK;
}

int crf_upload_members(crf_context* c, const float* const* host_members) {
    if (!c || !host_members) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (c->cs <= 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    for (int i = 0; i < c->cs; i++)
        if (!host_members[i]) return fail(c, CRF_ERR_ARGUMENT, fmt("member %d is a null pointer", i));
    if (int r = bind_device(c)) return r;
    release_members(c);
    // Member stride: volume size rounded up to 256 B so every member starts 256-B aligned (wide vector loads).
    const size_t stride = (c->num_voxels + 63) & ~size_t(63);
    c->owned_stride = stride;
    CRF_HIP(c, hipMalloc(&c->owned_block, stride * sizeof(float) * size_t(c->cs)));
    c->members.resize(size_t(c->cs));
    for (int i = 0; i < c->cs; i++) {
        float* dst = static_cast<float*>(c->owned_block) + stride * size_t(i);
        c->members[size_t(i)] = dst;
        CRF_HIP(c, hipMemcpyAsync(dst, host_members[i], c->num_voxels * sizeof(float), hipMemcpyHostToDevice,
                                  c->stream));
    }
    return install_member_table(c);
}

int crf_bind_members_device(crf_context* c, const void* const* device_members) {
    if (!c || !device_members) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (c->cs <= 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    for (int i = 0; i < c->cs; i++)
        if (!device_members[i]) return fail(c, CRF_ERR_ARGUMENT, fmt("member %d is a null pointer", i));
    if (int r = bind_device(c)) return r;
    release_members(c);
    c->members.resize(size_t(c->cs));
    for (int i = 0; i < c->cs; i++) c->members[size_t(i)] = static_cast<const float*>(device_members[i]);
    return install_member_table(c);
}

int crf_member_minmax(crf_context* c, float* out_min, float* out_max) {
    if (int r = check_ready(c)) return r;
    if (!out_min || !out_max) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (!c->minmax_valid) {
        if (int r = bind_device(c)) return r;
        CRF_HIP(c, crf::launch_minmax(c->d_member_table, c->cs, c->num_voxels, c->d_minmax, c->stream));
        uint32_t keys[2];
        CRF_HIP(c, hipMemcpyAsync(keys, c->d_minmax, sizeof keys, hipMemcpyDeviceToHost, c->stream));
        CRF_HIP(c, hipStreamSynchronize(c->stream));
        c->min_v = crf::minmax_key_to_float(keys[0]);
        c->max_v = crf::minmax_key_to_float(keys[1]);
        c->minmax_valid = true;
    }
    *out_min = c->min_v;
    *out_max = c->max_v;
    return CRF_OK;
}

int crf_member_minmax_divergent(crf_context* c, int secondary, float* out_min, float* out_max) {
    float mn = 0.f, mx = 0.f;
    if (int r = secondary ? crf_secondary_member_minmax(c, &mn, &mx) : crf_member_minmax(c, &mn, &mx)) return r;
    const float max_abs = std::max(std::abs(mn), std::abs(mx));  // VolumeData.cpp:1662-1666
    *out_min = -max_abs;
    *out_max = max_abs;
    return CRF_OK;
}

static int install_secondary_table(crf_context* c) {
    if (!c->d_sec_table)
        CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_sec_table), sizeof(float*) * size_t(c->cs)));
    CRF_HIP(c, hipMemcpyAsync(c->d_sec_table, c->sec_members.data(), sizeof(float*) * size_t(c->cs),
                              hipMemcpyHostToDevice, c->stream));
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    c->sec_minmax_valid = false;
    c->windows = 0;
    return CRF_OK;
}

int crf_upload_secondary_members(crf_context* c, const float* const* host_members) {
    if (!c || !host_members) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (c->cs <= 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    for (int i = 0; i < c->cs; i++)
        if (!host_members[i]) return fail(c, CRF_ERR_ARGUMENT, fmt("secondary member %d is a null pointer", i));
    if (int r = bind_device(c)) return r;
    release_secondary(c);
    const size_t stride = (c->num_voxels + 63) & ~size_t(63);
    CRF_HIP(c, hipMalloc(&c->sec_owned_block, stride * sizeof(float) * size_t(c->cs)));
    c->sec_members.resize(size_t(c->cs));
    for (int i = 0; i < c->cs; i++) {
        float* dst = static_cast<float*>(c->sec_owned_block) + stride * size_t(i);
        c->sec_members[size_t(i)] = dst;
        CRF_HIP(c, hipMemcpyAsync(dst, host_members[i], c->num_voxels * sizeof(float), hipMemcpyHostToDevice,
                                  c->stream));
    }
    return install_secondary_table(c);
}

int crf_bind_secondary_members_device(crf_context* c, const void* const* device_members) {
    if (!c || !device_members) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (c->cs <= 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    for (int i = 0; i < c->cs; i++)
        if (!device_members[i]) return fail(c, CRF_ERR_ARGUMENT, fmt("secondary member %d is a null pointer", i));
    if (int r = bind_device(c)) return r;
    release_secondary(c);
    c->sec_members.resize(size_t(c->cs));
    for (int i = 0; i < c->cs; i++) c->sec_members[size_t(i)] = static_cast<const float*>(device_members[i]);
    return install_secondary_table(c);
}

int crf_secondary_member_minmax(crf_context* c, float* out_min, float* out_max) {
    if (int r = check_ready(c)) return r;
    if (!out_min || !out_max) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (c->sec_members.empty()) return fail(c, CRF_ERR_STATE, "no secondary members are bound");
    if (!c->sec_minmax_valid) {
        if (int r = bind_device(c)) return r;
        CRF_HIP(c, crf::launch_minmax(c->d_sec_table, c->cs, c->num_voxels, c->d_minmax, c->stream));
        uint32_t keys[2];
        CRF_HIP(c, hipMemcpyAsync(keys, c->d_minmax, sizeof keys, hipMemcpyDeviceToHost, c->stream));
        CRF_HIP(c, hipStreamSynchronize(c->stream));
        c->sec_min_v = crf::minmax_key_to_float(keys[0]);
        c->sec_max_v = crf::minmax_key_to_float(keys[1]);
        c->sec_minmax_valid = true;
    }
    *out_min = c->sec_min_v;
    *out_max = c->sec_max_v;
    return CRF_OK;
}

static int ensure_workspace(crf_context* c, size_t need) {
    if (need > c->workspace_bytes) {
        CRF_HIP(c, hipDeviceSynchronize());  // an earlier evaluation on a caller stream may still use the old workspace
        if (c->d_workspace) (void)hipFree(c->d_workspace);
        c->d_workspace = nullptr;
        c->workspace_bytes = 0;
        CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_workspace), need));
        c->workspace_bytes = need;
    }
    return CRF_OK;
}

}  // extern "C"

// ---- member volumes of 4 GiB and more: evaluation in windows ----------------------------------------------------------
constexpr size_t kWindowVoxels = size_t(1) << 29;  // 2 GiB of every member per launch

static int ensure_windows(crf_context* c) {
    if (c->windows > 0) return CRF_OK;
    const int windows = int((c->alloc_voxels + kWindowVoxels - 1) / kWindowVoxels);
    const bool sec = !c->sec_members.empty();
    std::vector<const float*> table(size_t(windows) * size_t(c->cs) * (sec ? 2 : 1));
    for (int w = 0; w < windows; w++)
        for (int m = 0; m < c->cs; m++) {
            table[(size_t(w) * (sec ? 2 : 1)) * size_t(c->cs) + size_t(m)] = c->members[size_t(m)] + size_t(w) * kWindowVoxels;
            if (sec) table[(size_t(w) * 2 + 1) * size_t(c->cs) + size_t(m)] = c->sec_members[size_t(m)] + size_t(w) * kWindowVoxels;
        }
    if (c->d_window_tables) (void)hipFree(c->d_window_tables);
    c->d_window_tables = nullptr;
    CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_window_tables), table.size() * sizeof(float*)));
    CRF_HIP(c, hipMemcpy(c->d_window_tables, table.data(), table.size() * sizeof(float*), hipMemcpyHostToDevice));
    c->windows = windows;
    c->window_has_secondary = sec;
    return CRF_OK;
}

// narrows the context to one window for the duration of a launch; restores it on every exit path
struct WindowScope {
    crf_context* c;
    const float** table;
    const float** sec_table;
    size_t voxels;
    explicit WindowScope(crf_context* ctx) : c(ctx), table(ctx->d_member_table), sec_table(ctx->d_sec_table), voxels(ctx->num_voxels) {}
    size_t select(int w) {  // returns the window's first voxel
        const size_t per = size_t(c->window_has_secondary ? 2 : 1) * size_t(c->cs);
        c->d_member_table = c->d_window_tables + size_t(w) * per;
        if (c->window_has_secondary) c->d_sec_table = c->d_window_tables + size_t(w) * per + size_t(c->cs);
        c->num_voxels = std::min(kWindowVoxels, c->alloc_voxels - size_t(w) * kWindowVoxels);
        return size_t(w) * kWindowVoxels;
    }
    ~WindowScope() {
        c->d_member_table = table;
        c->d_sec_table = sec_table;
        c->num_voxels = voxels;
    }
};

// runs launch(out + first voxel of the window) for every window of a >= 4 GiB grid, or once for an ordinary grid
template <class Launch>
static int for_each_window(crf_context* c, float* out, Launch&& launch) {
    if (!c->windowed) return launch(out);
    if (int r = ensure_windows(c)) return r;
    WindowScope scope(c);
    for (int w = 0; w < c->windows; w++) {
        const size_t first = scope.select(w);
        if (int r = launch(out + first)) return r;
    }
    return CRF_OK;
}

extern "C" {

// CRF_FLAG_SYMMETRIC: measure(primary members at v, secondary members at v) for every voxel v
static int compute_symmetric(crf_context* c, const crf_params* p, float* out, hipStream_t s) {
    if (c->sec_members.empty())
        return fail(c, CRF_ERR_STATE, "CRF_FLAG_SYMMETRIC needs secondary members (crf_upload_secondary_members)");
    if (c->cs > crf::kMaxGenericMembers)
        return fail(c, CRF_ERR_UNSUPPORTED, fmt("the symmetric mode supports at most %d members", crf::kMaxGenericMembers));
    if ((p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC) && (p->num_bins < 1 || p->num_bins > 255))
        return fail(c, CRF_ERR_ARGUMENT, fmt("num_bins %d outside [1,255]", p->num_bins));
    // any k >= 1, like the reference (its k+1-nearest-neighbour query returns at most cs points; psi(k) itself is used)
    if ((p->measure == CRF_MI_KRASKOV || p->measure == CRF_KMI_CC) && p->k < 1)
        return fail(c, CRF_ERR_ARGUMENT, fmt("k=%d must be at least 1", p->k));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        e0 = take_event(c);
        e1 = take_event(c);
        (void)hipEventRecord(e0, s);
    }
    hipError_t e = hipErrorNotSupported;
    if (p->measure == CRF_PEARSON) {
        e = crf::launch_pearson_symmetric(c->d_member_table, c->d_sec_table, c->cs, c->num_voxels, out, s);
        c->last_kernel = "pearson_symmetric_kernel";
    }
    if (p->measure == CRF_MI_KRASKOV || p->measure == CRF_KMI_CC) {
        e = crf::launch_mi_kraskov_symmetric(c->d_member_table, c->d_sec_table, c->cs, c->num_voxels, p->k,
                                             kraskov_c_term(p->k, 1), p->measure == CRF_KMI_CC, c->d_tables, out, s);
        c->last_kernel = "kraskov_direct_kernel";
    }
    if (p->measure == CRF_SPEARMAN || p->measure == CRF_KENDALL || p->measure == CRF_MI_BINNED ||
        p->measure == CRF_BINNED_MI_CC) {
        const char* force_direct = getenv("CRF_SYMMETRIC_DIRECT");  // tuning / tests: the any-member-count kernel
        if (!(force_direct && *force_direct == '1')) {
            e = crf::launch_sorted_symmetric(c->d_member_table, c->d_sec_table, c->cs, c->num_voxels, p->measure,
                                             p->num_bins, p->min_ref, p->max_ref, p->min_query, p->max_query, c->d_tables,
                                             out, s);
            c->last_kernel = "sorted_symmetric_kernel";
        }
    }
    if (e == hipErrorNotSupported && (p->measure == CRF_SPEARMAN || p->measure == CRF_KENDALL ||
                                      p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC)) {
        if (int r = ensure_workspace(c, crf::direct_symmetric_workspace_bytes(c->cs, c->num_voxels, p->measure))) return r;
        e = crf::launch_direct_symmetric(c->d_member_table, c->d_sec_table, c->cs, c->num_voxels, p->measure, p->num_bins,
                                         p->min_ref, p->max_ref, p->min_query, p->max_query, c->d_tables, c->d_workspace,
                                         out, s);
        c->last_kernel = "direct_symmetric_kernel";
    }
    if (e == hipErrorNotSupported) {
        if (int r = ensure_workspace(c, crf::pair_workspace_bytes(c->cs, c->num_voxels))) return r;
        const crf::PairArgs a{p->measure, p->num_bins, p->k, 0, 1, p->min_ref, p->max_ref, p->min_query, p->max_query,
                              kraskov_c_term(p->k > 0 ? p->k : 1, 1)};
        e = crf::launch_pair_requests(c->d_member_table, c->d_sec_table, c->cs, c->xs, c->ys, c->num_voxels, nullptr,
                                      c->num_voxels, a, c->d_tables, c->d_workspace, out, s);
        c->last_kernel = "pair_request_kernel";
    }
    if (e0 && e1) {
        (void)hipEventRecord(e1, s);
        c->ev_pending.emplace_back(e0, e1);
    }
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

static int ref_voxel(crf_context* c, int x, int y, int z, size_t* voxel) {
    if (x < 0 || y < 0 || z < 0 || x >= c->xs || y >= c->ys || z >= c->zs)
        return fail(c, CRF_ERR_ARGUMENT,
                    fmt("reference point (%d,%d,%d) outside the local grid %dx%dx%d", x, y, z, c->xs, c->ys, c->zs));
    *voxel = (size_t(z) * size_t(c->ys) + size_t(y)) * size_t(c->xs) + size_t(x);  // IDXS, DataSet.hpp:37
    return CRF_OK;
}

int crf_gather_reference_device(crf_context* c, int x, int y, int z, void* device_out, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!device_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    size_t voxel;
    if (int r = ref_voxel(c, x, y, z, &voxel)) return r;
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    CRF_HIP(c, crf::launch_gather_reference(c->d_member_table, c->cs, voxel, static_cast<float*>(device_out), s));
    return CRF_OK;
}

int crf_gather_reference_rows_device(crf_context* c, const int32_t* xyz, int num_rows, void* device_rows, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!xyz || !device_rows) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (num_rows < 0 || num_rows > crf::kMaxGatherRows)
        return fail(c, CRF_ERR_ARGUMENT, fmt("num_rows %d outside [0,%d]", num_rows, crf::kMaxGatherRows));
    crf::GatherRows rows;
    for (int r = 0; r < num_rows; r++) {
        rows.voxel[r] = crf::kNoVoxel;
        if (xyz[3 * r + 2] >= 0)
            if (int e = ref_voxel(c, xyz[3 * r], xyz[3 * r + 1], xyz[3 * r + 2], &rows.voxel[r])) return e;
    }
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    CRF_HIP(c, crf::launch_gather_reference_rows(c->d_member_table, c->cs, rows, num_rows,
                                                 static_cast<float*>(device_rows), s));
    return CRF_OK;
}

int crf_gather_reference(crf_context* c, int x, int y, int z, float* host_out) {
    if (!host_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (int r = crf_gather_reference_device(c, x, y, z, c ? c->d_ref : nullptr, nullptr)) return r;
    CRF_HIP(c, hipMemcpyAsync(host_out, c->d_ref, sizeof(float) * size_t(c->cs), hipMemcpyDeviceToHost, c->stream));
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    return CRF_OK;
}

// ---- device result -> the caller's (pageable) host buffer -------------------------------------------------------------
// The caller of calculateCpu owns a freshly allocated `new float[xs*ys*zs]` (VolumeData.cpp:1222-1226): pageable and never
// touched.  Measured on MI355X hosts at 256^3 (67 MB, profiles/r03_host_boundary.md): one DMA into pinned memory 1.19 ms
// (56.5 GB/s, the PCIe floor); a kernel storing straight into device-mapped pinned memory 1.22 ms; 8 host threads move
// pinned -> resident pageable memory at 120 GB/s but only at 14 GB/s into never-touched pages (first-touch faults).
//
// So a host-output evaluation is a pipeline over a few voxel RANGES (ensure_host_ranges):
//   GPU     the per-voxel kernel of each range stores its results straight into a pinned, device-mapped staging buffer
//           (no device-side result buffer, no DMA engine: the stores cross PCIe while the kernel runs -- the kernel is
//           simply throttled to the link rate, which is the floor anyway);
//   host    a persistent pool of copier threads (crf_pool.h) moves each finished range from the staging buffer into
//           the caller's buffer, and while it waits for a range it faults the destination pages of the ranges ahead in
//           (MADV_POPULATE_WRITE batches the faults; transparent huge pages are requested for the buffer first).
// Range sizes are staggered over two streams so that kernel ends alternate and the last copy is small (ensure_host_ranges).
// CRF_HOST_PATH=dma keeps results in HBM and copies each range with the DMA engine instead (also used when a post-pass
// has to read the result back: CRF_FLAG_ABSOLUTE_VALUE).
constexpr int kMadvPopulateWrite = 23;  // MADV_POPULATE_WRITE (Linux 5.14), not in every libc header
constexpr size_t kPage = 4096;

static int env_int_or(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

// faults [lo, hi) of the caller's buffer in (write access); mode 0: leave it to the copy, 1: touch, 2: populate
static void fault_in(char* base, size_t lo, size_t hi, int mode) {
    if (mode == 0 || lo >= hi) return;
    const uintptr_t a = (reinterpret_cast<uintptr_t>(base) + lo) & ~(kPage - 1);
    const uintptr_t e = (reinterpret_cast<uintptr_t>(base) + hi + kPage - 1) & ~(kPage - 1);
    if (mode == 2 && madvise(reinterpret_cast<void*>(a), e - a, kMadvPopulateWrite) == 0) return;
    // fallback / mode 1: one write per page.  The bytes written are inside the caller's buffer and are overwritten by
    // the result afterwards.
    for (size_t off = lo; off < hi; off += kPage) static_cast<volatile char*>(base)[off] = 0;
    static_cast<volatile char*>(base)[hi - 1] = 0;
}

// thread w's share of byte range [r_lo, r_hi) among `workers` threads, split on page boundaries
static void thread_share(size_t r_lo, size_t r_hi, int w, int workers, size_t* lo, size_t* hi) {
    const size_t per = (((r_hi - r_lo) + size_t(workers) - 1) / size_t(workers) + kPage - 1) & ~(kPage - 1);
    *lo = std::min(r_hi, r_lo + size_t(w) * per);
    *hi = std::min(r_hi, *lo + per);
}

static int ensure_copy_pool(crf_context* c) {
    if (c->copy_pool) return CRF_OK;
    const unsigned hw = std::thread::hardware_concurrency();
    int cap = int(std::min<unsigned>(16u, std::max(1u, hw / 2)));
    if (c->copy_threads_cap > 0) cap = std::min(cap, c->copy_threads_cap);
    const int forced = env_int_or("CRF_COPY_THREADS", 0);
    if (forced >= 1) cap = std::min(forced, 64);
    // copier threads idle-spin for a short while only: back-to-back evaluations hand over within ~0.1 ms
    c->copy_pool = std::make_unique<crf::SpinPool>(cap, nullptr, 300e-6);
    c->copy_threads = cap;
    if (forced >= 1 || cap <= 2) return CRF_OK;
    // one-off calibration: how many of the pool's threads move pinned -> pageable memory fastest on this host
    const size_t bytes = std::min<size_t>(c->alloc_voxels * sizeof(float), size_t(16) << 20);
    std::vector<char> dst(bytes, 1);
    const char* src = reinterpret_cast<const char*>(c->h_staging);
    double best = 1e30;
    for (int t : {2, 4, 8, 12, 16}) {
        if (t > cap) break;
        double fastest = 1e30;
        for (int rep = 0; rep < 3; rep++) {
            const auto t0 = std::chrono::steady_clock::now();
            c->copy_pool->run([&](int w) -> int {
                if (w >= t) return 0;
                size_t lo, hi;
                thread_share(0, bytes, w, t, &lo, &hi);
                if (lo < hi) memcpy(dst.data() + lo, src + lo, hi - lo);
                return 0;
            });
            fastest = std::min(fastest, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
        }
        if (fastest < best * 0.93) {  // more threads only for a real gain
            best = fastest;
            c->copy_threads = t;
        }
    }
    return CRF_OK;
}

// Plain form for small results and the sibling reductions: kernel into HBM, one copy.
static int copy_result_to_host(crf_context* c, const float* d_src, float* host_out, size_t count) {
    CRF_HIP(c, hipMemcpyAsync(host_out, d_src, count * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    return CRF_OK;
}

// phase: bit 0 = reference-side preparation, bit 1 = per-voxel kernel (crf_internal.h RefSource::phase);
// slot < 0: the context's own preparation buffer
static int compute_impl_one(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                            void* stream, unsigned phase, int slot, const crf::RefOverride* ov);

// An ordinary grid: one call.  A grid whose members are 4 GiB or larger: the reference side once, from the whole grid
// (the reference point indexes it with 64 bits), then the per-voxel kernel window by window.
static int compute_impl(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                        void* stream, unsigned phase, int slot, const crf::RefOverride* ov = nullptr) {
    if (!c || !c->windowed || !p || (p->flags & CRF_FLAG_SYMMETRIC))
        return compute_impl_one(c, p, device_reference_values, device_out, stream, phase, slot, ov);
    if (phase & 1u)
        if (int r = compute_impl_one(c, p, device_reference_values, nullptr, stream, 1u, slot, ov)) return r;
    if (!(phase & 2u)) return CRF_OK;
    if (!device_out) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    return for_each_window(c, static_cast<float*>(device_out), [&](float* o) {
        return compute_impl_one(c, p, nullptr, o, stream, 2u, slot, nullptr);
    });
}

static int compute_impl_one(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                            void* stream, unsigned phase, int slot, const crf::RefOverride* ov) {
    if (int r = check_ready(c)) return r;
    if (!p || (!device_out && (phase & 2u))) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (p->measure < CRF_PEARSON || p->measure > CRF_KMI_CC)
        return fail(c, CRF_ERR_ARGUMENT, fmt("unknown measure %d", p->measure));
    for (int v : p->reserved)
        if (v != 0) return fail(c, CRF_ERR_ARGUMENT, "crf_params.reserved must be zero");
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    float* out = static_cast<float*>(device_out);
    if (p->flags & CRF_FLAG_SYMMETRIC) {
        if (phase != 3u) return fail(c, CRF_ERR_ARGUMENT, "the symmetric mode has no reference-side preparation");
        return for_each_window(c, out, [&](float* o) { return compute_symmetric(c, p, o, s); });
    }
    float* prep = c->d_prep;
    if (slot >= 0) {
        if (!c->d_prep_slots)
            CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_prep_slots), size_t(CRF_PREPARED_SLOTS) * crf::kPrepBytes));
        prep = c->d_prep_slots + size_t(slot) * (crf::kPrepBytes / sizeof(float));
    }

    // 1. reference vector (CorrelationCalculator.cpp:802-818): a device array, a host array (copied stream-ordered),
    //    or the reference point -- then the gather is fused into the estimator's preparation kernel.
    crf::RefSource ref{static_cast<const float*>(device_reference_values), 0};
    ref.phase = phase;
    if (!(phase & 1u)) {
        ref.values = nullptr;  // prepared earlier: no reference vector is read
    } else if (!ref.values && ov) {
        // crf_group, direct exchange: the preparation kernel reads the values out of another context's members
        ref.table = ov->table;
        ref.voxel = ov->voxel;
    } else if (!ref.values && (p->flags & CRF_FLAG_REFERENCE_FROM_SECONDARY)) {
        if (c->sec_members.empty())
            return fail(c, CRF_ERR_STATE, "CRF_FLAG_REFERENCE_FROM_SECONDARY needs secondary members");
        size_t voxel;
        if (int r = ref_voxel(c, p->ref_x, p->ref_y, p->ref_z, &voxel)) return r;
        CRF_HIP(c, crf::launch_gather_reference(c->d_sec_table, c->cs, voxel, c->d_ref, s));
        ref.values = c->d_ref;
    }
    if ((phase & 1u) && !ref.values && !ref.table) {
        if (p->reference_values) {
            CRF_HIP(c, hipMemcpyAsync(c->d_ref, p->reference_values, sizeof(float) * size_t(c->cs),
                                      hipMemcpyHostToDevice, s));
            ref.values = c->d_ref;
        } else {
            if (int r = ref_voxel(c, p->ref_x, p->ref_y, p->ref_z, &ref.voxel)) return r;
        }
    }

    // 2. estimator
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling && (phase & 2u)) {
        e0 = take_event(c);
        e1 = take_event(c);
    }
    const int vpt = std::min(c->max_vpt, alignment_vpt(out));
    crf::LaunchInfo info;
    hipError_t e = hipSuccess;
    if (p->measure != CRF_PEARSON && c->cs > crf::kMaxSortMembers) {
        // any-member-count path (kernels_generic.hip)
        if (c->cs > crf::kMaxGenericMembers)
            return fail(c, CRF_ERR_UNSUPPORTED, fmt("measure %d supports at most %d members (got %d)", p->measure,
                                                    crf::kMaxGenericMembers, c->cs));
        if ((p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC) && (p->num_bins < 1 || p->num_bins > 255))
            return fail(c, CRF_ERR_ARGUMENT, fmt("num_bins %d outside [1,255]", p->num_bins));
        if ((p->measure == CRF_MI_KRASKOV || p->measure == CRF_KMI_CC) && p->k < 1)
            return fail(c, CRF_ERR_ARGUMENT, fmt("k=%d must be at least 1", p->k));
        if (p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC) {  // O(cs) histogram kernel
            crf::BinnedArgs ba{p->num_bins, p->min_ref, p->max_ref, p->min_query, p->max_query,
                               p->measure == CRF_BINNED_MI_CC};
            e = crf::launch_mi_binned_hist(c->d_member_table, c->cs, c->num_voxels, ref, ba, c->d_tables, prep, out, s, e0,
                                           e1, &info);
            if (e != hipErrorNotSupported) {
                c->last_kernel = info.kernel_name ? info.kernel_name : "";
                if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
                if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
                return CRF_OK;
            }
            e = hipSuccess;  // too many bins for the LDS rows: the O(cs^2) kernel below
        }
        if (p->measure == CRF_MI_KRASKOV || p->measure == CRF_KMI_CC) {  // tile-free single-sweep top-K kernel
            const int est = p->kraskov_estimator_index == 2 ? 2 : 1;
            crf::KraskovArgs ka{p->k, est, p->measure == CRF_KMI_CC, kraskov_c_term(p->k, est)};
            e = crf::launch_mi_kraskov_direct(c->d_member_table, c->cs, c->num_voxels, ref, ka, c->d_tables, prep, out, s,
                                              e0, e1, &info);
            if (e != hipErrorNotSupported) {
                c->last_kernel = info.kernel_name ? info.kernel_name : "";
                if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
                if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
                return CRF_OK;
            }
            e = hipSuccess;  // k > 128 or tables beyond LDS: the repeated-minimum kernel below
        }
        if (int r = ensure_workspace(c, crf::generic_workspace_bytes(c->cs, c->num_voxels))) return r;
        crf::GenericArgs ga{p->measure, p->num_bins, p->min_ref, p->max_ref, p->min_query, p->max_query, p->k,
                            p->kraskov_estimator_index == 2 ? 2 : 1,
                            kraskov_c_term(p->k > 0 ? p->k : 1, p->kraskov_estimator_index == 2 ? 2 : 1)};
        const bool rank_measure = p->measure == CRF_SPEARMAN || p->measure == CRF_KENDALL;
        if (rank_measure && c->cs <= 256 && !c->d_todo)  // the sort-based kernels' list of deferred voxels
            CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_todo), (c->alloc_voxels + 1) * sizeof(uint32_t)));
        e = crf::launch_generic(c->d_member_table, c->cs, c->num_voxels, ref, ga, c->d_tables, prep, c->d_workspace,
                                out, s, e0, e1, &info, rank_measure ? c->d_todo : nullptr);
        c->last_kernel = info.kernel_name ? info.kernel_name : "";
        if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
        if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
        return CRF_OK;
    }
    switch (p->measure) {
        case CRF_PEARSON:
            e = crf::launch_pearson(c->d_member_table, c->cs, c->num_voxels, vpt, ref, prep, out, s, e0, e1,
                                    &info);
            break;
        case CRF_SPEARMAN:
            if (c->cs > crf::kMaxSortMembers)
                return fail(c, CRF_ERR_UNSUPPORTED, fmt("Spearman supports at most %d members", crf::kMaxSortMembers));
            if (c->cs > 16 && !c->d_todo)
                CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_todo), (c->alloc_voxels + 1) * sizeof(uint32_t)));
            e = crf::launch_spearman(c->d_member_table, c->cs, c->num_voxels, ref, prep, c->d_todo, out, s, e0, e1,
                                     &info);
            break;
        case CRF_KENDALL:
            if (c->cs > crf::kMaxSortMembers)
                return fail(c, CRF_ERR_UNSUPPORTED, fmt("Kendall supports at most %d members", crf::kMaxSortMembers));
            if (c->cs > 16 && !c->d_todo)
                CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_todo), (c->alloc_voxels + 1) * sizeof(uint32_t)));
            e = crf::launch_kendall(c->d_member_table, c->cs, c->num_voxels, ref, prep, c->d_todo, out, s, e0, e1,
                                    &info);
            break;
        case CRF_MI_BINNED:
        case CRF_BINNED_MI_CC: {
            if (p->num_bins < 1 || p->num_bins > 255)
                return fail(c, CRF_ERR_ARGUMENT, fmt("num_bins %d outside [1,255]", p->num_bins));
            if (c->cs > crf::kMaxSortMembers)
                return fail(c, CRF_ERR_UNSUPPORTED, fmt("binned MI supports at most %d members", crf::kMaxSortMembers));
            crf::BinnedArgs a{p->num_bins, p->min_ref, p->max_ref, p->min_query, p->max_query,
                              p->measure == CRF_BINNED_MI_CC};
            if (const char* hv = getenv("CRF_BINNED_HIST"); hv && *hv == '1') {  // tuning: histogram kernel for any cs
                e = crf::launch_mi_binned_hist(c->d_member_table, c->cs, c->num_voxels, ref, a, c->d_tables, prep, out, s,
                                               e0, e1, &info);
                if (e != hipErrorNotSupported) break;
            }
            e = crf::launch_mi_binned(c->d_member_table, c->cs, c->num_voxels, ref, a, c->d_tables, prep, out, s,
                                      e0, e1, &info);
            break;
        }
        case CRF_MI_KRASKOV:
        case CRF_KMI_CC: {
            if (p->k < 1) return fail(c, CRF_ERR_ARGUMENT, fmt("k=%d must be at least 1", p->k));
            if (c->cs > crf::kMaxSortMembers)
                return fail(c, CRF_ERR_UNSUPPORTED, fmt("Kraskov MI supports at most %d members", crf::kMaxSortMembers));
            const int est = p->kraskov_estimator_index == 2 ? 2 : 1;  // clamp as CorrelationCalculator.cpp:765
            crf::KraskovArgs a{p->k, est, p->measure == CRF_KMI_CC, kraskov_c_term(p->k, est)};
            e = crf::launch_mi_kraskov(c->d_member_table, c->cs, c->num_voxels, ref, a, c->d_tables, prep, out, s,
                                       e0, e1, &info);
            break;
        }
    }
    c->last_kernel = info.kernel_name ? info.kernel_name : "";
    if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
    if (e == hipErrorNotSupported)
        return fail(c, CRF_ERR_UNSUPPORTED, fmt("measure %d is not implemented by this build", p->measure));
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_compute_device(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                       void* stream) {
    int rc;
    if (p && p->prepared_slot != 0) {
        if (p->prepared_slot < 0 || p->prepared_slot > CRF_PREPARED_SLOTS)
            return fail(c, CRF_ERR_ARGUMENT, fmt("prepared_slot %d outside [0,%d]", p->prepared_slot, CRF_PREPARED_SLOTS));
        if (c && !c->d_prep_slots) return fail(c, CRF_ERR_STATE, "prepared_slot given but crf_prepare_device was never called");
        rc = compute_impl(c, p, nullptr, device_out, stream, 2u, p->prepared_slot - 1);
    } else {
        rc = compute_impl(c, p, device_reference_values, device_out, stream, 3u, -1);
    }
    if (rc == CRF_OK && (p->flags & CRF_FLAG_ABSOLUTE_VALUE)) {  // opt-in: what the reference's accelerator paths do
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
        CRF_HIP(c, crf::launch_abs(static_cast<float*>(device_out), c->num_voxels, s));
    }
    return rc;
}

int crf_prepare_rows_device(crf_context* c, const crf_params* p, const void* device_rows, int first_slot, int count,
                            void* stream) {
    if (!c || !p || !device_rows) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (count < 0 || first_slot < 0 || first_slot + count > CRF_PREPARED_SLOTS)
        return fail(c, CRF_ERR_ARGUMENT, fmt("slots [%d, %d) outside [0, %d)", first_slot, first_slot + count, CRF_PREPARED_SLOTS));
    crf_params local = *p;
    local.reference_values = nullptr;
    local.prepared_slot = 0;
    const float* rows = static_cast<const float*>(device_rows);
    for (int i = 0; i < count; i++)
        if (int r = crf_prepare_device(c, &local, rows + size_t(i) * size_t(c->cs), first_slot + i, stream)) return r;
    return CRF_OK;
}

int crf_compute_prepared_device(crf_context* c, const crf_params* p, int first_slot, int count, void* const* device_outs,
                                void* stream) {
    if (!c || !p || !device_outs) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (count < 0 || first_slot < 0 || first_slot + count > CRF_PREPARED_SLOTS)
        return fail(c, CRF_ERR_ARGUMENT, fmt("slots [%d, %d) outside [0, %d)", first_slot, first_slot + count, CRF_PREPARED_SLOTS));
    crf_params local = *p;
    for (int i = 0; i < count; i++) {
        local.prepared_slot = first_slot + i + 1;
        if (int r = crf_compute_device(c, &local, nullptr, device_outs[i], stream)) return r;
    }
    return CRF_OK;
}

int crf_prepare_device(crf_context* c, const crf_params* p, const void* device_reference_values, int slot, void* stream) {
    if (slot < 0 || slot >= CRF_PREPARED_SLOTS)
        return fail(c, CRF_ERR_ARGUMENT, fmt("slot %d outside [0,%d)", slot, CRF_PREPARED_SLOTS));
    return compute_impl(c, p, device_reference_values, nullptr, stream, 1u, slot);
}

int crf_compute(crf_context* c, const crf_params* p, float* host_out) {
    return crf::compute_to_host(c, p, nullptr, host_out, nullptr);
}

}  // extern "C"

namespace {
// The voxel ranges of a host-output evaluation: one member-pointer table per range (pointers advanced by the range's
// first voxel), so that every per-voxel kernel can be launched on a range without knowing about ranges.  Range lengths
// are multiples of 1024 voxels (4 KiB: every range stays as aligned as the members themselves) and shrink towards the
// end: the copy of the last range into the caller's buffer is the only host work no kernel hides.
int ensure_host_ranges(crf_context* c) {
    if (c->host_chunks > 0) return CRF_OK;
    const size_t n = c->alloc_voxels;
    std::vector<size_t> first{0};
    const int forced = env_int_or("CRF_HOST_CHUNKS", 0);
    if (forced >= 1) {  // experiments: equal ranges
        size_t per = (n + size_t(forced) - 1) / size_t(forced);
        per = (per + 1023) & ~size_t(1023);
        for (size_t at = per; at < n && int(first.size()) < kMaxHostChunks; at += per) first.push_back(at);
    } else {
        // Shares of 64, consecutive ranges alternating between two streams: 3 6 6 6 6 6 6 6 6 5 4 2 2.  The two streams'
        // kernels run concurrently and share the link; with the FIRST range half the size of the others the kernel ends
        // alternate (B0 A0 B1 A1 ...), so results land every ~1/11 of the run from early on, the copier threads always
        // have a landed range to move, and the last ranges are small: only their copy is not hidden behind a kernel.
        // Same-process A/B at 256^3 (tools/measure_host_path.py ab, profiles/r03_host_boundary_variants.txt), resident /
        // fresh destination: 13 ranges 1.326 / 1.391 ms; 11 ranges (4 8x6 6 3 2 1) 1.336 / 1.452; 8 staggered ranges
        // 1.369 / 1.467; 8 shrinking ranges 16 14 11 8 6 4 3 2 (pairs end together) 1.378 / 1.527; 8 equal 1.450 / 1.644.
        std::vector<int> kShares = {3, 6, 6, 6, 6, 6, 6, 6, 6, 5, 4, 2, 2};
        if (const char* e = getenv("CRF_HOST_SHARES")) {  // experiments: comma-separated shares of 64
            std::vector<int> v;
            int sum = 0;
            for (const char* q = e; *q;) {
                v.push_back(atoi(q));
                sum += v.back();
                while (*q && *q != ',') q++;
                if (*q == ',') q++;
            }
            if (sum == 64 && v.size() >= 1 && v.size() <= size_t(kMaxHostChunks)) kShares = v;
        }
        size_t acc = 0;
        for (int j = 0; j + 1 < int(kShares.size()); j++) {
            acc += size_t(kShares[size_t(j)]);
            const size_t at = (n / 64 * acc + 1023) & ~size_t(1023);
            if (at > first.back() && at < n) first.push_back(at);
        }
    }
    const int ranges = int(first.size());
    first.push_back(n);
    std::vector<const float*> table(size_t(ranges) * size_t(c->cs));
    for (int j = 0; j < ranges; j++)
        for (int m = 0; m < c->cs; m++) table[size_t(j) * size_t(c->cs) + size_t(m)] = c->members[size_t(m)] + first[size_t(j)];
    if (c->d_chunk_tables) (void)hipFree(c->d_chunk_tables);
    c->d_chunk_tables = nullptr;
    CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_chunk_tables), table.size() * sizeof(float*)));
    CRF_HIP(c, hipMemcpy(c->d_chunk_tables, table.data(), table.size() * sizeof(float*), hipMemcpyHostToDevice));
    if (!c->copy_stream) CRF_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int j = 0; j < ranges; j++)
        if (!c->chunk_done[j]) CRF_HIP(c, hipEventCreateWithFlags(&c->chunk_done[j], hipEventDisableTiming));
    for (int j = 0; j < ranges; j++)
        if (!c->chunk_copied[j]) CRF_HIP(c, hipEventCreateWithFlags(&c->chunk_copied[j], hipEventDisableTiming));
    if (!c->prep_done) CRF_HIP(c, hipEventCreateWithFlags(&c->prep_done, hipEventDisableTiming));
    for (int j = 0; j <= ranges; j++) c->chunk_first[j] = first[size_t(j)];
    c->host_chunks = ranges;
    return CRF_OK;
}

// narrows the context to one voxel range for the duration of a launch; restores it on every exit path
struct RangeScope {
    crf_context* c;
    const float** table;
    size_t voxels;
    int vpt;
    explicit RangeScope(crf_context* ctx) : c(ctx), table(ctx->d_member_table), voxels(ctx->num_voxels), vpt(ctx->max_vpt) {}
    void select(int j) {
        c->d_member_table = c->d_chunk_tables + size_t(j) * size_t(c->cs);
        c->num_voxels = c->chunk_first[j + 1] - c->chunk_first[j];
    }
    ~RangeScope() {
        c->d_member_table = table;
        c->num_voxels = voxels;
        c->max_vpt = vpt;
    }
};

}  // namespace

namespace crf {

int gather_reference_to(crf_context* c, bool secondary, int x, int y, int z, float* device_out, hipStream_t s) {
    if (int r = check_ready(c)) return r;
    if (secondary && c->sec_members.empty()) return fail(c, CRF_ERR_STATE, "no secondary members are bound");
    size_t voxel;
    if (int r = ref_voxel(c, x, y, z, &voxel)) return r;
    if (int r = bind_device(c)) return r;
    CRF_HIP(c, launch_gather_reference(secondary ? c->d_sec_table : c->d_member_table, c->cs, voxel, device_out, s));
    return CRF_OK;
}

int second_stream(crf_context* c, hipStream_t* out) {
    if (int r = bind_device(c)) return r;
    if (!c->stream2) CRF_HIP(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
    *out = c->stream2;
    return CRF_OK;
}

int reference_override(crf_context* owner, bool secondary, int x, int y, int z, RefOverride* out) {
    if (int r = check_ready(owner)) return r;
    if (secondary && owner->sec_members.empty()) return fail(owner, CRF_ERR_STATE, "no secondary members are bound");
    if (int r = ref_voxel(owner, x, y, z, &out->voxel)) return r;
    out->table = secondary ? owner->d_sec_table : owner->d_member_table;
    return CRF_OK;
}

int compute_device_ex(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                      void* stream, const RefOverride* ov) {
    if (!ov) return crf_compute_device(c, p, device_reference_values, device_out, stream);
    if (!p || p->prepared_slot != 0) return fail(c, CRF_ERR_ARGUMENT, "a direct reference read cannot use a prepared slot");
    const int rc = compute_impl(c, p, device_reference_values, device_out, stream, 3u, -1, ov);
    if (rc == CRF_OK && (p->flags & CRF_FLAG_ABSOLUTE_VALUE)) {
        hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
        CRF_HIP(c, crf::launch_abs(static_cast<float*>(device_out), c->num_voxels, s));
    }
    return rc;
}

int prepare_device_ex(crf_context* c, const crf_params* p, const void* device_reference_values, int slot, void* stream,
                      const RefOverride* ov) {
    if (slot < 0 || slot >= CRF_PREPARED_SLOTS)
        return fail(c, CRF_ERR_ARGUMENT, fmt("slot %d outside [0,%d)", slot, CRF_PREPARED_SLOTS));
    return compute_impl(c, p, device_reference_values, nullptr, stream, 1u, slot, ov);
}

// What calculateCpu(t, e, buffer) does, into the caller's host buffer (Calculator.hpp:123-124, VolumeData.cpp:1222-1226):
// the pipeline described above copy_result_to_host.  The reference-side preparation runs once, from the whole-grid
// member table (the reference point indexes the whole grid); the per-voxel kernels run range by range.
int compute_to_host(crf_context* c, const crf_params* p, const void* device_reference_values, float* host_out,
                    const RefOverride* ov) {
    if (int r = check_ready(c)) return r;
    if (!host_out || !p) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (int r = bind_device(c)) return r;
    const size_t bytes = c->alloc_voxels * sizeof(float);
    const bool ranged = !(p->flags & CRF_FLAG_SYMMETRIC) && p->prepared_slot == 0 && bytes >= (size_t(8) << 20) &&
                        !c->windowed && env_int_or("CRF_PLAIN_D2H", 0) != 1;
    if (!ranged) {
        if (!c->d_out) CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), bytes));
        if (int r = compute_device_ex(c, p, device_reference_values, c->d_out, nullptr, ov)) return r;
        return copy_result_to_host(c, c->d_out, host_out, c->alloc_voxels);
    }
    if (int r = ensure_host_ranges(c)) return r;
    const int ranges = c->host_chunks;
    // pinned, device-mapped staging for the whole local result
    if (!c->h_staging) {
        CRF_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_staging), bytes, hipHostMallocMapped));
        void* dev = nullptr;
        CRF_HIP(c, hipHostGetDevicePointer(&dev, c->h_staging, 0));
        c->d_staging = static_cast<float*>(dev);
    }
    if (int r = ensure_copy_pool(c)) return r;
    const char* path_env = getenv("CRF_HOST_PATH");
    const bool dma = (path_env && strcmp(path_env, "dma") == 0) || (p->flags & CRF_FLAG_ABSOLUTE_VALUE);
    if (dma && !c->d_out) CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), bytes));
    {
        hipStream_t unused;
        if (int r = second_stream(c, &unused)) return r;
    }
    const bool two_streams = env_int_or("CRF_HOST_STREAMS", 2) == 2;
    const int fault_mode = env_int_or("CRF_HOST_FAULT", 2);           // 0 none, 1 touch, 2 MADV_POPULATE_WRITE
    const bool huge = env_int_or("CRF_HOST_HUGEPAGE", 1) == 1;        // ask for transparent huge pages first
    const bool trace = env_int_or("CRF_HOST_TRACE", 0) == 1;
    const auto t_call = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count(); };

    // 0. host side first: the copier threads start faulting the destination in while the launches below are issued
    char* dst = reinterpret_cast<char*>(host_out);
    const char* src = reinterpret_cast<const char*>(c->h_staging);
    if (huge && fault_mode != 0) {
        const uintptr_t a = (reinterpret_cast<uintptr_t>(dst) + (size_t(2) << 20) - 1) & ~((uintptr_t(2) << 20) - 1);
        const uintptr_t e = (reinterpret_cast<uintptr_t>(dst) + bytes) & ~((uintptr_t(2) << 20) - 1);
        if (e > a) (void)madvise(reinterpret_cast<void*>(a), e - a, MADV_HUGEPAGE);  // a hint; failure is harmless
    }
    for (int j = 0; j < ranges; j++) c->chunk_ready[j].store(0, std::memory_order_relaxed);
    const int threads = c->copy_threads;
    std::atomic<int>* ready = c->chunk_ready;
    const size_t* first = c->chunk_first;
    const std::function<int(int)> copy_job = [=](int w) -> int {
        if (w >= threads) return 0;
        int faulted = 0;  // ranges whose share this thread has faulted in already
        for (int j = 0; j < ranges; j++) {
            size_t lo, hi;
            while (ready[j].load(std::memory_order_acquire) == 0) {
                if (faulted < ranges) {
                    thread_share(first[faulted] * sizeof(float), first[faulted + 1] * sizeof(float), w, threads, &lo, &hi);
                    fault_in(dst, lo, hi, fault_mode);
                    faulted++;
                } else {
                    _mm_pause();
                }
            }
            if (ready[j].load(std::memory_order_acquire) < 0) return 0;  // the evaluation failed: nothing to copy
            thread_share(first[j] * sizeof(float), first[j + 1] * sizeof(float), w, threads, &lo, &hi);
            if (lo < hi) memcpy(dst + lo, src + lo, hi - lo);
        }
        return 0;
    };
    c->copy_pool->start(copy_job);
    const double t_pool = trace ? since() : 0.0;
    // every exit below has to release the copier threads first
    auto abort_copy = [&](int rc) {
        for (int j = 0; j < ranges; j++) c->chunk_ready[j].store(-1, std::memory_order_release);
        c->copy_pool->wait();
        return rc;
    };

    // 1. reference-side tables, once
    if (int r = compute_impl(c, p, device_reference_values, nullptr, nullptr, 1u, -1, ov)) return abort_copy(r);
    const double t_prep = trace ? since() : 0.0;
    if (two_streams) {
        if (hipEventRecord(c->prep_done, c->stream) != hipSuccess || hipStreamWaitEvent(c->stream2, c->prep_done, 0) != hipSuccess)
            return abort_copy(fail(c, CRF_ERR_DEVICE, "ordering the second stream after the preparation failed"));
    }
    // 2. per-voxel kernels, range by range, alternating between two streams (the next range fills the GPU while the
    //    last waves of the previous one drain); results go straight to the mapped staging buffer, or to HBM + DMA
    float* out_base = dma ? c->d_out : c->d_staging;
    {
        RangeScope scope(c);
        for (int j = 0; j < ranges; j++) {
            scope.select(j);
            hipStream_t s = (two_streams && !(j & 1)) ? c->stream2 : c->stream;  // range 0 on the second stream
            float* out = out_base + c->chunk_first[j];
            if (int r = compute_impl(c, p, nullptr, out, s, 2u, -1)) return abort_copy(r);
            if (p->flags & CRF_FLAG_ABSOLUTE_VALUE)
                if (launch_abs(out, c->num_voxels, s) != hipSuccess) return abort_copy(fail(c, CRF_ERR_DEVICE, "launch_abs failed"));
            if (hipEventRecord(c->chunk_done[j], s) != hipSuccess) return abort_copy(fail(c, CRF_ERR_DEVICE, "hipEventRecord failed"));
            if (dma) {
                const size_t off = c->chunk_first[j], count = c->chunk_first[j + 1] - off;
                if (hipStreamWaitEvent(c->copy_stream, c->chunk_done[j], 0) != hipSuccess ||
                    hipMemcpyAsync(c->h_staging + off, c->d_out + off, count * sizeof(float), hipMemcpyDeviceToHost,
                                   c->copy_stream) != hipSuccess ||
                    hipEventRecord(c->chunk_copied[j], c->copy_stream) != hipSuccess)
                    return abort_copy(fail(c, CRF_ERR_DEVICE, "enqueueing the copy of a result range failed"));
            }
        }
    }
    const double t_issued = trace ? since() : 0.0;
    // 3. release each range to the copier threads as it lands in the staging buffer
    for (int j = 0; j < ranges; j++) {
        const hipError_t e = crf::spin_on_event(dma ? c->chunk_copied[j] : c->chunk_done[j]);
        if (e != hipSuccess) return abort_copy(fail(c, CRF_ERR_DEVICE, fmt("waiting for result range %d failed: %s", j, hipGetErrorString(e))));
        c->chunk_ready[j].store(1, std::memory_order_release);
        if (trace) fprintf(stderr, "crf_compute: range %d (%zu voxels) landed at %.0f us\n", j, c->chunk_first[j + 1] - c->chunk_first[j], since());
    }
    c->copy_pool->wait();
    if (trace) fprintf(stderr, "crf_compute: copier pool started by %.0f us, preparation launched by %.0f us, launches issued by %.0f us, "
                               "copied out by %.0f us (%d ranges, %d copier threads)\n", t_pool, t_prep, t_issued, since(), ranges, threads);
    return CRF_OK;
}

}  // namespace crf

extern "C" {

int crf_compute_requests_device(crf_context* c, const crf_params* p, const void* device_requests, size_t num_requests,
                                void* device_out, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!p || (num_requests && (!device_requests || !device_out))) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (p->measure < CRF_PEARSON || p->measure > CRF_KMI_CC)
        return fail(c, CRF_ERR_ARGUMENT, fmt("unknown measure %d", p->measure));
    for (int v : p->reserved)
        if (v != 0) return fail(c, CRF_ERR_ARGUMENT, "crf_params.reserved must be zero");
    if (c->cs > crf::kMaxGenericMembers)
        return fail(c, CRF_ERR_UNSUPPORTED, fmt("pair requests support at most %d members", crf::kMaxGenericMembers));
    if (c->windowed)
        return fail(c, CRF_ERR_UNSUPPORTED, "pair requests address voxels with 32-bit byte offsets: member volumes of 4 GiB or more are not supported in request mode");
    if ((p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC) && (p->num_bins < 1 || p->num_bins > 255))
        return fail(c, CRF_ERR_ARGUMENT, fmt("num_bins %d outside [1,255]", p->num_bins));
    if ((p->measure == CRF_MI_KRASKOV || p->measure == CRF_KMI_CC) && p->k < 1)
        return fail(c, CRF_ERR_ARGUMENT, fmt("k=%d must be at least 1", p->k));
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (int r = ensure_workspace(c, crf::pair_workspace_bytes(c->cs, num_requests))) return r;
    // two-field request mode: the j side reads the secondary members (CRF_FLAG_QUERY_FROM_SECONDARY)
    const float* const* members_j = c->d_member_table;
    if (p->flags & CRF_FLAG_QUERY_FROM_SECONDARY) {
        if (c->sec_members.empty())
            return fail(c, CRF_ERR_STATE, "CRF_FLAG_QUERY_FROM_SECONDARY needs secondary members (crf_upload_secondary_members)");
        members_j = c->d_sec_table;
    }
    const crf::PairArgs a{p->measure, p->num_bins, p->k, (p->flags & CRF_FLAG_ABSOLUTE_VALUE) ? 1 : 0, 0, 0.f, 0.f, 0.f, 0.f,
                          kraskov_c_term(p->k > 0 ? p->k : 1, 1)};
    // Spearman / Kendall up to 128 members: the sort-based two-vector kernels (kernels_symmetric.hip) in request mode
    hipError_t e = hipErrorNotSupported;
    const char* force_generic = getenv("CRF_REQUESTS_GENERIC");  // tuning / tests: the counting kernel
    if (!(force_generic && *force_generic == '1') && p->measure == CRF_PEARSON) {
        e = crf::launch_pearson_requests(c->d_member_table, members_j, c->cs, c->xs, c->ys, c->num_voxels,
                                         static_cast<const uint32_t*>(device_requests), num_requests, a.use_abs,
                                         static_cast<float*>(device_out), s);
        c->last_kernel = "pearson_request_kernel";
    } else if (!(force_generic && *force_generic == '1') && (p->measure == CRF_MI_BINNED || p->measure == CRF_BINNED_MI_CC)) {
        e = crf::launch_sorted_requests_binned(c->d_member_table, members_j, c->cs, c->xs, c->ys, c->num_voxels,
                                               static_cast<const uint32_t*>(device_requests), num_requests, p->measure,
                                               p->num_bins, a.use_abs, c->d_tables, static_cast<float*>(device_out), s);
        c->last_kernel = "sorted_request_kernel";
    } else if (!(force_generic && *force_generic == '1')) {
        e = crf::launch_sorted_requests(c->d_member_table, members_j, c->cs, c->xs, c->ys, c->num_voxels,
                                        static_cast<const uint32_t*>(device_requests), num_requests, p->measure,
                                        a.use_abs, static_cast<float*>(device_out), s);
        c->last_kernel = "sorted_request_kernel";
    }
    if (e == hipErrorNotSupported) {
        e = crf::launch_pair_requests(c->d_member_table, members_j, c->cs, c->xs, c->ys, c->num_voxels,
                                      static_cast<const uint32_t*>(device_requests), num_requests, a, c->d_tables,
                                      c->d_workspace, static_cast<float*>(device_out), s);
        c->last_kernel = "pair_request_kernel";
    }
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_compute_requests(crf_context* c, const crf_params* p, const crf_request* host_requests, size_t num_requests,
                         float* host_out) {
    if (int r = check_ready(c)) return r;
    if (num_requests == 0) return CRF_OK;
    if (!host_requests || !host_out) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    for (size_t r = 0; r < num_requests; r++) {
        const crf_request& q = host_requests[r];
        if (q.xi >= uint32_t(c->xs) || q.yi >= uint32_t(c->ys) || q.zi >= uint32_t(c->zs) || q.xj >= uint32_t(c->xs) ||
            q.yj >= uint32_t(c->ys) || q.zj >= uint32_t(c->zs))
            return fail(c, CRF_ERR_ARGUMENT, fmt("request %zu addresses a voxel outside the grid", r));
    }
    if (int r = bind_device(c)) return r;
    if (num_requests > c->request_capacity) {
        if (c->d_requests) (void)hipFree(c->d_requests);
        if (c->d_request_out) (void)hipFree(c->d_request_out);
        c->d_requests = nullptr;
        c->d_request_out = nullptr;
        c->request_capacity = 0;
        CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_requests), num_requests * sizeof(crf_request)));
        CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_request_out), num_requests * sizeof(float)));
        c->request_capacity = num_requests;
    }
    CRF_HIP(c, hipMemcpyAsync(c->d_requests, host_requests, num_requests * sizeof(crf_request), hipMemcpyHostToDevice,
                              c->stream));
    if (int r = crf_compute_requests_device(c, p, c->d_requests, num_requests, c->d_request_out, nullptr)) return r;
    CRF_HIP(c, hipMemcpyAsync(host_out, c->d_request_out, num_requests * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CRF_HIP(c, hipStreamSynchronize(c->stream));
    return CRF_OK;
}

int crf_compute_ensemble_stat_device(crf_context* c, int stat, void* device_out, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!device_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (stat != CRF_ENSEMBLE_MEAN && stat != CRF_ENSEMBLE_SPREAD)
        return fail(c, CRF_ERR_ARGUMENT, fmt("unknown ensemble statistic %d", stat));
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        e0 = take_event(c);
        e1 = take_event(c);
    }
    crf::LaunchInfo info;
    hipError_t e = hipSuccess;
    (void)for_each_window(c, static_cast<float*>(device_out), [&](float* o) {
        if (e == hipSuccess) e = crf::launch_ensemble_stat(stat, c->d_member_table, c->cs, c->num_voxels, o, s, e0, e1, &info);
        return CRF_OK;
    });
    c->last_kernel = info.kernel_name ? info.kernel_name : "";
    if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_compute_ensemble_stat(crf_context* c, int stat, float* host_out) {
    if (int r = check_ready(c)) return r;
    if (!host_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (int r = bind_device(c)) return r;
    if (!c->d_out) CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), c->num_voxels * sizeof(float)));
    if (int r = crf_compute_ensemble_stat_device(c, stat, c->d_out, nullptr)) return r;
    return copy_result_to_host(c, c->d_out, host_out, c->num_voxels);
}

int crf_compute_set_predicate_device(crf_context* c, int op, float comparison_value, int count_lower, int count_upper,
                                     void* device_out, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!device_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (op < CRF_CMP_GREATER || op > CRF_CMP_NOT_EQUAL)
        return fail(c, CRF_ERR_ARGUMENT, fmt("unknown comparison operator %d", op));
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        e0 = take_event(c);
        e1 = take_event(c);
    }
    crf::LaunchInfo info;
    hipError_t e = hipSuccess;
    (void)for_each_window(c, static_cast<float*>(device_out), [&](float* o) {
        if (e == hipSuccess)
            e = crf::launch_set_predicate(c->d_member_table, c->cs, c->num_voxels, op, comparison_value, count_lower, count_upper,
                                          o, s, e0, e1, &info);
        return CRF_OK;
    });
    c->last_kernel = info.kernel_name ? info.kernel_name : "";
    if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_compute_set_predicate(crf_context* c, int op, float comparison_value, int count_lower, int count_upper,
                              float* host_out) {
    if (int r = check_ready(c)) return r;
    if (!host_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (int r = bind_device(c)) return r;
    if (!c->d_out) CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), c->num_voxels * sizeof(float)));
    if (int r = crf_compute_set_predicate_device(c, op, comparison_value, count_lower, count_upper, c->d_out, nullptr))
        return r;
    return copy_result_to_host(c, c->d_out, host_out, c->num_voxels);
}

int crf_compute_dkl_device(crf_context* c, int estimator, int num_bins, int k, void* device_out, void* stream) {
    if (int r = check_ready(c)) return r;
    if (!device_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (estimator != CRF_DKL_BINNED && estimator != CRF_DKL_ENTROPY_KNN)
        return fail(c, CRF_ERR_ARGUMENT, fmt("unknown DKL estimator %d", estimator));
    if (c->cs > crf::kMaxGenericMembers)
        return fail(c, CRF_ERR_UNSUPPORTED, fmt("DKL supports at most %d members", crf::kMaxGenericMembers));
    if (estimator == CRF_DKL_BINNED && (num_bins < 1 || num_bins > 1024))
        return fail(c, CRF_ERR_ARGUMENT, fmt("num_bins %d outside [1,1024]", num_bins));
    if (estimator == CRF_DKL_ENTROPY_KNN && c->cs > 1 && (k < 1 || k >= c->cs))
        return fail(c, CRF_ERR_ARGUMENT, fmt("k=%d must be in [1, cs-1=%d]", k, c->cs - 1));
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (int r = ensure_workspace(c, crf::dkl_workspace_bytes(c->cs, estimator, num_bins, std::min(c->num_voxels, kWindowVoxels)))) return r;
    // psi(n) = -gamma + H_{n-1} (boost::math::digamma at positive integers, DKL.cpp:156)
    const double knn_const =
        estimator == CRF_DKL_ENTROPY_KNN && c->cs > 1 ? psi_int(c->cs) - psi_int(k) + std::log(2.0) : 0.0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->profiling) {
        e0 = take_event(c);
        e1 = take_event(c);
    }
    crf::LaunchInfo info;
    hipError_t e = hipSuccess;
    (void)for_each_window(c, static_cast<float*>(device_out), [&](float* o) {
        if (e == hipSuccess)
            e = crf::launch_dkl(c->d_member_table, c->cs, c->num_voxels, estimator, num_bins, k, knn_const, c->d_workspace, o, s,
                                e0, e1, &info);
        return CRF_OK;
    });
    c->last_kernel = info.kernel_name ? info.kernel_name : "";
    if (e0 && e1) c->ev_pending.emplace_back(e0, e1);
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_compute_dkl(crf_context* c, int estimator, int num_bins, int k, float* host_out) {
    if (int r = check_ready(c)) return r;
    if (!host_out) return fail(c, CRF_ERR_ARGUMENT, "null output");
    if (int r = bind_device(c)) return r;
    if (!c->d_out) CRF_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_out), c->num_voxels * sizeof(float)));
    if (int r = crf_compute_dkl_device(c, estimator, num_bins, k, c->d_out, nullptr)) return r;
    return copy_result_to_host(c, c->d_out, host_out, c->num_voxels);
}

double crf_max_mutual_information_kraskov(int k, int cs) {
    if (k < 1 || cs < 1) return std::numeric_limits<double>::quiet_NaN();
    return psi_int(cs) - psi_int(k);
}

size_t crf_tiled_element_count(int xs, int ys, int zs) {
    if (xs <= 0 || ys <= 0 || zs <= 0) return 0;
    return size_t((xs + 7) / 8) * size_t((ys + 7) / 8) * size_t((zs + 3) / 4) * 256;
}

int crf_tile_field_device(crf_context* c, const void* device_linear, void* device_tiled, void* stream) {
    if (!c) return CRF_ERR_ARGUMENT;
    if (c->cs <= 0) return fail(c, CRF_ERR_STATE, "crf_set_grid has not been called");
    if (c->windowed) return fail(c, CRF_ERR_UNSUPPORTED, "re-tiling a field of 4 GiB or more is not supported");
    if (!device_linear || !device_tiled) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    hipError_t e = crf::launch_tile_field(static_cast<const float*>(device_linear), static_cast<float*>(device_tiled),
                                          c->xs, c->ys, c->zs, s);
    if (e != hipSuccess) return fail(c, CRF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return CRF_OK;
}

int crf_set_profiling(crf_context* c, int enabled) {
    if (!c) return CRF_ERR_ARGUMENT;
    c->profiling = enabled != 0;
    return CRF_OK;
}

int crf_take_kernel_time(crf_context* c, double* out_ms_sum, int* out_launches) {
    if (!c || !out_ms_sum || !out_launches) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (int r = bind_device(c)) return r;
    double sum = 0.0;
    int n = 0;
    for (auto& p : c->ev_pending) {
        CRF_HIP(c, hipEventSynchronize(p.second));
        float ms = 0.f;
        CRF_HIP(c, hipEventElapsedTime(&ms, p.first, p.second));
        sum += double(ms);
        n++;
        c->ev_free.push_back(p.first);
        c->ev_free.push_back(p.second);
    }
    c->ev_pending.clear();
    *out_ms_sum = sum;
    *out_launches = n;
    return CRF_OK;
}

const char* crf_last_kernel_name(const crf_context* c) { return c ? c->last_kernel.c_str() : ""; }

int crf_synth_box_member(crf_context* c, void* device_out, int xs, int ys, int zs_local, int z_begin, int zs_global,
                         int member, int cs, uint64_t seed, void* stream) {
    if (!c || !device_out) return fail(c, CRF_ERR_ARGUMENT, "null argument");
    if (xs <= 0 || ys <= 0 || zs_local <= 0 || zs_global <= 0 || z_begin < 0 || member < 0 || member >= cs)
        return fail(c, CRF_ERR_ARGUMENT, "invalid synthetic volume description");
    if (int r = bind_device(c)) return r;
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    CRF_HIP(c, crf::launch_synth_box_member(static_cast<float*>(device_out), xs, ys, zs_local, z_begin, zs_global,
                                            member, cs, seed, s));
    return CRF_OK;
}

}  // extern "C"
