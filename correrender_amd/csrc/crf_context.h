// crf_context.h -- the state behind a crf_context (include/corrfield.h), shared by api.cpp and group.cpp.
// Internal to libcorrfield.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/corrfield.h"
#include "crf_internal.h"
#include "crf_pool.h"

constexpr int kMaxHostChunks = 16;  // z-chunks of a host-output evaluation (kernel of chunk i+1 under the D2H of chunk i)

struct crf_context {
    int device = -1;
    hipStream_t stream = nullptr;  // the context's own stream (used when the caller passes none)
    std::string err;
    int xs = 0, ys = 0, zs = 0, cs = 0;
    size_t num_voxels = 0;
    // members
    void* owned_block = nullptr;  // one allocation holding every uploaded member (stride owned_stride floats)
    size_t owned_stride = 0;
    std::vector<const float*> members;  // cs device pointers (owned or borrowed)
    const float** d_member_table = nullptr;
    int max_vpt = 1;
    // secondary members (second scalar field of the SEPARATE / SEPARATE_SYMMETRIC modes), optional
    void* sec_owned_block = nullptr;
    std::vector<const float*> sec_members;
    const float** d_sec_table = nullptr;
    bool sec_minmax_valid = false;
    float sec_min_v = 0.f, sec_max_v = 0.f;
    // scratch
    float* d_ref = nullptr;    // cs reference values
    float* d_prep = nullptr;   // crf::kPrepBytes
    float* d_prep_slots = nullptr;  // CRF_PREPARED_SLOTS x crf::kPrepBytes, lazily (crf_prepare_device)
    float* d_out = nullptr;    // num_voxels floats, lazily (crf_compute only)
    double* d_tables = nullptr;  // psi / p ln p / noise tables for this member count (crf_internal.h)
    uint32_t* d_todo = nullptr;  // deferred-voxel list of the split-sort rank kernels, lazily (num_voxels + 1)
    unsigned char* d_workspace = nullptr;  // voxel tiles of the generic (cs > 128) kernels, lazily
    size_t workspace_bytes = 0;
    uint32_t* d_requests = nullptr;  // staging of host pair requests / their results, lazily
    float* d_request_out = nullptr;
    size_t request_capacity = 0;
    uint32_t* d_minmax = nullptr;
    bool minmax_valid = false;
    float min_v = 0.f, max_v = 0.f;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev_free;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::string last_kernel;
    // host-output evaluations (crf_compute): the grid in up to kMaxHostChunks voxel ranges, one member-pointer table per
    // range; the per-voxel kernel of a range stores into the pinned, device-mapped staging buffer and a pool of host
    // threads moves finished ranges into the caller's buffer (api.cpp: compute_to_host)
    const float** d_chunk_tables = nullptr;  // host_chunks x cs pointers
    int host_chunks = 0;                     // 0: tables not built for the current members
    size_t chunk_first[kMaxHostChunks + 1] = {};  // first voxel of every range; [host_chunks] = alloc_voxels
    std::atomic<int> chunk_ready[kMaxHostChunks] = {};  // 1: range landed in the staging buffer, -1: evaluation failed
    hipStream_t stream2 = nullptr;           // odd ranges (the next range fills the GPU while the previous one drains)
    hipStream_t copy_stream = nullptr;       // DMA form only (CRF_HOST_PATH=dma, CRF_FLAG_ABSOLUTE_VALUE)
    hipEvent_t prep_done = nullptr;
    hipEvent_t chunk_done[kMaxHostChunks] = {};    // range evaluated
    hipEvent_t chunk_copied[kMaxHostChunks] = {};  // DMA form: range landed in the staging buffer (copy stream)
    float* h_staging = nullptr;                    // pinned + mapped, alloc_voxels floats, lazily
    float* d_staging = nullptr;                    // its device address
    std::unique_ptr<crf::SpinPool> copy_pool;      // copier threads, lazily
    int copy_threads = 0;                          // how many of them a copy uses (calibrated at first use)
    int copy_threads_cap = 0;                      // > 0: upper bound set by the owner (a device group shares the host)
    // member volumes of 4 GiB or more: evaluated in windows (api.cpp: ensure_windows)
    bool windowed = false;
    const float** d_window_tables = nullptr;  // windows x (1 or 2) x cs pointers (primary [, secondary] members)
    int windows = 0;                          // 0: tables not built for the current members
    bool window_has_secondary = false;
    size_t alloc_voxels = 0;  // voxels of the whole local grid (num_voxels is narrowed while a chunk is being launched)
};


// api.cpp internals used by the device group (group.cpp)
namespace crf {
// waits for an event by polling hipEventQuery (a render thread blocked in calculateCpu has nothing else to do, and the
// runtime's sleeping wait costs tens of microseconds per wake-up); falls back to hipEventSynchronize after 50 ms
inline hipError_t spin_on_event(hipEvent_t e) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned i = 1;; i++) {
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        if ((i & 1023u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.05)
            return hipEventSynchronize(e);
        _mm_pause();
    }
}
// One evaluation straight into a caller-owned HOST buffer of the local grid: reference-side preparation once, then the
// per-voxel kernel range by range with the D2H copies overlapped.  device_reference_values: device pointer to cs floats
// or null (then params->reference_values / the reference point are used).
// ref_override (crf_group, direct exchange): the reference-side preparation reads its cs values from table[c][voxel], the
// member table of ANOTHER context (the slab that holds the reference point), instead of a vector or the local members.
struct RefOverride {
    const float* const* table = nullptr;
    size_t voxel = 0;
};
int compute_to_host(crf_context* c, const crf_params* p, const void* device_reference_values, float* host_out,
                    const RefOverride* ref_override);
// crf_compute_device / crf_prepare_device with a direct reference read (ref_override may be null: the plain calls)
int compute_device_ex(crf_context* c, const crf_params* p, const void* device_reference_values, void* device_out,
                      void* stream, const RefOverride* ref_override);
int prepare_device_ex(crf_context* c, const crf_params* p, const void* device_reference_values, int slot, void* stream,
                      const RefOverride* ref_override);
// the context's second stream (created on first use)
int second_stream(crf_context* c, hipStream_t* out);
// the override that makes other contexts read the reference values of local point (x, y, z) out of `owner`'s members
int reference_override(crf_context* owner, bool secondary, int x, int y, int z, RefOverride* out);
// referenceValues[c] = (secondary ? secondary members : members)[c][IDXS(x,y,z)] into a device buffer, stream-ordered
int gather_reference_to(crf_context* c, bool secondary, int x, int y, int z, float* device_out, hipStream_t s);
}  // namespace crf
