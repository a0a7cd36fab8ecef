// crf_internal.h -- declarations shared by the C-ABI (api.cpp) and the gfx950 kernel translation units.
// Everything here is internal to libcorrfield.so; the public surface is include/corrfield.h.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace crf {

// Largest member count the register-resident kernels are instantiated for; above it the streaming
// (re-reading) variants run.
constexpr int kMaxRegisterMembers = 384;  // 257..384: VGPRs + AGPRs + a little scratch at one wave per SIMD
// Largest member count supported at all by the sort-based estimators (LDS / register budgets).
constexpr int kMaxSortMembers = 128;
// Prepared reference-derived table: floats (see each kernels_*.hip for its layout).
constexpr size_t kPrepBytes = 160 * 1024;  // (r02: room for the Kraskov x-distance table of 128 members, 133 KB)
// Largest member count of the generic (any-cs) kernels: bounded by the preparation scratch (2*cs ints / doubles).
constexpr int kMaxGenericMembers = 2048;
// where binned_prep_kernel leaves the reference-side entropy sum (fp64) inside the preparation scratch
constexpr size_t kBinnedSxOffset = kPrepBytes - 16;

// Where a preparation kernel takes the reference vector from: an explicit device array of cs floats (SEPARATE mode,
// or a vector received from another rank), or -- fused gather -- members[c][voxel]
// (referenceValues[c] = fields[c][IDXS(ref)], CorrelationCalculator.cpp:802,815-817).
struct RefSource {
    const float* values;  // non-null: use values[c]
    size_t voxel;         // else: members[c][voxel]
    // which halves of an evaluation a launcher performs: the reference-side preparation (into d_prep) and/or the
    // per-voxel kernel (reading d_prep).  Split so that a caller can prepare several evaluations ahead of time on
    // another stream (crf_prepare_device) and keep only the per-voxel kernels on the critical path.
    unsigned phase = 3;
    // non-null: the fused gather reads table[c][voxel] instead of the launcher's own member table -- the member table of
    // ANOTHER context whose slab holds the reference point (crf_group, direct exchange: same device, or a peer device
    // whose memory this one may read over xGMI)
    const float* const* table = nullptr;
    bool prepare() const { return (phase & 1u) != 0; }
    bool run() const { return (phase & 2u) != 0; }
};

struct LaunchInfo {
    const char* kernel_name = "";  // dominant per-voxel kernel (for rocprof row matching)
};

// ---- kernels_common.hip -----------------------------------------------------------------------------------
hipError_t launch_gather_reference(const float* const* d_members, int cs, size_t voxel, float* d_out, hipStream_t s);
constexpr int kMaxGatherRows = 32;
constexpr size_t kNoVoxel = ~size_t(0);
struct GatherRows {
    size_t voxel[kMaxGatherRows];  // kNoVoxel: the row is zero-filled
};
hipError_t launch_gather_reference_rows(const float* const* d_members, int cs, const GatherRows& rows, int num_rows,
                                        float* d_out, hipStream_t s);
hipError_t launch_minmax(const float* const* d_members, int cs, size_t num_voxels, uint32_t* d_keys /*[2]*/,
                         hipStream_t s);
float minmax_key_to_float(uint32_t key);
hipError_t launch_synth_box_member(float* d_out, int xs, int ys, int zs_local, int z_begin, int zs_global, int c,
                                   int cs, uint64_t seed, hipStream_t s);

// ---- kernels_pearson.hip ----------------------------------------------------------------------------------
// d_ref: cs reference values on the device.  d_prep: scratch of kPrepBytes.  Writes num_voxels floats to d_out.
// max_vpt: widest per-lane vector (1, 2 or 4 floats) the member/output pointers are aligned for.
hipError_t launch_pearson(const float* const* d_members, int cs, size_t num_voxels, int max_vpt, const RefSource& ref,
                          float* d_prep, float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                          LaunchInfo* info);
hipError_t launch_fill(float* d_out, size_t n, float value, hipStream_t s);
hipError_t launch_abs(float* d_out, size_t n, hipStream_t s);  // in place |.| (CRF_FLAG_ABSOLUTE_VALUE on a field)

// ---- kernels_rank.hip (Spearman, Kendall) ---------------------------------------------------------------
// d_todo: num_voxels + 1 uint32 (count, then voxel indices) used by the split-sort kernels (64 < cs <= 128) to defer
// voxels that contain ties to the monolithic kernel; may be null (then the monolithic kernel does everything).
hipError_t launch_spearman(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                           float* d_prep, uint32_t* d_todo, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                           hipEvent_t ev_end, LaunchInfo* info);
hipError_t launch_kendall(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                          float* d_prep, uint32_t* d_todo, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                          hipEvent_t ev_end, LaunchInfo* info);

// ---- kernels_binned.hip / kernels_kraskov.hip -------------------------------------------------------------------
struct BinnedArgs {
    int num_bins;
    float min_ref, max_ref, min_query, max_query;
    bool to_cc;
};
// d_tables (fp64, built on the host per member count, see build_tables() in api.cpp):
//   [0, cs]            psi(n) = -gamma + H_{n-1}  (psi(0) = NaN)
//   [cs+1, 2cs+1]      T[c] = (c/cs) * ln(c/cs)   (T[0] = 0)
//   [2cs+2, 3cs+2)     noise_ref[e]   = double(u_e) * 1e-10, xorshift32 stream seeded 617406168
//   [3cs+2, 4cs+2)     noise_query[e] = double(u_e) * 1e-10, xorshift32 stream seeded 864730169
hipError_t launch_mi_binned(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                            const BinnedArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                            hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info);
hipError_t launch_mi_binned_hist(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                 const BinnedArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                                 hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info);
struct KraskovArgs {
    int k;
    int estimator;  // 1 or 2
    bool to_cc;
    // psi(k) for KSG-1, psi(k) - 1/k for KSG-2 (MutualInformation.cpp:438,503), evaluated on the host: k may exceed the
    // member count (the reference accepts any k >= 1; its kd-tree just returns at most cs points), the device psi
    // table ends at cs
    double c_term;
};
hipError_t launch_mi_kraskov(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                             const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                             hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info);

hipError_t launch_mi_kraskov_direct(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                    const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out,
                                    hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info);

hipError_t launch_mi_kraskov_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                       size_t num_voxels, int k, double c_term, bool to_cc, const double* d_tables,
                                       float* d_out, hipStream_t s);

// ---- kernels_generic.hip: any member count (O(cs^2) counting algorithms, runtime loops) ------------------
struct GenericArgs {
    int measure;  // crf_measure value
    int num_bins;
    float min_ref, max_ref, min_query, max_query;
    int k, estimator;
    double kraskov_c;  // KraskovArgs::c_term
};
// workspace: generic_workspace_bytes(cs, num_voxels) bytes of device memory (per-block voxel tiles)
size_t generic_workspace_bytes(int cs, size_t num_voxels);
// d_todo: num_voxels + 1 uint32, or null (Spearman / Kendall at 129..256 members defer voxels with ties through it)
hipError_t launch_generic(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                          const GenericArgs& a, const double* d_tables, float* d_prep, unsigned char* d_workspace,
                          float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info,
                          uint32_t* d_todo = nullptr);
// kernels_rank.hip: Spearman at 129..256 members (spearman_pair_kernel); false if it does not apply.  The caller runs the
// counting kernel over d_todo afterwards.
bool launch_spearman_pair(const float* const* d_members, const float* d_prep, float* d_out, size_t num_voxels, int cs,
                          uint32_t* d_todo, hipStream_t s);
bool launch_kendall_pair(const float* const* d_members, const int* d_prep, float* d_out, size_t num_voxels, int cs,
                         uint32_t* d_todo, hipStream_t s);
// ---- kernels_stats.hip: ensemble mean (kind 0) / spread (kind 1) ------------------------------------------
hipError_t launch_ensemble_stat(int kind, const float* const* d_members, int cs, size_t num_voxels, float* d_out,
                                hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info);

hipError_t launch_set_predicate(const float* const* d_members, int cs, size_t num_voxels, int op, float comparison_value,
                                int count_lower, int count_upper, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                                hipEvent_t ev_end, LaunchInfo* info);
hipError_t launch_tile_field(const float* d_linear, float* d_tiled, int xs, int ys, int zs, hipStream_t s);

// ---- kernels_dkl.hip: DKLCalculator (estimator 0 = binned, 1 = entropy k-NN) --------------------------------
size_t dkl_workspace_bytes(int cs, int estimator, int num_bins, size_t num_voxels);
hipError_t launch_dkl(const float* const* d_members, int cs, size_t num_voxels, int estimator, int num_bins, int k,
                      double knn_const, unsigned char* d_workspace, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                      hipEvent_t ev_end, LaunchInfo* info);

// pair-request mode (kernels_generic.hip): requests = 8 uint32 each {xi,yi,zi,i,xj,yj,zj,j}; voxel i is read from
// d_members_i, voxel j from d_members_j.  d_requests == nullptr: request r = voxel pair (r, r) (symmetric field mode).
struct PairArgs {
    int measure, num_bins, k, use_abs;
    int fixed_ranges;  // binned MI: 0 = normalise with the pair's own extrema (HEBChart), 1 = with the ranges below
    float min_ref, max_ref, min_query, max_query;
    double kraskov_c;  // psi(k) (pair requests are KSG-1), host-evaluated like KraskovArgs::c_term
};
size_t pair_workspace_bytes(int cs, size_t num_requests);
hipError_t launch_pair_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                size_t num_voxels, const uint32_t* d_requests, size_t num_requests, const PairArgs& a,
                                const double* d_tables, unsigned char* d_workspace, float* d_out, hipStream_t s);
// symmetric field mode, Spearman (measure 1) / Kendall (2) / binned MI (3, 5), any member count (kernels_generic.hip)
size_t direct_symmetric_workspace_bytes(int cs, size_t num_voxels, int measure);
// sort-based symmetric kernels (kernels_symmetric.hip): 2 <= cs <= kMaxSortMembers, measures 1, 2, 3, 5
hipError_t launch_sorted_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                   size_t num_voxels, int measure, int num_bins, float min_x, float max_x, float min_y,
                                   float max_y, const double* d_tables, float* d_out, hipStream_t s);
hipError_t launch_sorted_requests_binned(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs,
                                         int ys, size_t num_voxels, const uint32_t* d_requests, size_t num_requests,
                                         int measure, int num_bins, int use_abs, const double* d_tables, float* d_out,
                                         hipStream_t s);
hipError_t launch_pearson_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                   size_t num_voxels, const uint32_t* d_requests, size_t num_requests, int use_abs,
                                   float* d_out, hipStream_t s);
hipError_t launch_sorted_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                  size_t num_voxels, const uint32_t* d_requests, size_t num_requests, int measure,
                                  int use_abs, float* d_out, hipStream_t s);
hipError_t launch_sorted_symmetric_binned(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                          size_t num_voxels, int measure, int num_bins, float min_x, float max_x,
                                          float min_y, float max_y, const double* d_tables, float* d_out, hipStream_t s);
hipError_t launch_direct_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                   size_t num_voxels, int measure, int num_bins, float min_x, float max_x, float min_y,
                                   float max_y, const double* d_tables, unsigned char* d_workspace, float* d_out,
                                   hipStream_t s);
// symmetric field mode, Pearson, members resident in registers (kernels_pearson.hip); hipErrorNotSupported above
// kMaxSymmetricRegisterMembers (the caller then uses launch_pair_requests)
constexpr int kMaxSymmetricRegisterMembers = 128;
hipError_t launch_pearson_symmetric(const float* const* d_members_ref, const float* const* d_members_query, int cs,
                                    size_t num_voxels, float* d_out, hipStream_t s);
// preparation launchers shared with the generic path (kernels_rank.hip / kernels_binned.hip / kernels_kraskov.hip); n_pad = table stride
void launch_spearman_prep(const RefSource& ref, const float* const* d_members, int cs, float* d_prep, hipStream_t s);
void launch_kendall_prep(const RefSource& ref, const float* const* d_members, int cs, int n_pad, int* d_prep,
                         hipStream_t s);
void launch_binned_prep(const RefSource& ref, const float* const* d_members, int cs, int n_pad, const BinnedArgs& a,
                        const double* tableT, int* d_prep, hipStream_t s);
void launch_kraskov_prep(const RefSource& ref, const float* const* d_members, int cs, const double* noise_ref,
                         double* d_prep, hipStream_t s);

}  // namespace crf
