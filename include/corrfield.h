/*
 * corrfield.h -- C ABI of libcorrfield.so, the MI355X (gfx950) correlation-field engine.
 *
 * Drop-in boundary for ONE path of chrismile/Correrender: the per-voxel ensemble correlation estimators evaluated
 * between one reference grid point and every voxel of a 3-D grid, i.e. what
 *     void CorrelationCalculator::calculateCpu(int timeStepIdx, int ensembleIdx, float* buffer)
 * (reference: src/Calculators/CorrelationCalculator.cpp:781-1154, declared src/Calculators/Calculator.hpp:123-124)
 * computes.  The reference has no C ABI (calculators are compiled-in C++ classes registered as factories,
 * src/Volume/VolumeData.cpp:198-232); these entry points are what a `Calculator` subclass living in the reference
 * tree would bind to hand the hot loop to the GPU -- see INTEGRATION.md for that subclass and
 * correrender_amd/csrc/host/ for this repo's mirror of the ICorrelationCalculator/VolumeData surface built on top.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ / torch types; every function returns 0 on success, non-zero on error
 *     (the reference reports errors through sgl::Logfile::throwError, e.g. VolumeData.cpp:1217-1221; the adapter
 *     maps a non-zero status + crf_last_error() onto that).
 *   - single caller thread per context (the reference calls calculators from the render thread only,
 *     VolumeData.cpp:1225,1469-1472); no re-entrancy.
 *   - the *_device entry points are asynchronous on the stream they are given, but a context owns ONE set of scratch
 *     buffers (reference vector, reference-side tables, deferred-voxel list, workspace): at most one evaluation per
 *     context may be in flight at a time unless consecutive calls are ordered on the same stream (or by events).  The
 *     only state meant for overlap across streams are the prepared slots of crf_prepare_device.  crf_set_grid and any
 *     call that has to grow a scratch buffer synchronise the device first.
 *   - volumes are fp32, x fastest: voxel (x,y,z) at z*xs*ys + y*xs + x  (IDXS, src/Loaders/DataSet.hpp:37);
 *     the ensemble is member-major SoA: one contiguous volume per member
 *     (std::vector<const float*> fields, CorrelationCalculator.cpp:791-800).
 *   - outputs are caller-owned: exactly xs*ys*zs floats are written, nothing is retained
 *     (VolumeData.cpp:1222-1226 allocates the buffer and wraps it in a HostCacheEntry).
 *   - There is NO CPU fallback: every compute entry point fails with CRF_ERR_DEVICE if no gfx950 device is usable.
 */
#ifndef CORRFIELD_H
#define CORRFIELD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct crf_context crf_context;

/* Same order and meaning as enum class CorrelationMeasureType (src/Calculators/CorrelationDefines.hpp:41-45);
 * the string ids are CORRELATION_MEASURE_TYPE_IDS (CorrelationDefines.hpp:54-57). */
typedef enum crf_measure {
    CRF_PEARSON = 0,                 /* "pearson"   computePearson2<float>, Correlation.cpp:100-133 */
    CRF_SPEARMAN = 1,                /* "spearman"  computeRanks + computePearson2<float>, :277-303,:141-174 */
    CRF_KENDALL = 2,                 /* "kendall"   computeKendall<int32_t> (tau-b), :423-455 */
    CRF_MI_BINNED = 3,               /* "mi_binned" computeMutualInformationBinned<double>, MutualInformation.cpp:45-143 */
    CRF_MI_KRASKOV = 4,              /* "mi_kraskov" computeMutualInformationKraskov[2]<double>, :399-509 */
    CRF_BINNED_MI_CC = 5,            /* "binned_mi_correlation_coefficient"  sqrt(1-exp(-2 MI)), CorrelationCalculator.cpp:1071-1073 */
    CRF_KMI_CC = 6                   /* "kmi_correlation_coefficient"        same map, :1130-1132 */
} crf_measure;

enum {
    CRF_OK = 0,
    CRF_ERR_ARGUMENT = 1,   /* bad pointer / size / enum */
    CRF_ERR_STATE = 2,      /* grid or members not set */
    CRF_ERR_DEVICE = 3,     /* HIP error or no usable gfx950 device */
    CRF_ERR_UNSUPPORTED = 4 /* configuration outside what the kernels implement (message says which) */
};

/* One evaluation = the state CorrelationCalculator holds when calculateCpu runs
 * (CorrelationCalculator.hpp:200-213 and ICorrelationCalculator members :120-140). */
typedef struct crf_params {
    int32_t measure;                 /* crf_measure */
    int32_t ref_x, ref_y, ref_z;     /* referencePointIndex in LOCAL grid coordinates; used when reference_values==NULL */
    int32_t k;                       /* kmi_neighbors (Kraskov); default max(ceil(3*cs/100),1), CorrelationCalculator.cpp:592-598 */
    int32_t kraskov_estimator_index; /* 1 = KSG-1 (default), 2 = KSG-2 */
    int32_t num_bins;                /* mi_bins, default 80 (CorrelationCalculator.hpp:209) */
    float min_ref, max_ref;          /* binned MI: extrema used to normalise the reference vector (:820-833) */
    float min_query, max_query;      /* binned MI: extrema used to normalise each voxel's vector (:834-846,:1061-1062) */
    const float* reference_values;   /* HOST pointer to cs floats, or NULL.  Non-NULL = CorrelationFieldMode::SEPARATE
                                        / time-lag (reference vector taken from another field, :804-813) or a vector
                                        received from another rank. */
    int32_t flags;                   /* CRF_FLAG_* */
    int32_t prepared_slot;           /* 0: the evaluation prepares its reference-side tables itself.  s+1: use the
                                        tables crf_prepare_device left in slot s (the reference vector arguments are
                                        then not read; the other fields must equal those given to crf_prepare_device) */
    int32_t reserved[2];             /* must be 0 */
} crf_params;

/* |.| of the result.  Pair requests: useAbsoluteCorrelationMeasure (HEBChartCorrelation.cpp:583-585).  Full-grid
 * evaluations: OPT-IN -- the reference's calculateCpu ignores calculate_absolute_value (it is a shader define of the
 * accelerator paths only, CorrelationCalculator.cpp:1662-1664), so the C++ adapter never sets the flag; a caller that
 * wants the accelerator paths' behaviour sets it and gets |value| (NaN stays NaN). */
#define CRF_FLAG_ABSOLUTE_VALUE 1
/* CorrelationFieldMode::SEPARATE_SYMMETRIC (CorrelationCalculator.hpp:59-64): crf_compute[_device] evaluate, at every
 * voxel v, the measure between X[c] = member_c[v] (the reference field = the primary members) and Y[c] =
 * secondary_member_c[v] (the query field) -- `#define referencePointIdx currentPointIdx`, CorrelationMain.glsl:10-15;
 * scalarFieldsRef = scalarFields, scalarFieldsQuery = scalarFieldsSecondary, ScalarFields.glsl:107-117.  The
 * reference implements this mode on its Vulkan path only (CorrelationCalculator.cpp:1182-1229; calculateCpu falls into
 * the SINGLE branch), so the arithmetic here is DEFINED as calculateCpu's per-voxel computation with the reference
 * vector taken from the primary field at the same voxel; a NaN in either vector gives NaN; Kraskov is KSG-1 (the
 * shaders ignore the estimator index).  Binned MI normalises X with min_ref/max_ref and Y with min_query/max_query.
 * The reference point / reference_values are not read. */
#define CRF_FLAG_SYMMETRIC 2
/* CorrelationFieldMode::SEPARATE on the device: the reference vector is gathered from the SECONDARY members at
 * (ref_x, ref_y, ref_z) -- referenceValues[c] = field2_c[IDXS(ref)], CorrelationCalculator.cpp:804-813 (for time-lag
 * correlations bind the secondary field's members of the lagged time step). */
#define CRF_FLAG_REFERENCE_FROM_SECONDARY 4
/* Pair requests over two fields (crf_compute_requests*): the i side of every request reads the primary members, the j
 * side the SECONDARY members -- the request mode with setUseSecondaryFields(true) / setFieldBuffersSecondary
 * (CorrelationCalculator.hpp:255-257, HEBChartCorrelation.cpp:1164-1169; scalarFieldsRef / scalarFieldsQuery in
 * Data/Shaders/Correlation/ScalarFields.glsl:105-120).  Both member sets share the local grid. */
#define CRF_FLAG_QUERY_FROM_SECONDARY 8

/* One pair request: the estimator between the ensemble vectors of voxel (xi,yi,zi) and voxel (xj,yj,zj).  Same layout
 * as struct CorrelationRequestData {xi,yi,zi,i,xj,yj,zj,j} (src/Renderers/Diagram/HEBChart.hpp:166-168,
 * Data/Shaders/Correlation/RequestsBuffer.glsl:22-36); i and j (linear indices) are carried along, not read. */
typedef struct crf_request {
    uint32_t xi, yi, zi, i, xj, yj, zj, j;
} crf_request;

/* ---- lifetime ------------------------------------------------------------------------------------------- */
/* Creates a context bound to HIP device `device_ordinal`.  Fails (CRF_ERR_DEVICE) when the device is absent or is
 * not gfx950.  *out_ctx is set to NULL on failure; crf_last_error(NULL) then holds the message. */
int crf_create(int device_ordinal, crf_context** out_ctx);
void crf_destroy(crf_context* ctx);
/* Last error message of this context (or of the calling thread's last failed crf_create when ctx==NULL). */
const char* crf_last_error(const crf_context* ctx);
/* ABI version of this header, bumped on incompatible change. */
int crf_abi_version(void);  /* 5: crf_group_compute_batch[_device], crf_member_minmax_divergent;
                               4: crf_group_* (several devices behind one caller thread), crf_set_kraskov_noise;
                               3: crf_params.reserved[0] became prepared_slot (same layout; 0 keeps the old meaning) */

/* ---- the ensemble (replaces the fields[] gather of CorrelationCalculator.cpp:791-800) --------------------- */
/* Declares the LOCAL grid (a whole grid, or this process's z-slab of one) and member count; drops any members. */
int crf_set_grid(crf_context* ctx, int xs, int ys, int zs, int cs);
/* Copies cs host volumes (each xs*ys*zs floats) into HBM owned by the context (the role the reference's LRU
 * field cache plays across reference-point moves, VolumeData.cpp:1202-1226). */
int crf_upload_members(crf_context* ctx, const float* const* host_members);
/* Uses cs caller-owned DEVICE volumes in place (borrowed until the next set_grid/upload/bind/destroy). */
int crf_bind_members_device(crf_context* ctx, const void* const* device_members);
/* min of per-member minima / max of per-member maxima over the local grid (CorrelationCalculator.cpp:822-829 on
 * top of VolumeData::getMinMaxScalarFieldValue, VolumeData.cpp:1632-1670); computed on the device, cached until
 * the members change. */
int crf_member_minmax(crf_context* ctx, float* out_min, float* out_max);
/* The same extrema for a field the reference flags as DIVERGENT (VolumeData::getIsScalarFieldDivergent,
 * VolumeData.cpp:616-621: the field named "Helicity"): getMinMaxScalarFieldValue centres its range at zero,
 * min = -max|.|, max = +max|.| (VolumeData.cpp:1661-1666), per member -- and the min of the mins / max of the maxes the
 * calculator takes (CorrelationCalculator.cpp:822-829) is then -A / +A with A = max(|min|, |max|) over all members.
 * secondary != 0: the secondary members.  The caller decides whether a field is divergent (it knows the field's name). */
int crf_member_minmax_divergent(crf_context* ctx, int secondary, float* out_min, float* out_max);
/* The second scalar field of the SEPARATE / SEPARATE_SYMMETRIC field modes (fieldIndex2Gui; fieldEntriesSecondary,
 * CorrelationCalculator.cpp:1182-1229): cs volumes of the same local grid, uploaded or borrowed like the primary
 * members; dropped by crf_set_grid.  Used by CRF_FLAG_SYMMETRIC / CRF_FLAG_REFERENCE_FROM_SECONDARY. */
int crf_upload_secondary_members(crf_context* ctx, const float* const* host_members);
int crf_bind_secondary_members_device(crf_context* ctx, const void* const* device_members);
int crf_secondary_member_minmax(crf_context* ctx, float* out_min, float* out_max);

/* ---- reference vector (CorrelationCalculator.cpp:802,815-817) --------------------------------------------- */
/* referenceValues[c] = member_c[IDXS(x,y,z)] -> cs floats to a host buffer (synchronous) ... */
int crf_gather_reference(crf_context* ctx, int x, int y, int z, float* host_out);
/* ... or to a device buffer, asynchronously on `stream` (a hipStream_t, NULL = the context's own stream). */
int crf_gather_reference_device(crf_context* ctx, int x, int y, int z, void* device_out, void* stream);

/* Batched form for the multi-GPU exchange: row r of device_rows (num_rows x cs floats, num_rows <= 32) receives the
 * reference values of LOCAL grid point (xyz[3r], xyz[3r+1], xyz[3r+2]), or zeros when xyz[3r+2] < 0 (a point whose
 * z-slice another process owns) -- so that one all-reduce(sum) over the processes yields every row everywhere. */
int crf_gather_reference_rows_device(crf_context* ctx, const int32_t* xyz, int num_rows, void* device_rows, void* stream);

/* ---- Kraskov tie-breaking noise ------------------------------------------------------------------------------------
 * The KSG estimators add `u * 1e-10`, u drawn per member from sgl::XorshiftRandomGenerator(617406168) for the reference
 * vector and (864730169) for every query vector, to break exact ties (MutualInformation.cpp:409-420, 167-185).  sgl is
 * not vendored by the reference and not available to this build, so the library's DEFAULT tables come from a
 * documented stand-in stream (Marsaglia xorshift32) -- results on exactly tied data then differ from a
 * reference build in the last digits.  An integrator who has sgl passes the real stream here: ref_noise[e] and
 * query_noise[e] are the noise VALUES (already multiplied by 1e-10, as double) of member e, cs of each; both NULL
 * restores the default.  Applies to every later Kraskov evaluation of this context (field, symmetric and pair-request
 * modes) until the next crf_set_grid.  Stream-ordered on the context's own stream; call it between evaluations. */
int crf_set_kraskov_noise(crf_context* ctx, const double* ref_noise, const double* query_noise);

/* ---- evaluation (replaces the hot loop CorrelationCalculator.cpp:868-1142) ------------------------------- */
/* Synchronous, host output: what calculateCpu(t, e, buffer) does.  host_out receives xs*ys*zs floats. */
int crf_compute(crf_context* ctx, const crf_params* params, float* host_out);
/* Asynchronous, device output, stream-ordered, no host synchronisation: the form the multi-GPU path and the
 * benchmark use.  device_reference_values: DEVICE pointer to cs floats or NULL (then params->reference_values or
 * the reference point are used).  device_out receives xs*ys*zs floats. */
int crf_compute_device(crf_context* ctx, const crf_params* params, const void* device_reference_values,
                       void* device_out, void* stream);

/* Two-phase form for pipelined callers (the multi-GPU driver): crf_prepare_device runs only the reference-side
 * preparation of an evaluation (the reference-derived tables: Pearson/Spearman a_e, Kendall's x order and tie groups,
 * binned reference bins, Kraskov's noisy reference coordinates) into slot `slot` (0 <= slot < CRF_PREPARED_SLOTS), e.g.
 * on a communication stream right after the reference vectors arrive; a later crf_compute_device with
 * params->prepared_slot = slot + 1 launches only the per-voxel kernel.  The caller orders the two calls (same stream
 * or an event) and must not re-prepare a slot before the evaluation that reads it has been launched and ordered. */
#define CRF_PREPARED_SLOTS 64
int crf_prepare_device(crf_context* ctx, const crf_params* params, const void* device_reference_values, int slot,
                       void* stream);

/* crf_prepare_device for `count` reference vectors with ONE call: row i of device_rows (cs floats each, contiguous rows --
 * e.g. the buffer crf_gather_reference_rows_device filled and an all-reduce completed) is prepared into slot
 * first_slot + i.  params: the evaluation's settings (reference point and reference_values are not read). */
int crf_prepare_rows_device(crf_context* ctx, const crf_params* params, const void* device_rows, int first_slot, int count,
                            void* stream);
/* Launches `count` prepared evaluations back to back with ONE call: evaluation i reads the tables crf_prepare_device left
 * in slot first_slot + i and writes device_outs[i] (params: as given to crf_prepare_device, prepared_slot is ignored).
 * The host-side cost of a pipelined driver then is one call per batch instead of one per evaluation -- at 8 GPUs a
 * 256^3 x 64 evaluation is 0.09 ms per device, the same order as a Python-level call. */
int crf_compute_prepared_device(crf_context* ctx, const crf_params* params, int first_slot, int count,
                                void* const* device_outs, void* stream);

/* ---- several GPUs behind ONE caller thread ------------------------------------------------------------------------
 * The reference is a single process that calls its calculators from the render thread with a caller-owned host buffer
 * (src/Volume/VolumeData.cpp:1214-1226, 1469-1472).  A crf_group gives such a caller N devices: the GLOBAL grid is cut
 * into z-slabs (the first zs % N slabs one slice longer -- with x-fastest volumes a slab of every member is contiguous
 * and the result slabs concatenate in the caller's buffer), each device keeps its slab of every member resident in its
 * own HBM and is driven by its own worker thread inside the library, and every evaluation has one exchange step: the
 * device whose slab holds the reference point (the owner) has the cs reference values
 * (CorrelationCalculator.cpp:802,815-817).  By default every other device READS them directly out of the owner's member
 * volumes -- peer access over xGMI, fused into its reference-side preparation kernel: no collective, no copy, no
 * rendezvous.  CRF_GROUP_EXCHANGE=rccl (and the default when some pair of devices lacks peer access): the owner gathers
 * and ncclBroadcast distributes them on a persistent single-process RCCL communicator.  CRF_GROUP_EXCHANGE=copy: staged
 * peer copies.  crf_group_exchange() names the form in use.  Each device then evaluates its slab and copies it straight
 * into its part
 * of the caller's buffer (the copies of the N devices run concurrently).  Results are bit-identical to a single
 * context's.  Same conventions as above: one caller thread, status codes, crf_group_last_error. */
typedef struct crf_group crf_group;
int crf_group_create(const int* device_ordinals, int num_devices, crf_group** out_group);
void crf_group_destroy(crf_group* group);
const char* crf_group_last_error(const crf_group* group);   /* group == NULL: last failed crf_group_create of this thread */
int crf_group_size(const crf_group* group);
/* "peer read (...)", "rccl (...)", "peer copy (...)" or "none (one device)": how the reference vector travels. */
const char* crf_group_exchange(const crf_group* group);
/* The context that drives device slot `slot` (its LOCAL grid is the slab): for per-device helpers such as
 * crf_bind_members_device or crf_last_kernel_name.  Owned by the group. */
crf_context* crf_group_context(crf_group* group, int slot);
/* Declares the GLOBAL grid and member count (zs >= number of devices); drops any members. */
int crf_group_set_grid(crf_group* group, int xs, int ys, int zs, int cs);
int crf_group_slab(const crf_group* group, int slot, int* z_begin, int* z_count);
/* cs host volumes of the WHOLE grid; every device uploads its slab of each (concurrently). */
int crf_group_upload_members(crf_group* group, const float* const* host_members);
int crf_group_upload_secondary_members(crf_group* group, const float* const* host_members);
/* Extrema over all slabs (binned-MI normalisation range, CorrelationCalculator.cpp:822-829). */
int crf_group_member_minmax(crf_group* group, float* out_min, float* out_max);
int crf_group_secondary_member_minmax(crf_group* group, float* out_min, float* out_max);
/* calculateCpu(t, e, buffer) on N devices: params as for crf_compute with the reference point in GLOBAL coordinates;
 * host_out receives xs*ys*zs floats.  Supports the reference point, a host reference vector (no exchange),
 * CRF_FLAG_REFERENCE_FROM_SECONDARY, CRF_FLAG_SYMMETRIC (no exchange) and CRF_FLAG_ABSOLUTE_VALUE. */
int crf_group_compute(crf_group* group, const crf_params* params, float* host_out);
/* The same evaluation with DEVICE-resident results (what crf_compute_device is to crf_compute): device_outs[slot] is a
 * buffer on the device of that slot and receives the slot's slab, xs*ys*z_count floats (crf_group_slab); returns when
 * every device has finished.  For consumers that keep the field on the GPUs. */
int crf_group_compute_device(crf_group* group, const crf_params* params, void* const* device_outs);
/* MANY reference points per call -- a diagram or an animation that evaluates a list of points (the reference's
 * HEBChart / time-series consumers call calculateCpu point after point from the same thread): params[i] as for
 * crf_group_compute, evaluation i writes host_outs[i] (xs*ys*zs floats each).  One hand-off to the device workers for
 * the whole list; the reference vectors of up to 32 points travel in ONE collective (owners fill their rows, one
 * ncclAllReduce(sum)) or are read directly from the owner's members (peer exchange). */
int crf_group_compute_batch(crf_group* group, const crf_params* params, int count, float* const* host_outs);
/* ... with device-resident results: device_outs[i * crf_group_size(group) + slot] receives the slab of slot `slot` of
 * evaluation i.  Per device the reference-side preparations of a block of evaluations run first, the per-voxel kernels
 * follow back to back and the workers synchronise once, at the end of the call. */
int crf_group_compute_batch_device(crf_group* group, const crf_params* params, int count, void* const* device_outs);
int crf_group_set_profiling(crf_group* group, int enabled);
/* Slowest device's summed kernel time (ms) and its launch count since the last call. */
int crf_group_take_kernel_time(crf_group* group, double* out_ms_max, int* out_launches);
/* crf_set_kraskov_noise on every device of the group. */
int crf_group_set_kraskov_noise(crf_group* group, const double* ref_noise, const double* query_noise);

/* ---- pair-request evaluation (CorrelationComputePass request mode, CorrelationCalculator.hpp:250-258; CPU twin
 * HEBChart::computeCorrelations, src/Renderers/Diagram/HEBChartCorrelation.cpp:493-600) ---------------------------- */
/* out[r] = measure(X_r, Y_r) with X_r[c] = member_c[IDXS(xi,yi,zi)], Y_r[c] = member_c[IDXS(xj,yj,zj)].  Semantics of the
 * CPU twin: Pearson/Spearman/Kendall as in the full-grid path but with both vectors per request; binned MI normalises
 * both vectors with the extrema of the pair (:556-566); Kraskov is KSG-1; MI-CC variants map sqrt(1-exp(-2 MI));
 * CRF_FLAG_ABSOLUTE_VALUE takes |.|; a NaN in either vector gives NaN (the CPU twin emits no entry for that pair).
 * CRF_FLAG_QUERY_FROM_SECONDARY: the j side reads the secondary members (two-field request mode).
 * Read from params: measure, k, num_bins, flags. */
int crf_compute_requests(crf_context* ctx, const crf_params* params, const crf_request* host_requests,
                         size_t num_requests, float* host_out);
int crf_compute_requests_device(crf_context* ctx, const crf_params* params, const void* device_requests,
                                size_t num_requests, void* device_out, void* stream);

/* ---- sibling per-voxel ensemble reductions (same access pattern) ------------------------------ */
typedef enum crf_ensemble_stat {
    CRF_ENSEMBLE_MEAN = 0,   /* EnsembleMeanCalculator::calculateCpu,   src/Calculators/EnsembleMeanCalculator.cpp:94-138 */
    CRF_ENSEMBLE_SPREAD = 1  /* EnsembleSpreadCalculator::calculateCpu, src/Calculators/EnsembleSpreadCalculator.cpp:94-149 */
} crf_ensemble_stat;
/* NaN-skipping mean / sample standard deviation over the cs members at every voxel; xs*ys*zs floats out. */
int crf_compute_ensemble_stat(crf_context* ctx, int stat, float* host_out);
int crf_compute_ensemble_stat_device(crf_context* ctx, int stat, void* device_out, void* stream);

/* SetPredicateCalculator::calculateCpu (src/Calculators/SetPredicateCalculator.cpp:154-210): count = number of members
 * with `value OP comparison_value`; out = clamp(count - count_lower, 0, 1) when count_lower == count_upper, else
 * clamp((count - count_lower) / (count_upper - count_lower), 0, 1), all in fp32.  Operator values follow
 * ComparisonOperatorType (SetPredicateCalculator.hpp:41-43); NaN compares as in C. */
typedef enum crf_comparison_operator {
    CRF_CMP_GREATER = 0, CRF_CMP_GREATER_EQUAL = 1, CRF_CMP_LESS = 2, CRF_CMP_LESS_EQUAL = 3, CRF_CMP_EQUAL = 4,
    CRF_CMP_NOT_EQUAL = 5
} crf_comparison_operator;
int crf_compute_set_predicate(crf_context* ctx, int comparison_operator, float comparison_value, int count_lower,
                              int count_upper, float* host_out);
int crf_compute_set_predicate_device(crf_context* ctx, int comparison_operator, float comparison_value, int count_lower,
                                     int count_upper, void* device_out, void* stream);

/* DKLCalculator::calculateCpu (src/Calculators/DKLCalculator.cpp:134-262): Kullback-Leibler divergence between the
 * normalised distribution of the cs member values of a voxel and N(0,1).  Estimators (DKLEstimatorType,
 * DKLCalculator.hpp:39-41): binned = computeDKLBinned<double> (DKL.cpp:38-84, num_bins in [1, 1024], default 80),
 * entropy k-NN = computeDKLKNNEstimate<double> (DKL.cpp:98-165, 1 <= k < cs, default max(ceil(3 cs / 100), 1),
 * DKLCalculator.cpp:94-101).  cs == 1 -> 1; a NaN member -> NaN; at most 2048 members. */
typedef enum crf_dkl_estimator { CRF_DKL_BINNED = 0, CRF_DKL_ENTROPY_KNN = 1 } crf_dkl_estimator;
int crf_compute_dkl(crf_context* ctx, int estimator, int num_bins, int k, float* host_out);
int crf_compute_dkl_device(crf_context* ctx, int estimator, int num_bins, int k, void* device_out, void* stream);

/* ---- result layout for the renderer -------------------------------------------------------------------------------- */
/* The reference keeps device fields in 8x8x4 tiles (bufferTileSize, VolumeData.cpp:1581-1621; addressed by IDXS of
 * Data/Shaders/Correlation/ScalarFields.glsl:32-50): tiles in x-fastest order, x-fastest inside a tile, grid padded up
 * to whole tiles with zeros.  crf_tiled_element_count = ceil(xs/8)*ceil(ys/8)*ceil(zs/4)*256 floats;
 * crf_tile_field_device re-lays a linear (IDXS) device field of the context's grid into that layout, stream-ordered. */
size_t crf_tiled_element_count(int xs, int ys, int zs);
int crf_tile_field_device(crf_context* ctx, const void* device_linear, void* device_tiled, void* stream);

/* Upper bound of the KSG estimate used for the colour range of the diagrams:
 * computeMaximumMutualInformationKraskov(k, es) = psi(es) - psi(k) (MutualInformation.cpp:526-528); psi(n) = -gamma + H_{n-1}.
 * Host-only helper, no context needed; NaN for k < 1 or cs < 1. */
double crf_max_mutual_information_kraskov(int k, int cs);

/* ---- instrumentation --------------------------------------------------------------------------------------- */
/* When enabled, every crf_compute* brackets its dominant (per-voxel) kernel with HIP events on the launch stream. */
int crf_set_profiling(crf_context* ctx, int enabled);
/* Sum of the recorded kernel durations (ms) and their count since the last call; synchronises the events.  Resets. */
int crf_take_kernel_time(crf_context* ctx, double* out_ms_sum, int* out_launches);
/* Name of the dominant kernel of the last compute (for matching rocprofv3 rows), valid until the next call. */
const char* crf_last_kernel_name(const crf_context* ctx);
/* Fills device memory with a deterministic synthetic "box ensemble" member volume (recipe of the
 * reference's scripts/generate_synth_box_ensembles.py:57-136) for a z-slab [z_begin, z_begin+zs_local) of a global
 * xs*ys*zs_global grid: member `c` of `cs`.  Benchmark/test input generation only; not on the timed path. */
int crf_synth_box_member(crf_context* ctx, void* device_out, int xs, int ys, int zs_local, int z_begin, int zs_global,
                         int c, int cs, uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CORRFIELD_H */
