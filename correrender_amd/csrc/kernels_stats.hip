// kernels_stats.hip -- the sibling per-voxel ensemble reductions of the correlation path:
//   ensemble mean    EnsembleMeanCalculator::calculateCpu   (src/Calculators/EnsembleMeanCalculator.cpp:94-138)
//   ensemble spread  EnsembleSpreadCalculator::calculateCpu (src/Calculators/EnsembleSpreadCalculator.cpp:94-149)
//   set predicate    SetPredicateCalculator::calculateCpu   (src/Calculators/SetPredicateCalculator.cpp:154-210)
// plus the linear -> 8x8x4-tiled re-layout of a result field (VolumeData.cpp:1581-1621).
// Same access pattern and roofline as Pearson (cs member streams in, one float per voxel out, 4*cs + 4 bytes/voxel),
// same loader (buffer descriptors, shared 32-bit offset, non-temporal).  fp32, NaN values skipped, sums in member order
// exactly like the reference:  mean = (sum of valid) / numValid (NaN if none);  spread = sqrt( sum (mean - v)^2 /
// (numValid - 1) ) (NaN if fewer than two valid values).
#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

// kind 0: mean, 1: spread.  Members resident in registers (one voxel per lane), loops fully unrolled to CS_PAD.
template <int CS_PAD, int KIND>
__global__ __launch_bounds__(256) void ensemble_stat_reg_kernel(const float* const* __restrict__ members,
                                                                float* __restrict__ out, uint32_t num_voxels, int cs) {
    const uint32_t v0 = blockIdx.x * 256u + threadIdx.x;
    const uint32_t byte_offset = v0 * 4u, bytes = num_voxels * 4u;
    float y[CS_PAD];
#pragma unroll
    for (int e = 0; e < CS_PAD; e++)
        y[e] = load_member_nt(members[e < cs ? e : cs - 1], bytes, byte_offset);  // slots past cs re-read a valid member
    int num_valid = 0;
    float mean = 0.0f;
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
        const bool ok = e < cs && !(y[e] != y[e]);
        mean = ok ? mean + y[e] : mean;
        num_valid += ok ? 1 : 0;
    }
    float res;
    if (KIND == 0) {
        res = num_valid >= 1 ? mean / float(num_valid) : __uint_as_float(0x7FC00000u);
    } else {
        mean = mean / float(num_valid);
        float var_sum = 0.0f;
#pragma unroll
        for (int e = 0; e < CS_PAD; e++) {
            const bool ok = e < cs && !(y[e] != y[e]);
            const float diff = mean - y[e];
            var_sum = ok ? var_sum + diff * diff : var_sum;
        }
        res = num_valid > 1 ? sqrtf(var_sum / float(num_valid - 1)) : __uint_as_float(0x7FC00000u);
    }
    if (v0 < num_voxels) store_result_nt(out + v0, res);
}

// Any member count: streaming passes (the spread re-reads the members once; mostly served by L2 / Infinity Cache).
template <int KIND>
__global__ __launch_bounds__(256) void ensemble_stat_stream_kernel(const float* const* __restrict__ members,
                                                                   float* __restrict__ out, uint32_t num_voxels, int cs) {
    const uint32_t v0 = blockIdx.x * 256u + threadIdx.x;
    const uint32_t byte_offset = v0 * 4u, bytes = num_voxels * 4u;
    int num_valid = 0;
    float mean = 0.0f;
#pragma unroll 16
    for (int e = 0; e < cs; e++) {
        const float v = load_member_nt(members[e], bytes, byte_offset);
        const bool ok = !(v != v);
        mean = ok ? mean + v : mean;
        num_valid += ok ? 1 : 0;
    }
    float res;
    if (KIND == 0) {
        res = num_valid >= 1 ? mean / float(num_valid) : __uint_as_float(0x7FC00000u);
    } else {
        mean = mean / float(num_valid);
        float var_sum = 0.0f;
#pragma unroll 16
        for (int e = 0; e < cs; e++) {
            const float v = load_member_nt(members[e], bytes, byte_offset);
            const float diff = mean - v;
            var_sum = !(v != v) ? var_sum + diff * diff : var_sum;
        }
        res = num_valid > 1 ? sqrtf(var_sum / float(num_valid - 1)) : __uint_as_float(0x7FC00000u);
    }
    if (v0 < num_voxels) store_result_nt(out + v0, res);
}

// Set predicate (SetPredicateCalculator::calculateCpu, src/Calculators/SetPredicateCalculator.cpp:154-210): count the
// members whose value satisfies `value OP comparison_value`, map the count to [0, 1] between countLower and
// countUpper.  One streaming pass, nothing to hold: 4*cs + 4 bytes per voxel.
template <int OP>
__device__ __forceinline__ bool set_predicate_compare(float a, float b) {
    if constexpr (OP == 0) return a > b;
    else if constexpr (OP == 1) return a >= b;
    else if constexpr (OP == 2) return a < b;
    else if constexpr (OP == 3) return a <= b;
    else if constexpr (OP == 4) return a == b;
    else return a != b;
}

template <int OP>
__global__ __launch_bounds__(256) void set_predicate_kernel(const float* const* __restrict__ members,
                                                            float* __restrict__ out, uint32_t num_voxels, int cs,
                                                            float comparison_value, int count_lower, int count_upper) {
    const uint32_t v0 = blockIdx.x * 256u + threadIdx.x;
    const uint32_t byte_offset = v0 * 4u, bytes = num_voxels * 4u;
    int count = 0;
#pragma unroll 16
    for (int e = 0; e < cs; e++)
        count += set_predicate_compare<OP>(load_member_nt(members[e], bytes, byte_offset), comparison_value) ? 1 : 0;
    float res = float(count) - float(count_lower);
    if (count_lower != count_upper) res = res / (float(count_upper) - float(count_lower));
    res = res < 0.0f ? 0.0f : (1.0f < res ? 1.0f : res);  // std::clamp: -0.0 stays -0.0
    if (v0 < num_voxels) store_result_nt(out + v0, res);
}

hipError_t launch_set_predicate(const float* const* d_members, int cs, size_t num_voxels, int op, float comparison_value,
                                int count_lower, int count_upper, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                                hipEvent_t ev_end, LaunchInfo* info) {
    const unsigned blocks = unsigned((num_voxels + 255) / 256);
    const uint32_t n = uint32_t(num_voxels);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
#define CRF_SETPRED(OP)                                                                                       \
    case OP:                                                                                                  \
        hipLaunchKernelGGL((set_predicate_kernel<OP>), dim3(blocks), dim3(256), 0, s, d_members, d_out, n, cs, \
                           comparison_value, count_lower, count_upper);                                       \
        break;
    switch (op) {
        CRF_SETPRED(0)
        CRF_SETPRED(1)
        CRF_SETPRED(2)
        CRF_SETPRED(3)
        CRF_SETPRED(4)
        CRF_SETPRED(5)
        default: return hipErrorInvalidValue;
    }
#undef CRF_SETPRED
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "set_predicate_kernel";
    return hipGetLastError();
}

// Linear IDXS order -> the 8x8x4-tiled buffer layout of the reference's device field cache (VolumeData.cpp:1581-1621,
// IDXS of Data/Shaders/Correlation/ScalarFields.glsl:32-50): tiles in x-fastest tile order, x-fastest inside a tile,
// the grid padded up to whole tiles with zeros.  One wave writes one 256-B row of 64 tiled elements.
__global__ __launch_bounds__(256) void tile_field_kernel(const float* __restrict__ linear, float* __restrict__ tiled,
                                                         uint32_t xs, uint32_t ys, uint32_t zs, uint32_t xst,
                                                         uint32_t yst, size_t num_tiled) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= num_tiled) return;
    const uint32_t tile = uint32_t(i >> 8), voxel = uint32_t(i & 255u);
    const uint32_t xt = tile % xst, yt = (tile / xst) % yst, zt = tile / (xst * yst);
    const uint32_t x = (voxel & 7u) + xt * 8u, y = ((voxel >> 3) & 7u) + yt * 8u, z = (voxel >> 6) + zt * 4u;
    float value = 0.0f;
    if (x < xs && y < ys && z < zs) value = linear[(size_t(z) * ys + y) * xs + x];
    tiled[i] = value;
}

hipError_t launch_tile_field(const float* d_linear, float* d_tiled, int xs, int ys, int zs, hipStream_t s) {
    const uint32_t xst = (uint32_t(xs) + 7u) / 8u, yst = (uint32_t(ys) + 7u) / 8u, zst = (uint32_t(zs) + 3u) / 4u;
    const size_t num_tiled = size_t(xst) * yst * zst * 256;
    hipLaunchKernelGGL(tile_field_kernel, dim3(unsigned((num_tiled + 255) / 256)), dim3(256), 0, s, d_linear, d_tiled,
                       uint32_t(xs), uint32_t(ys), uint32_t(zs), xst, yst, num_tiled);
    return hipGetLastError();
}

hipError_t launch_ensemble_stat(int kind, const float* const* d_members, int cs, size_t num_voxels, float* d_out,
                                hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    const unsigned blocks = unsigned((num_voxels + 255) / 256);
    const uint32_t n = uint32_t(num_voxels);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
#define CRF_STAT(CSP)                                                                                              \
    if (kind == 0)                                                                                                 \
        hipLaunchKernelGGL((ensemble_stat_reg_kernel<CSP, 0>), dim3(blocks), dim3(256), 0, s, d_members, d_out, n, cs); \
    else                                                                                                           \
        hipLaunchKernelGGL((ensemble_stat_reg_kernel<CSP, 1>), dim3(blocks), dim3(256), 0, s, d_members, d_out, n, cs);
    if (kind == 0 || cs > 128) {  // the mean needs one pass only: no reason to hold the values
        if (kind == 0)
            hipLaunchKernelGGL((ensemble_stat_stream_kernel<0>), dim3(blocks), dim3(256), 0, s, d_members, d_out, n, cs);
        else
            hipLaunchKernelGGL((ensemble_stat_stream_kernel<1>), dim3(blocks), dim3(256), 0, s, d_members, d_out, n, cs);
        if (info) info->kernel_name = "ensemble_stat_stream_kernel";
    } else {
        if (cs <= 16) {
            CRF_STAT(16)
        } else if (cs <= 32) {
            CRF_STAT(32)
        } else if (cs <= 64) {
            CRF_STAT(64)
        } else {
            CRF_STAT(128)
        }
        if (info) info->kernel_name = "ensemble_stat_reg_kernel";
    }
#undef CRF_STAT
    if (ev_end) (void)hipEventRecord(ev_end, s);
    return hipGetLastError();
}

}  // namespace crf
