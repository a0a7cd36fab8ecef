// kernels_stats.hip -- the sibling per-voxel ensemble reductions of the correlation path:
//   ensemble mean    EnsembleMeanCalculator::calculateCpu   (src/Calculators/EnsembleMeanCalculator.cpp:94-138)
//   ensemble spread  EnsembleSpreadCalculator::calculateCpu (src/Calculators/EnsembleSpreadCalculator.cpp:94-149)
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
    if (v0 < num_voxels) out[v0] = res;
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
    if (v0 < num_voxels) out[v0] = res;
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
