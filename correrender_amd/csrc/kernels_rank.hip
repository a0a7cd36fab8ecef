// kernels_rank.hip -- the two rank-based estimators on gfx950: Spearman and Kendall tau-b.
//
// Both are integer/ordering problems per voxel followed by a short fp32 tail, and both are reproduced bit-exactly:
//   Spearman  (CorrelationCalculator.cpp:902-962): fractional ranks of the voxel's cs values (computeRanks,
//             Correlation.cpp:277-303: a run of m equal values starting at 1-based rank R gets R + (m-1)/2), then
//             computePearson2<float>(referenceRanks, ranks, cs) (Correlation.cpp:141-174) in member order.
//   Kendall   (CorrelationCalculator.cpp:963-1025, computeKendall<int32_t>, Correlation.cpp:423-455):
//             num = n0 - n1 - n2 - 2*S_y (joint ties n3 := 0), tau = float(num) / (sqrtf(n0-n1) * sqrtf(n0-n2)),
//             S_y = strict inversions of y after sorting the (x, y) pairs lexicographically = number of pairs with
//             x_a < x_b and y_a > y_b.
//
// Mapping: ONE LANE PER VOXEL, the voxel's cs values live in that lane's registers as 64-bit composites
// (order-preserving key of the value << 32 | slot) and are sorted by a fully unrolled Batcher merge-exchange
// network (tools/gen_sortnet.py): static register indices only, no divergence, no LDS traffic in the sort.  Loads
// stay coalesced (a wave load = 64 consecutive voxels of one member).
//   Spearman: a forward and a backward scan over the sorted registers find each tie run [start, end]; 2*rank =
//     start + end + 2 is scattered as a 16-bit value to LDS row `slot` (column = lane, so the scatter is
//     conflict-free and private to the lane: LDS is used as per-lane indexed scratch, no barrier), read back in
//     member order and fed to the same fp32 three-pass Pearson tail as the Pearson kernel.
//   Kendall: members are LOADED in reference-sorted order (slot = position in the x order, permutation prepared once
//     per evaluation), so after sorting by (y, slot) the slot sequence is a permutation whose inversions are the
//     discordant pairs; they are counted with a per-lane bitset of seen slots (popcount of the bits above the
//     current slot's x-tie group), ties in y by run lengths, ties in x once per evaluation.
// NaN in the voxel's values -> quiet NaN (CorrelationCalculator.cpp:929-940, 1002-1013).
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

constexpr uint32_t kPadKey = 0xFFFFFFFFu;  // sorts after every real value (orderable_key(+inf) = 0xFF800000)

// ---------------------------------------------------------------------------------------------------------
// Reference-side preparation
// ---------------------------------------------------------------------------------------------------------
// Spearman: prep[e] = a_e = invNm1 * ((rx_e - mean) / sd) over the reference RANKS rx (CorrelationCalculator.cpp:859-865).
__global__ __launch_bounds__(256) void spearman_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                            int cs, float* __restrict__ prep) {
    __shared__ float ref[kMaxSortMembers];
    __shared__ float rx[kMaxSortMembers];
    __shared__ float sh[2];
    for (int i = threadIdx.x; i < cs; i += blockDim.x) ref[i] = load_ref(src, members, i);
    __syncthreads();
    for (int i = threadIdx.x; i < cs; i += blockDim.x) {
        const float v = ref[i];
        int s = 0;  // sum over j of sign(v_i - v_j); 2*rank_i = cs + 1 + s
        for (int j = 0; j < cs; j++) {
            const float w = ref[j];
            s += (w < v) ? 1 : ((v < w) ? -1 : 0);
        }
        rx[i] = 0.5f * float(cs + 1 + s);
    }
    __syncthreads();
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    if (threadIdx.x == 0) {
        float mean = 0.0f;
        for (int e = 0; e < cs; e++) mean += invN * rx[e];
        float var = 0.0f;
        for (int e = 0; e < cs; e++) {
            const float d = rx[e] - mean;
            var += invNm1 * d * d;
        }
        sh[0] = mean;
        sh[1] = sqrtf(var);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) prep[e] = invNm1 * ((rx[e] - sh[0]) / sh[1]);
}

// Kendall: prep as int32: [0, N) perm (slot -> member), [N, 2N) gend (slot -> last slot of its x-tie group),
// [2N] n1 = sum over x-tie groups t(t-1)/2 (computeTiesB, Correlation.cpp:305-329), [2N+1] 1 if x has ties.
__global__ __launch_bounds__(256) void kendall_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                           int cs, int n_pad, int* __restrict__ prep) {
    __shared__ float ref[kMaxSortMembers];
    __shared__ int n1;
    if (threadIdx.x == 0) n1 = 0;
    for (int i = threadIdx.x; i < cs; i += blockDim.x) ref[i] = load_ref(src, members, i);
    __syncthreads();
    for (int i = threadIdx.x; i < n_pad; i += blockDim.x) {
        if (i >= cs) {
            prep[i] = 0;
            prep[n_pad + i] = i;
        }
    }
    for (int i = threadIdx.x; i < cs; i += blockDim.x) {
        const float v = ref[i];
        int less = 0, eq_before = 0, eq_total = 0;
        for (int j = 0; j < cs; j++) {
            const float w = ref[j];
            less += (w < v) ? 1 : 0;
            const int eq = (w == v) ? 1 : 0;
            eq_total += eq;
            eq_before += (j < i) ? eq : 0;
        }
        const int pos = less + eq_before;
        prep[pos] = i;
        prep[n_pad + pos] = less + eq_total - 1;
        if (eq_before) atomicAdd(&n1, eq_before);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        prep[2 * n_pad] = n1;
        prep[2 * n_pad + 1] = n1 != 0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Spearman
// ---------------------------------------------------------------------------------------------------------
template <int N, bool EXACT, int MIN_WAVES>
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_kernel(const float* const* __restrict__ members,
                                                                 const float* __restrict__ prep,
                                                                 float* __restrict__ out, size_t num_voxels, int cs) {
    __shared__ uint16_t rank2[N * 64];
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;  // lanes past the end read 0

    composite_t a[N];
    bool is_nan = false;
#pragma unroll
    for (int e = 0; e < N; e++) {
        if (EXACT || e < cs) {
            float y = load_member_nt(members[e], bytes, byte_offset);
            is_nan |= (y != y);
            y += 0.0f;  // -0.0 -> +0.0 so that key equality is float equality
            a[e] = make_composite(orderable_key(y), uint32_t(e));
        } else {
            a[e] = make_composite(kPadKey, uint32_t(e));
        }
        if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(a);
    __builtin_amdgcn_sched_barrier(0);

    // forward scan: first position of the tie run each sorted position belongs to, parked in bits 8..15 of the low word
    uint32_t run_start = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
        if (EXACT || p < cs) {
            if (p > 0) {
                const bool same = composite_key(a[p]) == composite_key(a[p - 1]);
                run_start = same ? run_start : uint32_t(p);
            }
            a[p] = composite_or_low(a[p], run_start << 8);
        }
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // bound the scheduler's window: register pressure
    }
    // backward scan: last position of the run; 2*rank = start + end + 2; scatter to the member's LDS row
    uint32_t run_end = 0;
#pragma unroll
    for (int p = N - 1; p >= 0; p--) {
        if (EXACT || p < cs) {
            bool same = false;
            if (p < N - 1 && (EXACT || p + 1 < cs)) same = composite_key(a[p]) == composite_key(a[p + 1]);
            run_end = same ? run_end : uint32_t(p);
            const uint32_t low = composite_low(a[p]);
            const uint32_t slot = low & 0xFFu;
            const uint32_t start = (low >> 8) & 0xFFu;
            rank2[slot * 64 + lane] = uint16_t(start + run_end + 2u);
        }
        if ((p & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ranks back in member order (same lane wrote them: program order suffices, no barrier)
    float r[N];
#pragma unroll
    for (int e = 0; e < N; e++) r[e] = (EXACT || e < cs) ? 0.5f * float(rank2[e * 64 + lane]) : 0.0f;
    float res = pearson_tail<N, EXACT>(r, prep, cs);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) out[v] = res;
}

// ---------------------------------------------------------------------------------------------------------
// Kendall
// ---------------------------------------------------------------------------------------------------------
template <int N, bool EXACT, int MIN_WAVES>
__global__ __launch_bounds__(64, MIN_WAVES) void kendall_kernel(const float* const* __restrict__ members,
                                                                const int* __restrict__ prep, float* __restrict__ out,
                                                                size_t num_voxels, int cs) {
    __shared__ uint8_t gend_lds[N];
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    const bool x_ties = prep[2 * N + 1] != 0;  // wave-uniform
    if (x_ties) {
        for (int i = lane; i < N; i += 64) gend_lds[i] = uint8_t(prep[N + i]);
        __syncthreads();
    }

    composite_t a[N];
    bool is_nan = false;
#pragma unroll
    for (int e = 0; e < N; e++) {
        if (EXACT || e < cs) {
            float y = load_member_nt(members[prep[e]], bytes, byte_offset);  // slot e = e-th smallest reference value
            is_nan |= (y != y);
            y += 0.0f;
            a[e] = make_composite(orderable_key(y), uint32_t(e));
        } else {
            a[e] = make_composite(kPadKey, uint32_t(e));
        }
        if ((e & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(a);
    __builtin_amdgcn_sched_barrier(0);

    constexpr int W = (N + 31) / 32;
    uint32_t seen[W];
#pragma unroll
    for (int w = 0; w < W; w++) seen[w] = 0u;
    int32_t discordant = 0, n2 = 0, run = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
        if (EXACT || p < cs) {
            // ties in y: a run of t equal values contributes 0+1+...+(t-1) = t(t-1)/2
            if (p > 0) {
                const bool same = composite_key(a[p]) == composite_key(a[p - 1]);
                run = same ? run + 1 : 0;
                n2 += run;
            }
            const uint32_t slot = composite_low(a[p]) & 0xFFu;
            const uint32_t g = x_ties ? uint32_t(gend_lds[slot]) : slot;  // last slot with the same x
            // already-seen slots (smaller y, or equal y and smaller slot) with strictly larger x: slot' > g
            const uint32_t gw = g >> 5;
            const uint32_t gm = 0xFFFFFFFEu << (g & 31u);
            const uint32_t sw = slot >> 5;
            const uint32_t sbit = 1u << (slot & 31u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                const uint32_t mask = (uint32_t(w) > gw) ? 0xFFFFFFFFu : ((uint32_t(w) == gw) ? gm : 0u);
                discordant += __popc(seen[w] & mask);
                seen[w] |= (uint32_t(w) == sw) ? sbit : 0u;
            }
        }
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t n1 = prep[2 * N];
    const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
    const float denominator = sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2));
    float res = float(numerator) / denominator;
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) out[v] = res;
}

namespace {

template <int N, int MIN_WAVES>
void launch_spearman_n(const float* const* d_members, const float* d_prep, float* d_out, size_t num_voxels, int cs,
                       hipStream_t s) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    if (cs == N)
        hipLaunchKernelGGL((spearman_kernel<N, true, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, d_prep,
                           d_out, num_voxels, cs);
    else
        hipLaunchKernelGGL((spearman_kernel<N, false, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, d_prep,
                           d_out, num_voxels, cs);
}

template <int N, int MIN_WAVES>
void launch_kendall_n(const float* const* d_members, const int* d_prep, float* d_out, size_t num_voxels, int cs,
                      hipStream_t s) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    if (cs == N)
        hipLaunchKernelGGL((kendall_kernel<N, true, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, d_prep,
                           d_out, num_voxels, cs);
    else
        hipLaunchKernelGGL((kendall_kernel<N, false, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, d_prep,
                           d_out, num_voxels, cs);
}

int pad_pow2(int cs) { return cs <= 16 ? 16 : cs <= 32 ? 32 : cs <= 64 ? 64 : 128; }

// waves/SIMD the 64-member kernels are compiled for (register cap 512/256/168); CRF_RANK_WAVES overrides for tuning.
int env_waves(int fallback) {
    const char* v = getenv("CRF_RANK_WAVES");
    return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

hipError_t launch_spearman(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref, float* d_prep,
                           float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    hipLaunchKernelGGL(spearman_prep_kernel, dim3(1), dim3(256), 0, s, ref, d_members, cs, d_prep);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    switch (pad_pow2(cs)) {
        case 16: launch_spearman_n<16, 4>(d_members, d_prep, d_out, num_voxels, cs, s); break;
        case 32: launch_spearman_n<32, 4>(d_members, d_prep, d_out, num_voxels, cs, s); break;
        case 64:
            switch (env_waves(2)) {
                case 1: launch_spearman_n<64, 1>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                case 3: launch_spearman_n<64, 3>(d_members, d_prep, d_out, num_voxels, cs, s); break;
                default: launch_spearman_n<64, 2>(d_members, d_prep, d_out, num_voxels, cs, s); break;
            }
            break;
        default: launch_spearman_n<128, 1>(d_members, d_prep, d_out, num_voxels, cs, s); break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "spearman_kernel";
    return hipGetLastError();
}

hipError_t launch_kendall(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref, float* d_prep,
                          float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const int n_pad = pad_pow2(cs);
    int* prep = reinterpret_cast<int*>(d_prep);
    hipLaunchKernelGGL(kendall_prep_kernel, dim3(1), dim3(256), 0, s, ref, d_members, cs, n_pad, prep);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    switch (n_pad) {
        case 16: launch_kendall_n<16, 4>(d_members, prep, d_out, num_voxels, cs, s); break;
        case 32: launch_kendall_n<32, 4>(d_members, prep, d_out, num_voxels, cs, s); break;
        case 64:
            switch (env_waves(1)) {
                case 3: launch_kendall_n<64, 3>(d_members, prep, d_out, num_voxels, cs, s); break;
                case 2: launch_kendall_n<64, 2>(d_members, prep, d_out, num_voxels, cs, s); break;
                default: launch_kendall_n<64, 1>(d_members, prep, d_out, num_voxels, cs, s); break;
            }
            break;
        default: launch_kendall_n<128, 1>(d_members, prep, d_out, num_voxels, cs, s); break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "kendall_kernel";
    return hipGetLastError();
}

}  // namespace crf
