// kernels_symmetric_binned.hip -- binned mutual information of the SEPARATE_SYMMETRIC field mode (see
// kernels_symmetric.hip; its own translation unit so that the two compile side by side).
//   codes b1 << 8 | b0 sorted -> joint cells and Y bins as runs; the low bytes sorted again -> X bins.
//   Voxels with skipped samples (NaN after normalisation) take the O(cs^2) path over the LDS code column.
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"
#include "crf_mi_device.h"

namespace crf {

namespace {

// the first N - 16 slots are members whatever cs is: the launcher picks the smallest multiple of 16 that holds cs
template <int N>
constexpr int sure_slots() {
    return N - 16;
}

}  // namespace

struct SymmetricSortBinnedArgs {
    int num_bins;  // <= 255
    float min_x, max_x, min_y, max_y;
    int to_cc;
};

template <int N, bool EXACT, int MIN_WAVES>
__global__ __launch_bounds__(64, MIN_WAVES) void binned_symmetric_kernel(const float* const* __restrict__ members_x,
                                                                         const float* const* __restrict__ members_y,
                                                                         const double* __restrict__ tableT,
                                                                         float* __restrict__ out, size_t num_voxels,
                                                                         int cs, SymmetricSortBinnedArgs ba) {
    __shared__ double T[N + 1];  // T[c] = (c/cs) ln(c/cs), T[0] = 0
    __shared__ uint16_t codes[N * 64];
    constexpr int SURE = sure_slots<N>();
    const int lane = threadIdx.x;
    for (int i = lane; i <= N; i += 64) T[i] = i <= cs ? tableT[i] : 0.0;
    __syncthreads();
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;
    const auto is_member = [cs](int e) { return EXACT || e < SURE || e < cs; };
    const float range_x = ba.max_x - ba.min_x, range_y = ba.max_y - ba.min_y;
    const double nbd = double(ba.num_bins);
    const int nb = ba.num_bins;
    uint32_t a[N];
    bool is_nan = false;
    int total = 0;
    {
        float x[N];
#pragma unroll
        for (int e = 0; e < N; e++)
            x[e] = load_member_nt(members_x[is_member(e) ? e : cs - 1], bytes,
                                  is_member(e) ? byte_offset : kOutOfRangeOffset);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const bool member = is_member(e);
            is_nan |= member && (x[e] != x[e]);
            const float x01 = (x[e] - ba.min_x) / range_x;  // CorrelationCalculator.cpp:1061-1062
            int b0 = int(double(x01) * nbd);
            b0 = b0 < 0 ? 0 : (b0 > nb - 1 ? nb - 1 : b0);
            a[e] = (member && x01 == x01) ? uint32_t(b0) : kPadCode;
        }
        uint32_t nan_flag = is_nan ? 1u : 0u;  // pinned: otherwise the samples stay alive to the end (kernels_binned.hip)
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float y[N];
#pragma unroll
        for (int e = 0; e < N; e++)
            y[e] = load_member_nt(members_y[is_member(e) ? e : cs - 1], bytes,
                                  is_member(e) ? byte_offset : kOutOfRangeOffset);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const bool member = is_member(e);
            is_nan |= member && (y[e] != y[e]);
            const float y01 = (y[e] - ba.min_y) / range_y;
            int b1 = int(double(y01) * nbd);
            b1 = b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
            const bool valid = member && (y01 == y01) && a[e] != kPadCode;
            a[e] = valid ? (uint32_t(b1) << 8) | a[e] : kPadCode;
            total += valid ? 1 : 0;
        }
        uint32_t nan_flag = is_nan ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    }
    const bool slow = total != cs;
    const bool any_slow = __any(slow);
    if (any_slow) {
#pragma unroll
        for (int e = 0; e < N; e++)
            if (is_member(e)) codes[e * 64 + lane] = uint16_t(a[e] & 0xFFFFu);  // pad -> 0xFFFF
    }
    __builtin_amdgcn_sched_barrier(0);
    SortNet32<N>::sort(a);
    __builtin_amdgcn_sched_barrier(0);
    // runs of equal codes = joint cells, runs of equal high bytes = Y bins; a pad position contributes T[0] = 0
    double h_y = 0.0, joint = 0.0;
    {
        uint32_t cell_len = 0, col_len = 0;
#pragma unroll
        for (int p = 0; p < N; p++) {
            const bool member = is_member(p);
            uint32_t next = kPadCode;
            if (p + 1 < N) next = is_member(p + 1) ? a[p + 1] : kPadCode;
            cell_len++;
            col_len++;
            const bool end_cell = member && next != a[p];
            const bool end_col = member && (next >> 8) != (a[p] >> 8);
            joint += T[end_cell ? cell_len : 0u];
            h_y += T[end_col ? col_len : 0u];
            cell_len = end_cell ? 0u : cell_len;
            col_len = end_col ? 0u : col_len;
        }
    }
    // X bins: the low bytes sorted on their own (pads stay the largest code)
#pragma unroll
    for (int e = 0; e < N; e++) a[e] = a[e] == kPadCode ? kPadCode : (a[e] & 0xFFu);
    __builtin_amdgcn_sched_barrier(0);
    SortNet32<N>::sort(a);
    __builtin_amdgcn_sched_barrier(0);
    double h_x = 0.0;
    {
        uint32_t len = 0;
#pragma unroll
        for (int p = 0; p < N; p++) {
            const bool member = is_member(p);
            uint32_t next = kPadCode;
            if (p + 1 < N) next = is_member(p + 1) ? a[p + 1] : kPadCode;
            len++;
            const bool end = member && next != a[p];
            h_x += T[end ? len : 0u];
            len = end ? 0u : len;
        }
    }
    double mi = joint - h_x - h_y;

    if (any_slow && slow) {
        // Samples were skipped: probabilities are c/total with total < cs.  Direct O(cs^2) evaluation over the lane's
        // LDS column; the first occurrence of each bin / cell contributes its term (as mi_binned_kernel).
        mi = 0.0;
        if (total > 0) {
            const double tot = double(total);
            const double eps1 = 0.5 / double(cs);
            const double eps2 = 0.5 / double(cs * cs);
#pragma unroll 1
            for (int i = 0; i < cs; i++) {
                const uint32_t ci = codes[i * 64 + lane];
                if (ci == 0xFFFFu) continue;
                int cx = 0, cy = 0, cxy = 0;
                bool fx = true, fy = true, fxy = true;
#pragma unroll 1
                for (int j = 0; j < cs; j++) {
                    const uint32_t cj = codes[j * 64 + lane];
                    if (cj == 0xFFFFu) continue;
                    const bool ex = (cj & 0xFFu) == (ci & 0xFFu);
                    const bool ey = (cj >> 8) == (ci >> 8);
                    cx += ex;
                    cy += ey;
                    cxy += (ex && ey);
                    if (j < i) {
                        fx = fx && !ex;
                        fy = fy && !ey;
                        fxy = fxy && !(ex && ey);
                    }
                }
                if (fx) {
                    const double p = double(cx) / tot;
                    if (p > eps1) mi -= p * log(p);
                }
                if (fy) {
                    const double p = double(cy) / tot;
                    if (p > eps1) mi -= p * log(p);
                }
                if (fxy) {
                    const double p = double(cxy) / tot;
                    if (p > eps2) mi += p * log(p);
                }
            }
        }
    }
    float res = float(mi);
    if (ba.to_cc) res = mi_to_cc(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (cs == 1) res = 1.0f;
    if (v < num_voxels) store_result_nt(out + v, res);
}

namespace {

template <template <int, bool, int> class Launcher, int N, int WAVES, class... Args>
void launch_exact_or_guarded(int cs, Args... args) {
    if (cs == N)
        Launcher<N, true, WAVES>::launch(args...);
    else
        Launcher<N, false, WAVES>::launch(args...);
}

template <int N, bool EXACT, int WAVES>
struct BinnedLauncher {
    static void launch(const float* const* mx, const float* const* my, const double* tableT, float* out,
                       size_t num_voxels, int cs, SymmetricSortBinnedArgs ba, hipStream_t s) {
        hipLaunchKernelGGL((binned_symmetric_kernel<N, EXACT, WAVES>), dim3(unsigned((num_voxels + 63) / 64)), dim3(64), 0,
                           s, mx, my, tableT, out, num_voxels, cs, ba);
    }
};

}  // namespace

hipError_t launch_sorted_symmetric_binned(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                          size_t num_voxels, int measure, int num_bins, float min_x, float max_x,
                                          float min_y, float max_y, const double* d_tables, float* d_out, hipStream_t s) {
    if (cs < 2 || cs > kMaxSortMembers || (measure != 3 && measure != 5)) return hipErrorNotSupported;
    const int n = (cs + 15) / 16 * 16;
#define CRF_SYM_CASE(L, N, W, ...) \
    case N: launch_exact_or_guarded<L, N, W>(cs, __VA_ARGS__); break
    {
        if (num_bins < 1 || num_bins > 255) return hipErrorNotSupported;
        const SymmetricSortBinnedArgs ba{num_bins, min_x, max_x, min_y, max_y, measure == 5};
        const double* tableT = d_tables + (cs + 1);
        switch (n) {
            CRF_SYM_CASE(BinnedLauncher, 16, 4, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 32, 3, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 48, 3, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 64, 3, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 80, 2, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 96, 2, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            CRF_SYM_CASE(BinnedLauncher, 112, 2, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s);
            default: launch_exact_or_guarded<BinnedLauncher, 128, 2>(cs, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, s); break;
        }
    }
#undef CRF_SYM_CASE
    return hipGetLastError();
}

}  // namespace crf
