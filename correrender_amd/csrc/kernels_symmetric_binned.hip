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

// REQ: pair-request mode (crf_compute_requests; request layout {xi, yi, zi, i, xj, yj, zj, j}, HEBChart.hpp:166-168):
// X = the members at voxel i, Y = the members at voxel j, both normalised with the PAIR's own extrema over the 2 cs
// values (HEBChartCorrelation.cpp:493-600), cached loads, optional |.|.
struct BinnedRequestArgs {
    const uint32_t* requests;
    int xs, ys;
    int use_abs;
};

// UDIV (field mode): both ranges lie in [2^-60, 2^60] (the launcher checks): the per-sample division is formed from one
// reciprocal per evaluation with the Markstein correction, exactly as in mi_binned_kernel (kernels_binned.hip: rcp_q)
template <int N, bool EXACT, int MIN_WAVES, bool REQ = false, bool UDIV = false>
__global__ __launch_bounds__(64, MIN_WAVES) void binned_symmetric_kernel(const float* const* __restrict__ members_x,
                                                                         const float* const* __restrict__ members_y,
                                                                         const double* __restrict__ tableT,
                                                                         float* __restrict__ out, size_t num_voxels,
                                                                         int cs, SymmetricSortBinnedArgs ba,
                                                                         size_t num_items, BinnedRequestArgs ra) {
    __shared__ double T[N + 1];  // T[c] = (c/cs) ln(c/cs), T[0] = 0
    __shared__ uint16_t codes[N * 64];
    constexpr int SURE = sure_slots<N>();
    const int lane = threadIdx.x;
    for (int i = lane; i <= N; i += 64) T[i] = i <= cs ? tableT[i] : 0.0;
    __syncthreads();
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const auto is_member = [cs](int e) { return EXACT || e < SURE || e < cs; };
    const float range_x = ba.max_x - ba.min_x, range_y = ba.max_y - ba.min_y;
    const float rcp_x = UDIV ? 1.0f / range_x : 0.0f, rcp_y = UDIV ? 1.0f / range_y : 0.0f;
    const double nbd = double(ba.num_bins);
    const int nb = ba.num_bins;
    uint32_t a[N];
    bool is_nan = false;
    int total = 0;
    if constexpr (REQ) {
        uint32_t offset_x = kOutOfRangeOffset, offset_y = kOutOfRangeOffset;  // items past the end read 0, store nothing
        if (v < num_items) {
            const uint32_t* q = ra.requests + v * 8;
            offset_x = ((q[2] * uint32_t(ra.ys) + q[1]) * uint32_t(ra.xs) + q[0]) * 4u;  // IDXS
            offset_y = ((q[6] * uint32_t(ra.ys) + q[5]) * uint32_t(ra.xs) + q[4]) * 4u;
        }
        float x[N], y[N];
#pragma unroll
        for (int e = 0; e < N; e++) {
            x[e] = load_member_cached(members_x[is_member(e) ? e : cs - 1], bytes, is_member(e) ? offset_x : kOutOfRangeOffset);
            y[e] = load_member_cached(members_y[is_member(e) ? e : cs - 1], bytes, is_member(e) ? offset_y : kOutOfRangeOffset);
        }
        float mn = __uint_as_float(0x7F7FFFFFu), mx = __uint_as_float(0xFF7FFFFFu);  // the pair's own extrema
#pragma unroll
        for (int e = 0; e < N; e++) {
            if (is_member(e)) {
                mn = fminf(mn, fminf(x[e], y[e]));
                mx = fmaxf(mx, fmaxf(x[e], y[e]));
            }
        }
        const float range = mx - mn;
#pragma unroll
        for (int e = 0; e < N; e++) {
            const bool member = is_member(e);
            is_nan |= member && ((x[e] != x[e]) || (y[e] != y[e]));
            const float x01 = (x[e] - mn) / range, y01 = (y[e] - mn) / range;
            int b0 = bin_index_x86(double(x01) * nbd), b1 = bin_index_x86(double(y01) * nbd);
            b0 = b0 < 0 ? 0 : (b0 > nb - 1 ? nb - 1 : b0);
            b1 = b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
            const bool valid = member && (x01 == x01) && (y01 == y01);
            a[e] = valid ? (uint32_t(b1) << 8) | uint32_t(b0) : kPadCode;
            total += valid ? 1 : 0;
        }
        uint32_t nan_flag = is_nan ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    } else {
    const uint32_t byte_offset = uint32_t(v) * 4u;
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
            const float dx = x[e] - ba.min_x;  // CorrelationCalculator.cpp:1061-1062
            float x01;
            if constexpr (UDIV) {
                const float q0 = dx * rcp_x;
                x01 = fmaf(fmaf(-q0, range_x, dx), rcp_x, q0);
            } else {
                x01 = dx / range_x;
            }
            int b0 = bin_index_x86(double(x01) * nbd);
            b0 = b0 < 0 ? 0 : (b0 > nb - 1 ? nb - 1 : b0);
            // UDIV: with finite min and range the division is NaN iff the sample is (an infinite sample makes the fma
            // chain NaN where the division gives inf: both fall into bin 0)
            a[e] = (member && (UDIV ? dx == dx : x01 == x01)) ? uint32_t(b0) : kPadCode;
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
            const float dy = y[e] - ba.min_y;
            float y01;
            if constexpr (UDIV) {
                const float q0 = dy * rcp_y;
                y01 = fmaf(fmaf(-q0, range_y, dy), rcp_y, q0);
            } else {
                y01 = dy / range_y;
            }
            int b1 = bin_index_x86(double(y01) * nbd);
            b1 = b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
            const bool valid = member && (UDIV ? dy == dy : y01 == y01) && a[e] != kPadCode;
            a[e] = valid ? (uint32_t(b1) << 8) | a[e] : kPadCode;
            total += valid ? 1 : 0;
        }
        uint32_t nan_flag = is_nan ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    }
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
    if (REQ && ra.use_abs) res = fabsf(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (cs == 1) res = 1.0f;
    if (v < num_items) store_result_nt(out + v, res);
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
                       size_t num_voxels, int cs, SymmetricSortBinnedArgs ba, size_t num_items, BinnedRequestArgs ra,
                       hipStream_t s) {
        const dim3 grid(unsigned((num_items + 63) / 64));
        if (ra.requests) {
            // request mode holds both vectors' samples at once (2 N + N registers): one wave fewer than field mode
            constexpr int RW = WAVES > 1 ? WAVES - 1 : 1;
            if constexpr (!EXACT)
                hipLaunchKernelGGL((binned_symmetric_kernel<N, false, RW, true>), grid, dim3(64), 0, s, mx, my, tableT,
                                   out, num_voxels, cs, ba, num_items, ra);
        } else {
            const float rx = ba.max_x - ba.min_x, ry = ba.max_y - ba.min_y;  // the kernel's own fp32 subtractions
            const char* plain = getenv("CRF_BINNED_PLAIN_DIV");              // tuning / tests
            const bool udiv = rx >= 0x1p-60f && rx <= 0x1p60f && ry >= 0x1p-60f && ry <= 0x1p60f && !(plain && *plain == '1');
            if (udiv)
                hipLaunchKernelGGL((binned_symmetric_kernel<N, EXACT, WAVES, false, true>), grid, dim3(64), 0, s, mx, my,
                                   tableT, out, num_voxels, cs, ba, num_items, ra);
            else
                hipLaunchKernelGGL((binned_symmetric_kernel<N, EXACT, WAVES, false, false>), grid, dim3(64), 0, s, mx, my,
                                   tableT, out, num_voxels, cs, ba, num_items, ra);
        }
    }
};

}  // namespace

namespace {

hipError_t launch_sorted_binned(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                size_t num_voxels, const SymmetricSortBinnedArgs& ba, const double* d_tables, float* d_out,
                                size_t num_items, const BinnedRequestArgs& ra, hipStream_t s) {
    const int n = (cs + 15) / 16 * 16;
    const int cs_sel = ra.requests ? -1 : cs;  // request mode: the guarded instantiations only
    const double* tableT = d_tables + (cs + 1);
#define CRF_SYM_CASE(N, W) \
    case N: launch_exact_or_guarded<BinnedLauncher, N, W>(cs_sel, d_members_x, d_members_y, tableT, d_out, num_voxels, cs, ba, num_items, ra, s); break
    switch (n) {
        CRF_SYM_CASE(16, 4);
        CRF_SYM_CASE(32, 3);
        CRF_SYM_CASE(48, 3);
        CRF_SYM_CASE(64, 3);
        CRF_SYM_CASE(80, 2);
        CRF_SYM_CASE(96, 2);
        CRF_SYM_CASE(112, 2);
        CRF_SYM_CASE(128, 2);
        default: return hipErrorNotSupported;
    }
#undef CRF_SYM_CASE
    return hipGetLastError();
}

}  // namespace

hipError_t launch_sorted_symmetric_binned(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                          size_t num_voxels, int measure, int num_bins, float min_x, float max_x,
                                          float min_y, float max_y, const double* d_tables, float* d_out, hipStream_t s) {
    if (cs < 2 || cs > kMaxSortMembers || (measure != 3 && measure != 5)) return hipErrorNotSupported;
    if (num_bins < 1 || num_bins > 255) return hipErrorNotSupported;
    const SymmetricSortBinnedArgs ba{num_bins, min_x, max_x, min_y, max_y, measure == 5};
    return launch_sorted_binned(d_members_x, d_members_y, cs, num_voxels, ba, d_tables, d_out, num_voxels,
                                BinnedRequestArgs{nullptr, 0, 0, 0}, s);
}

// binned MI / its correlation coefficient for pair requests; hipErrorNotSupported -> pair_request_kernel
hipError_t launch_sorted_requests_binned(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs,
                                         int ys, size_t num_voxels, const uint32_t* d_requests, size_t num_requests,
                                         int measure, int num_bins, int use_abs, const double* d_tables, float* d_out,
                                         hipStream_t s) {
    if (cs < 2 || cs > kMaxSortMembers || (measure != 3 && measure != 5) || !d_requests) return hipErrorNotSupported;
    if (num_bins < 1 || num_bins > 255) return hipErrorNotSupported;
    if (num_requests == 0) return hipSuccess;
    const SymmetricSortBinnedArgs ba{num_bins, 0.f, 0.f, 0.f, 0.f, measure == 5};
    return launch_sorted_binned(d_members_i, d_members_j, cs, num_voxels, ba, d_tables, d_out, num_requests,
                                BinnedRequestArgs{d_requests, xs, ys, use_abs}, s);
}

}  // namespace crf
