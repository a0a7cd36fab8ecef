// kernels_symmetric.hip -- SEPARATE_SYMMETRIC field mode (CRF_FLAG_SYMMETRIC; CorrelationMain.glsl:10-15 in the reference:
// first field vs second field AT THE SAME VOXEL) for Spearman, Kendall tau-b and binned mutual information, member
// counts up to 128, by per-lane sorting networks.
//
// Both vectors of a voxel depend on the voxel, so nothing can be prepared once per evaluation: the one-reference
// kernels' preparation (ranks / sort order / bins of the reference vector) happens per voxel here, with the same
// register-resident networks (crf_device.h).  The any-member-count fallback (direct_symmetric_kernel,
// kernels_generic.hip) counts in O(cs^2) per voxel; at 256^3 x 64 it needs 25 / 21 / 37 ms where these need a few.
//   Spearman: two sorts -> doubled fractional ranks of X and of Y in two 16-bit LDS columns -> computePearson2<float>
//             on the two rank vectors in member order (Correlation.cpp:141-174).
//   Kendall : sort X as (key, member) -> each member's position in the X order and the last position of its X-tie group
//             in one 16-bit LDS column; sort Y as (key, member); walk in ascending Y with a bitset of seen X positions:
//             positions beyond the visited element's X-tie group are its discordant pairs (elements of the current Y-tie
//             run wait in a second bitset until the run ends); n1 / n2 from the tie runs of the two sorted sequences.
//   Binned  : (kernels_symmetric_binned.hip) codes b1 << 8 | b0 sorted -> joint cells and Y bins as runs; the byte-swapped codes sorted again -> X bins.
//             Voxels with skipped samples (NaN after normalisation) take the O(cs^2) path over the LDS code column.
// One lane per voxel, one wave per block; pads (slots >= cs) load at kOutOfRangeOffset and sort last.
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

namespace {

constexpr uint32_t kPadKey = 0xFFFFFFFFu;  // sorts after every real value (orderable_key(+inf) = 0xFF800000)

// the first N - 16 slots are members whatever cs is: launch_* picks the smallest N (a multiple of 16) that holds cs
template <int N>
constexpr int sure_slots() {
    return N - 16;
}

// CACHED: default cache policy instead of non-temporal loads (pair requests revisit voxels)
template <int N, bool EXACT, bool CACHED = false>
__device__ __forceinline__ void load_composites(composite_t (&a)[N], const float* const* __restrict__ members, int cs,
                                                uint32_t bytes, uint32_t byte_offset) {
    constexpr int SURE = sure_slots<N>();
    float y[N];
#pragma unroll
    for (int e = 0; e < N; e++) {
        const bool real = EXACT || e < SURE || e < cs;
        const uint32_t off = real ? byte_offset : kOutOfRangeOffset;
        y[e] = CACHED ? load_member_cached(members[real ? e : cs - 1], bytes, off)
                      : load_member_nt(members[real ? e : cs - 1], bytes, off);
    }
#pragma unroll
    for (int e = 0; e < N; e++) {
        const float yc = y[e] + 0.0f;  // -0.0 -> +0.0: key equality is float equality
        a[e] = make_composite((EXACT || e < SURE || e < cs) ? orderable_key(yc) : kPadKey, uint32_t(e));
    }
}

// NaNs sort to the ends (see kernels_rank.hip): smallest key below key(-inf), or largest REAL key above key(+inf)
template <int N, bool EXACT>
__device__ __forceinline__ bool sorted_keys_hold_nan(const composite_t (&a)[N], int cs) {
    constexpr int SURE = sure_slots<N>();
    bool is_nan = composite_key(a[0]) < 0x007FFFFFu;
    if constexpr (EXACT) {
        is_nan |= composite_key(a[N - 1]) > 0xFF800000u;
    } else {
#pragma unroll
        for (int p = (SURE > 0 ? SURE - 1 : 0); p < N; p++)
            is_nan |= composite_key(a[p]) > ((p == cs - 1) ? 0xFF800000u : 0xFFFFFFFFu);
    }
    return is_nan;
}

// doubled fractional ranks (2 * rank, computeRanks, Correlation.cpp:277-303) of one side into the lane's LDS column
template <int N, bool EXACT, bool CACHED = false>
__device__ __forceinline__ bool rank_side(const float* const* __restrict__ members, int cs, uint32_t bytes,
                                          uint32_t byte_offset, uint16_t* __restrict__ rank2) {
    composite_t a[N];
    load_composites<N, EXACT, CACHED>(a, members, cs, bytes, byte_offset);
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(a);
    __builtin_amdgcn_sched_barrier(0);
    const bool is_nan = sorted_keys_hold_nan<N, EXACT>(a, cs);
    // Tie-free fast path (as in spearman_kernel): when no lane of the wave has two equal values among its members,
    // 2 * rank = 2 * position + 2 and the two tie-run scans (~8 vector instructions per element) are skipped.  Pads are
    // left out of the test (they tie with each other) and their rank slots are never read as ranks.
    {
        constexpr int SURE = sure_slots<N>();
        uint32_t tie_min = 0xFFFFFFFFu;  // min over neighbouring members of (key ^ previous key): 0 iff a tie
#pragma unroll
        for (int p = 1; p < N; p++) {
            const uint32_t x = composite_key(a[p]) ^ composite_key(a[p - 1]);
            tie_min = min(tie_min, (EXACT || p < SURE || p < cs) ? x : 0xFFFFFFFFu);
        }
        if (!__any(tie_min == 0u)) {
#pragma unroll
            for (int p = 0; p < N; p++)
                if (EXACT || p < SURE || p < cs) rank2[(composite_low(a[p]) & 0xFFu) * 64] = uint16_t(2 * p + 2);
            return is_nan;
        }
    }
    uint32_t run_start = 0;  // forward scan: first position of the tie run, parked in bits 8..15 of the low word
#pragma unroll
    for (int p = 0; p < N; p++) {
        if (p > 0) {
            const bool same = composite_key(a[p]) == composite_key(a[p - 1]);
            run_start = same ? run_start : uint32_t(p);
        }
        a[p] = composite_or_low(a[p], run_start << 8);
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t run_end = 0;  // backward scan: last position of the run; 2 * rank = start + end + 2
#pragma unroll
    for (int p = N - 1; p >= 0; p--) {
        bool same = false;
        if (p < N - 1) same = composite_key(a[p]) == composite_key(a[p + 1]);  // real vs pad: equal only for a NaN
        run_end = same ? run_end : uint32_t(p);
        const uint32_t low = composite_low(a[p]);
        rank2[(low & 0xFFu) * 64] = uint16_t(((low >> 8) & 0xFFu) + run_end + 2u);
        if ((p & 3) == 0) __builtin_amdgcn_sched_barrier(0);
    }
    return is_nan;
}

// What a lane works on.  Field mode (REQ = false): voxel `item` of both fields.  Request mode: pair request `item`
// (layout {xi, yi, zi, i, xj, yj, zj, j}, HEBChart.hpp:166-168): X = the members at voxel i, Y = the members at voxel j.
struct RequestArgs {
    const uint32_t* requests;
    int xs, ys;
    int use_abs;
};
template <bool REQ>
__device__ __forceinline__ void lane_offsets(const RequestArgs& ra, size_t item, size_t num_items, uint32_t& offset_x,
                                             uint32_t& offset_y) {
    if constexpr (REQ) {
        offset_x = offset_y = kOutOfRangeOffset;  // lanes past the end read 0 and store nothing
        if (item < num_items) {
            const uint32_t* q = ra.requests + item * 8;
            offset_x = ((q[2] * uint32_t(ra.ys) + q[1]) * uint32_t(ra.xs) + q[0]) * 4u;  // IDXS
            offset_y = ((q[6] * uint32_t(ra.ys) + q[5]) * uint32_t(ra.xs) + q[4]) * 4u;
        }
    } else {
        offset_x = offset_y = uint32_t(item) * 4u;
    }
}

}  // namespace

template <int N, bool EXACT, int MIN_WAVES, bool REQ = false>
__global__ __launch_bounds__(64, MIN_WAVES) void spearman_symmetric_kernel(const float* const* __restrict__ members_x,
                                                                           const float* const* __restrict__ members_y,
                                                                           float* __restrict__ out, size_t num_voxels,
                                                                           int cs, size_t num_items, RequestArgs ra) {
    __shared__ uint16_t rank2[2 * N * 64];
    constexpr int SURE = sure_slots<N>();
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;  // lanes past the end read 0
    uint32_t byte_offset, offset_y;
    lane_offsets<REQ>(ra, v, num_items, byte_offset, offset_y);
    bool is_nan = rank_side<N, EXACT, REQ>(members_x, cs, bytes, byte_offset, rank2 + lane);
    uint32_t nan_flag = is_nan ? 1u : 0u;
    order_after(nan_flag, offset_y);  // the second side starts after the first is complete
    __builtin_amdgcn_sched_barrier(0);
    is_nan = rank_side<N, EXACT, REQ>(members_y, cs, bytes, offset_y, rank2 + N * 64 + lane);
    is_nan |= nan_flag != 0u;
    __builtin_amdgcn_sched_barrier(0);

    // computePearson2<float>(ranksX, ranksY, cs), sequential fp32, member order; pads contribute +0
    // (N > 64: reading the ranks back from LDS in three rolled passes instead of holding 2N floats was measured and is
    // not faster -- 100 members 11.1 vs 9.4 ms, 128 members 11.8 vs 11.2 ms: the registers are the sort's, not the tail's)
    float rx[N], ry[N];
#pragma unroll
    for (int e = 0; e < N; e++) {
        const bool member = EXACT || e < SURE || e < cs;
        rx[e] = member ? 0.5f * float(rank2[e * 64 + lane]) : 0.0f;
        ry[e] = member ? 0.5f * float(rank2[(N + e) * 64 + lane]) : 0.0f;
    }
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanX = 0.0f, meanY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++) {
        meanX += invN * rx[e];
        meanY += invN * ry[e];
    }
    float varX = 0.0f, varY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++) {
        const bool member = EXACT || e < SURE || e < cs;
        rx[e] = member ? rx[e] - meanX : 0.0f;
        ry[e] = member ? ry[e] - meanY : 0.0f;
        varX += invNm1 * rx[e] * rx[e];
        varY += invNm1 * ry[e] * ry[e];
    }
    const float sdX = sqrtf(varX), sdY = sqrtf(varY);
    float r = 0.0f;
    if (__all(exact_div_guard(meanX, sdX) && exact_div_guard(meanY, sdY))) {
        const float rcpX = 1.0f / sdX, rcpY = 1.0f / sdY;
#pragma unroll
        for (int e = 0; e < N; e++) r += invNm1 * exact_div(rx[e], sdX, rcpX) * exact_div(ry[e], sdY, rcpY);
    } else {
#pragma unroll
        for (int e = 0; e < N; e++) {
            const bool member = EXACT || e < SURE || e < cs;
            r += member ? invNm1 * (rx[e] / sdX) * (ry[e] / sdY) : 0.0f;
        }
    }
    if (REQ && ra.use_abs) r = fabsf(r);
    if (is_nan) r = __uint_as_float(0x7FC00000u);
    if (v < num_items) store_result_nt(out + v, r);
}

template <int N, bool EXACT, int MIN_WAVES, bool REQ = false>
__global__ __launch_bounds__(64, MIN_WAVES) void kendall_symmetric_kernel(const float* const* __restrict__ members_x,
                                                                          const float* const* __restrict__ members_y,
                                                                          float* __restrict__ out, size_t num_voxels,
                                                                          int cs, size_t num_items, RequestArgs ra) {
    __shared__ uint16_t xinfo[N * 64];  // [member][lane]: position in the X order | last position of its X-tie group << 8
    // N > 64: the Y order as a second column, [position][lane]: member | (same Y as the previous position) << 8
    constexpr bool kRolledWalk = N > 64;
    __shared__ uint16_t yseq[kRolledWalk ? N * 64 : 64];
    constexpr int SURE = sure_slots<N>();
    const int lane = threadIdx.x;
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    uint32_t byte_offset, offset_y;
    lane_offsets<REQ>(ra, v, num_items, byte_offset, offset_y);
    bool is_nan;
    int32_t n1 = 0;
    {
        composite_t a[N];
        load_composites<N, EXACT, REQ>(a, members_x, cs, bytes, byte_offset);
        __builtin_amdgcn_sched_barrier(0);
        SortNet<N>::sort(a);
        __builtin_amdgcn_sched_barrier(0);
        is_nan = sorted_keys_hold_nan<N, EXACT>(a, cs);
        // ties in X: a run of t equal values contributes t(t-1)/2 (computeTiesB, Correlation.cpp:305-329); pads excluded
        int32_t run = 0;
#pragma unroll
        for (int p = 1; p < N; p++) {
            const bool same = composite_key(a[p]) == composite_key(a[p - 1]);
            run = ((EXACT || p < SURE || p < cs) && same) ? run + 1 : 0;
            n1 += run;
        }
        uint32_t run_end = 0;
#pragma unroll
        for (int p = N - 1; p >= 0; p--) {
            bool same = false;
            if (p < N - 1) same = composite_key(a[p]) == composite_key(a[p + 1]);
            run_end = same ? run_end : uint32_t(p);
            xinfo[(composite_low(a[p]) & 0xFFu) * 64 + lane] = uint16_t(uint32_t(p) | (run_end << 8));
            if ((p & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    uint32_t nan_flag = is_nan ? 1u : 0u;
    order_after(nan_flag, offset_y);  // the Y side starts after the X composites are dead
    int32_t n1_pinned = n1;
    uint32_t lane_b = uint32_t(lane);
    order_after(n1_pinned, lane_b);
    is_nan = nan_flag != 0u;
    n1 = n1_pinned;
    composite_t b[N];  // (Y key, member)
    load_composites<N, EXACT, REQ>(b, members_y, cs, bytes, offset_y);
    __builtin_amdgcn_sched_barrier(0);
    SortNet<N>::sort(b);
    __builtin_amdgcn_sched_barrier(0);
    is_nan |= sorted_keys_hold_nan<N, EXACT>(b, cs);

    // Walk in ascending Y.  A visited element is discordant with every element of strictly smaller Y whose X position
    // lies beyond its own X-tie group; elements of the current Y-tie run must not count, so they wait in `pending` and
    // join `seen` when the run ends (the one-reference kernels order equal Y by X position instead, which needs the Y
    // values permuted into X order first).  Pads sort last on both sides and their X-tie group ends at N - 1.
    constexpr int W = (N + 63) / 64;
    uint64_t seen[W], pending[W];
#pragma unroll
    for (int w = 0; w < W; w++) seen[w] = pending[w] = 0ull;
    int32_t discordant = 0, n2 = 0, run = 0;
    uint32_t prev_key = 0;
    uint32_t info[8];
    if constexpr (kRolledWalk) {
        // More than 64 members: the 2N registers of the sorted composites leave no room for an unrolled walk (r01: every
        // look-up and mask hoisted, 1-2 KB of scratch per lane, 19 / 27 ms at 100 / 128 members).  The Y order goes to
        // LDS instead -- member and tie flag per position -- and the walk is a rolled loop over the two columns with a
        // handful of registers; the tie runs of Y are still counted on the sorted keys here.
#pragma unroll
        for (int p = 0; p < N; p++) {
            const uint32_t key = composite_key(b[p]);
            const bool same = p > 0 && key == prev_key;
            if (p > 0) {
                run = ((EXACT || p < SURE || p < cs) && same) ? run + 1 : 0;  // ties in Y: t(t-1)/2 per run
                n2 += run;
            }
            prev_key = key;
            yseq[p * 64 + lane] = uint16_t((composite_low(b[p]) & 0xFFu) | (same ? 0x100u : 0u));
            if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int p0 = 0; p0 < N; p0 += 8) {
            uint32_t ys[8], xi[8];
#pragma unroll
            for (int q = 0; q < 8; q++) ys[q] = yseq[(p0 + q) * 64 + lane];  // two levels of independent reads
#pragma unroll
            for (int q = 0; q < 8; q++) xi[q] = xinfo[(ys[q] & 0xFFu) * 64 + lane];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const bool same = (ys[q] & 0x100u) != 0u;
                const uint32_t slot = xi[q] & 0xFFu, g = xi[q] >> 8;
                const uint64_t gm = 0xFFFFFFFFFFFFFFFEull << (g & 63u);
                const uint64_t sbit = 1ull << (slot & 63u);
#pragma unroll
                for (int w = 0; w < W; w++) {
                    seen[w] |= same ? 0ull : pending[w];
                    pending[w] = same ? pending[w] : 0ull;
                    const uint64_t mask = (uint32_t(w) > (g >> 6)) ? ~0ull : ((uint32_t(w) == (g >> 6)) ? gm : 0ull);
                    discordant += __popcll(seen[w] & mask);
                    pending[w] |= (uint32_t(w) == (slot >> 6)) ? sbit : 0ull;
                }
            }
        }
    } else {
#pragma unroll
    for (int p = 0; p < N; p++) {
        const uint32_t key = composite_key(b[p]);
        const bool same = p > 0 && key == prev_key;
        if (p > 0) {
            run = ((EXACT || p < SURE || p < cs) && same) ? run + 1 : 0;  // ties in Y: t(t-1)/2 per run
            n2 += run;
        }
        prev_key = key;
        // the X-order look-ups are fetched 8 at a time, each batch ordered behind the walk of the previous one (left
        // alone, all N look-ups and masks are hoisted en bloc: scratch); N > 64 (one wave, AGPR overflow): every fourth
        if constexpr (N <= 64) {
            if ((p & 7) == 0) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    uint32_t mq = composite_low(b[p + q]) & 0xFFu;
                    order_after(mq, seen[0]);
                    info[q] = uint32_t(xinfo[mq * 64 + lane_b]);
                }
            }
        } else {
            uint32_t mq = composite_low(b[p]) & 0xFFu;
            if ((p & 3) == 0) order_after(mq, seen[0]);
            info[p & 7] = uint32_t(xinfo[mq * 64 + lane_b]);
        }
        const uint32_t slot = info[p & 7] & 0xFFu, g = info[p & 7] >> 8;
        if constexpr (W == 1) {
            seen[0] |= same ? 0ull : pending[0];
            pending[0] = same ? pending[0] : 0ull;
            discordant += __popcll(seen[0] & (0xFFFFFFFFFFFFFFFEull << g));
            pending[0] |= 1ull << slot;
        } else {
            const uint64_t gm = 0xFFFFFFFFFFFFFFFEull << (g & 63u);
            const uint64_t sbit = 1ull << (slot & 63u);
#pragma unroll
            for (int w = 0; w < W; w++) {
                seen[w] |= same ? 0ull : pending[w];
                pending[w] = same ? pending[w] : 0ull;
                const uint64_t mask = (uint32_t(w) > (g >> 6)) ? ~0ull : ((uint32_t(w) == (g >> 6)) ? gm : 0ull);
                discordant += __popcll(seen[w] & mask);
                pending[w] |= (uint32_t(w) == (slot >> 6)) ? sbit : 0ull;
            }
        }
        if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    }
    const int32_t n = cs;
    const int32_t n0 = (n * (n - 1)) / 2;
    const int32_t numerator = n0 - n1 - n2 - 2 * discordant;
    float res = float(numerator) / (sqrtf(float(n0 - n1)) * sqrtf(float(n0 - n2)));
    if (REQ && ra.use_abs) res = fabsf(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
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
struct SpearmanLauncher {
    static void launch(const float* const* mx, const float* const* my, float* out, size_t num_voxels, int cs,
                       size_t num_items, RequestArgs ra, hipStream_t s) {
        const dim3 grid(unsigned((num_items + 63) / 64));
        if (ra.requests) {
            if constexpr (!EXACT)  // request mode: the guarded instantiations only
                hipLaunchKernelGGL((spearman_symmetric_kernel<N, false, WAVES, true>), grid, dim3(64), 0, s, mx, my, out,
                                   num_voxels, cs, num_items, ra);
        } else {
            hipLaunchKernelGGL((spearman_symmetric_kernel<N, EXACT, WAVES, false>), grid, dim3(64), 0, s, mx, my, out,
                               num_voxels, cs, num_items, ra);
        }
    }
};
template <int N, bool EXACT, int WAVES>
struct KendallLauncher {
    static void launch(const float* const* mx, const float* const* my, float* out, size_t num_voxels, int cs,
                       size_t num_items, RequestArgs ra, hipStream_t s) {
        const dim3 grid(unsigned((num_items + 63) / 64));
        if (ra.requests) {
            if constexpr (!EXACT)
                hipLaunchKernelGGL((kendall_symmetric_kernel<N, false, WAVES, true>), grid, dim3(64), 0, s, mx, my, out,
                                   num_voxels, cs, num_items, ra);
        } else {
            hipLaunchKernelGGL((kendall_symmetric_kernel<N, EXACT, WAVES, false>), grid, dim3(64), 0, s, mx, my, out,
                               num_voxels, cs, num_items, ra);
        }
    }
};

}  // namespace

namespace {

// exact = may the unguarded instantiation be used (field mode with cs a multiple of 16)
hipError_t launch_sorted_rank(const float* const* d_members_x, const float* const* d_members_y, int cs, size_t num_voxels,
                              int measure, float* d_out, size_t num_items, const RequestArgs& ra, hipStream_t s) {
    const int n = (cs + 15) / 16 * 16;
    const int cs_sel = ra.requests ? -1 : cs;  // request mode: never the unguarded instantiation
#define CRF_SYM_CASE(L, N, W) \
    case N: launch_exact_or_guarded<L, N, W>(cs_sel, d_members_x, d_members_y, d_out, num_voxels, cs, num_items, ra, s); break
    if (measure == 1) {
        switch (n) {
            CRF_SYM_CASE(SpearmanLauncher, 16, 4);
            CRF_SYM_CASE(SpearmanLauncher, 32, 3);
            CRF_SYM_CASE(SpearmanLauncher, 48, 2);
            CRF_SYM_CASE(SpearmanLauncher, 64, 2);
            CRF_SYM_CASE(SpearmanLauncher, 80, 2);
            CRF_SYM_CASE(SpearmanLauncher, 96, 1);
            CRF_SYM_CASE(SpearmanLauncher, 112, 1);
            CRF_SYM_CASE(SpearmanLauncher, 128, 1);
            default: return hipErrorNotSupported;
        }
    } else {
        switch (n) {
            CRF_SYM_CASE(KendallLauncher, 16, 4);
            CRF_SYM_CASE(KendallLauncher, 32, 3);
            CRF_SYM_CASE(KendallLauncher, 48, 2);
            CRF_SYM_CASE(KendallLauncher, 64, 2);
            CRF_SYM_CASE(KendallLauncher, 80, 1);
            CRF_SYM_CASE(KendallLauncher, 96, 1);
            CRF_SYM_CASE(KendallLauncher, 112, 1);
            CRF_SYM_CASE(KendallLauncher, 128, 1);
            default: return hipErrorNotSupported;
        }
    }
#undef CRF_SYM_CASE
    return hipGetLastError();
}

}  // namespace

// cs in [2, 128]; measure 1 Spearman, 2 Kendall, 3 / 5 binned MI / its correlation coefficient; hipErrorNotSupported
// otherwise (the caller then uses direct_symmetric_kernel)
hipError_t launch_sorted_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                   size_t num_voxels, int measure, int num_bins, float min_x, float max_x, float min_y,
                                   float max_y, const double* d_tables, float* d_out, hipStream_t s) {
    if (cs < 2 || cs > kMaxSortMembers) return hipErrorNotSupported;
    if (measure == 3 || measure == 5)
        return launch_sorted_symmetric_binned(d_members_x, d_members_y, cs, num_voxels, measure, num_bins, min_x, max_x,
                                              min_y, max_y, d_tables, d_out, s);
    if (measure != 1 && measure != 2) return hipErrorNotSupported;
    return launch_sorted_rank(d_members_x, d_members_y, cs, num_voxels, measure, d_out, num_voxels,
                              RequestArgs{nullptr, 0, 0, 0}, s);
}

// Pair requests (crf_compute_requests) through the same kernels: Spearman / Kendall, cs in [2, 128]; X = members_i at
// voxel i, Y = members_j at voxel j of each request; hipErrorNotSupported otherwise (-> pair_request_kernel)
hipError_t launch_sorted_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                  size_t num_voxels, const uint32_t* d_requests, size_t num_requests, int measure,
                                  int use_abs, float* d_out, hipStream_t s) {
    if (cs < 2 || cs > kMaxSortMembers || (measure != 1 && measure != 2) || !d_requests) return hipErrorNotSupported;
    if (num_requests == 0) return hipSuccess;
    return launch_sorted_rank(d_members_i, d_members_j, cs, num_voxels, measure, d_out, num_requests,
                              RequestArgs{d_requests, xs, ys, use_abs}, s);
}

}  // namespace crf
