// kernels_binned.hip -- binned mutual information on gfx950 (the Kraskov estimator: kernels_kraskov.hip).
//
// Binned MI (computeMutualInformationBinned<double>, MutualInformation.cpp:45-143; driver
// CorrelationCalculator.cpp:820-846,1026-1085).  The reference fills an 80x80 fp64 histogram per voxel although at
// most cs cells are occupied.  Here one lane owns one voxel and works on the <= cs occupied cells only: each sample
// becomes a 16-bit cell code (b1 << 8 | b0), the codes are sorted by a register min/max network, and one scan over
// the sorted codes yields the run lengths of equal cells (joint histogram) and of equal b1 (query marginal).  With
// every sample valid, a cell/marginal probability is c/cs for an integer count c, so p*ln(p) comes from a cs+1 entry
// fp64 table built on the host with the same libm log as the reference; the reference-vector marginal is voxel
// independent and is summed once per evaluation.  MI = -sum_x - sum_y + sum_xy, accumulated in fp64, returned as
// float.  fp64 sums are taken in sorted-cell order, which is not the reference's bin-index order: a difference of
// a few 1e-16 relative before the final cast to float (tolerance 1e-5 relative per the north star; in practice the
// float results are bit-identical except at rounding boundaries).  Voxels with skipped samples (normalised value
// NaN, MutualInformation.cpp:64 -- needs infinities in the data or max == min) take a compact O(cs^2) path.
//
// Kraskov kNN MI (KSG-1 computeMutualInformationKraskov<double>, MutualInformation.cpp:399-444; KSG-2 :449-509;
// averageDigamma :167-259).  fp64 throughout, as the reference.  One lane owns one voxel; the voxel's cs values are
// parked in that lane's LDS column; for each point a brute-force Chebyshev k-select over all points keeps the k+1
// smallest distances in registers (sorted insertion, min/max only), then the marginal counts are brute-force
// compares against [c-r, c+r); psi(n) comes from a table.  The tie-breaking noise (1e-10 * u) uses this repo's
// documented xorshift32 stream, identical to oracle/corr_oracle.cpp (sgl's generator is not available).
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

#include <cstdlib>

#include "crf_mi_device.h"

namespace crf {

// Ends a sorting network for the compiler: an empty asm "modifies" every element, so nothing that follows is mixed into the
// network's last stages.  Left alone the compiler starts the scans / searches that consume the sorted array while the last
// exchanges are still pending, the live ranges of both overlap, and a network over N values needs ~2N registers
// (spearman_u32_kernel: 116 B of scratch per lane at 128 members, 25.1 -> 21.5 ms at 512^3 x 128 once they were gone;
// mi_binned_kernel: 40 B of scratch at 64 / 128 members gone, three waves per SIMD instead of two at 96 members: 2.33 ->
// 2.15 ms at 256^3).  Measured neutral for the other rank kernels, slightly negative (+2 %) for the two-field kernels of
// kernels_symmetric*.hip, which therefore do without it.
template <class T, int N>
__device__ __forceinline__ void pin_array(T (&a)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(a[i]));
}

// ---------------------------------------------------------------------------------------------------------
// Binned MI: reference-side preparation.
//   prep (int32 view): [0, N) b0_e (kInvalidBin when the normalised reference value is NaN), [N] = 1 if every reference
//   sample is valid; prep (fp64 view) at byte offset kBinnedSxOffset: SX = sum over occupied reference bins of p ln p.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void binned_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                         int cs, int n_pad, int nb, float min_ref, float max_ref,
                                                         const double* __restrict__ tableT, int* __restrict__ prep) {
    extern __shared__ int b0s[];  // cs ints
    __shared__ int all_valid;
    if (threadIdx.x == 0) all_valid = 1;
    __syncthreads();
    for (int e = threadIdx.x; e < n_pad; e += 64) {
        int b = kInvalidBin;
        if (e < cs) {
            const float r01 = (load_ref(src, members, e) - min_ref) / (max_ref - min_ref);  // CorrelationCalculator.cpp:830-832
            if (r01 == r01) {
                int t = bin_index_x86(double(r01) * double(nb));
                b = t < 0 ? 0 : (t > nb - 1 ? nb - 1 : t);
            } else {
                atomicAnd(&all_valid, 0);
            }
            b0s[e] = b;
        }
        prep[e] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sx = 0.0;
        for (int b = 0; b < nb; b++) {
            int c = 0;
            for (int e = 0; e < cs; e++) c += (b0s[e] == b);
            sx += tableT[c];  // tableT[0] == 0
        }
        prep[n_pad] = all_valid;
        *reinterpret_cast<double*>(reinterpret_cast<char*>(prep) + kBinnedSxOffset) = sx;
    }
}

// LC: slots loaded per batch (N = all loads first; a smaller batch bounds the live values to N codes + LC samples, which
// keeps the wide instantiations inside the 256 registers of two waves per SIMD -- 63 loads in flight is the hardware
// limit anyway)
// UDIV: the launcher found range = max_q - min_q inside [2^-60, 2^60] (see the comment at rcp_q below)
template <int N, bool EXACT, int MIN_WAVES, int LC = N, bool UDIV = false>
__global__ __launch_bounds__(64, MIN_WAVES) void mi_binned_kernel(const float* const* __restrict__ members,
                                                                  const int* __restrict__ prep,
                                                                  const double* __restrict__ tableT,
                                                                  float* __restrict__ out, size_t num_voxels, int cs,
                                                                  int nb, float min_q, float max_q, int to_cc) {
    __shared__ double T[N + 1];
    __shared__ uint16_t codes[N * 64];
    const int lane = threadIdx.x;
    for (int i = lane; i <= N; i += 64) T[i] = i <= cs ? tableT[i] : 0.0;
    __syncthreads();
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;  // lanes past the end read 0
    const bool ref_all_valid = prep[N] != 0;
    const double sx = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(prep) + kBinnedSxOffset);

    // cs is padded to the next multiple of 16 (launch_mi_binned): only the last 16 slots can be padding
    constexpr int kFirstGuarded = EXACT ? N : N - 16;
    const auto is_member = [cs](int e) { return e < kFirstGuarded || e < cs; };  // folds in the unrolled loops
    uint32_t a[N];
    bool is_nan = false;
    int total = 0;
    const float range_q = max_q - min_q;
    const double nbd = double(nb);
    // (y - min) / range with ONE true division per evaluation: range is the same for every sample, and with
    // rcp = RN(1 / range) the quotient q = fma(fma(-q0, range, d), rcp, q0), q0 = RN(d * rcp), is the correctly rounded
    // d / range (Markstein; crf_device.h: exact_div) whenever the remainder is exact.  Where it is not -- |d| < 2^-100,
    // or a quotient that leaves the normal range -- the quotient is below 2^-40 or infinite either way and lands in the
    // same bin (0) as the division's; for d = +-inf (or an overflowing q0) the chain yields NaN where the division
    // yields +-inf: both convert to bin 0 (bin_index_x86), and whether the sample counts is decided from the sample
    // itself (with finite min and range the division is NaN iff the sample is).  range outside [2^-60, 2^60] (incl. 0, NaN: max == min, infinite extrema) keeps the division:
    // the launcher picks the instantiation (a branch per sample in the unrolled loop costs registers).
    const float rcp_q = UDIV ? 1.0f / range_q : 0.0f;
    static_assert(N % LC == 0, "whole batches");
#pragma unroll
    for (int base = 0; base < N; base += LC) {
        // a batch of loads first, branch free: a slot past cs loads at an out-of-range offset (0, no memory request)
        // and its code is forced to the pad code below
        float y[LC];
#pragma unroll
        for (int i = 0; i < LC; i++) {
            const int e = base + i;
            y[i] = load_member_nt(members[is_member(e) ? e : cs - 1], bytes,
                                  is_member(e) ? byte_offset : kOutOfRangeOffset);
        }
#pragma unroll
        for (int i = 0; i < LC; i++) {
            const int e = base + i;
            const bool member = is_member(e);
            is_nan |= member && (y[i] != y[i]);
            const float d = y[i] - min_q;  // CorrelationCalculator.cpp:1061-1062
            float q01;
            bool not_nan;
            if constexpr (UDIV) {
                const float q0 = d * rcp_q;
                q01 = fmaf(fmaf(-q0, range_q, d), rcp_q, q0);
                not_nan = d == d;  // min and range are finite here: the division is NaN iff the sample (and so d) is
            } else {
                q01 = d / range_q;
                not_nan = q01 == q01;
            }
            // reference bin of the member; kInvalidBin = 0xFFFF for a skipped reference sample (and for pads): OR-ed
            // into the code it leaves the low 16 bits all ones, which IS the pad marker of the slow path below -- no
            // per-member test needed here (64 uniform conditions held in SGPR pairs cost lane spills); such an
            // evaluation has ref_all_valid == false and every lane recounts its samples on the slow path
            const int b0 = prep[e];
            const bool valid = member && not_nan;
            int b1 = bin_index_x86(double(q01) * nbd);
            b1 = b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
            a[e] = valid ? (uint32_t(b1) << 8) | uint32_t(b0) : kPadCode;
            total += valid ? 1 : 0;
        }
        if (LC < N) __builtin_amdgcn_sched_barrier(0);  // the next batch is not hoisted above this one's conversion
        // pin the NaN flag here: left alone the compiler sinks the y != y compares to the end of the kernel and keeps
        // all N samples alive (in scratch from N = 96 on) across the sort
        uint32_t nan_flag = is_nan ? 1u : 0u;
        asm volatile("" : "+v"(nan_flag));
        is_nan = nan_flag != 0u;
    }
    // (a lane with a NaN member stores NaN whatever the histogram says: it must not drag itself -- and with it its wave,
    // for cs^2 steps -- onto the recount; missing values come in whole regions: 256^3 x 128 with 30 % NaN voxels 75.8 ms)
    const bool slow = !is_nan && ((total != cs) || !ref_all_valid);
    const bool any_slow = __any(slow);
    if (any_slow) {
#pragma unroll
        for (int e = 0; e < N; e++)
            if (is_member(e)) codes[e * 64 + lane] = uint16_t(a[e] & 0xFFFFu);  // pad -> 0xFFFF
    }

    __builtin_amdgcn_sched_barrier(0);
    SortNet32<N>::sort(a);
    pin_array(a);  // the network ends here (crf_device.h)
    __builtin_amdgcn_sched_barrier(0);
    double mi_y = -sx, joint = 0.0;
    uint32_t cell_len = 0, col_len = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
        if (p % 8 == 0) __builtin_amdgcn_sched_barrier(0);  // bounds the table look-ups hoisted ahead of the sums
        // guarded instantiation: the cs - total .. pads (kPadCode, the largest code) sort behind the real samples;
        // a pad position contributes T[0] = 0
        const bool member = is_member(p);
        uint32_t next = kPadCode;
        if (p + 1 < N) next = is_member(p + 1) ? a[p + 1] : kPadCode;
        cell_len++;
        col_len++;
        const bool end_cell = member && next != a[p];
        const bool end_col = member && (next >> 8) != (a[p] >> 8);
        joint += T[end_cell ? cell_len : 0u];
        mi_y -= T[end_col ? col_len : 0u];
        cell_len = end_cell ? 0u : cell_len;
        col_len = end_col ? 0u : col_len;
    }
    double mi = mi_y + joint;

    if (any_slow && slow) {
        // Samples were skipped: probabilities are c/total with total < cs.  Direct O(cs^2) evaluation over the
        // lane's LDS column; first occurrence of each bin/cell contributes its term.
        mi = 0.0;
        int counted = 0;  // samples with a query bin AND a reference bin
#pragma unroll 1
        for (int i = 0; i < cs; i++) counted += codes[i * 64 + lane] != 0xFFFFu;
        if (counted > 0) {
            const double tot = double(counted);
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
    if (to_cc) res = mi_to_cc(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) store_result_nt(out + v, res);
}

// ---------------------------------------------------------------------------------------------------------
// Binned MI in O(cs) per voxel for ANY member count: per-lane histograms in LDS instead of a sort.
//
// The members are visited in the order of their reference bin b0 (binned_hist_prep_kernel sorts them once per
// evaluation; the order and the group boundaries are voxel independent), so the samples of one reference bin -- one
// COLUMN of the joint histogram -- are consecutive and a single per-lane row of num_bins counters, tagged with the
// group index so that it never has to be cleared, holds the cell counts of the current column.  With T[c] = (c/cs)
// ln(c/cs) the cell sum  sum_cells T[count]  is accumulated incrementally: raising a count from c to c+1 adds
// T[c+1] - T[c] (telescoping).  The marginal of the voxel's own values is a second per-lane row, summed at the end.
// Voxels with skipped samples (NaN after normalisation; total < cs, so the table does not apply) and evaluations whose
// reference vector has invalid samples take an O(cs^2) path that re-reads the members from memory -- rare by design.
// LDS per wave: num_bins * 64 * (2 + 4) bytes + the 8 (cs + 1)-byte difference table when it fits.
//   prep (int32 view): [0, cs) perm: member of sorted slot e; [cs, 2cs) b0 of sorted slot e (kInvalidBin last);
//   [2cs] 1 if every reference sample is valid; SX at kBinnedSxOffset as before.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void binned_hist_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                               int cs, int nb, float min_ref, float max_ref,
                                                               const double* __restrict__ tableT,
                                                               int* __restrict__ prep) {
    extern __shared__ int b0s[];  // cs ints
    __shared__ int all_valid;
    if (threadIdx.x == 0) all_valid = 1;
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) {
        const float r01 = (load_ref(src, members, e) - min_ref) / (max_ref - min_ref);  // CorrelationCalculator.cpp:830-832
        int b = kInvalidBin;
        if (r01 == r01) {
            const int t = bin_index_x86(double(r01) * double(nb));
            b = t < 0 ? 0 : (t > nb - 1 ? nb - 1 : t);
        } else {
            atomicAnd(&all_valid, 0);
        }
        b0s[e] = b;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) {
        const int b = b0s[e];
        int pos = 0;
        for (int j = 0; j < cs; j++) pos += (b0s[j] < b || (b0s[j] == b && j < e)) ? 1 : 0;  // kInvalidBin sorts last
        prep[pos] = e;
        prep[cs + pos] = b;
    }
    if (threadIdx.x == 0) {
        double sx = 0.0;
        for (int b = 0; b < nb; b++) {
            int c = 0;
            for (int e = 0; e < cs; e++) c += (b0s[e] == b);
            sx += tableT[c];  // tableT[0] == 0
        }
        prep[2 * cs] = all_valid;
        *reinterpret_cast<double*>(reinterpret_cast<char*>(prep) + kBinnedSxOffset) = sx;
    }
}

__device__ __forceinline__ int binned_query_bin(float y, float min_q, float range_q, double nbd, int nb, bool& valid) {
    const float q01 = (y - min_q) / range_q;  // CorrelationCalculator.cpp:1061-1062
    valid = q01 == q01;
    int b1 = bin_index_x86(double(q01) * nbd);
    return b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
}

__global__ __launch_bounds__(64) void mi_binned_hist_kernel(const float* const* __restrict__ members,
                                                            const int* __restrict__ prep,
                                                            const double* __restrict__ tableT, float* __restrict__ out,
                                                            size_t num_voxels, int cs, int nb, float min_q, float max_q,
                                                            int to_cc, int table_in_lds) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x;
    uint32_t* hist_c = reinterpret_cast<uint32_t*>(smem) + lane;                               // [nb][64] epoch<<16 | n
    uint16_t* hist_y = reinterpret_cast<uint16_t*>(smem + size_t(nb) * 64 * 4) + lane;         // [nb][64]
    double* t_diff = reinterpret_cast<double*>(smem + size_t(nb) * 64 * 6);                    // [cs] T[c+1] - T[c]
    if (table_in_lds)
        for (int c = lane; c < cs; c += 64) t_diff[c] = tableT[c + 1] - tableT[c];
    const int* perm = prep;
    const int* b0s = prep + cs;
    const bool ref_all_valid = prep[2 * cs] != 0;
    const double sx = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(prep) + kBinnedSxOffset);
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const float range_q = max_q - min_q;
    const double nbd = double(nb);
    const size_t tiles = (num_voxels + 63) / 64;
    __syncthreads();
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t v = t * 64 + lane;
        const uint32_t byte_offset = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
#pragma unroll 4
        for (int b = 0; b < nb; b++) {
            hist_y[b * 64] = 0;
            hist_c[b * 64] = 0u;  // epoch 0 is never used (groups are numbered from 1)
        }
        bool is_nan = false;
        int total = 0;
        double joint = 0.0;
        uint32_t epoch = 0u;
        int prev_b0 = -1;
        // chunks of 32 samples: lane l fetches the member pointer and reference bin of sample e0 + l with vector loads
        // (one dependent chain per chunk instead of one per sample), the 32 value loads go out back to back, then the
        // histogram updates run from registers
#pragma unroll 1
        for (int e0 = 0; e0 < cs; e0 += 32) {
            const int mine = e0 + (lane & 31) < cs ? e0 + (lane & 31) : cs - 1;
            const uint64_t ptr = reinterpret_cast<uint64_t>(members[perm[mine]]);
            const uint32_t ptr_lo = uint32_t(ptr), ptr_hi = uint32_t(ptr >> 32);
            const int b0_mine = b0s[mine];
            float y[32];
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const uint64_t base = (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_hi), i))) << 32) |
                                      uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_lo), i)));
                y[i] = load_member_nt(reinterpret_cast<const float*>(base), bytes,
                                      e0 + i < cs ? byte_offset : kOutOfRangeOffset);
            }
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const int b0 = __builtin_amdgcn_readlane(b0_mine, i);  // wave uniform
                const bool member = e0 + i < cs;
                epoch += (member && b0 != prev_b0) ? 1u : 0u;
                prev_b0 = member ? b0 : prev_b0;
                is_nan |= member && (y[i] != y[i]);
                bool valid;
                const int b1 = binned_query_bin(y[i], min_q, range_q, nbd, nb, valid);
                valid = valid && member && b0 != kInvalidBin;
                total += valid ? 1 : 0;
                if (valid) {
                    const uint32_t h = hist_c[b1 * 64];
                    const uint32_t cnt = (h >> 16) == epoch ? (h & 0xFFFFu) : 0u;
                    joint += table_in_lds ? t_diff[cnt] : tableT[cnt + 1] - tableT[cnt];
                    hist_c[b1 * 64] = (epoch << 16) | (cnt + 1u);
                    hist_y[b1 * 64] = uint16_t(hist_y[b1 * 64] + 1u);
                }
            }
        }
        double mi = joint - sx;
#pragma unroll 2
        for (int b = 0; b < nb; b++) mi -= tableT[hist_y[b * 64]];  // tableT[0] == 0
        const bool slow = total != cs || !ref_all_valid;
        if (slow) {
            // probabilities are c/total: direct evaluation, first occurrence of each bin / cell contributes its term
            mi = 0.0;
            if (total > 0) {
                const double tot = double(total);
                const double eps1 = 0.5 / double(cs);
                const double eps2 = 0.5 / double(cs * cs);
#pragma unroll 1
                for (int i = 0; i < cs; i++) {
                    bool vi;
                    const int b1i = binned_query_bin(load_member_nt(members[perm[i]], bytes, byte_offset), min_q, range_q,
                                                     nbd, nb, vi);
                    const int b0i = b0s[i];
                    if (!vi || b0i == kInvalidBin) continue;
                    int cx = 0, cy = 0, cxy = 0;
                    bool fx = true, fy = true, fxy = true;
#pragma unroll 1
                    for (int j = 0; j < cs; j++) {
                        bool vj;
                        const int b1j = binned_query_bin(load_member_nt(members[perm[j]], bytes, byte_offset), min_q,
                                                         range_q, nbd, nb, vj);
                        const int b0j = b0s[j];
                        const bool ok = vj && b0j != kInvalidBin;
                        const bool ex = ok && b0j == b0i;
                        const bool ey = ok && b1j == b1i;
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
        if (to_cc) res = mi_to_cc(res);
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (v < num_voxels) store_result_nt(out + v, res);
    }
}

namespace {

template <int N, int MIN_WAVES, int LC = N>
void launch_binned_n(const float* const* d_members, const int* prep, const double* tableT, float* d_out,
                     size_t num_voxels, int cs, const BinnedArgs& a, hipStream_t s) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    const float range = a.max_query - a.min_query;  // the same fp32 subtraction the kernel performs
    const char* plain = getenv("CRF_BINNED_PLAIN_DIV");  // tuning / tests: always the per-sample division
    const bool udiv = range >= 0x1p-60f && range <= 0x1p60f && !(plain && *plain == '1');
#define CRF_LAUNCH_BINNED(EX, UD)                                                                                      \
    hipLaunchKernelGGL((mi_binned_kernel<N, EX, MIN_WAVES, LC, UD>), dim3(blocks), dim3(64), 0, s, d_members, prep, tableT, \
                       d_out, num_voxels, cs, a.num_bins, a.min_query, a.max_query, int(a.to_cc))
    if (cs == N) {
        if (udiv) CRF_LAUNCH_BINNED(true, true); else CRF_LAUNCH_BINNED(true, false);
    } else {
        if (udiv) CRF_LAUNCH_BINNED(false, true); else CRF_LAUNCH_BINNED(false, false);
    }
#undef CRF_LAUNCH_BINNED
}


}  // namespace

void launch_binned_prep(const RefSource& ref, const float* const* d_members, int cs, int n_pad, const BinnedArgs& a,
                        const double* tableT, int* d_prep, hipStream_t s) {
    hipLaunchKernelGGL(binned_prep_kernel, dim3(1), dim3(64), size_t(cs) * sizeof(int), s, ref, d_members, cs, n_pad,
                       a.num_bins, a.min_ref, a.max_ref, tableT, d_prep);
}


hipError_t launch_mi_binned_hist(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                 const BinnedArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                                 hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const size_t rows = size_t(a.num_bins) * 64 * 6;
    if (rows > 56 * 1024 || size_t(2 * cs + 1) * sizeof(int) > kBinnedSxOffset) return hipErrorNotSupported;
    int* prep = reinterpret_cast<int*>(d_prep);
    const double* tableT = d_tables + (cs + 1);
    if (ref.prepare())
        hipLaunchKernelGGL(binned_hist_prep_kernel, dim3(1), dim3(256), size_t(cs) * sizeof(int), s, ref, d_members, cs,
                           a.num_bins, a.min_ref, a.max_ref, tableT, prep);
    if (!ref.run()) return hipGetLastError();
    const size_t with_table = rows + size_t(cs) * sizeof(double);
    const bool table_in_lds = with_table <= 60 * 1024;
    const size_t tiles = (num_voxels + 63) / 64;
    const unsigned blocks = unsigned(tiles < 16384 ? tiles : 16384);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    hipLaunchKernelGGL(mi_binned_hist_kernel, dim3(blocks), dim3(64), table_in_lds ? with_table : rows, s, d_members, prep,
                       tableT, d_out, num_voxels, cs, a.num_bins, a.min_query, a.max_query, int(a.to_cc),
                       int(table_in_lds));
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "mi_binned_hist_kernel";
    return hipGetLastError();
}

hipError_t launch_mi_binned(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                            const BinnedArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                            hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const int n_pad = (cs + 15) / 16 * 16;
    int* prep = reinterpret_cast<int*>(d_prep);
    const double* tableT = d_tables + (cs + 1);
    if (ref.prepare()) launch_binned_prep(ref, d_members, cs, n_pad, a, tableT, prep, s);
    if (!ref.run()) return hipGetLastError();
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    // waves/SIMD per size from measurements at 256^3 (profiles/tuning_r01.md), after the NaN flag was pinned (the kernel
    // then needs ~2 registers per member): 64 members 1.41 ms at 4 waves (1.46 at 2), 80: 1.89 ms at 3 (2.14 at 2),
    // 128: 3.50 ms at 2 (5.16 at 1)
    switch (n_pad) {
        case 16: launch_binned_n<16, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 32: launch_binned_n<32, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 48: launch_binned_n<48, 3>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 64: launch_binned_n<64, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 80: launch_binned_n<80, 3>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 96: launch_binned_n<96, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 112: launch_binned_n<112, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        default: launch_binned_n<128, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "mi_binned_kernel";
    return hipGetLastError();
}

// any cs (tables must fit LDS: cs <= 2048), k <= 128; hipErrorNotSupported otherwise
}  // namespace crf
