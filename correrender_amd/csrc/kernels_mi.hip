// kernels_mi.hip -- the two mutual-information estimators on gfx950.
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

namespace crf {

constexpr uint32_t kPadCode = 0xFFFFFFFFu;
constexpr int kInvalidBin = 0xFFFF;

// expf as the reference's host libm computes it.  The MI-correlation-coefficient map sqrt(1 - exp(-2 MI)) cancels
// catastrophically for small MI (1 - exp(-2e-4) keeps ~11 bits), so a 1-ulp difference between two expf
// implementations shows up as a 1e-4 relative difference in the result -- outside the 1e-5 tolerance.  glibc >= 2.27
// evaluates expf in double precision with a 32-entry table of 2^(i/32) and a cubic (the ARM optimized-routines
// algorithm: z = x*32/ln2, k = round(z), r = z-k, 2^(k/32) * (C0 r^3 + C1 r^2 + C2 r + 1)); the same IEEE fp64
// operations in the same order give the same float (checked on the host against libm expf on 5e7 inputs).  The table
// is 2^(i/32) rounded to double with i << 47 subtracted from its bits.
__device__ const uint64_t kExp2Tab32[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

__device__ __forceinline__ float expf_host_libm(float x) {
    if (!(x > -80.0f && x < 80.0f)) return expf(x);  // NaN, overflow/underflow range: never reached by -2*MI
    const double inv_ln2_n = 0x1.71547652b82fep+0 * 32.0;
    const double shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = inv_ln2_n * double(x);
    double kd = z + shift;
    const uint64_t ki = uint64_t(__double_as_longlong(kd));
    kd -= shift;
    const double r = z - kd;
    const uint64_t t = kExp2Tab32[ki & 31u] + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    z = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = z * r2 + y;
    y = y * s;
    return float(y);
}

__device__ __forceinline__ float mi_to_cc(float mi) {  // CorrelationCalculator.cpp:1071-1073,1130-1132
    return sqrtf(1.0f - expf_host_libm(-2.0f * mi));
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
                int t = int(double(r01) * double(nb));
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

template <int N, bool EXACT, int MIN_WAVES>
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
    {
        // all loads first, branch free: a slot past cs loads at an out-of-range offset (0, no memory request) and its
        // code is forced to the pad code below
        float y[N];
#pragma unroll
        for (int e = 0; e < N; e++)
            y[e] = load_member_nt(members[is_member(e) ? e : cs - 1], bytes,
                                  is_member(e) ? byte_offset : kOutOfRangeOffset);
#pragma unroll
        for (int e = 0; e < N; e++) {
            const bool member = is_member(e);
            is_nan |= member && (y[e] != y[e]);
            const float q01 = (y[e] - min_q) / range_q;  // CorrelationCalculator.cpp:1061-1062
            const int b0 = prep[e];                      // pads: kInvalidBin (binned_prep_kernel)
            const bool valid = member && (q01 == q01) && b0 != kInvalidBin;
            int b1 = int(double(q01) * nbd);
            b1 = b1 < 0 ? 0 : (b1 > nb - 1 ? nb - 1 : b1);
            a[e] = valid ? (uint32_t(b1) << 8) | uint32_t(b0) : kPadCode;
            total += valid ? 1 : 0;
        }
    }
    const bool slow = (total != cs) || !ref_all_valid;
    const bool any_slow = __any(slow);
    if (any_slow) {
#pragma unroll
        for (int e = 0; e < N; e++)
            if (is_member(e)) codes[e * 64 + lane] = uint16_t(a[e] & 0xFFFFu);  // pad -> 0xFFFF
    }

    SortNet32<N>::sort(a);
    double mi_y = -sx, joint = 0.0;
    uint32_t cell_len = 0, col_len = 0;
#pragma unroll
    for (int p = 0; p < N; p++) {
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
    if (to_cc) res = mi_to_cc(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) out[v] = res;
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
            const int t = int(double(r01) * double(nb));
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
    int b1 = int(double(q01) * nbd);
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
        if (v < num_voxels) out[v] = res;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kraskov
// ---------------------------------------------------------------------------------------------------------
// prep (fp64 view): [0, cs) px_e = double(ref_e) + noise_ref_e (MutualInformation.cpp:417-420), member order;
//                   [cs, 2cs) the same values sorted ascending (the reference sorts them for its 1-D range counts,
//                   MutualInformation.cpp:187; voxel independent, so sorted once per evaluation)
__global__ __launch_bounds__(256) void kraskov_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                           int cs, const double* __restrict__ noise_ref,
                                                           double* __restrict__ prep) {
    extern __shared__ double px[];  // cs doubles
    for (int e = threadIdx.x; e < cs; e += blockDim.x) px[e] = double(load_ref(src, members, e)) + noise_ref[e];
    __syncthreads();
    for (int e = threadIdx.x; e < cs; e += blockDim.x) {
        const double v = px[e];
        int rank = 0;
        for (int j = 0; j < cs; j++) rank += (px[j] < v || (px[j] == v && j < e)) ? 1 : 0;
        prep[e] = v;
        prep[cs + rank] = v;
    }
}

constexpr double kCountSlack = 1e-15;  // default_epsilon<double>::value, MutualInformation.cpp:163

// fp64 min / max / Chebyshev distance as single instructions.  fmin()/fmax() make hipcc canonicalise each operand first
// (an extra v_max_f64 x, x); the operands here are never signalling NaNs and the k-select only needs "one of the two".
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double max_f64(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double chebyshev_f64(double dx, double dy) {  // max(|dx|, |dy|)
    double r;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(dx), "v"(dy));
    return r;
}

// #{ t : tab[t] < v } for an ascending table of n entries; top = largest power of two <= n.  Branch-free binary search;
// the table is shared by the wave (LDS), the probe index is per lane.
__device__ __forceinline__ int count_less(const double* tab, int n, int top, double v) {
    int pos = 0;
    for (int step = top; step >= 1; step >>= 1) {
        const int idx = pos + step;
        const int probe = idx <= n ? idx : n;  // keep the read in range; the result is discarded when idx > n
        pos = (idx <= n && tab[probe - 1] < v) ? idx : pos;
    }
    return pos;
}

// K > 0: the K = k nearest OTHER points are kept in registers (sorted insertion: min/max only).  K == 0: any k,
// selection by repeated minimum passes.  TI points are processed concurrently per lane.
template <int K, int TI>
__global__ __launch_bounds__(64) void mi_kraskov_kernel(const float* const* __restrict__ members,
                                                        const double* __restrict__ prep_px,
                                                        const double* __restrict__ table_psi,
                                                        const double* __restrict__ noise_query,
                                                        float* __restrict__ out, size_t num_voxels, int cs, int k,
                                                        int estimator, int to_cc) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_px = reinterpret_cast<double*>(smem);  // member order
    double* s_spx = s_px + cs;                       // ascending
    double* s_nq = s_spx + cs;
    double* s_psi = s_nq + cs;                                                  // cs + 1 entries: psi(0..cs)
    float* s_y = reinterpret_cast<float*>(s_psi + (cs + 1) + ((cs + 1) & 1));  // keep 16-byte alignment
    const int lane = threadIdx.x;
    for (int i = lane; i < cs; i += 64) {
        s_px[i] = prep_px[i];
        s_spx[i] = prep_px[cs + i];
        s_nq[i] = noise_query[i];
    }
    for (int i = lane; i <= cs; i += 64) s_psi[i] = table_psi[i];
    const size_t v = size_t(blockIdx.x) * 64 + lane;
    const uint32_t byte_offset = uint32_t(v) * 4u, bytes = uint32_t(num_voxels) * 4u;  // lanes past the end read 0
    bool is_nan = false;
#pragma unroll 8
    for (int e = 0; e < cs; e++) {
        const float y = load_member_nt(members[e], bytes, byte_offset);
        is_nan |= (y != y);
        s_y[e * 64 + lane] = y;
    }
    __syncthreads();

    const int kk = k < cs - 1 ? k : cs - 1;  // neighbours besides the point itself (a kd-tree returns at most cs points)
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    double sum_x = 0.0, sum_y = 0.0;

#pragma unroll 1
    for (int i0 = 0; i0 < cs; i0 += TI) {
        double pxi[TI], pyi[TI], rx[TI], ry[TI];
#pragma unroll
        for (int t = 0; t < TI; t++) {
            const int ii = (i0 + t < cs) ? i0 + t : cs - 1;
            pxi[t] = s_px[ii];
            pyi[t] = double(s_y[ii * 64 + lane]) + s_nq[ii];
        }
        const int i1 = (i0 + TI < cs) ? i0 + TI : cs;
        // ---- pass A: Chebyshev distance to the k-th neighbour (the (k+1)-th smallest distance including the zero
        //      distance to the point itself, MutualInformation.cpp:430-434)
        double dk[TI];
        if constexpr (K > 0) {
            double best[TI][K];
#pragma unroll
            for (int t = 0; t < TI; t++)
#pragma unroll
                for (int q = 0; q < K; q++) best[t][q] = inf;
            auto visit = [&](int j, bool may_be_self) {
                const double pxj = s_px[j];
                const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    double d = chebyshev_f64(pxi[t] - pxj, pyi[t] - pyj);
                    if (may_be_self) d = (j == i0 + t) ? inf : d;
#pragma unroll
                    for (int q = 0; q < K; q++) {
                        const double lo = min_f64(best[t][q], d);
                        if (q + 1 < K) d = max_f64(best[t][q], d);
                        best[t][q] = lo;
                    }
                }
            };
#pragma unroll 2
            for (int j = 0; j < i0; j++) visit(j, false);
#pragma unroll 1
            for (int j = i0; j < i1; j++) visit(j, true);
#pragma unroll 2
            for (int j = i1; j < cs; j++) visit(j, false);
#pragma unroll
            for (int t = 0; t < TI; t++) dk[t] = best[t][K - 1];
        } else {
#pragma unroll
            for (int t = 0; t < TI; t++) {
                double cur = -1.0, m = 0.0;
                int cnt = 0;
#pragma unroll 1
                for (int pass = 0; pass < kk; pass++) {
                    m = inf;
                    int c = 0;
#pragma unroll 2
                    for (int j = 0; j < cs; j++) {
                        const double pxj = s_px[j];
                        const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
                        const double d = fmax(fabs(pxi[t] - pxj), fabs(pyi[t] - pyj));
                        if (d > cur && j != i0 + t) {
                            c = (d < m) ? 1 : (d == m ? c + 1 : c);
                            m = fmin(m, d);
                        }
                    }
                    cnt += c;
                    if (cnt >= kk) break;
                    cur = m;
                }
                dk[t] = m;
            }
        }
        // ---- search radii
        if (estimator == 1) {
#pragma unroll
            for (int t = 0; t < TI; t++) rx[t] = ry[t] = dk[t] - kCountSlack;  // includeCenter, :196-197
        } else {
            // KSG-2: extents of the k+1 nearest points (incl. itself) per dimension (:485-499), then +slack (:198-199)
            double ex[TI], ey[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) ex[t] = ey[t] = 0.0;
#pragma unroll 2
            for (int j = 0; j < cs; j++) {
                const double pxj = s_px[j];
                const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    const double ax = fabs(pxi[t] - pxj), ay = fabs(pyi[t] - pyj);
                    const bool in = fmax(ax, ay) <= dk[t];
                    ex[t] = in ? fmax(ex[t], ax) : ex[t];
                    ey[t] = in ? fmax(ey[t], ay) : ey[t];
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                rx[t] = ex[t] + kCountSlack;
                ry[t] = ey[t] + kCountSlack;
            }
        }
        // ---- pass C: marginal counts  #{ j : c - r <= v_j < c + r }  (:201-233).  x: two binary searches in the sorted
        //      reference coordinates (like the reference); y: compares against every point.
        double loy[TI], hiy[TI];
        int cx[TI], cy[TI];
#pragma unroll
        for (int t = 0; t < TI; t++) {
            const double lox = pxi[t] - rx[t], hix = pxi[t] + rx[t];
            cx[t] = count_less(s_spx, cs, top, hix) - count_less(s_spx, cs, top, lox);
            loy[t] = pyi[t] - ry[t];
            hiy[t] = pyi[t] + ry[t];
            cy[t] = 0;
        }
#pragma unroll 2
        for (int j = 0; j < cs; j++) {
            const double pyj = double(s_y[j * 64 + lane]) + s_nq[j];
#pragma unroll
            for (int t = 0; t < TI; t++) cy[t] += (pyj >= loy[t] && pyj < hiy[t]) ? 1 : 0;
        }
#pragma unroll
        for (int t = 0; t < TI; t++) {
            if (i0 + t < cs) {
                int nx = cx[t] > 1 ? cx[t] : 1;
                int ny = cy[t] > 1 ? cy[t] : 1;
                if (estimator != 1) {
                    nx -= 1;  // psi(n - 1), psi(0) = NaN (pole)
                    ny -= 1;
                }
                sum_x += factor * s_psi[nx];
                sum_y += factor * s_psi[ny];
            }
        }
    }
    double c = s_psi[k <= cs ? k : cs];
    if (estimator != 1) c -= 1.0 / double(k);
    const double d = s_psi[cs];
    const double mi = -sum_x - sum_y + c + d;
    float res = float(mi);
    res = (res < 0.0f) ? 0.0f : res;  // std::max(float(mi), 0.0f), :443
    if (to_cc) res = mi_to_cc(res);
    if (is_nan) res = __uint_as_float(0x7FC00000u);
    if (v < num_voxels) out[v] = res;
}

// ---------------------------------------------------------------------------------------------------------------
// Kraskov for any member count and any k <= 128: no per-voxel tile at all.  A lane keeps, for TI points at a time, the
// K >= k smallest distances in registers (sorted insertion, 2K min/max per candidate) while it sweeps the members
// straight from the member volumes (coalesced 256 B per wave and member, as a tile read would be); the marginal counts
// take a second sweep.  One sweep per TI points instead of the k sweeps per point of the repeated-minimum selection:
// at 1000 members and k = 30 that is ~60x fewer distance evaluations.  LDS holds the voxel-independent tables only.
// ---------------------------------------------------------------------------------------------------------------
// Blocks of 4 waves share one copy of the tables in LDS (24 KB at 1000 members: with one wave per block LDS would cap
// the occupancy at 1.5 waves per SIMD).
// SYM (symmetric field mode): the X side is voxel dependent too -- x_e = members_x[e][v] + noise_ref[e] (prep_px then
// points at the noise_ref table), X counts by comparison like the Y counts instead of the binary search.
template <int K, int TI, bool SYM = false>
__global__ __launch_bounds__(256) void kraskov_direct_kernel(const float* const* __restrict__ members,
                                                            const float* const* __restrict__ members_x,
                                                            const double* __restrict__ prep_px,
                                                            const double* __restrict__ table_psi,
                                                            const double* __restrict__ noise_query,
                                                            float* __restrict__ out, size_t num_voxels, int cs, int k,
                                                            int estimator, int to_cc) {
    extern __shared__ __align__(16) unsigned char smem[];
    double* s_px = reinterpret_cast<double*>(smem);  // member order
    double* s_spx = s_px + cs;                       // ascending
    double* s_nq = s_spx + cs;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < cs; i += 256) {
        s_px[i] = prep_px[i];                         // SYM: noise_ref[i]
        s_spx[i] = SYM ? 0.0 : prep_px[cs + i];
        s_nq[i] = noise_query[i];
    }
    __syncthreads();
    constexpr int JB = 16;  // member values fetched per batch of the sweeps
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const int kk = k < cs - 1 ? k : cs - 1;
    int top = 1;
    while (top * 2 <= cs) top *= 2;
    const double factor = 1.0 / double(cs);
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    const size_t tiles = (num_voxels + 63) / 64;
#pragma unroll 1
    for (size_t tile = size_t(blockIdx.x) * 4 + (threadIdx.x >> 6); tile < tiles; tile += size_t(gridDim.x) * 4) {
        const size_t v = tile * 64 + lane;
        const uint32_t off = v < num_voxels ? uint32_t(v) * 4u : kOutOfRangeOffset;
        bool is_nan = false;
        double sum_x = 0.0, sum_y = 0.0;
#pragma unroll 1
        for (int i0 = 0; i0 < cs; i0 += TI) {
            double pxi[TI], pyi[TI], dk[TI], rx[TI], ry[TI];
            double best[TI][K];
#pragma unroll
            for (int t = 0; t < TI; t++) {
                const int ii = (i0 + t < cs) ? i0 + t : cs - 1;
                const float y = load_member_cached(members[ii], bytes, off);
                is_nan |= y != y;
                if constexpr (SYM) {
                    const float x = load_member_cached(members_x[ii], bytes, off);
                    is_nan |= x != x;
                    pxi[t] = double(x) + s_px[ii];
                } else {
                    pxi[t] = s_px[ii];
                }
                pyi[t] = double(y) + s_nq[ii];
#pragma unroll
                for (int q = 0; q < K; q++) best[t][q] = inf;
            }
            // ---- sweep A: the K smallest Chebyshev distances to OTHER points (MutualInformation.cpp:430-434)
            //      (the member values come in batches of JB loads issued back to back: one memory latency per batch)
#pragma unroll 1
            for (int j0 = 0; j0 < cs; j0 += JB) {
                float yb[JB], xb[SYM ? JB : 1];
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    yb[u] = load_member_cached(members[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                    if constexpr (SYM) xb[u] = load_member_cached(members_x[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                }
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    const int j = j0 + u;
                    const int jc = j < cs ? j : cs - 1;
                    const double pxj = SYM ? double(xb[SYM ? u : 0]) + s_px[jc] : s_px[jc];
                    const double pyj = double(yb[u]) + s_nq[jc];
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        double d = chebyshev_f64(pxi[t] - pxj, pyi[t] - pyj);
                        d = (j == i0 + t || j >= cs) ? inf : d;
#pragma unroll
                        for (int q = 0; q < K; q++) {
                            const double lo = min_f64(best[t][q], d);
                            if (q + 1 < K) d = max_f64(best[t][q], d);
                            best[t][q] = lo;
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                double sel = best[t][0];
#pragma unroll
                for (int q = 1; q < K; q++) sel = (q == kk - 1) ? best[t][q] : sel;
                dk[t] = sel;
            }
            if (estimator == 1) {
#pragma unroll
                for (int t = 0; t < TI; t++) rx[t] = ry[t] = dk[t] - kCountSlack;  // includeCenter, :196-197
            } else {
                double ex[TI], ey[TI];
#pragma unroll
                for (int t = 0; t < TI; t++) ex[t] = ey[t] = 0.0;
#pragma unroll 2
                for (int j = 0; j < cs; j++) {
                    const double pxj = SYM ? double(load_member_cached(members_x[j], bytes, off)) + s_px[j] : s_px[j];
                    const double pyj = double(load_member_cached(members[j], bytes, off)) + s_nq[j];
#pragma unroll
                    for (int t = 0; t < TI; t++) {
                        const double ax = fabs(pxi[t] - pxj), ay = fabs(pyi[t] - pyj);
                        const bool in = fmax(ax, ay) <= dk[t];
                        ex[t] = in ? fmax(ex[t], ax) : ex[t];
                        ey[t] = in ? fmax(ey[t], ay) : ey[t];
                    }
                }
#pragma unroll
                for (int t = 0; t < TI; t++) {
                    rx[t] = ex[t] + kCountSlack;
                    ry[t] = ey[t] + kCountSlack;
                }
            }
            // ---- sweep C: marginal counts (:201-233)
            double loy[TI], hiy[TI], lox[TI], hix[TI];
            int cx[TI], cy[TI];
#pragma unroll
            for (int t = 0; t < TI; t++) {
                lox[t] = pxi[t] - rx[t];
                hix[t] = pxi[t] + rx[t];
                cx[t] = SYM ? 0 : count_less(s_spx, cs, top, hix[t]) - count_less(s_spx, cs, top, lox[t]);
                loy[t] = pyi[t] - ry[t];
                hiy[t] = pyi[t] + ry[t];
                cy[t] = 0;
            }
#pragma unroll 1
            for (int j0 = 0; j0 < cs; j0 += JB) {
                float yb[JB], xb[SYM ? JB : 1];
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    yb[u] = load_member_cached(members[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                    if constexpr (SYM) xb[u] = load_member_cached(members_x[j0 + u < cs ? j0 + u : cs - 1], bytes, off);
                }
#pragma unroll
                for (int u = 0; u < JB; u++) {
                    const int jc = j0 + u < cs ? j0 + u : cs - 1;
                    const double pyj = double(yb[u]) + s_nq[jc];
#pragma unroll
                    for (int t = 0; t < TI; t++) cy[t] += (j0 + u < cs && pyj >= loy[t] && pyj < hiy[t]) ? 1 : 0;
                    if constexpr (SYM) {
                        const double pxj = double(xb[u]) + s_px[jc];
#pragma unroll
                        for (int t = 0; t < TI; t++) cx[t] += (j0 + u < cs && pxj >= lox[t] && pxj < hix[t]) ? 1 : 0;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < TI; t++) {
                if (i0 + t < cs) {
                    int nx = cx[t] > 1 ? cx[t] : 1;
                    int ny = cy[t] > 1 ? cy[t] : 1;
                    if (estimator != 1) {
                        nx -= 1;  // psi(n - 1), psi(0) = NaN (pole)
                        ny -= 1;
                    }
                    sum_x += factor * table_psi[nx];
                    sum_y += factor * table_psi[ny];
                }
            }
        }
        double c = table_psi[k <= cs ? k : cs];
        if (estimator != 1) c -= 1.0 / double(k);
        const double mi = -sum_x - sum_y + c + table_psi[cs];
        float res = float(mi);
        res = (res < 0.0f) ? 0.0f : res;  // std::max(float(mi), 0.0f), :443
        if (to_cc) res = mi_to_cc(res);
        if (is_nan) res = __uint_as_float(0x7FC00000u);
        if (v < num_voxels) out[v] = res;
    }
}

namespace {

template <int N, int MIN_WAVES>
void launch_binned_n(const float* const* d_members, const int* prep, const double* tableT, float* d_out,
                     size_t num_voxels, int cs, const BinnedArgs& a, hipStream_t s) {
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    if (cs == N)
        hipLaunchKernelGGL((mi_binned_kernel<N, true, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, prep,
                           tableT, d_out, num_voxels, cs, a.num_bins, a.min_query, a.max_query, int(a.to_cc));
    else
        hipLaunchKernelGGL((mi_binned_kernel<N, false, MIN_WAVES>), dim3(blocks), dim3(64), 0, s, d_members, prep,
                           tableT, d_out, num_voxels, cs, a.num_bins, a.min_query, a.max_query, int(a.to_cc));
}


// waves/SIMD the binned kernel is compiled for; CRF_BINNED_WAVES overrides for tuning (tools/tune_pearson.py)
int env_binned_waves(int fallback) {
    const char* v = getenv("CRF_BINNED_WAVES");
    return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

void launch_binned_prep(const RefSource& ref, const float* const* d_members, int cs, int n_pad, const BinnedArgs& a,
                        const double* tableT, int* d_prep, hipStream_t s) {
    hipLaunchKernelGGL(binned_prep_kernel, dim3(1), dim3(64), size_t(cs) * sizeof(int), s, ref, d_members, cs, n_pad,
                       a.num_bins, a.min_ref, a.max_ref, tableT, d_prep);
}

void launch_kraskov_prep(const RefSource& ref, const float* const* d_members, int cs, const double* noise_ref,
                         double* d_prep, hipStream_t s) {
    hipLaunchKernelGGL(kraskov_prep_kernel, dim3(1), dim3(256), size_t(cs) * sizeof(double), s, ref, d_members, cs,
                       noise_ref, d_prep);
}

// O(cs) histogram kernel for any member count; hipErrorNotSupported when num_bins is too large for its LDS rows
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
    // waves/SIMD per size from measurements at 256^3 (profiles/tuning_r01.md): 64 members 1.61 ms at 2 waves (2.0 ms at
    // 4, with scratch), 128 members 5.0 ms at 1 wave (5.7 ms at 2)
    const int waves = env_binned_waves(0);
    switch (n_pad) {
        case 16: launch_binned_n<16, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 32: launch_binned_n<32, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 48: launch_binned_n<48, 3>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 64:
            switch (waves) {
                case 3: launch_binned_n<64, 3>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
                case 4: launch_binned_n<64, 4>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
                default: launch_binned_n<64, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
            }
            break;
        case 80: launch_binned_n<80, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 96: launch_binned_n<96, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
        case 112:  // 2 waves with ~330 B of scratch beat 1 wave with AGPRs: 100 members 5.56 -> 4.44 ms, 112: 5.46 -> 4.72 ms
            if (waves == 1)
                launch_binned_n<112, 1>(d_members, prep, tableT, d_out, num_voxels, cs, a, s);
            else
                launch_binned_n<112, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s);
            break;
        default:
            switch (waves) {
                case 2: launch_binned_n<128, 2>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
                default: launch_binned_n<128, 1>(d_members, prep, tableT, d_out, num_voxels, cs, a, s); break;
            }
            break;
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "mi_binned_kernel";
    return hipGetLastError();
}

// any cs (tables must fit LDS: cs <= 2048), k <= 128; hipErrorNotSupported otherwise
hipError_t launch_mi_kraskov_direct(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                                    const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out,
                                    hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    const int kk = a.k < cs - 1 ? a.k : cs - 1;
    const size_t lds = size_t(3 * cs) * sizeof(double);
    if (kk > 128 || lds > 60 * 1024) return hipErrorNotSupported;
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    double* prep = reinterpret_cast<double*>(d_prep);
    if (ref.prepare()) launch_kraskov_prep(ref, d_members, cs, noise_ref, prep, s);
    if (!ref.run()) return hipGetLastError();
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t groups = (tiles + 3) / 4;
    const unsigned blocks = unsigned(groups < 4096 ? groups : 4096);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
#define CRF_LAUNCH_DIRECT(K, TI)                                                                                    \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, TI>), dim3(blocks), dim3(256), lds, s, d_members, nullptr, prep, psi, \
                       noise_query, d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc))
    if (kk <= 4) {
        CRF_LAUNCH_DIRECT(4, 8);
    } else if (kk <= 8) {
        CRF_LAUNCH_DIRECT(8, 4);
    } else if (kk <= 16) {
        CRF_LAUNCH_DIRECT(16, 2);
    } else if (kk <= 32) {
        CRF_LAUNCH_DIRECT(32, 1);
    } else if (kk <= 64) {
        CRF_LAUNCH_DIRECT(64, 1);
    } else {
        CRF_LAUNCH_DIRECT(128, 1);
    }
#undef CRF_LAUNCH_DIRECT
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "kraskov_direct_kernel";
    return hipGetLastError();
}

// symmetric field mode: X = d_members_x (reference field), Y = d_members_y (query field); KSG-1
hipError_t launch_mi_kraskov_symmetric(const float* const* d_members_x, const float* const* d_members_y, int cs,
                                       size_t num_voxels, int k, bool to_cc, const double* d_tables, float* d_out,
                                       hipStream_t s) {
    if (cs == 1) return launch_fill(d_out, num_voxels, 1.0f, s);
    const int kk = k < cs - 1 ? k : cs - 1;
    const size_t lds = size_t(3 * cs) * sizeof(double);
    if (kk > 64 || lds > 60 * 1024) return hipErrorNotSupported;
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    const size_t tiles = (num_voxels + 63) / 64;
    const size_t groups = (tiles + 3) / 4;
    const unsigned blocks = unsigned(groups < 4096 ? groups : 4096);
#define CRF_LAUNCH_SYM(K, TI)                                                                                        \
    hipLaunchKernelGGL((kraskov_direct_kernel<K, TI, true>), dim3(blocks), dim3(256), lds, s, d_members_y, d_members_x, \
                       noise_ref, psi, noise_query, d_out, num_voxels, cs, k, 1, int(to_cc))
    if (kk <= 4) {
        CRF_LAUNCH_SYM(4, 8);
    } else if (kk <= 8) {
        CRF_LAUNCH_SYM(8, 4);
    } else if (kk <= 16) {
        CRF_LAUNCH_SYM(16, 2);
    } else if (kk <= 32) {
        CRF_LAUNCH_SYM(32, 1);
    } else {
        CRF_LAUNCH_SYM(64, 1);
    }
#undef CRF_LAUNCH_SYM
    return hipGetLastError();
}

hipError_t launch_mi_kraskov(const float* const* d_members, int cs, size_t num_voxels, const RefSource& ref,
                             const KraskovArgs& a, const double* d_tables, float* d_prep, float* d_out, hipStream_t s,
                             hipEvent_t ev_begin, hipEvent_t ev_end, LaunchInfo* info) {
    if (cs == 1) {
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    const int kk = a.k < cs - 1 ? a.k : cs - 1;
    const char* force_direct = getenv("CRF_KRASKOV_DIRECT");  // tuning: the tile-free kernel for every k
    // The LDS-tile kernels below are instantiated for k <= 4 and hold a 256*cs-byte column per wave, which caps the
    // occupancy beyond ~80 members (measured at 256^3, k = 3: 80 members 71 vs 72 ms, 96: 112 vs 100 ms, 128: 226 vs
    // 171 ms, tile vs tile-free): the tile-free kernel takes over there and for every larger k.
    if (kk > 4 || cs > 80 || (force_direct && *force_direct == '1')) {
        hipError_t e = launch_mi_kraskov_direct(d_members, cs, num_voxels, ref, a, d_tables, d_prep, d_out, s, ev_begin,
                                                ev_end, info);
        if (e != hipErrorNotSupported) return e;
    }
    const double* psi = d_tables;
    const double* noise_ref = d_tables + 2 * (cs + 1);
    const double* noise_query = noise_ref + cs;
    double* prep = reinterpret_cast<double*>(d_prep);
    if (ref.prepare()) launch_kraskov_prep(ref, d_members, cs, noise_ref, prep, s);
    if (!ref.run()) return hipGetLastError();
    const unsigned blocks = unsigned((num_voxels + 63) / 64);
    const size_t lds = size_t(4 * cs + 1 + ((cs + 1) & 1)) * sizeof(double) + size_t(cs) * 64 * sizeof(float);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    const bool wide = cs % 16 == 0 || cs % 16 > 8;
#define CRF_LAUNCH_KRASKOV(K, TI)                                                                                     \
    hipLaunchKernelGGL((mi_kraskov_kernel<K, TI>), dim3(blocks), dim3(64), lds, s, d_members, prep, psi, noise_query, \
                       d_out, num_voxels, cs, a.k, a.estimator, int(a.to_cc))
    switch (kk) {
        case 1: CRF_LAUNCH_KRASKOV(1, 8); break;
        // 16 points per sweep where the member count fills the last tile well: 253 VGPRs still give the 2 waves per SIMD
        // that the LDS column allows anyway (256^3 x 64, k = 3: 37.4 ms vs 39.7 ms at 8 points, 45.3 ms at 4)
        case 2:
            if (wide) {
                CRF_LAUNCH_KRASKOV(2, 16);
            } else {
                CRF_LAUNCH_KRASKOV(2, 8);
            }
            break;
        case 3:
            if (wide) {
                CRF_LAUNCH_KRASKOV(3, 16);
            } else {
                CRF_LAUNCH_KRASKOV(3, 8);
            }
            break;
        case 4: CRF_LAUNCH_KRASKOV(4, 4); break;
        default: CRF_LAUNCH_KRASKOV(0, 1); break;
    }
#undef CRF_LAUNCH_KRASKOV
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "mi_kraskov_kernel";
    return hipGetLastError();
}

}  // namespace crf
