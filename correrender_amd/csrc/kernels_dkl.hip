// kernels_dkl.hip -- DKLCalculator on gfx950: the Kullback-Leibler divergence between the
// normalised distribution of a voxel's cs member values and N(0,1), two estimators
//   binned        computeDKLBinned<double>       (src/Calculators/DKL.cpp:38-84)
//   entropy k-NN  computeDKLKNNEstimate<double>  (src/Calculators/DKL.cpp:98-165; Kozachenko-Leonenko entropy)
// driven like DKLCalculator::calculateCpu (src/Calculators/DKLCalculator.cpp:134-262): cs == 1 -> 1, NaN member -> NaN.
//
// Same mapping as the generic kernels (kernels_generic.hip): one lane = one voxel, the voxel's values in a per-lane
// column [member][lane] of a tile in LDS (or, when it does not fit, in a global workspace slice owned by the persistent
// block).  fp64 moments in member order exactly like the reference; the k-NN estimator sorts by counting (O(cs^2)
// compares) and then plays the reference's step-halving window descent (KnnWindow below) in fp32.  4*cs + 4 algorithmic bytes
// per voxel; VALU/LDS bound.  fp64 log/exp come from the device math library (<= 1 ulp from the host's): results agree
// with the CPU restatement to ~1e-15 before the cast to float (tolerance 1e-5 relative, tests/test_gpu_dkl.py).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

namespace {
constexpr int kDklBlocks = 1024;
constexpr size_t kDklLdsLimit = 60 * 1024;

// ln(h) for the h = 0..cs samples a bin can hold (binned estimator), rounded up to 16 bytes; 0 for the k-NN estimator
__host__ __device__ inline size_t dkl_log_table_bytes(int cs, int estimator) {
    return estimator == 0 ? ((size_t(cs) + 1) * sizeof(double) + 15) & ~size_t(15) : 0;
}

// in_place: the k-NN estimator's network sort holds the whole column in registers and writes the sorted values back
// over it -- one column instead of two (twice the resident blocks per CU for this latency-bound kernel)
__host__ __device__ inline size_t dkl_tile_bytes(int cs, int estimator, int num_bins, bool in_place = false) {
    const size_t column = size_t(cs) * 64 * sizeof(float);
    return estimator == 0 ? column + size_t(num_bins) * 64 * sizeof(uint16_t) : (in_place ? column : 2 * column);
}

// sgl's Math.hpp declares PI and TWO_PI as float (see oracle/corr_oracle.cpp for the pinning note)
constexpr float kSglPi = 3.1415926535897932f;
constexpr float kSglTwoPi = kSglPi * 2.0f;

// Sorts the lane's column (stride 64 floats, cs <= N values) into `sorted` with the N-input min/max network: keys that
// order like the floats (crf_device.h: orderable_key), pads = 0xFFFFFFFF behind every real value, keys mapped back.
template <int N>
__device__ __forceinline__ void sort_column(const float* vals, float* sorted, int cs) {
    uint32_t a[N];
#pragma unroll
    for (int e = 0; e < N; e++) a[e] = e < cs ? orderable_key(vals[(e < cs ? e : 0) * 64]) : 0xFFFFFFFFu;
    __builtin_amdgcn_sched_barrier(0);  // the network is straight-line code: left to the scheduler it is interleaved
    SortNet32<N>::sort(a);              // with the loads and stores around it and runs out of registers
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < N; p++) {
        const uint32_t key = a[p];
        const uint32_t bits = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
        if (p < cs) sorted[p * 64] = __uint_as_float(bits);
    }
}

// Distance from sorted element i to its k-th nearest neighbour as the reference's heuristic finds it
// (findKNearestNeighbors<float>, DKL.cpp:98-130): NOT an exact k-NN query but a descent over windows of k + 1 consecutive
// sorted elements that hold element i, whose outcome -- including where it stalls -- is part of the estimator's
// definition.  Restated here as a descent on a window COST: the window [lo, lo + k] costs the larger of the two gaps
// from element i to its ends, an end that leaves the column or passes element i costs FLT_MAX.  Each round compares
// the windows `step` to the left and to the right, moves towards the cheaper one unless the current window is
// strictly cheaper than both, then halves the step (rounding up) until a round with step 1 has been played.  `col` is
// the lane's ascending column (stride 64 floats).
struct KnnWindow {
    const float* col;
    int n, k, i;
    float centre;
    __device__ float left_gap(int lo) const { return (lo >= 0 && lo <= i) ? centre - col[lo * 64] : 3.402823466e+38f; }
    __device__ float right_gap(int hi) const { return (hi >= i && hi < n) ? col[hi * 64] - centre : 3.402823466e+38f; }
    __device__ float cost(int lo) const { return fmaxf(left_gap(lo), right_gap(lo + k)); }
};

// The same descent for E sorted elements at once (e0 .. e0 + E - 1, clamped to the column): the step sequence depends
// on k only, so the rounds are the outer loop and the E x 6 column reads of a round are independent -- one element after
// the other, every round was a dependent LDS round trip (r02: the kernel is bound by exactly that latency).
template <int E>
__device__ __forceinline__ void dkl_window_distances(const float* data, int N, int k, int e0, float (&dist)[E]) {
    KnnWindow w[E];
    int lo[E];
    const int first_step = (k + 1) / 2;
#pragma unroll
    for (int u = 0; u < E; u++) {
        const int i = e0 + u < N ? e0 + u : N - 1;
        w[u] = KnnWindow{data, N, k, i, data[i * 64]};
        lo[u] = i - first_step > 0 ? i - first_step : 0;
        lo[u] = lo[u] + k >= N ? N - k - 1 : lo[u];
    }
    int step = first_step;
#pragma unroll 1
    for (bool last = false; !last; step = (step + 1) / 2) {
        last = step == 1;
        float here[E], left[E], right[E];
#pragma unroll
        for (int u = 0; u < E; u++) {
            here[u] = w[u].cost(lo[u]);
            left[u] = w[u].cost(lo[u] - step);
            right[u] = w[u].cost(lo[u] + step);
        }
#pragma unroll
        for (int u = 0; u < E; u++) {
            const int towards = (left[u] > right[u]) - (left[u] < right[u]);
            lo[u] += (here[u] < left[u] && here[u] < right[u]) ? 0 : towards * step;
        }
    }
#pragma unroll
    for (int u = 0; u < E; u++) dist[u] = w[u].cost(lo[u]);
}

__device__ float dkl_window_distance(const float* data, int N, int k, int i) {
    const KnnWindow w{data, N, k, i, data[i * 64]};
    int step = (k + 1) / 2;
    // start: element i sits `step` from the left end, clamped into the column
    int lo = i - step > 0 ? i - step : 0;
    lo = lo + k >= N ? N - k - 1 : lo;
#pragma unroll 1
    for (bool last = false; !last; step = (step + 1) / 2) {
        last = step == 1;
        const float here = w.cost(lo), left = w.cost(lo - step), right = w.cost(lo + step);
        const int towards = (left > right) - (left < right);  // +1: the right window is cheaper; 0: a draw
        lo += (here < left && here < right) ? 0 : towards * step;
    }
    return w.cost(lo);
}
}  // namespace

size_t dkl_workspace_bytes(int cs, int estimator, int num_bins, size_t num_voxels) {
    const size_t tile = dkl_tile_bytes(cs, estimator, num_bins);
    if (tile <= kDklLdsLimit) return 0;
    const size_t tiles = (num_voxels + 63) / 64;
    return tile * (tiles < size_t(kDklBlocks) ? tiles : size_t(kDklBlocks));
}

// knn_const = psi(cs) - psi(k) + ln 2, half_log_two_pi = 0.5 * double(std::log(float TWO_PI)): host libm, fp64
// NS: size of the sorting network of the k-NN estimator's column sort (a multiple of 16 >= cs, at most 128), or 0: sort by
// counting (any member count; also the instantiation the binned estimator runs).  One kernel per network size -- all
// eight networks inside one kernel cost 512 registers and scratch.
template <int NS>
__global__ __launch_bounds__(64, (NS > 0 && NS <= 64) ? 2 : 1) void dkl_kernel(const float* const* __restrict__ members, float* __restrict__ out,
                                                 size_t num_voxels, int cs, int estimator, int num_bins, int k,
                                                 double knn_const, double half_log_two_pi,
                                                 unsigned char* __restrict__ workspace) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS: [ln(h), h = 0..cs, of the binned estimator (dkl_log_table_bytes)] [the tile, unless it lives in the workspace]
    double* s_logn = reinterpret_cast<double*>(smem);
    unsigned char* lds_tile = smem + dkl_log_table_bytes(cs, estimator);
    // the network instantiations are only launched with the tile in LDS: with a plain LDS pointer the column accesses
    // become ds_read / ds_write with immediate offsets; through the generic pointer of the workspace fallback the
    // compiler materialises (and, the tile loop being persistent, hoists) one 64-bit address per column slot
    unsigned char* tile = (NS > 0 || !workspace) ? lds_tile : workspace + size_t(blockIdx.x) * dkl_tile_bytes(cs, estimator, num_bins);
    constexpr bool kInPlace = NS > 0;  // network sort: `sorted` is the value column itself
    const int lane = threadIdx.x;
    if (estimator == 0) {
        for (int h = lane; h <= cs; h += 64) s_logn[h] = h > 0 ? log(double(h)) : 0.0;
        __syncthreads();
    }
    float* vals = reinterpret_cast<float*>(tile) + lane;
    float* sorted = kInPlace ? vals : vals + size_t(cs) * 64;                                  // k-NN
    uint16_t* hist = reinterpret_cast<uint16_t*>(tile + size_t(cs) * 64 * sizeof(float)) + lane;  // binned
    const uint32_t bytes = uint32_t(num_voxels) * 4u;
    const size_t tiles = (num_voxels + 63) / 64;
    const float qnan = __uint_as_float(0x7FC00000u);
    const double factor = 1.0 / double(cs);
#pragma unroll 1
    for (size_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t v = t * 64 + lane;
        const bool active = v < num_voxels;
        const uint32_t byte_offset = active ? uint32_t(v) * 4u : kOutOfRangeOffset;
        bool is_nan = false;
        double mean = 0.0;
        double variance = 0.0;
        // network instantiations (k-NN estimator, cs <= NS <= 128): the voxel's values stay in registers from the load
        // to the sorted column -- the passes over the LDS column were one dependent LDS round trip per 2-4 members each
        float x[NS > 0 ? NS : 1];
        if constexpr (NS > 0) {
            int cs_p = cs;  // (laundered per phase: keeps the compiler from carrying NS uniform member-count tests as
            asm volatile("" : "+s"(cs_p));  // SGPR pairs across the persistent tile loop)
#pragma unroll
            for (int e = 0; e < NS; e++)
                x[e] = load_member_nt(members[e < cs_p ? e : cs_p - 1], bytes, e < cs_p ? byte_offset : kOutOfRangeOffset);
            asm volatile("" : "+s"(cs_p));
#pragma unroll
            for (int e = 0; e < NS; e++) {
                if (e < cs_p) {
                    is_nan |= x[e] != x[e];
                    mean += factor * double(x[e]);
                }
            }
            asm volatile("" : "+s"(cs_p));
#pragma unroll
            for (int e = 0; e < NS; e++) {
                if (e < cs_p) {
                    const double diff = mean - double(x[e]);
                    variance += factor * diff * diff;
                }
            }
        } else {
#pragma unroll 4
            for (int e = 0; e < cs; e++) {
                const float xe = load_member_nt(members[e], bytes, byte_offset);
                is_nan |= xe != xe;
                vals[e * 64] = xe;
                mean += factor * double(xe);
            }
#pragma unroll 2
            for (int e = 0; e < cs; e++) {
                const double diff = mean - double(vals[e * 64]);
                variance += factor * diff * diff;
            }
        }
        const double stdev = sqrt(variance);
        // (v - mean) / stdev for the cs members of the voxel: ONE division (the reciprocal) per voxel, the quotients
        // from q0 = RN(a * rcp), q = fma(fma(-q0, stdev, a), rcp, q0) -- the correctly rounded a / stdev (Markstein)
        // whenever the remainder is exact, which a standard deviation in [2^-400, 2^400] guarantees for every
        // deviation that is 0 or not below 2^-500; everything else (incl. stdev = 0: a constant voxel) divides.
        const bool by_reciprocal = __all(stdev >= 0x1p-400 && stdev <= 0x1p400);
        const double rcp_sd = 1.0 / stdev;
        auto normalised = [&](double a) {
            if (by_reciprocal) {
                const double q0 = a * rcp_sd;
                const double q = fma(fma(-q0, stdev, a), rcp_sd, q0);
                return (fabs(a) >= 0x1p-500 || a == 0.0) ? q : a / stdev;
            }
            return a / stdev;
        };
        float res;
        if (NS > 0 && NS <= 96 && estimator == 0) {
            // Binned estimator, up to 96 members, all in registers: the samples' bin indices are sorted with the
            // min/max network and the histogram is read off as run lengths (the form of mi_binned_kernel) -- no LDS
            // column, no element-by-element read-modify-write of an LDS histogram.  Runs come out in ascending bin
            // order, the order of the reference's loop over the bins, so the sum is the same sum.
            if constexpr (NS > 0 && NS <= 96) {
                int cs_p = cs;
                asm volatile("" : "+s"(cs_p));
                double min_val = 1.7976931348623157e308, max_val = -1.7976931348623157e308;
#pragma unroll
                for (int e = 0; e < NS; e++) {
                    if (e < cs_p) {
                        const double val = normalised(double(x[e]) - mean);
                        x[e] = float(val);
                        min_val = (val < min_val) ? val : min_val;  // a NaN never replaces the extremum
                        max_val = (max_val < val) ? val : max_val;
                    }
                }
                min_val -= 0.01;
                max_val += 0.01;
                const double bin_factor = double(num_bins) / (max_val - min_val);
                const double bin_factor_inv = (max_val - min_val) / double(num_bins);
                uint32_t a[NS];
                asm volatile("" : "+s"(cs_p));
#pragma unroll
                for (int e = 0; e < NS; e++) {
                    const double tt = (double(x[e]) - min_val) * bin_factor;
                    int b = (tt > -2147483649.0 && tt < 2147483648.0) ? int(tt) : 0;  // x86 int(): see below
                    b = b < 0 ? 0 : (b > num_bins - 1 ? num_bins - 1 : b);
                    a[e] = e < cs_p ? uint32_t(b) : 0xFFFFFFFFu;
                }
                __builtin_amdgcn_sched_barrier(0);
                SortNet32<NS>::sort(a);
                __builtin_amdgcn_sched_barrier(0);
                const double gauss_norm = sqrt(0.5 / double(kSglPi));
                const double log_scale = log(bin_factor / (double(cs) * gauss_norm));
                double dkl = 0.0;
                bool overflow = false;
                uint32_t run = 0;  // samples of the current bin seen so far
                asm volatile("" : "+s"(cs_p));
#pragma unroll
                for (int p = 0; p < NS; p++) {
                    if (p < cs_p) {
                        run += 1u;
                        const bool last_of_bin = (p + 1 < NS) ? (p + 1 >= cs_p || a[p + 1 < NS ? p + 1 : p] != a[p]) : true;
                        const uint32_t h = last_of_bin ? run : 0u;
                        const double px = double(h) / double(cs);
                        const double center = (double(a[p]) + 0.5) * bin_factor_inv + min_val;
                        const double half_sq = 0.5 * (center * center);
                        overflow |= last_of_bin && half_sq > 745.1332191019411;
                        const double term = (s_logn[h] + log_scale + half_sq) * px;
                        dkl = last_of_bin ? dkl + term : dkl;
                        run = last_of_bin ? 0u : run;
                    }
                }
                res = (overflow || isinf(dkl)) ? qnan : float(dkl);
            }
        } else if (estimator == 0) {
            double min_val = 1.7976931348623157e308, max_val = -1.7976931348623157e308;
#pragma unroll 2
            for (int e = 0; e < cs; e++) {
                const double val = normalised(double(vals[e * 64]) - mean);
                vals[e * 64] = float(val);
                min_val = (val < min_val) ? val : min_val;  // std::min / std::max: a NaN never replaces the extremum
                max_val = (max_val < val) ? val : max_val;
            }
            min_val -= 0.01;
            max_val += 0.01;
            const double bin_factor = double(num_bins) / (max_val - min_val);
            const double bin_factor_inv = (max_val - min_val) / double(num_bins);
#pragma unroll 1
            for (int b = 0; b < num_bins; b++) hist[b * 64] = 0;
#pragma unroll 2
            for (int e = 0; e < cs; e++) {
                const double tt = (double(vals[e * 64]) - min_val) * bin_factor;
                // the reference's int(t) is INT_MIN for NaN / out-of-range on x86-64 and clamps to bin 0
                int b = (tt > -2147483649.0 && tt < 2147483648.0) ? int(tt) : 0;
                b = b < 0 ? 0 : (b > num_bins - 1 ? num_bins - 1 : b);
                hist[b * 64] = uint16_t(hist[b * 64] + 1u);
            }
            // ln(px * binFactor / (gaussNorm * exp(-c^2 / 2))) = ln(h) + ln(binFactor / (cs * gaussNorm)) + c^2 / 2: one
            // logarithm per voxel and a table of ln(h) instead of a logarithm, an exponential and a division per
            // occupied bin (the reference's form, DKL.cpp:70-78; agreement ~1e-15 relative before the cast to float).
            // Its exp underflows to 0 beyond c^2 / 2 = 745.13..., the quotient becomes inf and the result NaN (:80-82):
            // kept as an explicit test.
            const double gauss_norm = sqrt(0.5 / double(kSglPi));
            const double log_scale = log(bin_factor / (double(cs) * gauss_norm));
            double dkl = 0.0;
            bool overflow = false;
#pragma unroll 1
            for (int b = 0; b < num_bins; b++) {
                const uint32_t h = hist[b * 64];
                if (h > 0u) {
                    const double px = double(h) / double(cs);
                    const double center = (double(b) + 0.5) * bin_factor_inv + min_val;
                    const double half_sq = 0.5 * (center * center);
                    overflow |= half_sq > 745.1332191019411;
                    dkl += (s_logn[h] + log_scale + half_sq) * px;
                }
            }
            res = (overflow || isinf(dkl)) ? qnan : float(dkl);
        } else {
            bool any_nan = false;
            if constexpr (NS == 0) {
#pragma unroll 2
                for (int e = 0; e < cs; e++) {
                    const float val = float(normalised(double(vals[e * 64]) - mean));
                    any_nan |= val != val;
                    vals[e * 64] = val;
                }
            }
            // ascending sort of the lane's column: up to 128 members through a register min/max network on
            // order-preserving 32-bit keys (r02; the estimator only reads the sorted VALUES, so the order among equal
            // values is immaterial; with a NaN the result is NaN whatever the column holds), beyond that by counting
            if constexpr (NS > 0) {
                // normalise, map to order-preserving keys (pads behind every real value), sort, park the sorted VALUES
                uint32_t a[NS];
                int cs_p = cs;
                asm volatile("" : "+s"(cs_p));
#pragma unroll
                for (int e = 0; e < NS; e++) {
                    const float val = float(normalised(double(x[e]) - mean));
                    any_nan |= (e < cs_p) && (val != val);
                    a[e] = e < cs_p ? orderable_key(val) : 0xFFFFFFFFu;
                }
                __builtin_amdgcn_sched_barrier(0);
                SortNet32<NS>::sort(a);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+s"(cs_p));
                if (NS <= 96 && k <= 2 && cs >= 4) {  // (NS = 112, 128: scratch)
                    // k = 1, 2: the descent is ONE round with step 1 (see KnnWindow), so the window it ends in is the
                    // start window or a neighbour, all within [i - 3, i + 3] of element i: with the sorted values in
                    // registers and i a compile-time index nothing is read from LDS at all.  c[m] = cost of the window
                    // that starts at i + m; the start window is i - 1 (0 for i = 0; i - 2 for k = 2 at the last
                    // element), chosen by wave-uniform selects.  Same comparisons as dkl_window_distance().
                    float sv[NS];
#pragma unroll
                    for (int p = 0; p < NS; p++) {
                        const uint32_t key = a[p];
                        sv[p] = __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
                    }
                    const float big = 3.402823466e+38f;
                    double second_moment = 0.0, mant_prod = 1.0;
                    int exp_sum = 0;
#pragma unroll
                    for (int i = 0; i < NS; i++) {
                        if (i < cs_p) {
                            const float centre = sv[i];
                            float c[5];  // m = -3 .. +1
#pragma unroll
                            for (int m = -3; m <= 1; m++) {
                                const int lo = i + m;
                                // left_gap(lo): lo in [0, i]; right_gap(lo + k): lo + k in [i, n)
                                const float left = (lo >= 0 && lo <= i) ? centre - sv[lo >= 0 && lo < NS ? lo : 0] : big;
                                const int h1 = lo + 1, h2 = lo + 2;
                                const float r1 = (h1 >= i && h1 < NS && h1 < cs_p) ? sv[h1 >= 0 && h1 < NS ? h1 : 0] - centre : big;
                                const float r2 = (h2 >= i && h2 < NS && h2 < cs_p) ? sv[h2 >= 0 && h2 < NS ? h2 : 0] - centre : big;
                                c[m + 3] = fmaxf(left, k == 2 ? r2 : r1);
                            }
                            int lo0 = i - 1 > 0 ? i - 1 : 0;
                            lo0 = lo0 + k >= cs_p ? cs_p - k - 1 : lo0;
                            const int d0 = lo0 - i;  // -1; 0 for i = 0; -2 for k = 2 at the last element
                            const float here = d0 == -1 ? c[2] : (d0 == 0 ? c[3] : c[1]);
                            const float left = d0 == -1 ? c[1] : (d0 == 0 ? c[2] : c[0]);
                            const float right = d0 == -1 ? c[3] : (d0 == 0 ? c[4] : c[2]);
                            const int towards = (left > right) - (left < right);
                            const bool stay = here < left && here < right;
                            const float dist = stay ? here : (towards > 0 ? right : (towards < 0 ? left : here));
                            int xp;
                            mant_prod *= __builtin_frexp(double(dist), &xp);
                            exp_sum += xp;
                            second_moment += factor * double(centre) * double(centre);
                        }
                    }
                    const double entropy = factor * (log(mant_prod) + 0.693147180559945309417 * double(exp_sum)) + knn_const;
                    const float dkl = float(-entropy + half_log_two_pi + 0.5 * second_moment);
                    res = isinf(dkl) ? qnan : ((dkl < 0.0f) ? 0.0f : dkl);
                    if (any_nan || is_nan) res = qnan;
                    if (active) store_result_nt(out + v, res);
                    continue;
                }
#pragma unroll
                for (int p = 0; p < NS; p++) {
                    const uint32_t key = a[p];
                    const uint32_t bits = (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key;
                    if (p < cs_p) sorted[p * 64] = __uint_as_float(bits);
                }
            } else {
#pragma unroll 1
                for (int e = 0; e < cs; e++) {  // position = #{smaller} + #{equal with a lower index}
                    const float ve = vals[e * 64];
                    int pos = 0;
#pragma unroll 4
                    for (int j = 0; j < cs; j++) {
                        const float vj = vals[j * 64];
                        pos += (vj < ve || (vj == ve && j < e)) ? 1 : 0;
                    }
                    sorted[(any_nan ? e : pos) * 64] = ve;
                }
            }
            // sum_e ln(d_e) as ONE logarithm: d_e = m_e * 2^x_e (frexp), sum = ln(prod m_e) + ln 2 * sum x_e.  The product
            // of up to 256 mantissas in [0.5, 1) stays a normal double and carries ~256 ulp of relative error -- 1e-14
            // against the 1e-5 tolerance of this float-valued estimator; a zero distance gives ln 0 = -inf, an infinite
            // or NaN one propagates, as with one logarithm per element (which was ~100 fp64 instructions each).
            double second_moment = 0.0, log_sum = 0.0, mant_prod = 1.0;
            int exp_sum = 0;
            constexpr int E = 4;  // elements per descent batch
#pragma unroll 1
            for (int e0 = 0; e0 < cs; e0 += E) {
                float dist[E];
                dkl_window_distances<E>(sorted, cs, k, e0, dist);
#pragma unroll
                for (int u = 0; u < E; u++) {  // member order
                    if (e0 + u < cs) {
                        int x;
                        mant_prod *= __builtin_frexp(double(dist[u]), &x);
                        exp_sum += x;
                        const double value = double(sorted[(e0 + u) * 64]);
                        second_moment += factor * value * value;
                    }
                }
                if ((e0 & 255) == 256 - E || e0 + E >= cs) {  // fold the product every 256 elements and at the end
                    log_sum += log(mant_prod) + 0.693147180559945309417 * double(exp_sum);
                    mant_prod = 1.0;
                    exp_sum = 0;
                }
            }
            double entropy = factor * log_sum + knn_const;
            const float dkl = float(-entropy + half_log_two_pi + 0.5 * second_moment);
            res = isinf(dkl) ? qnan : ((dkl < 0.0f) ? 0.0f : dkl);  // std::max(dkl, 0.0f): NaN stays NaN
            if (any_nan) res = qnan;
        }
        if (is_nan) res = qnan;
        if (cs == 1) res = 1.0f;
        if (active) store_result_nt(out + v, res);
    }
}

hipError_t launch_dkl(const float* const* d_members, int cs, size_t num_voxels, int estimator, int num_bins, int k,
                      double knn_const, unsigned char* d_workspace, float* d_out, hipStream_t s, hipEvent_t ev_begin,
                      hipEvent_t ev_end, LaunchInfo* info) {
    const size_t tiles = (num_voxels + 63) / 64;
    unsigned blocks = unsigned(tiles < size_t(kDklBlocks) ? tiles : size_t(kDklBlocks));
    const char* counting = getenv("CRF_DKL_COUNTING_SORT");  // tuning / tests: the O(cs^2) column sort for every cs
    const bool plain = counting && *counting == '1';
    // network instantiations: the k-NN estimator up to 128 members, the binned estimator up to 96 (registers only)
    const bool binned_registers = estimator == 0 && cs <= 96 && !plain;
    const bool network_sort = (estimator == 1 && cs <= 128 && !plain) || binned_registers;
    // k-NN with k <= 2 up to 96 members runs entirely in registers (see the kernel): no LDS column, so the occupancy is
    // the registers' alone
    const bool register_descent = estimator == 1 && network_sort && k <= 2 && cs >= 4 && cs <= 96;
    const size_t tile = (register_descent || binned_registers) ? 0 : dkl_tile_bytes(cs, estimator, num_bins, network_sort);
    const bool use_lds = tile <= kDklLdsLimit;  // (one column of <= 128 members always fits)
    if (!use_lds && !d_workspace) return hipErrorInvalidValue;
    const size_t lds_bytes = dkl_log_table_bytes(cs, estimator) + (use_lds ? tile : 0);
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    // persistent grid = the blocks the chip holds at once, asked from the runtime per instantiation (registers and the
    // LDS tile decide; a grid one block per CU larger than that runs a second round: measured 13.9 -> 18.5 ms)
#define CRF_LAUNCH_DKL(NS)                                                                                              \
    if (use_lds) {                                                                                                      \
        int per_cu = 0, cus = 256;                                                                                      \
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dkl_kernel<NS>, 64, lds_bytes) == hipSuccess &&      \
            per_cu > 0) {                                                                                               \
            /* the runtime's answer was one block high for both estimators (LDS is handed out in 1280-byte granules:  \
               160 KB / 128); a grid one block per CU too large runs a second round: 11.1 -> 18.6 ms (binned), 17.5 -> \
               31.0 ms (k-NN, two columns) */                                                                          \
            const size_t granules = (lds_bytes + 1279) / 1280;                                                          \
            const int by_lds = granules ? int(128 / granules) : per_cu;                                                 \
            per_cu = std::max(1, std::min(per_cu, by_lds));                                                             \
            int dev = 0;                                                                                                \
            (void)hipGetDevice(&dev);                                                                                   \
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);                              \
            if (const char* ov = getenv("CRF_DKL_BLOCKS_PER_CU"); ov && atoi(ov) > 0) per_cu = atoi(ov); /* tuning */    \
            const size_t resident = size_t(cus > 0 ? cus : 256) * size_t(per_cu);                                       \
            blocks = unsigned(tiles < resident ? tiles : resident);                                                     \
        }                                                                                                               \
    }                                                                                                                   \
    hipLaunchKernelGGL(dkl_kernel<NS>, dim3(blocks), dim3(64), lds_bytes, s, d_members, d_out, num_voxels, cs,      \
                       estimator, num_bins, k, knn_const, 0.5 * double(std::log(kSglTwoPi)),                              \
                       use_lds ? nullptr : d_workspace)
    const int network = network_sort ? (cs + 15) / 16 : 0;
    switch (network) {
        case 1: CRF_LAUNCH_DKL(16); break;
        case 2: CRF_LAUNCH_DKL(32); break;
        case 3: CRF_LAUNCH_DKL(48); break;
        case 4: CRF_LAUNCH_DKL(64); break;
        case 5: CRF_LAUNCH_DKL(80); break;
        case 6: CRF_LAUNCH_DKL(96); break;
        case 7: CRF_LAUNCH_DKL(112); break;
        case 8: CRF_LAUNCH_DKL(128); break;
        default: CRF_LAUNCH_DKL(0); break;
    }
#undef CRF_LAUNCH_DKL
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "dkl_kernel";
    return hipGetLastError();
}

}  // namespace crf
