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
#include <cmath>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

namespace {
constexpr int kDklBlocks = 1024;
constexpr size_t kDklLdsLimit = 60 * 1024;

__host__ __device__ inline size_t dkl_tile_bytes(int cs, int estimator, int num_bins) {
    const size_t column = size_t(cs) * 64 * sizeof(float);
    return estimator == 0 ? column + size_t(num_bins) * 64 * sizeof(uint16_t) : 2 * column;
}

// sgl's Math.hpp declares PI and TWO_PI as float (see oracle/corr_oracle.cpp for the pinning note)
constexpr float kSglPi = 3.1415926535897932f;
constexpr float kSglTwoPi = kSglPi * 2.0f;

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
__global__ __launch_bounds__(64) void dkl_kernel(const float* const* __restrict__ members, float* __restrict__ out,
                                                 size_t num_voxels, int cs, int estimator, int num_bins, int k,
                                                 double knn_const, double half_log_two_pi,
                                                 unsigned char* __restrict__ workspace) {
    extern __shared__ __align__(16) unsigned char smem[];
    unsigned char* tile = workspace ? workspace + size_t(blockIdx.x) * dkl_tile_bytes(cs, estimator, num_bins) : smem;
    const int lane = threadIdx.x;
    float* vals = reinterpret_cast<float*>(tile) + lane;
    float* sorted = vals + size_t(cs) * 64;                                                    // k-NN
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
#pragma unroll 4
        for (int e = 0; e < cs; e++) {
            const float x = load_member_nt(members[e], bytes, byte_offset);
            is_nan |= x != x;
            vals[e * 64] = x;
            mean += factor * double(x);
        }
        double variance = 0.0;
#pragma unroll 2
        for (int e = 0; e < cs; e++) {
            const double diff = mean - double(vals[e * 64]);
            variance += factor * diff * diff;
        }
        const double stdev = sqrt(variance);
        float res;
        if (estimator == 0) {
            double min_val = 1.7976931348623157e308, max_val = -1.7976931348623157e308;
#pragma unroll 2
            for (int e = 0; e < cs; e++) {
                const double val = (double(vals[e * 64]) - mean) / stdev;
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
            const double gauss_norm = sqrt(0.5 / double(kSglPi));
            double dkl = 0.0;
#pragma unroll 1
            for (int b = 0; b < num_bins; b++) {
                const uint32_t h = hist[b * 64];
                if (h > 0u) {
                    const double px = double(h) / double(cs);
                    const double center = (double(b) + 0.5) * bin_factor_inv + min_val;
                    dkl += log(px * bin_factor / (gauss_norm * exp(-0.5 * (center * center)))) * px;
                }
            }
            res = isinf(dkl) ? qnan : float(dkl);
        } else {
            bool any_nan = false;
#pragma unroll 2
            for (int e = 0; e < cs; e++) {
                const float val = float((double(vals[e * 64]) - mean) / stdev);
                any_nan |= val != val;
                vals[e * 64] = val;
            }
            // ascending sort by counting: position = #{smaller} + #{equal with a lower index}
#pragma unroll 1
            for (int e = 0; e < cs; e++) {
                const float ve = vals[e * 64];
                int pos = 0;
#pragma unroll 4
                for (int j = 0; j < cs; j++) {
                    const float vj = vals[j * 64];
                    pos += (vj < ve || (vj == ve && j < e)) ? 1 : 0;
                }
                sorted[(any_nan ? e : pos) * 64] = ve;
            }
            double entropy = 0.0, second_moment = 0.0;
#pragma unroll 1
            for (int e = 0; e < cs; e++) {
                const double nn_dist = double(dkl_window_distance(sorted, cs, k, e));
                entropy += factor * log(nn_dist);
                const double value = double(sorted[e * 64]);
                second_moment += factor * value * value;
            }
            entropy += knn_const;
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
    const unsigned blocks = unsigned(tiles < size_t(kDklBlocks) ? tiles : size_t(kDklBlocks));
    const size_t tile = dkl_tile_bytes(cs, estimator, num_bins);
    const bool use_lds = tile <= kDklLdsLimit;
    if (!use_lds && !d_workspace) return hipErrorInvalidValue;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    hipLaunchKernelGGL(dkl_kernel, dim3(blocks), dim3(64), use_lds ? tile : 0, s, d_members, d_out, num_voxels, cs,
                       estimator, num_bins, k, knn_const, 0.5 * double(std::log(kSglTwoPi)),
                       use_lds ? nullptr : d_workspace);
    if (ev_end) (void)hipEventRecord(ev_end, s);
    if (info) info->kernel_name = "dkl_kernel";
    return hipGetLastError();
}

}  // namespace crf
