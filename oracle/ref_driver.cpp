/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE.  A thin C-ABI driver around the REFERENCE's own estimator code.
 *
 * It is compiled together with /root/reference/src/Calculators/Correlation.cpp *where that file lies* (never
 * copied) into oracle/_ref/libref_corr.so by oracle/Makefile, and exists only in the build container: the
 * reference tree does not travel to the GPU box, the built .so does.  It is used to (1) validate
 * oracle/corr_oracle.cpp bit-for-bit and (2) produce the golden vectors under tests/golden/
 * (oracle/make_golden.py), and may be used as the "reference"-kind CPU baseline of bench.py.
 *
 * Coverage: Pearson, Spearman, Kendall -- everything Correlation.cpp holds.  MutualInformation.cpp is
 * unbuildable here (needs boost, sgl, glm, which the image lacks) and is NOT part of this library.
 *
 * The voxel loops below restate CorrelationCalculator.cpp:868-1025 (that file itself cannot be compiled: sgl::vk,
 * ImGui, VolumeData) and call the reference functions declared in Correlation.hpp.
 */
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "Correlation.hpp"  // -I/root/reference/src/Calculators

extern "C" {

float ref_pearson2(const float* x, const float* y, int n) { return computePearson2<float>(x, y, n); }

void ref_ranks(const float* v, float* ranks, int n) {
    std::vector<std::pair<float, int>> tmp;
    tmp.reserve(size_t(n));
    computeRanks(v, ranks, tmp, n);
}

float ref_kendall(const float* x, const float* y, int n) {
    std::vector<std::pair<float, float>> joint;
    std::vector<float> ord, yy, sortArray;
    std::vector<std::pair<int, int>> stack;
    return computeKendall<int32_t>(x, y, n, joint, ord, yy, sortArray, stack);
}

float ref_kendall_slow(const float* x, const float* y, int n) { return computeKendallSlow(x, y, n); }

// measure: 0 Pearson, 1 Spearman, 2 Kendall
int ref_correlation_field(
        int measure, const float* const* fieldPtrs, int cs, size_t voxelBegin, size_t voxelEnd,
        const float* referenceValues, float* out) {
    if (measure < 0 || measure > 2) return 1;
    std::vector<const float*> fields(fieldPtrs, fieldPtrs + cs);
    std::vector<float> referenceRanks;
    if (measure == 1) {
        referenceRanks.resize(size_t(cs));
        ref_ranks(referenceValues, referenceRanks.data(), cs);
    }
    const long long n = (long long)(voxelEnd - voxelBegin);
#pragma omp parallel
    {
        std::vector<float> q((size_t)cs), qr((size_t)cs);
        std::vector<std::pair<float, int>> rankTmp;
        std::vector<std::pair<float, float>> joint;
        std::vector<float> ord, yy, sortArray;
        std::vector<std::pair<int, int>> stack;
#pragma omp for
        for (long long i = 0; i < n; i++) {
            const size_t idx = voxelBegin + size_t(i);
            if (cs == 1) {
                out[i] = 1.0f;
                continue;
            }
            if (measure == 0) {
                out[i] = computePearson2<float>(referenceValues, fields, cs, idx);
                continue;
            }
            bool isNan = false;
            for (int c = 0; c < cs; c++) {
                q[size_t(c)] = fields[size_t(c)][idx];
                if (std::isnan(q[size_t(c)])) {
                    isNan = true;
                    break;
                }
            }
            if (isNan) {
                out[i] = std::numeric_limits<float>::quiet_NaN();
                continue;
            }
            if (measure == 1) {
                computeRanks(q.data(), qr.data(), rankTmp, cs);
                out[i] = computePearson2<float>(referenceRanks.data(), qr.data(), cs);
            } else {
                out[i] = computeKendall<int32_t>(referenceValues, q.data(), cs, joint, ord, yy, sortArray, stack);
            }
        }
    }
    return 0;
}

// Pair requests with the reference's own primitives (the per-pair body of HEBChartCorrelation.cpp:493-600 for
// measure 0 Pearson, 1 Spearman, 2 Kendall).
int ref_pair_requests(int measure, const float* const* fieldPtrs, int cs, const size_t* idxI, const size_t* idxJ,
                      size_t numRequests, float* out) {
    if (measure < 0 || measure > 2) return 1;
    std::vector<float> X((size_t)cs), Y((size_t)cs), rx((size_t)cs), ry((size_t)cs);
    std::vector<std::pair<float, int>> rankTmp;
    std::vector<std::pair<float, float>> joint;
    std::vector<float> ord, yy, sortArray;
    std::vector<std::pair<int, int>> stack;
    for (size_t r = 0; r < numRequests; r++) {
        bool isNan = false;
        for (int c = 0; c < cs; c++) {
            X[size_t(c)] = fieldPtrs[c][idxI[r]];
            Y[size_t(c)] = fieldPtrs[c][idxJ[r]];
            isNan = isNan || std::isnan(X[size_t(c)]) || std::isnan(Y[size_t(c)]);
        }
        if (isNan) {
            out[r] = std::numeric_limits<float>::quiet_NaN();
        } else if (cs == 1) {
            out[r] = 1.0f;
        } else if (measure == 0) {
            out[r] = computePearson2<float>(X.data(), Y.data(), cs);
        } else if (measure == 1) {
            computeRanks(X.data(), rx.data(), rankTmp, cs);
            computeRanks(Y.data(), ry.data(), rankTmp, cs);
            out[r] = computePearson2<float>(rx.data(), ry.data(), cs);
        } else {
            out[r] = computeKendall<int32_t>(X.data(), Y.data(), cs, joint, ord, yy, sortArray, stack);
        }
    }
    return 0;
}

}  // extern "C"
