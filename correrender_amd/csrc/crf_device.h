// crf_device.h -- device-side building blocks shared by the gfx950 estimator kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "crf_internal.h"

namespace crf {

__device__ __forceinline__ float load_ref(const RefSource& r, const float* const* __restrict__ members, int c) {
    return r.values ? r.values[c] : members[c][r.voxel];
}

// Order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b) for non-NaN a, b; -0.0 must have been
// canonicalised to +0.0 by the caller with `y + 0.0f` so that key equality == float equality).
__device__ __forceinline__ uint32_t orderable_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}

// Per-lane sorting networks over 64-bit composites held in registers (static indices only).
#define CRF_CE(i, j)                       \
    {                                      \
        const uint64_t lo_ = a[i], hi_ = a[j]; \
        const bool sw_ = hi_ < lo_;        \
        a[i] = sw_ ? hi_ : lo_;            \
        a[j] = sw_ ? lo_ : hi_;            \
    }
template <int N>
struct SortNet;
template <>
struct SortNet<16> {
    static __device__ __forceinline__ void sort(uint64_t (&a)[16]) {
#define CRF_SORTNET_N 16
#include "sortnet.inc"
    }
};
template <>
struct SortNet<32> {
    static __device__ __forceinline__ void sort(uint64_t (&a)[32]) {
#define CRF_SORTNET_N 32
#include "sortnet.inc"
    }
};
template <>
struct SortNet<64> {
    static __device__ __forceinline__ void sort(uint64_t (&a)[64]) {
#define CRF_SORTNET_N 64
#include "sortnet.inc"
    }
};
template <>
struct SortNet<128> {
    static __device__ __forceinline__ void sort(uint64_t (&a)[128]) {
#define CRF_SORTNET_N 128
#include "sortnet.inc"
    }
};
#undef CRF_CE

// Same networks over plain 32-bit keys (min/max: 2 VALU per exchange).
#define CRF_CE(i, j)                      \
    {                                     \
        const uint32_t lo_ = a[i], hi_ = a[j]; \
        a[i] = lo_ < hi_ ? lo_ : hi_;     \
        a[j] = lo_ < hi_ ? hi_ : lo_;     \
    }
template <int N>
struct SortNet32;
template <>
struct SortNet32<16> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[16]) {
#define CRF_SORTNET_N 16
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<32> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[32]) {
#define CRF_SORTNET_N 32
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<64> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[64]) {
#define CRF_SORTNET_N 64
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<128> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[128]) {
#define CRF_SORTNET_N 128
#include "sortnet.inc"
    }
};
#undef CRF_CE

// The voxel side of computePearson2<float> (Correlation.cpp:141-174) for one lane: y[0..cs) in registers, a_e =
// invNm1 * ((x_e - meanX) / sdX) prepared once per evaluation.  Sequential fp32, no contraction.
template <int N, bool EXACT>
__device__ __forceinline__ float pearson_tail(float (&y)[N], const float* __restrict__ prep_a, int cs) {
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++)
        if (EXACT || e < cs) meanY += invN * y[e];
    float varY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++) {
        if (EXACT || e < cs) {
            const float d = y[e] - meanY;
            y[e] = d;
            varY += invNm1 * d * d;
        }
    }
    const float sdY = sqrtf(varY);
    float r = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++)
        if (EXACT || e < cs) r += prep_a[e] * (y[e] / sdY);
    return r;
}

}  // namespace crf
