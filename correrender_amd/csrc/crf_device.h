// crf_device.h -- device-side building blocks shared by the gfx950 estimator kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "crf_internal.h"

namespace crf {

// int(t) the way the reference's x86-64 build evaluates a bin index before clamping it to [0, numBins - 1]
// (MutualInformation.cpp:66-67): cvttsd2si yields INT_MIN for NaN and for every t outside the int range -- so a POSITIVE
// overflow (t >= 2^31: caller-supplied extrema far narrower than the data, or +inf data) lands in bin 0, where the
// GPU's saturating conversion would give INT_MAX and bin numBins - 1.  Negative overflow and NaN saturate to values
// that clamp to bin 0 either way, so one compare suffices.
__device__ __forceinline__ int bin_index_x86(double t) { return t < 2147483648.0 ? int(t) : 0; }


// Streaming member loads through buffer descriptors.  Member base pointers come out of a pointer table, so plain
// loads are flat/global loads with a 64-bit VGPR address each (two VGPRs + a 64-bit VALU add per load in flight).  A
// raw buffer load takes the wave-uniform base in a 128-bit SGPR descriptor and ONE shared 32-bit VGPR byte offset, and
// the hardware bounds check (num_records) makes lanes past the end of the volume read 0 instead of faulting, so the
// ragged tail needs no separate kernel.  aux = 2 is the non-temporal (`nt`) policy: every member value is read exactly
// once per evaluation and the ensemble dwarfs the 256 MiB Infinity Cache.  A member volume (or slab) is < 4 GiB
// (checked in crf_set_grid), so the 32-bit offset and num_records suffice.
typedef int32_t __attribute__((ext_vector_type(4))) buffer_rsrc_t;
#ifndef CRF_LOAD_AUX
#define CRF_LOAD_AUX 2
#endif
constexpr int kAuxNonTemporal = CRF_LOAD_AUX;  // buffer-load cache-policy bits: 1 = sc0, 2 = nt, 16 = sc1
__device__ __forceinline__ auto make_member_rsrc(const float* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), /*stride*/ short(0), int(bytes), 0x00020000);
}
template <class RSRC>
__device__ __forceinline__ float buffer_load_f32_nt(RSRC rsrc, uint32_t byte_offset) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, int(byte_offset), 0, kAuxNonTemporal));
}
// default cache policy through the same descriptor path: for values that are read again soon (kernels that re-read
// a voxel's members from L2 / Infinity Cache)
__device__ __forceinline__ float load_member_cached(const float* base, uint32_t bytes, uint32_t byte_offset) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(make_member_rsrc(base, bytes), int(byte_offset), 0, 0));
}
__device__ __forceinline__ float load_member_nt(const float* base, uint32_t bytes, uint32_t byte_offset) {
    return buffer_load_f32_nt(make_member_rsrc(base, bytes), byte_offset);
}
// Result stores of the bandwidth-bound kernels: written once, never re-read by the kernel (non-temporal)
__device__ __forceinline__ void store_result_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }

// Offset that is out of range for every member descriptor (num_records <= 4 * 2^30): a load through it returns 0
// without a memory request.  Padded slots of the guarded kernels select it (wave-uniform condition, one v_cndmask)
// instead of branching around the load or re-reading a valid member.
constexpr uint32_t kOutOfRangeOffset = 0xFFFFFFF0u;

// plain (cacheable) gather of one value: pair requests re-use voxels, so the default cache policy is kept
__device__ __forceinline__ float load_member(const float* base, uint32_t byte_offset) {
    typedef const float __attribute__((address_space(1)))* gptr;
    typedef const char __attribute__((address_space(1)))* gcptr;
    return *(gptr)((gcptr)base + byte_offset);
}

__device__ __forceinline__ float load_ref(const RefSource& r, const float* const* __restrict__ members, int c) {
    if (r.values) return r.values[c];
    return (r.table ? r.table : members)[c][r.voxel];
}

// Compiler ordering fence on VALUES: an empty asm that "modifies" a and b makes everything computed from them
// afterwards wait for both, which pins phase order in the fully unrolled straight-line kernels (the machine
// scheduler otherwise hoists hundreds of independent address/mask computations and runs out of registers;
// __builtin_amdgcn_sched_barrier does not stop IR-level and DAG-level reordering of pure register code).
template <class A, class B>
__device__ __forceinline__ void order_after(A& a, B& b) {
    asm volatile("" : "+v"(a), "+v"(b));
}

// Order-preserving map float -> uint32 (a < b  <=>  key(a) < key(b) for non-NaN a, b; -0.0 must have been
// canonicalised to +0.0 by the caller with `y + 0.0f` so that key equality == float equality).
__device__ __forceinline__ uint32_t orderable_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return b ^ ((b & 0x80000000u) ? 0xFFFFFFFFu : 0x80000000u);
}

// Per-lane sorting networks over 64-bit (key, slot) composites held in registers (static indices only).
//
// A composite is built so that it is a POSITIVE NORMAL fp64 number whose numeric order equals the lexicographic
// (key, slot) order:   bits = 0x4000000000000000 | (uint64(key32) << 29) | low        (low < 2^29; slot in bits 0..7)
// bit 62 set and bit 61 clear keep the exponent field away from 0 (denormal) and 0x7FF (inf/NaN).  A compare-exchange
// is then exactly two VALU instructions, v_min_f64 + v_max_f64 (they return one of their operands bit for bit),
// instead of a 64-bit integer compare plus four conditional moves.  Inline asm because fmin()/fmax() add a
// canonicalising v_max_f64 x, x per operand.
typedef double composite_t;
__device__ __forceinline__ composite_t make_composite(uint32_t key32, uint32_t low) {
    const uint64_t bits = 0x4000000000000000ull | (uint64_t(key32) << 29) | uint64_t(low);
    return __longlong_as_double((long long)bits);
}
__device__ __forceinline__ uint32_t composite_key(composite_t c) {
    return uint32_t(uint64_t(__double_as_longlong(c)) >> 29);
}
__device__ __forceinline__ uint32_t composite_low(composite_t c) {
    return uint32_t(uint64_t(__double_as_longlong(c))) & 0x1FFFFFFFu;
}
__device__ __forceinline__ composite_t composite_or_low(composite_t c, uint32_t bits) {
    return __longlong_as_double(__double_as_longlong(c) | (long long)bits);
}

#define CRF_CE(i, j)                                                               \
    {                                                                              \
        composite_t lo_, hi_;                                                      \
        asm("v_min_f64 %0, %1, %2" : "=v"(lo_) : "v"(a[i]), "v"(a[j]));            \
        asm("v_max_f64 %0, %1, %2" : "=v"(hi_) : "v"(a[i]), "v"(a[j]));            \
        a[i] = lo_;                                                                \
        a[j] = hi_;                                                                \
    }
template <int N>
struct SortNet;
template <>
struct SortNet<8> {
    static __device__ __forceinline__ void sort(composite_t (&a)[8]) {
#define CRF_SORTNET_N 8
#include "sortnet.inc"
    }
};
template <>
struct SortNet<16> {
    static __device__ __forceinline__ void sort(composite_t (&a)[16]) {
#define CRF_SORTNET_N 16
#include "sortnet.inc"
    }
};
template <>
struct SortNet<32> {
    static __device__ __forceinline__ void sort(composite_t (&a)[32]) {
#define CRF_SORTNET_N 32
#include "sortnet.inc"
    }
};
template <>
struct SortNet<24> {
    static __device__ __forceinline__ void sort(composite_t (&a)[24]) {
#define CRF_SORTNET_N 24
#include "sortnet.inc"
    }
};
template <>
struct SortNet<40> {
    static __device__ __forceinline__ void sort(composite_t (&a)[40]) {
#define CRF_SORTNET_N 40
#include "sortnet.inc"
    }
};
template <>
struct SortNet<48> {
    static __device__ __forceinline__ void sort(composite_t (&a)[48]) {
#define CRF_SORTNET_N 48
#include "sortnet.inc"
    }
};
template <>
struct SortNet<56> {
    static __device__ __forceinline__ void sort(composite_t (&a)[56]) {
#define CRF_SORTNET_N 56
#include "sortnet.inc"
    }
};
template <>
struct SortNet<64> {
    static __device__ __forceinline__ void sort(composite_t (&a)[64]) {
#define CRF_SORTNET_N 64
#include "sortnet.inc"
    }
};
template <>
struct SortNet<80> {
    static __device__ __forceinline__ void sort(composite_t (&a)[80]) {
#define CRF_SORTNET_N 80
#include "sortnet.inc"
    }
};
template <>
struct SortNet<96> {
    static __device__ __forceinline__ void sort(composite_t (&a)[96]) {
#define CRF_SORTNET_N 96
#include "sortnet.inc"
    }
};
template <>
struct SortNet<112> {
    static __device__ __forceinline__ void sort(composite_t (&a)[112]) {
#define CRF_SORTNET_N 112
#include "sortnet.inc"
    }
};
template <>
struct SortNet<128> {
    static __device__ __forceinline__ void sort(composite_t (&a)[128]) {
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
struct SortNet32<48> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[48]) {
#define CRF_SORTNET_N 48
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<80> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[80]) {
#define CRF_SORTNET_N 80
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<96> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[96]) {
#define CRF_SORTNET_N 96
#include "sortnet.inc"
    }
};
template <>
struct SortNet32<112> {
    static __device__ __forceinline__ void sort(uint32_t (&a)[112]) {
#define CRF_SORTNET_N 112
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

// Correctly rounded fp32 quotients a/b for MANY numerators and ONE denominator (the voxel's standard deviation) without
// the ~10-instruction IEEE division expansion per element: with rcp = RN(1/b) (one true division),
//     q0 = RN(a * rcp);  rem = fma(-q0, b, a) (exact);  q = fma(rem, rcp, q0)
// is the correctly rounded a/b (Markstein's theorem; checked against `/` on 2.4e9 operand pairs including every
// significand of b) PROVIDED rem is exactly representable, i.e. no underflow: |a| >= 2^-100 or a == 0, b in
// [2^-60, 2^60].  In the Pearson tail a = y_e - mean: a non-zero difference of two floats is at least half an ulp of
// the mean, so |mean| >= 2^-70 guarantees the bound on every a.  exact_div_guard() is that per-voxel test; when it
// fails for any lane of the wave the kernels take the plain-division path.
__device__ __forceinline__ bool exact_div_guard(float mean, float sd) {
    return fabsf(mean) >= 0x1p-70f && sd >= 0x1p-60f && sd <= 0x1p60f;  // false for NaN, 0, inf, tiny
}
__device__ __forceinline__ float exact_div(float a, float b, float rcp) {
    const float q0 = a * rcp;
    const float rem = fmaf(-q0, b, a);
    return fmaf(rem, rcp, q0);
}

// The voxel side of computePearson2<float> (Correlation.cpp:141-174) for one lane: y[0..cs) in registers, a_e =
// invNm1 * ((x_e - meanX) / sdX) prepared once per evaluation.  Sequential fp32, no contraction.
// SURE: slots 0..SURE-1 are members whatever cs is (the caller's contract), so only the slots behind them are guarded.
template <int N, bool EXACT, int SURE = 0>
__device__ __forceinline__ float pearson_tail(float (&y)[N], const float* __restrict__ prep_a, int cs) {
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    // Guarded use (cs < N), branch free: the caller passes y[e] = 0 and prep_a[e] = 0 for e >= cs and the deviation
    // of those slots is forced to 0, so every pass adds +0 for them (see pearson_reg_kernel).
    float meanY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++) meanY += invN * y[e];
    float varY = 0.0f;
#pragma unroll
    for (int e = 0; e < N; e++) {
        const float d = (EXACT || e < SURE || e < cs) ? y[e] - meanY : 0.0f;
        y[e] = d;
        varY += invNm1 * d * d;
    }
    const float sdY = sqrtf(varY);
    float r = 0.0f;
    if (__all(exact_div_guard(meanY, sdY))) {
        const float rcp = 1.0f / sdY;
#pragma unroll
        for (int e = 0; e < N; e++) r += prep_a[e] * exact_div(y[e], sdY, rcp);
    } else {
#pragma unroll
        for (int e = 0; e < N; e++) r += prep_a[e] * ((EXACT || e < SURE || e < cs) ? y[e] / sdY : 0.0f);
    }
    return r;
}

}  // namespace crf
