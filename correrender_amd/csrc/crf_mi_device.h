// crf_mi_device.h -- device helpers shared by the two mutual-information translation units (kernels_binned.hip,
// kernels_kraskov.hip): the glibc-compatible expf of the MI -> correlation-coefficient map.
#pragma once
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
static __device__ const uint64_t kExp2Tab32[32] = {
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


}  // namespace crf
