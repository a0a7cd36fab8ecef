// kernels_pearson.hip -- Pearson correlation field on gfx950.
//
// Semantics: computePearson2<float>(referenceValues, fields, cs, voxel) of the reference
// (src/Calculators/Correlation.cpp:100-133, selected by FORMULA_2_FLOAT at CorrelationCalculator.cpp:887-893),
// reproduced operation for operation in fp32 with contraction off (built with -ffp-contract=off) so that results
// are bit-identical to the reference's x86-64 build (no FMA: CMakeLists.txt:34-36 sets no -march):
//     pass 1   meanY += invN * y_e                         (e = 0..cs-1, sequential)
//     pass 2   varY  += (invNm1 * (y_e - meanY)) * (y_e - meanY)
//     pass 3   r     += (invNm1 * ((x_e - meanX) / sdX)) * ((y_e - meanY) / sdY)
// Every reference-only term -- meanX, sdX and a_e = invNm1 * ((x_e - meanX) / sdX) -- is voxel independent and is
// computed once by pearson_prep_kernel with the same sequential fp32 arithmetic; the per-voxel kernel then needs
// only the cs a_e values (scalar loads -> SGPRs).
//
// Data movement (the bound): each voxel reads its cs member values exactly once from HBM (4*cs bytes) and writes
// 4 bytes.  One lane owns VPT consecutive voxels and keeps all cs values in VGPRs across the three passes; a wave
// load instruction covers 64*VPT consecutive floats of ONE member volume (256 B / 512 B / 1 KiB contiguous), all
// cs loads of a wave are issued back to back before the first use, so a wave has cs*256*VPT bytes in flight.
// No LDS, no MFMA: ~4 flop/byte, HBM-read bound.
#include <cstdlib>

#include "crf_device.h"
#include "crf_internal.h"

namespace crf {

constexpr int kPrepZeroFilled = 1280;  // a_e = 0 for cs <= e < this: the padded slots of the register and split kernels
static_assert(kPrepZeroFilled >= kMaxRegisterMembers, "padded slots of the register kernels");

// ---------------------------------------------------------------------------------------------------------
// Reference-side preparation: one wave.  d_prep[e] = a_e for e < cs.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pearson_prep_kernel(RefSource src, const float* const* __restrict__ members,
                                                           int cs, float* __restrict__ prep) {
    extern __shared__ float x[];  // cs reference values
    __shared__ float sh[2];
    for (int e = threadIdx.x; e < cs; e += blockDim.x) x[e] = load_ref(src, members, e);
    __syncthreads();
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    if (threadIdx.x == 0) {
        float meanX = 0.0f;
        for (int e = 0; e < cs; e++) meanX += invN * x[e];
        float varX = 0.0f;
        for (int e = 0; e < cs; e++) {
            const float d = x[e] - meanX;
            varX += invNm1 * d * d;
        }
        sh[0] = meanX;
        sh[1] = sqrtf(varX);
    }
    __syncthreads();
    const float meanX = sh[0], sdX = sh[1];
    for (int e = threadIdx.x; e < cs; e += blockDim.x) prep[e] = invNm1 * ((x[e] - meanX) / sdX);
    // padded slots of the guarded register kernels multiply by a_e = 0 (see pearson_reg_kernel)
    for (int e = cs + threadIdx.x; e < kPrepZeroFilled; e += blockDim.x) prep[e] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------
// Per-voxel kernel, members resident in registers.
//   CS_PAD  compile-time upper bound of cs (loops fully unrolled to it); EXACT: cs == CS_PAD, no guards; otherwise
//           CS_PAD - pad_granule(CS_PAD) < cs < CS_PAD and only the last granule is guarded.
//   VPT     voxels per lane (1, 2 or 4) = width of each global load in dwords.
// ---------------------------------------------------------------------------------------------------------
template <int VPT>
struct VecT;
template <>
struct VecT<1> {
    using type = float;
};
template <>
struct VecT<2> {
    using type = float __attribute__((ext_vector_type(2)));
};
template <>
struct VecT<4> {
    using type = float __attribute__((ext_vector_type(4)));
};

// VPT consecutive voxels of one member for this lane (see crf_device.h: buffer descriptor + shared 32-bit offset).
// NT selects the non-temporal policy: measured 0.753 -> 0.705 ms at 256^3 x 64 on MI355X (profiles/tuning_r01.md).
template <int VPT, bool NT>
__device__ __forceinline__ void load_vec(const float* base, uint32_t bytes, uint32_t byte_offset, float (&dst)[VPT]) {
    const auto rsrc = make_member_rsrc(base, bytes);
    constexpr int aux = NT ? kAuxNonTemporal : 0;
    if constexpr (VPT == 1) {
        dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, int(byte_offset), 0, aux));
    } else if constexpr (VPT == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, int(byte_offset), 0, aux);
        dst[0] = __uint_as_float(v[0]);
        dst[1] = __uint_as_float(v[1]);
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, int(byte_offset), 0, aux);
#pragma unroll
        for (int i = 0; i < 4; i++) dst[i] = __uint_as_float(v[i]);
    }
}
template <int VPT>
__device__ __forceinline__ void store_vec(float* p, const float (&src)[VPT]) {
    using V = typename VecT<VPT>::type;
    V v;
    __builtin_memcpy(&v, src, sizeof(V));
    // non-temporal: the result is written once and not read by this kernel -- keeping it out of the caches' way measured
    // 0.692 -> 0.657 ms at 256^3 x 64 (78.9 -> 83.0 % of the HBM peak, same box, interleaved; profiles/tuning_r01.md)
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
}

// launch_pearson pads cs to the next multiple of this: only the last granule of a guarded instantiation can be padding
constexpr int pad_granule(int cs_pad) {
    return cs_pad <= 16 ? 8 : cs_pad <= 128 ? 16 : cs_pad <= 224 ? 32 : cs_pad <= 240 ? 16 : cs_pad <= 256 ? 8 : 64;
}

template <int CS_PAD, int VPT, bool EXACT, int MIN_WAVES, int BLOCK = 256, bool NT = true>
__global__ __launch_bounds__(BLOCK, MIN_WAVES) void pearson_reg_kernel(const float* const* __restrict__ members,
                                                                       const float* __restrict__ prep,
                                                                       float* __restrict__ out, uint32_t num_voxels,
                                                                       int cs) {
    constexpr int kFirstGuarded = EXACT ? CS_PAD : CS_PAD - pad_granule(CS_PAD);  // slots below are always members
    const auto is_member = [cs](int e) { return e < kFirstGuarded || e < cs; };      // folds in the unrolled loops
    const uint32_t v0 = (blockIdx.x * BLOCK + threadIdx.x) * VPT;
    const uint32_t byte_offset = v0 * 4u;       // one 32-bit offset serves all cs loads of the lane
    const uint32_t bytes = num_voxels * 4u;     // descriptor bound: lanes past the end read 0 and store nothing
    // Guarded instantiation (cs < CS_PAD), branch free: a padded slot loads at an out-of-range offset (the value is 0
    // and no memory request is made: crf_device.h kOutOfRangeOffset), its deviation is forced to 0 in pass 2 and its
    // a_e is 0 (pearson_prep_kernel), so each pass adds +0 for it -- an identity on the running sums, which start at
    // +0 and therefore are never -0.  In pass 3 exact_div(0, sd) = 0 on the fast path; the plain-division path (sd
    // may be 0 there: 0/0) selects 0 for the pads explicitly.
    float y[CS_PAD][VPT];
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
        if (e < kFirstGuarded) {
            load_vec<VPT, NT>(members[e], bytes, byte_offset, y[e]);
        } else if (CS_PAD >= 224) {
            // the widest kernels have no register to spare for per-slot offsets: a uniform branch around each of the
            // few guarded loads instead (they are the last loads issued)
#pragma unroll
            for (int v = 0; v < VPT; v++) y[e][v] = 0.0f;
            if (e < cs) load_vec<VPT, NT>(members[e], bytes, byte_offset, y[e]);
        } else {
            load_vec<VPT, NT>(members[e < cs ? e : cs - 1], bytes, e < cs ? byte_offset : kOutOfRangeOffset, y[e]);
        }
    }
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);

    float meanY[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) meanY[v] = 0.0f;
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
#pragma unroll
        for (int v = 0; v < VPT; v++) meanY[v] += invN * y[e][v];
    }
    float varY[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) varY[v] = 0.0f;
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
#pragma unroll
        for (int v = 0; v < VPT; v++) {
            const float d = is_member(e) ? y[e][v] - meanY[v] : 0.0f;
            y[e][v] = d;  // (y_e - meanY) is needed again, bit-identically, by pass 3
            varY[v] += invNm1 * d * d;
        }
    }
    float r[VPT];
    float sdY[VPT];
#pragma unroll
    for (int v = 0; v < VPT; v++) {
        sdY[v] = sqrtf(varY[v]);
        r[v] = 0.0f;
    }
    bool guard = true;
#pragma unroll
    for (int v = 0; v < VPT; v++) guard = guard && exact_div_guard(meanY[v], sdY[v]);
    if (__all(guard)) {  // exact quotients through one reciprocal per voxel (crf_device.h: exact_div)
        float rcp[VPT];
#pragma unroll
        for (int v = 0; v < VPT; v++) rcp[v] = 1.0f / sdY[v];
#pragma unroll
        for (int e = 0; e < CS_PAD; e++) {
            const float a = prep[e];
#pragma unroll
            for (int v = 0; v < VPT; v++) r[v] += a * exact_div(y[e][v], sdY[v], rcp[v]);
        }
    } else {
#pragma unroll
        for (int e = 0; e < CS_PAD; e++) {
            const float a = prep[e];
#pragma unroll
            for (int v = 0; v < VPT; v++) r[v] += a * (is_member(e) ? y[e][v] / sdY[v] : 0.0f);
        }
    }
    if (v0 + VPT <= num_voxels) store_vec<VPT>(out + v0, r);
}

// ---------------------------------------------------------------------------------------------------------
// R + up to L members: the first R values of a voxel in registers (always members: no guards), the remaining cs - R
// (1..L) in the lane's LDS column.  For member counts just above what fits two waves per SIMD in registers: 256
// register values spill 120 B - 1 KB under the 256-register cap, 240 + 16 in LDS do not (16 x 1 KB per block).
// The tail slots are handled in uniform branches (they are in LDS: no register live ranges to split).
// ---------------------------------------------------------------------------------------------------------
template <int R, int L, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void pearson_reg_lds_kernel(const float* const* __restrict__ members,
                                                                         const float* __restrict__ prep,
                                                                         float* __restrict__ out, uint32_t num_voxels,
                                                                         int cs) {
    extern __shared__ float tail_dyn[];  // L rows of 256 floats (dynamic: 80 rows exceed the 64 KB static limit)
    float(*tail)[256] = reinterpret_cast<float(*)[256]>(tail_dyn);
    const uint32_t v0 = blockIdx.x * 256u + threadIdx.x;
    const uint32_t byte_offset = v0 * 4u, bytes = num_voxels * 4u;
    const int nt = cs - R;  // members in the LDS tail, 1..L
    float y[R];
#pragma unroll
    for (int e = 0; e < R; e++) y[e] = load_member_nt(members[e], bytes, byte_offset);
    // the tail goes straight from memory to LDS (global_load_lds: no registers, issued back to back with the loads above;
    // the hardware writes lane l of a wave to lds_base + 4 l, i.e. the wave's own 64 consecutive floats of the row,
    // which only this wave reads again, so the only synchronisation is its own vmcnt wait).  Lanes past the end of the
    // grid read the last voxel instead (no descriptor bounds on this path); they store nothing.
    const int wave_first = int(threadIdx.x) & ~63;
    const uint32_t v_safe = v0 < num_voxels ? v0 : num_voxels - 1u;
#pragma unroll
    for (int t = 0; t < L; t++) {
        if (t < nt) {
            typedef const float __attribute__((address_space(1)))* gptr_t;
            typedef float __attribute__((address_space(3)))* lptr_t;
            __builtin_amdgcn_global_load_lds((gptr_t)(members[R + t] + v_safe), (lptr_t)&tail[t][wave_first], 4, 0,
                                             kAuxNonTemporal);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanY = 0.0f;
#pragma unroll
    for (int e = 0; e < R; e++) meanY += invN * y[e];
#pragma unroll
    for (int t = 0; t < L; t++)
        if (t < nt) meanY += invN * tail[t][threadIdx.x];
    float varY = 0.0f;
#pragma unroll
    for (int e = 0; e < R; e++) {
        const float d = y[e] - meanY;
        y[e] = d;
        varY += invNm1 * d * d;
    }
#pragma unroll
    for (int t = 0; t < L; t++) {
        if (t < nt) {
            const float d = tail[t][threadIdx.x] - meanY;
            tail[t][threadIdx.x] = d;
            varY += invNm1 * d * d;
        }
    }
    const float sdY = sqrtf(varY);
    float r = 0.0f;
    if (__all(exact_div_guard(meanY, sdY))) {
        const float rcp = 1.0f / sdY;
#pragma unroll
        for (int e = 0; e < R; e++) r += prep[e] * exact_div(y[e], sdY, rcp);
#pragma unroll
        for (int t = 0; t < L; t++)
            if (t < nt) r += prep[R + t] * exact_div(tail[t][threadIdx.x], sdY, rcp);
    } else {
#pragma unroll
        for (int e = 0; e < R; e++) r += prep[e] * (y[e] / sdY);
#pragma unroll
        for (int t = 0; t < L; t++)
            if (t < nt) r += prep[R + t] * (tail[t][threadIdx.x] / sdY);
    }
    if (v0 < num_voxels) store_result_nt(out + v0, r);
}

// ---------------------------------------------------------------------------------------------------------
// 289..1216 members (r03): G = 2 or 4 LANES per voxel.  A wave owns 64 / G voxels; lane group g (lanes g * 64 / G ...)
// holds members [g * S, g * S + S) of them, S = R + L slots per lane: R in registers and L in the lane's LDS column,
// exactly the storage of the 176..320-member kernels above, which run at two waves per SIMD and 70-88 % of the HBM
// peak -- instead of one wave per SIMD with 384 values in VGPRs + AGPRs (37-56 %), an 8-wave relay through LDS
// (385..512 members, 37 %) or three sweeps over the members (beyond 512: 3x the algorithmic bytes).
// The three passes of computePearson2<float> are sequential fp32 sums over the members, so each pass is a relay of
// G stages inside the wave: in stage gg every lane runs the chain over its own S slots, starting from the value that
// group gg - 1 handed over (ds_bpermute, no LDS memory), and only group gg's result is kept; the other groups compute
// throw-away values in that stage (no traps, nothing stored).  Same operations in the same order as the reference,
// every member value fetched once.  What does not lie on a chain -- products, deviations, quotients -- is done once
// and two slots at a time (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: the same IEEE operations per element), which
// pays for the repeated chain stages: 12.9 vector instructions per slot at G = 2 against 12 per member in
// pearson_reg_kernel.  A load instruction is issued per member under the owning group's exec mask (the descriptor is
// wave-uniform), 256 / G bytes each.
//   PAD   cs lies in (G * S - PAD, G * S]: only the last PAD slots of the last group can be padding (they read 0, their
//         deviation is forced to 0 and a_e = 0 for e >= cs: every pass adds +0 for them, as in pearson_reg_kernel).
// ---------------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float from_lane(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}

constexpr int kSplitMaxMembers = 4 * 304;  // 1216

template <int R, int L, int G, int PAD, int MIN_WAVES>
__global__ __launch_bounds__(256, MIN_WAVES) void pearson_split_kernel(const float* const* __restrict__ members,
                                                                       const float* __restrict__ prep,
                                                                       float* __restrict__ out, uint32_t num_voxels,
                                                                       int cs) {
    static_assert(R % 2 == 0 && L % 2 == 0 && PAD % 2 == 0 && 64 % G == 0, "slots are handled in pairs");
    constexpr int S = R + L;        // slots per lane
    constexpr int VW = 64 / G;      // voxels per wave
    constexpr int kSure = S - PAD;  // slots below are members in every lane group
    static_assert(kSure >= 0 && G * S <= kPrepZeroFilled, "a_e is zero-filled up to kPrepZeroFilled");
    extern __shared__ float tail_dyn[];  // L rows of 256 floats: slot R + t of thread x is tail[t][x]
    float(*tail)[256] = reinterpret_cast<float(*)[256]>(tail_dyn);
    const int lane = int(threadIdx.x) & 63;
    const int g = lane / VW;
    const uint32_t v0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * uint32_t(VW) + uint32_t(lane % VW);
    const uint32_t byte_offset = v0 * 4u, bytes = num_voxels * 4u;  // lanes past the end read 0 and store nothing
    // slot i of this lane is a member iff i < mine.  (Laundered before each phase that tests it: left alone the compiler
    // evaluates all PAD tests once and keeps them as SGPR pairs across the kernel, which spill into VGPR lanes.)
    int mine = cs - g * S;
    const auto is_member = [&mine](int i) { return i < kSure || i < mine; };
    f2 y[R / 2];
    // the LDS tail first (global_load_lds: memory -> LDS without registers; lane l of the wave lands at row base + 4 l):
    // loads return in order, so by the time the register loads issued below have been waited for these are done too
    if constexpr (L > 0) {
        const int wave_first = int(threadIdx.x) & ~63;
        const uint32_t v_safe = v0 < num_voxels ? v0 : num_voxels - 1u;  // no descriptor bounds on this path
#pragma unroll
        for (int gg = 0; gg < G; gg++) {
            if (g == gg) {
#pragma unroll
                for (int t = 0; t < L; t++) {
                    const int e = gg * S + R + t;
                    if (gg < G - 1 || R + t < kSure || e < cs) {
                        typedef const float __attribute__((address_space(1)))* gptr_t;
                        typedef float __attribute__((address_space(3)))* lptr_t;
                        __builtin_amdgcn_global_load_lds((gptr_t)(members[e] + v_safe), (lptr_t)&tail[t][wave_first], 4, 0,
                                                         kAuxNonTemporal);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
        if (g == gg) {  // this group's members, under its exec mask
#pragma unroll
            for (int i = 0; i < R; i++) {
                const int e = gg * S + i;
                float v = 0.0f;
                if (gg < G - 1 || i < kSure) {
                    v = load_member_nt(members[e], bytes, byte_offset);
                } else if (e < cs) {  // uniform branch around a load that may be padding
                    v = load_member_nt(members[e], bytes, byte_offset);
                }
                y[i / 2][i % 2] = v;
            }
        }
    }
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    const int last_group_lane = (G - 1) * VW + lane % VW;  // where a pass's result ends up for this lane's voxel
    // ---- pass 1: meanY += invN * y_e
    float m = 0.0f;
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
        if (gg > 0) m = from_lane(m, lane - VW);
        // laundered per stage: left alone the compiler computes the products once and keeps all S of them for the
        // other stages (spills)
        f2 scale = {invN, invN};
        asm volatile("" : "+v"(scale));
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            const f2 t = scale * y[k];
            m += t[0];
            m += t[1];
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L > 0) {
            if (gg == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail has landed (see above)
            asm volatile("" : "+v"(mine));
#pragma unroll
            for (int t = 0; t < L; t += 2) {
                f2 v = {tail[t][threadIdx.x], tail[t + 1][threadIdx.x]};
                if (R + t >= kSure) {  // a padded row was not loaded: whatever the row holds is not a member value
                    if (!is_member(R + t)) v[0] = 0.0f;
                    if (!is_member(R + t + 1)) v[1] = 0.0f;
                }
                const f2 p = scale * v;
                m += p[0];
                m += p[1];
                if ((t & 6) == 6) __builtin_amdgcn_sched_barrier(0);  // (else all L rows are read up front)
            }
        }
    }
    const float meanY = from_lane(m, last_group_lane);
    // ---- deviations in place (needed again, bit-identically, by pass 3)
    {
        const f2 mean2 = {meanY, meanY};
        asm volatile("" : "+v"(mine));
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            f2 d = y[k] - mean2;
            if (2 * k >= kSure) {
                if (!is_member(2 * k)) d[0] = 0.0f;
                if (!is_member(2 * k + 1)) d[1] = 0.0f;
            }
            y[k] = d;
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L > 0) {
            asm volatile("" : "+v"(mine));
#pragma unroll
            for (int t = 0; t < L; t++) {
                float d = tail[t][threadIdx.x] - meanY;
                if (R + t >= kSure && !is_member(R + t)) d = 0.0f;
                tail[t][threadIdx.x] = d;
                if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- pass 2: varY += (invNm1 * d_e) * d_e
    float var = 0.0f;
#pragma unroll
    for (int gg = 0; gg < G; gg++) {
        if (gg > 0) var = from_lane(var, lane - VW);
        f2 scale = {invNm1, invNm1};
        asm volatile("" : "+v"(scale));
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            const f2 t = (scale * y[k]) * y[k];
            var += t[0];
            var += t[1];
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L > 0) {
#pragma unroll
            for (int t = 0; t < L; t += 2) {
                const f2 d = {tail[t][threadIdx.x], tail[t + 1][threadIdx.x]};
                const f2 p = (scale * d) * d;
                var += p[0];
                var += p[1];
                if ((t & 6) == 6) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const float sdY = sqrtf(from_lane(var, last_group_lane));
    // ---- pass 3: r += a_e * (d_e / sdY)
    float r = 0.0f;
    // exact_div needs sd in [2^-60, 2^60] and every non-zero deviation >= 2^-100 in magnitude (crf_device.h).
    // exact_div_guard() infers the latter from |mean| >= 2^-70; a wave in which that fails -- a mean that is exactly 0
    // is enough, and the benchmark's box ensemble has such voxels at 512 members -- looks at its deviations themselves
    // before it gives up the exact path, whose alternative is expensive here (see the else branch).
    bool exact = exact_div_guard(meanY, sdY);
    constexpr bool kSecondLook = R + L < 304;  // (in the 304-slot instantiation it costs 0.3 KB of scratch per lane)
    if (kSecondLook && !__all(exact)) {
        uint32_t smallest = 0xFFFFFFFFu;  // min over the slots of (bits of |d|) - 1: a zero wraps to the maximum
#pragma unroll
        for (int k = 0; k < R / 2; k++) {
            smallest = min(smallest, (__float_as_uint(y[k][0]) & 0x7FFFFFFFu) - 1u);
            smallest = min(smallest, (__float_as_uint(y[k][1]) & 0x7FFFFFFFu) - 1u);
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L > 0) {
#pragma unroll
            for (int t = 0; t < L; t++) {
                smallest = min(smallest, (__float_as_uint(tail[t][threadIdx.x]) & 0x7FFFFFFFu) - 1u);
                if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (a NaN or infinite member makes the mean and therefore sd NaN: the range test fails)
        exact = sdY >= 0x1p-60f && sdY <= 0x1p60f && smallest >= __float_as_uint(0x1p-100f) - 1u;
        // Two kinds of voxels whose result is NaN on either path, so they need not drag their wave onto the slow one --
        // and they come in whole regions in real ensembles (missing values, masks): a NaN mean (some member is NaN:
        // every deviation is NaN), and sd = 0 with every deviation exactly 0 (all members equal, e.g. a zero mask:
        // the exact path computes 0 * (1 / 0) = NaN where the division gives 0 / 0 = NaN).
        exact = exact || meanY != meanY || (sdY == 0.0f && smallest == 0xFFFFFFFFu);
    }
    if (__all(exact)) {  // exact quotients through one reciprocal per voxel
        const float rcp = 1.0f / sdY;
        const f2 rcp2 = {rcp, rcp}, sd2 = {sdY, sdY};
#pragma unroll
        for (int k = 0; k < R / 2; k++) {  // exact_div in place, two slots at a time
            const f2 q0 = y[k] * rcp2;
            const f2 rem = __builtin_elementwise_fma(-q0, sd2, y[k]);
            y[k] = __builtin_elementwise_fma(rem, rcp2, q0);
            if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (L > 0) {
#pragma unroll
            for (int t = 0; t < L; t++) {
                tail[t][threadIdx.x] = exact_div(tail[t][threadIdx.x], sdY, rcp);
                if ((t & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int gg = 0; gg < G; gg++) {
            if (gg > 0) r = from_lane(r, lane - VW);
#pragma unroll
            for (int k = 0; k < R / 2; k++) {
                const f2 a = {prep[gg * S + 2 * k], prep[gg * S + 2 * k + 1]};  // the stage's group decides: wave-uniform
                const f2 t = a * y[k];
                r += t[0];
                r += t[1];
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (L > 0) {
#pragma unroll
                for (int t = 0; t < L; t += 2) {
                    const f2 a = {prep[gg * S + R + t], prep[gg * S + R + t + 1]};
                    const f2 q = {tail[t][threadIdx.x], tail[t + 1][threadIdx.x]};
                    const f2 p = a * q;
                    r += p[0];
                    r += p[1];
                    if ((t & 6) == 6) __builtin_amdgcn_sched_barrier(0);  // (else all L rows are read up front)
                }
            }
        }
    } else if (g == G - 1 && v0 < num_voxels) {
        // A constant voxel (sd = 0) or a tiny mean somewhere in the wave: plain divisions.  The storing lanes redo pass 3
        // from memory (the lines are still in L2), 16 or 8 members per step with their loads in flight together; y_e - meanY is
        // recomputed from the same operands: the same float.  (A second unrolled body over y[] -- quotients in place, or
        // taken on the fly per stage -- costs the whole kernel its register allocation: 0.9-2 KB of scratch per lane;
        // in groups of 8 slots under a uniform branch still 150-320 B.)
        constexpr int kStep = G == 2 ? 16 : 8;  // (16 live values cost the four-lane instantiations 0.1-0.4 KB of scratch)
        int e = 0;
#pragma unroll 1
        for (; e + kStep <= cs; e += kStep) {
            float v[kStep];
#pragma unroll
            for (int i = 0; i < kStep; i++) v[i] = load_member(members[e + i], byte_offset);
#pragma unroll
            for (int i = 0; i < kStep; i++) r += prep[e + i] * ((v[i] - meanY) / sdY);
        }
#pragma unroll 1
        for (; e < cs; e++) r += prep[e] * ((load_member(members[e], byte_offset) - meanY) / sdY);
    }
    if (g == G - 1 && v0 < num_voxels) store_result_nt(out + v0, r);
}

// ---------------------------------------------------------------------------------------------------------
// Symmetric field mode (CorrelationFieldMode::SEPARATE_SYMMETRIC, CorrelationMain.glsl:10-15): voxel v correlates
// the reference field's members at v with the query field's members at v -- computePearson2 on two arrays
// (Correlation.cpp:141-174), nothing to hoist.  2*cs loads per voxel (8*cs + 4 algorithmic bytes), both sides in
// registers.  Same guarded-slot scheme as pearson_reg_kernel.
// ---------------------------------------------------------------------------------------------------------
// REQ: pair-request mode (crf_compute_requests): item r works on request r = {xi, yi, zi, i, xj, yj, zj, j}
// (HEBChart.hpp:166-168), X = the members at voxel i, Y = the members at voxel j; default cache policy (requests
// revisit voxels); `requests`, `xs`, `ys`, `use_abs` are unused in field mode, where num_items = num_voxels.
template <int CS_PAD, bool EXACT, int MIN_WAVES, bool REQ = false>
__global__ __launch_bounds__(256, MIN_WAVES) void pearson_symmetric_kernel(const float* const* __restrict__ members_x,
                                                                           const float* const* __restrict__ members_y,
                                                                           float* __restrict__ out,
                                                                           uint32_t num_voxels, int cs,
                                                                           const uint32_t* __restrict__ requests,
                                                                           uint32_t num_items, int xs, int ys,
                                                                           int use_abs) {
    constexpr int kFirstGuarded = EXACT ? CS_PAD : CS_PAD - 16;
    const auto is_member = [cs](int e) { return e < kFirstGuarded || e < cs; };
    const uint32_t v0 = blockIdx.x * 256 + threadIdx.x;
    uint32_t offset_x = v0 * 4u, offset_y = v0 * 4u;
    if constexpr (REQ) {
        offset_x = offset_y = kOutOfRangeOffset;  // items past the end read 0 and store nothing
        if (v0 < num_items) {
            const uint32_t* q = requests + size_t(v0) * 8;
            offset_x = ((q[2] * uint32_t(ys) + q[1]) * uint32_t(xs) + q[0]) * 4u;  // IDXS
            offset_y = ((q[6] * uint32_t(ys) + q[5]) * uint32_t(xs) + q[4]) * 4u;
        }
    }
    const uint32_t bytes = num_voxels * 4u;
    float x[CS_PAD], y[CS_PAD];
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
        const int slot = (e < kFirstGuarded || e < cs) ? e : cs - 1;
        const uint32_t off_x = is_member(e) ? offset_x : kOutOfRangeOffset;
        const uint32_t off_y = is_member(e) ? offset_y : kOutOfRangeOffset;
        x[e] = REQ ? load_member_cached(members_x[slot], bytes, off_x) : load_member_nt(members_x[slot], bytes, off_x);
        y[e] = REQ ? load_member_cached(members_y[slot], bytes, off_y) : load_member_nt(members_y[slot], bytes, off_y);
    }
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanX = 0.0f, meanY = 0.0f;
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
        meanX += invN * x[e];
        meanY += invN * y[e];
    }
    float varX = 0.0f, varY = 0.0f;
#pragma unroll
    for (int e = 0; e < CS_PAD; e++) {
        const float dx = is_member(e) ? x[e] - meanX : 0.0f;
        const float dy = is_member(e) ? y[e] - meanY : 0.0f;
        x[e] = dx;
        y[e] = dy;
        varX += invNm1 * dx * dx;
        varY += invNm1 * dy * dy;
    }
    const float sdX = sqrtf(varX), sdY = sqrtf(varY);
    float r = 0.0f;
    if (__all(exact_div_guard(meanX, sdX) && exact_div_guard(meanY, sdY))) {
        const float rcpX = 1.0f / sdX, rcpY = 1.0f / sdY;
#pragma unroll
        for (int e = 0; e < CS_PAD; e++)
            r += (invNm1 * exact_div(x[e], sdX, rcpX)) * exact_div(y[e], sdY, rcpY);
    } else {
#pragma unroll
        for (int e = 0; e < CS_PAD; e++)
            r += (invNm1 * (is_member(e) ? x[e] / sdX : 0.0f)) * (is_member(e) ? y[e] / sdY : 0.0f);
    }
    if (REQ && use_abs) r = fabsf(r);
    if (v0 < num_items) store_result_nt(out + v0, r);
}

// ---------------------------------------------------------------------------------------------------------
// Streaming fallback for any cs (three passes re-reading the members; the second and third mostly hit L2/MALL) and
// for the ragged tail of the grid.  One voxel per lane, bounds-checked.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pearson_stream_kernel(const float* const* __restrict__ members,
                                                             const float* __restrict__ prep,
                                                             float* __restrict__ out, size_t voxel_offset,
                                                             size_t voxel_end, int cs) {
    const size_t v0 = voxel_offset + size_t(blockIdx.x) * 256 + threadIdx.x;
    if (v0 >= voxel_end) return;
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    float meanY = 0.0f;
    for (int e = 0; e < cs; e++) meanY += invN * members[e][v0];
    float varY = 0.0f;
    for (int e = 0; e < cs; e++) {
        const float d = members[e][v0] - meanY;
        varY += invNm1 * d * d;
    }
    const float sdY = sqrtf(varY);
    float r = 0.0f;
    for (int e = 0; e < cs; e++) r += prep[e] * ((members[e][v0] - meanY) / sdY);
    store_result_nt(out + v0, r);
}

// ---------------------------------------------------------------------------------------------------------
// Member counts beyond the register kernels (the reference's own synthetic data set has 1000 members): the three passes
// re-read the voxel's values, 3x the algorithmic bytes.  One wave owns 64 voxels at a time and walks the members in
// chunks of 64 with all 64 loads of a chunk in flight; measured at 128x128x64x1000: 2.12 ms = 5.9 TB/s moved (2.0 TB/s
// of algorithmic bytes), HBM-bound, vs 3.42 ms for the plain streaming kernel below.  A deliberately small persistent
// grid (512-1024 waves, so that the lines of pass 1 would still be in the 256 MiB Infinity Cache for passes 2 and 3)
// was measured too and is slower -- a wave moves only ~4 GB/s when it alternates load and compute phases, so the grid
// must fill the chip (3.6 ms at 1024 waves, 2.3 ms at 2048, flat from 4096; profiles/tuning_r01.md).
// Arithmetic: the same sequential fp32 passes as everywhere else.
// ---------------------------------------------------------------------------------------------------------
constexpr int kBigWaves = 8192;  // persistent beyond this many 64-voxel tiles

// One chunk = 64 members: the 64 member pointers are fetched by ONE coalesced vector load (lane l takes member e0 + l)
// and handed to the wave one at a time with v_readlane -- no dependent scalar-load chain in front of the 64 buffer
// loads, which all go out back to back (vmcnt allows 63 in flight).
__device__ __forceinline__ void load_chunk_64(const float* const* __restrict__ members, int e0, int cs, uint32_t bytes,
                                              uint32_t byte_offset, float (&buf)[64]) {
    const int mine = e0 + int(threadIdx.x) < cs ? e0 + int(threadIdx.x) : cs - 1;
    const uint64_t ptr = reinterpret_cast<uint64_t>(members[mine]);
    const uint32_t ptr_lo = uint32_t(ptr), ptr_hi = uint32_t(ptr >> 32);
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const uint64_t base = (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_hi), i))) << 32) |
                              uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_lo), i)));
        buf[i] = load_member_cached(reinterpret_cast<const float*>(base), bytes,
                                    e0 + i < cs ? byte_offset : kOutOfRangeOffset);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 385..512 members, one HBM read (r02): a block of 8 waves holds ONE 64-voxel tile, wave w keeps members
// [64 w, 64 w + 64) of it in registers.  The three passes of computePearson2<float> are sequential fp32 sums over the
// members, so a pass runs as a relay: wave 0 adds its 64 terms, hands the running value to wave 1 through LDS, ... --
// 8 stages per pass, a barrier between stages; what does not lie on the chain (deviations, quotients) is computed by
// all waves at once between the passes.  Same operations in the same order as the reference, every member value
// fetched once (pearson_big_kernel reads the volume three times: 3.0x the algorithmic bytes at the fabric counters).
// Only one wave of a block works during a stage, and a lone wave issues a dependent fp32 chain slowly: the kernel is
// bound by that latency, not by HBM -- 256x256x64: 385 / 512 members 2.72 / 2.91 ms (3.0 TB/s at 512) against 3.37 /
// 4.39 ms for the three-pass kernel.  Two blocks per CU (128-register cap) overlap one block's loads with the other's
// relay: 0.72 vs 1.68 ms at 128x128x64 x 400 with 143 registers and one block per CU.  With 128 members per wave
// (513..1024 members) the relay loses to the three-pass kernel (2.40 vs 2.22 ms at 1000 members): not instantiated.
// ---------------------------------------------------------------------------------------------------------
template <int W, int R>
__global__ __launch_bounds__(64 * W, 4)  // 4 waves per SIMD: 128 registers, two blocks per CU
    void pearson_relay_kernel(const float* const* __restrict__ members,
                                                               const float* __restrict__ prep,
                                                               float* __restrict__ out, uint32_t num_voxels, int cs) {
    // The products of a pass do not depend on the relay stage, so the compiler would hoist all R of them in front of the
    // relay (R more registers; the 128-register cap then spills).  The scale factor is laundered through an empty asm
    // inside the stage, which pins the products there.
    constexpr bool kPinned = true;
    __shared__ float s_relay[64];
    const int lane = int(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
    const int e0 = wave * R;
    const uint32_t bytes = num_voxels * 4u;
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    const uint32_t tiles = (num_voxels + 63u) / 64u;
    // this wave's member pointers and a_e, one per lane and group of 64 members (handed out with v_readlane)
    constexpr int G = R / 64;
    uint32_t ptr_lo[G], ptr_hi[G];
    float a_mine[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int e = e0 + g * 64 + lane;
        const uint64_t ptr = reinterpret_cast<uint64_t>(members[e < cs ? e : cs - 1]);
        ptr_lo[g] = uint32_t(ptr);
        ptr_hi[g] = uint32_t(ptr >> 32);
        a_mine[g] = e < cs ? prep[e] : 0.0f;
    }
#pragma unroll 1
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint32_t v0 = t * 64u + uint32_t(lane);
        const uint32_t byte_offset = v0 * 4u;  // lanes past the end read 0 and store nothing
        // (the member-count tests `e0 + i < cs` are uniform; left alone the compiler evaluates all R of them once, keeps
        // them as SGPR pairs across the tile loop and spills those to VGPR lanes: the count is laundered per phase)
        int cs_p = cs;
        asm volatile("" : "+s"(cs_p));
        float y[R];
#pragma unroll
        for (int i = 0; i < R; i++) {
            const uint64_t base = (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_hi[i / 64]), i % 64))) << 32) |
                                  uint64_t(uint32_t(__builtin_amdgcn_readlane(int(ptr_lo[i / 64]), i % 64)));
            y[i] = load_member_nt(reinterpret_cast<const float*>(base), bytes,
                                  e0 + i < cs_p ? byte_offset : kOutOfRangeOffset);  // a slot past cs reads 0
        }
        // ---- pass 1: meanY += invN * y_e
#pragma unroll 1
        for (int s = 0; s < W; s++) {
            if (wave == s) {
                float m = s == 0 ? 0.0f : s_relay[lane];
                float scale = invN;
                if constexpr (kPinned) asm volatile("" : "+v"(scale));
#pragma unroll
                for (int i = 0; i < R; i++) m += scale * y[i];  // a padded slot adds invN * 0 = +0
                s_relay[lane] = m;
            }
            __syncthreads();
        }
        const float meanY = s_relay[lane];
        __syncthreads();
        asm volatile("" : "+s"(cs_p));
#pragma unroll
        for (int i = 0; i < R; i++) {
            y[i] = e0 + i < cs_p ? y[i] - meanY : 0.0f;
            if (kPinned && (i & 7) == 7) __builtin_amdgcn_sched_barrier(0);  // in place, 8 at a time: register pressure
        }
        // ---- pass 2: varY += invNm1 * d * d
#pragma unroll 1
        for (int s = 0; s < W; s++) {
            if (wave == s) {
                float var = s == 0 ? 0.0f : s_relay[lane];
                float scale = invNm1;
                if constexpr (kPinned) asm volatile("" : "+v"(scale));
#pragma unroll
                for (int i = 0; i < R; i++) var += scale * y[i] * y[i];
                s_relay[lane] = var;
            }
            __syncthreads();
        }
        const float sdY = sqrtf(s_relay[lane]);
        __syncthreads();
        if (__all(exact_div_guard(meanY, sdY))) {  // the same lanes in every wave of the block: one decision
            const float rcp = 1.0f / sdY;
#pragma unroll
            for (int i = 0; i < R; i++) {
                y[i] = exact_div(y[i], sdY, rcp);
                if (kPinned && (i & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            asm volatile("" : "+s"(cs_p));
#pragma unroll
            for (int i = 0; i < R; i++) {
                y[i] = e0 + i < cs_p ? y[i] / sdY : 0.0f;
                if (kPinned && (i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- pass 3: r += a_e * ((y_e - meanY) / sdY)
#pragma unroll 1
        for (int s = 0; s < W; s++) {
            if (wave == s) {
                float r = s == 0 ? 0.0f : s_relay[lane];
                asm volatile("" : "+s"(cs_p));
                float a_lanes[G];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    a_lanes[g] = a_mine[g];
                    if constexpr (kPinned) asm volatile("" : "+v"(a_lanes[g]));
                }
#pragma unroll
                for (int i = 0; i < R; i++) {
                    const float a =
                        __uint_as_float(uint32_t(__builtin_amdgcn_readlane(int(__float_as_uint(a_lanes[i / 64])), i % 64)));
                    if (e0 + i < cs_p) r += a * y[i];
                }
                s_relay[lane] = r;
            }
            __syncthreads();
        }
        if (wave == 0 && v0 < num_voxels) store_result_nt(out + v0, s_relay[lane]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void pearson_big_kernel(const float* const* __restrict__ members,
                                                         const float* __restrict__ prep, float* __restrict__ out,
                                                         uint32_t num_voxels, int cs) {
    const uint32_t bytes = num_voxels * 4u;
    const float n = float(cs);
    const float invN = 1.0f / n;
    const float invNm1 = 1.0f / (n - 1.0f);
    const uint32_t tiles = (num_voxels + 63u) / 64u;
#pragma unroll 1
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint32_t v0 = t * 64u + threadIdx.x;
        const uint32_t byte_offset = v0 * 4u;  // lanes past the end read 0 and store nothing
        float buf[64];
        float meanY = 0.0f;
#pragma unroll 1
        for (int e0 = 0; e0 < cs; e0 += 64) {
            load_chunk_64(members, e0, cs, bytes, byte_offset, buf);
#pragma unroll
            for (int i = 0; i < 64; i++) meanY += invN * buf[i];  // a padded slot adds invN * 0 = +0
        }
        float varY = 0.0f;
#pragma unroll 1
        for (int e0 = 0; e0 < cs; e0 += 64) {
            load_chunk_64(members, e0, cs, bytes, byte_offset, buf);
#pragma unroll
            for (int i = 0; i < 64; i++) {
                const float d = e0 + i < cs ? buf[i] - meanY : 0.0f;
                varY += invNm1 * d * d;
            }
        }
        const float sdY = sqrtf(varY);
        float r = 0.0f;
#pragma unroll 1
        for (int e0 = 0; e0 < cs; e0 += 64) {
            load_chunk_64(members, e0, cs, bytes, byte_offset, buf);
            const float a_mine = e0 + int(threadIdx.x) < cs ? prep[e0 + threadIdx.x] : 0.0f;  // a_e, one per lane
#pragma unroll
            for (int i = 0; i < 64; i++) {
                const float a = __uint_as_float(uint32_t(__builtin_amdgcn_readlane(int(__float_as_uint(a_mine)), i)));
                if (e0 + i < cs) r += a * ((buf[i] - meanY) / sdY);
            }
        }
        if (v0 < num_voxels) store_result_nt(out + v0, r);
    }
}

__global__ void fill_kernel(float* __restrict__ out, size_t n, float value) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = value;
}

namespace {

int env_int(const char* name, int fallback);

template <int CS_PAD, int VPT>
void launch_reg(const float* const* d_members, const float* d_prep, float* d_out, size_t blocks, size_t num_voxels,
                int cs, hipStream_t s) {
    // occupancy request: data registers are CS_PAD*VPT per lane; ask for the waves/SIMD that budget allows.
    constexpr int kData = CS_PAD * VPT;
    constexpr int kMinWaves = kData <= 64 ? 4 : (kData <= 128 ? 2 : 1);
    // 161..256 values per lane: capped at 256 registers (two waves per SIMD; 12-120 B of scratch) the kernel is 1.2x
    // faster than with 256 VGPRs + AGPRs at one wave per SIMD, which cannot overlap its load and compute phases --
    // measured at 512x512x128: 256 members 8.15 -> 6.72 ms (64 % of the HBM peak), 224 members 5.60 -> 4.63 ms (81 %),
    // 200 (guarded 224) 5.61 -> 4.53 ms.  The guarded 248 / 256 instantiations spill ~1 KB under the cap (28 ms): they stay
    // at one wave.  CRF_PEARSON_WAVES=1 restores the one-wave kernels (tuning).
    constexpr bool kTwoWavesExact = kData > 160 && kData <= 256;
    constexpr bool kTwoWavesGuarded = kData > 160 && kData <= 240;
    const bool two = env_int("CRF_PEARSON_WAVES", 2) == 2;
    if (cs == CS_PAD) {
        if constexpr (kTwoWavesExact) {
            if (two) {
                hipLaunchKernelGGL((pearson_reg_kernel<CS_PAD, VPT, true, 2>), dim3(unsigned(blocks)), dim3(256), 0, s,
                                   d_members, d_prep, d_out, uint32_t(num_voxels), cs);
                return;
            }
        }
        hipLaunchKernelGGL((pearson_reg_kernel<CS_PAD, VPT, true, kMinWaves>), dim3(unsigned(blocks)), dim3(256), 0, s,
                           d_members, d_prep, d_out, uint32_t(num_voxels), cs);
    } else {
        if constexpr (kTwoWavesGuarded) {
            if (two) {
                hipLaunchKernelGGL((pearson_reg_kernel<CS_PAD, VPT, false, 2>), dim3(unsigned(blocks)), dim3(256), 0, s,
                                   d_members, d_prep, d_out, uint32_t(num_voxels), cs);
                return;
            }
        }
        hipLaunchKernelGGL((pearson_reg_kernel<CS_PAD, VPT, false, kMinWaves>), dim3(unsigned(blocks)), dim3(256), 0,
                           s, d_members, d_prep, d_out, uint32_t(num_voxels), cs);
    }
}

template <int CS_PAD>
void launch_reg_vpt(int vpt, const float* const* d_members, const float* d_prep, float* d_out, size_t blocks,
                    size_t num_voxels, int cs, hipStream_t s) {
    if constexpr (CS_PAD <= 64) {
        if (vpt == 4) return launch_reg<CS_PAD, 4>(d_members, d_prep, d_out, blocks, num_voxels, cs, s);
    }
    if constexpr (CS_PAD <= 64 || CS_PAD == 128) {
        if (vpt >= 2) return launch_reg<CS_PAD, 2>(d_members, d_prep, d_out, blocks, num_voxels, cs, s);
    }
    return launch_reg<CS_PAD, 1>(d_members, d_prep, d_out, blocks, num_voxels, cs, s);
}

int env_int(const char* name, int fallback) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

}  // namespace

__global__ void abs_kernel(float* __restrict__ out, size_t n) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fabsf(out[i]);  // NaN stays NaN
}

hipError_t launch_abs(float* d_out, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(abs_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, d_out, n);
    return hipGetLastError();
}

hipError_t launch_fill(float* d_out, size_t n, float value, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, d_out, n, value);
    return hipGetLastError();
}

namespace {
template <int CS_PAD>
void launch_symmetric(const float* const* mx, const float* const* my, int cs, size_t num_voxels, float* d_out,
                      hipStream_t s) {
    constexpr int kMinWaves = CS_PAD <= 32 ? 4 : (CS_PAD <= 64 ? 2 : 1);
    const unsigned blocks = unsigned((num_voxels + 255) / 256);
    // 2 * CS_PAD values per lane.  Two waves under a 256-register cap pay off at 96 (90 members: 2.55 -> 2.04 ms) but
    // not with the 72-670 B of scratch of the 112 / 128 instantiations (128 members: 3.92 -> 5.09 ms)
    if constexpr (CS_PAD == 96) {
        if (env_int("CRF_PEARSON_WAVES", 2) == 2) {
            if (cs == CS_PAD)
                hipLaunchKernelGGL((pearson_symmetric_kernel<CS_PAD, true, 2>), dim3(blocks), dim3(256), 0, s, mx, my,
                                   d_out, uint32_t(num_voxels), cs, nullptr, uint32_t(num_voxels), 0, 0, 0);
            else
                hipLaunchKernelGGL((pearson_symmetric_kernel<CS_PAD, false, 2>), dim3(blocks), dim3(256), 0, s, mx, my,
                                   d_out, uint32_t(num_voxels), cs, nullptr, uint32_t(num_voxels), 0, 0, 0);
            return;
        }
    }
    if (cs == CS_PAD)
        hipLaunchKernelGGL((pearson_symmetric_kernel<CS_PAD, true, kMinWaves>), dim3(blocks), dim3(256), 0, s, mx, my,
                           d_out, uint32_t(num_voxels), cs, nullptr, uint32_t(num_voxels), 0, 0, 0);
    else
        hipLaunchKernelGGL((pearson_symmetric_kernel<CS_PAD, false, kMinWaves>), dim3(blocks), dim3(256), 0, s, mx, my,
                           d_out, uint32_t(num_voxels), cs, nullptr, uint32_t(num_voxels), 0, 0, 0);
}
}  // namespace

namespace {
template <int CS_PAD>
void launch_symmetric_requests(const float* const* mi, const float* const* mj, int cs, size_t num_voxels,
                               const uint32_t* d_requests, size_t num_requests, int xs, int ys, int use_abs, float* d_out,
                               hipStream_t s) {
    constexpr int kMinWaves = CS_PAD <= 32 ? 4 : (CS_PAD <= 96 ? 2 : 1);
    hipLaunchKernelGGL((pearson_symmetric_kernel<CS_PAD, false, kMinWaves, true>), dim3(unsigned((num_requests + 255) / 256)),
                       dim3(256), 0, s, mi, mj, d_out, uint32_t(num_voxels), cs, d_requests, uint32_t(num_requests), xs, ys,
                       use_abs);
}
}  // namespace

// Pearson pair requests through the two-vector register kernel: 2 <= cs <= kMaxSymmetricRegisterMembers, fewer than
// 2^32 requests; hipErrorNotSupported otherwise (-> pair_request_kernel)
hipError_t launch_pearson_requests(const float* const* d_members_i, const float* const* d_members_j, int cs, int xs, int ys,
                                   size_t num_voxels, const uint32_t* d_requests, size_t num_requests, int use_abs,
                                   float* d_out, hipStream_t s) {
    if (cs < 2 || cs > kMaxSymmetricRegisterMembers || !d_requests || num_requests >= (size_t(1) << 32))
        return hipErrorNotSupported;
    if (num_requests == 0) return hipSuccess;
#define CRF_REQ_CASE(I, N) \
    case I: launch_symmetric_requests<N>(d_members_i, d_members_j, cs, num_voxels, d_requests, num_requests, xs, ys, use_abs, d_out, s); break
    switch ((cs + 15) / 16) {
        CRF_REQ_CASE(1, 16);
        CRF_REQ_CASE(2, 32);
        CRF_REQ_CASE(3, 48);
        CRF_REQ_CASE(4, 64);
        CRF_REQ_CASE(5, 80);
        CRF_REQ_CASE(6, 96);
        CRF_REQ_CASE(7, 112);
        default: launch_symmetric_requests<128>(d_members_i, d_members_j, cs, num_voxels, d_requests, num_requests, xs, ys, use_abs, d_out, s); break;
    }
#undef CRF_REQ_CASE
    return hipGetLastError();
}

hipError_t launch_pearson_symmetric(const float* const* d_members_ref, const float* const* d_members_query, int cs,
                                    size_t num_voxels, float* d_out, hipStream_t s) {
    if (cs == 1) return launch_fill(d_out, num_voxels, 1.0f, s);
    if (cs > kMaxSymmetricRegisterMembers) return hipErrorNotSupported;
    switch ((cs + 15) / 16) {
        case 1: launch_symmetric<16>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 2: launch_symmetric<32>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 3: launch_symmetric<48>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 4: launch_symmetric<64>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 5: launch_symmetric<80>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 6: launch_symmetric<96>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        case 7: launch_symmetric<112>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
        default: launch_symmetric<128>(d_members_ref, d_members_query, cs, num_voxels, d_out, s); break;
    }
    return hipGetLastError();
}

hipError_t launch_pearson(const float* const* d_members, int cs, size_t num_voxels, int max_vpt, const RefSource& ref,
                          float* d_prep, float* d_out, hipStream_t s, hipEvent_t ev_begin, hipEvent_t ev_end,
                          LaunchInfo* info) {
    if (cs == 1) {  // CorrelationCalculator.cpp:882-885
        if (!ref.run()) return hipSuccess;
        if (ev_begin) (void)hipEventRecord(ev_begin, s);
        hipError_t e = launch_fill(d_out, num_voxels, 1.0f, s);
        if (ev_end) (void)hipEventRecord(ev_end, s);
        if (info) info->kernel_name = "fill_kernel";
        return e;
    }
    if (ref.prepare())
        hipLaunchKernelGGL(pearson_prep_kernel, dim3(1), dim3(256), size_t(cs) * sizeof(float), s, ref, d_members, cs,
                           d_prep);
    if (!ref.run()) return hipGetLastError();

    size_t covered = 0;
    if (ev_begin) (void)hipEventRecord(ev_begin, s);
    // 289..1216 members: two (up to 576) or four lanes per voxel (pearson_split_kernel).  CRF_PEARSON_SPLIT=0 selects the
    // r02 kernels (VGPRs + AGPRs at one wave per SIMD up to 384 members, the 8-wave relay up to 512, three sweeps beyond).
    // From 289 members: measured at 512x512x128 against the one-lane kernels (registers + LDS rows), % of the HBM peak:
    // 160 members 78 vs 83, 192: 76 vs 83, 224: 75 vs 83, 256: 73 vs 87, 288: 74 vs 79, 320: 74 vs 71
    // (profiles/r03_pearson_two_lanes_from_129_members.txt).  CRF_PEARSON_SPLIT_FROM moves the threshold (>= 129; tuning).
    const int split_from = env_int("CRF_PEARSON_SPLIT_FROM", 289);
    if (cs >= split_from && cs > 128 && cs <= kSplitMaxMembers && env_int("CRF_PEARSON_SPLIT", 1) != 0) {
        const int lanes = cs > 576 ? 4 : 2;  // (two lanes x 304 slots = 224 + 80 LDS rows: 0.5 KB of scratch per lane)
        const int slots = ((cs + lanes - 1) / lanes + 15) / 16 * 16;  // per lane, in steps of 16: 160 .. 304
        const size_t per_block = size_t(4) * (64 / lanes);
        const size_t blocks_ = (num_voxels + per_block - 1) / per_block;
        hipError_t attr = hipSuccess;
        bool launched = false;
#define CRF_LAUNCH_SPLIT(R_, L_, G_, W_)                                                                          \
    {                                                                                                             \
        constexpr size_t kBytes = size_t(L_) * 256 * sizeof(float);                                               \
        const auto kern = &pearson_split_kernel<R_, L_, G_, 16 * G_, W_>;                                         \
        if (kBytes > 64 * 1024)                                                                                   \
            attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                       \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(kBytes));                  \
        if (attr == hipSuccess && blocks_ > 0)                                                                    \
            hipLaunchKernelGGL(kern, dim3(unsigned(blocks_)), dim3(256), kBytes, s, d_members, d_prep, d_out,      \
                               uint32_t(num_voxels), cs);                                                         \
        launched = attr == hipSuccess;                                                                            \
    }
#define CRF_LAUNCH_SPLIT_G(R_, L_, W_)                                                                            \
    if (lanes == 2) CRF_LAUNCH_SPLIT(R_, L_, 2, W_) else CRF_LAUNCH_SPLIT(R_, L_, 4, W_)
        switch (slots) {
            case 80: CRF_LAUNCH_SPLIT(80, 0, 2, 4); break;    // (tuning: CRF_PEARSON_SPLIT_FROM)
            case 96: CRF_LAUNCH_SPLIT(96, 0, 2, 4); break;
            case 112: CRF_LAUNCH_SPLIT(112, 0, 2, 3); break;
            case 128: CRF_LAUNCH_SPLIT(128, 0, 2, 3); break;
            case 144: CRF_LAUNCH_SPLIT(144, 0, 2, 2); break;
            case 160: if (lanes == 2) CRF_LAUNCH_SPLIT(160, 0, 2, 2) else CRF_LAUNCH_SPLIT(160, 0, 4, 2) break;  // 577..640 members
            case 176: CRF_LAUNCH_SPLIT_G(176, 0, 2); break;
            case 192: CRF_LAUNCH_SPLIT_G(192, 0, 2); break;
            case 208: CRF_LAUNCH_SPLIT_G(208, 0, 2); break;
            // beyond 208 slots the rest goes to LDS rows: with 224 register slots the allocation is at its edge and
            // whether 16 B or 0.5 KB of scratch come out depends on details of the rare path (measured: 448 members
            // 72 % of the peak without, 63 % with 52 B of scratch)
            case 224: CRF_LAUNCH_SPLIT_G(208, 16, 2); break;
            case 240: CRF_LAUNCH_SPLIT_G(216, 24, 2); break;
            case 256: CRF_LAUNCH_SPLIT_G(216, 40, 2); break;
            case 272: CRF_LAUNCH_SPLIT_G(216, 56, 2); break;
            case 288: CRF_LAUNCH_SPLIT_G(216, 72, 2); break;
            case 304: CRF_LAUNCH_SPLIT(224, 80, 4, 2); break;  // 80 KB of LDS per block: still two blocks per CU
            // (320 slots = 224 + 96 LDS rows: one block per CU, 17-32 % of the peak -- not instantiated)
            default: break;
        }
#undef CRF_LAUNCH_SPLIT_G
#undef CRF_LAUNCH_SPLIT
        if (launched) {
            covered = num_voxels;
            if (info) info->kernel_name = "pearson_split_kernel";
            goto tail;
        }
    }
    if (cs <= kMaxRegisterMembers) {
        const int cs_pad = cs <= 8 ? 8 : cs <= 128 ? (cs + 15) / 16 * 16 : cs <= 224 ? (cs + 31) / 32 * 32
                         : cs <= 240 ? 240 : cs <= 256 ? (cs + 7) / 8 * 8 : (cs + 63) / 64 * 64;
        // voxels per lane.  Measured on MI355X at 256^3 x 64 (profiles/): one voxel per lane (dword loads, 93 VGPRs,
        // 5 waves/SIMD) reaches 5.7 TB/s; 2 per lane (196 VGPRs, 2 waves/SIMD) 4.9 TB/s; 4 per lane 3.4 TB/s --
        // occupancy, not load width, is what keeps HBM busy here.  CRF_PEARSON_VPT overrides for tuning experiments.
        int vpt = cs_pad <= 16 ? 2 : 1;
        vpt = env_int("CRF_PEARSON_VPT", vpt);
        if (vpt > max_vpt) vpt = max_vpt;
        while (vpt > 1 && cs_pad * vpt > 256) vpt >>= 1;
        if (cs_pad > 64 && cs_pad != 128) vpt = 1;  // wider loads are instantiated for the tuning sizes only
        if (vpt != 1 && vpt != 2 && vpt != 4) vpt = 1;
        const int variant = env_int("CRF_PEARSON_VARIANT", 0);
        if (variant > 0 && cs == 64) {  // tuning experiments (tools/tune_pearson.py), one voxel per lane
#define CRF_VARIANT(MINW, BLK, NT_)                                                                              \
    {                                                                                                            \
        const size_t blocks_ = (num_voxels + BLK - 1) / BLK;                                                     \
        covered = num_voxels;                                                                                    \
        if (blocks_ > 0)                                                                                         \
            hipLaunchKernelGGL((pearson_reg_kernel<64, 1, true, MINW, BLK, NT_>), dim3(unsigned(blocks_)),       \
                               dim3(BLK), 0, s, d_members, d_prep, d_out, uint32_t(num_voxels), cs);                        \
    }
            switch (variant) {
                case 1: CRF_VARIANT(4, 256, false); break;  // temporal (default-policy) loads
                case 2: CRF_VARIANT(2, 512, true); break;   // 512-thread blocks
                case 3: CRF_VARIANT(6, 256, true); break;   // register cap for 6 waves/SIMD
                case 4: CRF_VARIANT(4, 128, true); break;   // 128-thread blocks
                case 5: CRF_VARIANT(4, 64, true); break;    // one wave per block
                default: CRF_VARIANT(4, 256, true); break;
            }
#undef CRF_VARIANT
            if (info) info->kernel_name = "pearson_reg_kernel";
            goto tail;
        }
        // 241..320 members: 240 values in registers + the rest in the lane's LDS column (pearson_reg_lds_kernel); measured at
        // 512x512x128: 256 members 85 % of the HBM peak (64 % with 256 register values under the two-wave cap, 53 % at one
        // wave), 272: 82 % (57 %), 288: 78 % (54 %), 300: 71 % (54 %), 320: 69 % (59 %).  CRF_PEARSON_LDS_TAIL=0 selects the pure register kernels.
        if (cs > 224 && cs <= 320 && env_int("CRF_PEARSON_LDS_TAIL", 1) != 0) {
            const size_t blocks_ = (num_voxels + 255) / 256;
            hipError_t attr = hipSuccess;
#define CRF_LAUNCH_REG_LDS(R_, L_)                                                                                \
    {                                                                                                             \
        constexpr size_t kBytes = size_t(L_) * 256 * sizeof(float);                                               \
        if (kBytes > 64 * 1024)                                                                                   \
            attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&pearson_reg_lds_kernel<R_, L_, 2>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(kBytes));                  \
        if (attr == hipSuccess)                                                                                   \
            hipLaunchKernelGGL((pearson_reg_lds_kernel<R_, L_, 2>), dim3(unsigned(blocks_)), dim3(256), kBytes, s,  \
                               d_members, d_prep, d_out, uint32_t(num_voxels), cs);                               \
    }
            if (cs <= 240) {
                CRF_LAUNCH_REG_LDS(224, 16)
            } else if (cs <= 256) {
                CRF_LAUNCH_REG_LDS(240, 16)
            } else if (cs <= 288) {
                CRF_LAUNCH_REG_LDS(240, 48)
            } else {
                CRF_LAUNCH_REG_LDS(240, 80)  // (144 rows for 321..384 were measured: one block per CU, 31-33 % -- not kept)
            }
#undef CRF_LAUNCH_REG_LDS
            if (attr == hipSuccess) {
                covered = num_voxels;
                if (info) info->kernel_name = "pearson_reg_lds_kernel";
                goto tail;
            }
        }
        const size_t per_block = size_t(256) * vpt;
        covered = num_voxels / vpt * vpt;                       // whole vectors; the descriptor bounds the last block
        const size_t blocks = (covered + per_block - 1) / per_block;
        if (blocks > 0) {
            switch (cs_pad) {
                case 8: launch_reg_vpt<8>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 16: launch_reg_vpt<16>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 32: launch_reg_vpt<32>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 48: launch_reg_vpt<48>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 64: launch_reg_vpt<64>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 80: launch_reg_vpt<80>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 96: launch_reg_vpt<96>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 112: launch_reg_vpt<112>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 128: launch_reg_vpt<128>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 160: launch_reg_vpt<160>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 192: launch_reg_vpt<192>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 224: launch_reg_vpt<224>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 240: launch_reg_vpt<240>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 248: launch_reg_vpt<248>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 256: launch_reg_vpt<256>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                case 320: launch_reg_vpt<320>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
                default: launch_reg_vpt<384>(vpt, d_members, d_prep, d_out, blocks, num_voxels, cs, s); break;
            }
        }
        if (info) info->kernel_name = "pearson_reg_kernel";
    } else {
        // chunked three-pass kernel (see pearson_big_kernel); CRF_PEARSON_BIG=0 selects the plain streaming kernel,
        // CRF_PEARSON_BIG_WAVES overrides the grid (tuning)
        if (cs <= 512 && env_int("CRF_PEARSON_RELAY", 1) != 0) {
            // one block per 64-voxel tile at a time, as many blocks as the chip holds
            const size_t tiles = (num_voxels + 63) / 64;
            int per_cu = 1, device = 0, cus = 256;
            (void)hipGetDevice(&device);
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pearson_relay_kernel<8, 64>, 512, 0);
            const size_t want = size_t(cus) * size_t(per_cu > 0 ? per_cu : 1);
            const unsigned blocks = unsigned(tiles < want ? tiles : want);
            hipLaunchKernelGGL((pearson_relay_kernel<8, 64>), dim3(blocks), dim3(512), 0, s, d_members, d_prep, d_out,
                               uint32_t(num_voxels), cs);
            covered = num_voxels;
            if (info) info->kernel_name = "pearson_relay_kernel";
        } else if (env_int("CRF_PEARSON_BIG", 1) != 0) {
            const size_t tiles = (num_voxels + 63) / 64;
            const size_t want = size_t(env_int("CRF_PEARSON_BIG_WAVES", kBigWaves));
            const unsigned blocks = unsigned(tiles < want ? tiles : want);
            hipLaunchKernelGGL(pearson_big_kernel, dim3(blocks), dim3(64), 0, s, d_members, d_prep, d_out,
                               uint32_t(num_voxels), cs);
            covered = num_voxels;
            if (info) info->kernel_name = "pearson_big_kernel";
        } else if (info) {
            info->kernel_name = "pearson_stream_kernel";
        }
    }
tail:
    if (covered < num_voxels) {
        const size_t rest = num_voxels - covered;
        hipLaunchKernelGGL(pearson_stream_kernel, dim3(unsigned((rest + 255) / 256)), dim3(256), 0, s, d_members,
                           d_prep, d_out, covered, num_voxels, cs);
    }
    if (ev_end) (void)hipEventRecord(ev_end, s);
    return hipGetLastError();
}

}  // namespace crf
