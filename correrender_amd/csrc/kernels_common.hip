// kernels_common.hip -- small gfx950 helper kernels around the per-voxel estimators:
//   * reference-vector gather      (referenceValues[c] = fields[c][IDXS(ref)], CorrelationCalculator.cpp:802,815-817)
//   * member min/max reduction     (getMinMaxScalarFieldValue per member + min/max over members,
//                                   VolumeData.cpp:1632-1670, CorrelationCalculator.cpp:822-829)
//   * synthetic box-ensemble fill  (recipe of scripts/generate_synth_box_ensembles.py:57-136; input generation only)
#include "crf_internal.h"

namespace crf {

// ---------------------------------------------------------------------------------------------------------
__global__ void gather_reference_kernel(const float* const* __restrict__ members, int cs, size_t voxel,
                                        float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cs) out[c] = members[c][voxel];
}

hipError_t launch_gather_reference(const float* const* d_members, int cs, size_t voxel, float* d_out, hipStream_t s) {
    const int block = 64;
    hipLaunchKernelGGL(gather_reference_kernel, dim3((cs + block - 1) / block), dim3(block), 0, s, d_members, cs,
                       voxel, d_out);
    return hipGetLastError();
}

// Rows of a batched reference-vector exchange (multi-GPU: one collective per several evaluations): row r receives the
// reference values of voxel[r], or zeros when voxel[r] == kNoVoxel (a point whose slice another rank owns) -- one
// launch instead of a memset plus one gather per owned row.
__global__ void gather_reference_rows_kernel(const float* const* __restrict__ members, int cs, GatherRows rows,
                                             float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (c < cs) {
        const size_t voxel = rows.voxel[r];
        out[size_t(r) * size_t(cs) + c] = voxel == kNoVoxel ? 0.0f : members[c][voxel];
    }
}

hipError_t launch_gather_reference_rows(const float* const* d_members, int cs, const GatherRows& rows, int num_rows,
                                        float* d_out, hipStream_t s) {
    if (num_rows <= 0) return hipSuccess;
    const int block = 64;
    hipLaunchKernelGGL(gather_reference_rows_kernel, dim3((cs + block - 1) / block, num_rows), dim3(block), 0, s,
                       d_members, cs, rows, d_out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Order-preserving float -> uint32 key so that unsigned atomicMin/atomicMax order like the floats do.
__device__ __forceinline__ uint32_t float_to_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
float minmax_key_to_float(uint32_t key) {
    const uint32_t b = (key & 0x80000000u) ? (key & 0x7FFFFFFFu) : ~key;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}

// grid = (blocks_per_member, cs).  Each block strides over one member with 16-byte loads; NaNs are skipped
// (fminf/fmaxf), one pair of atomics per wave.
__global__ __launch_bounds__(256) void minmax_kernel(const float* const* __restrict__ members, size_t num_voxels,
                                                     uint32_t* __restrict__ keys) {
    const float* __restrict__ p = members[blockIdx.y];
    float mn = INFINITY, mx = -INFINITY;
    const size_t n4 = num_voxels / 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    const size_t t = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (aligned) {
        const float4* __restrict__ p4 = reinterpret_cast<const float4*>(p);
        for (size_t i = t; i < n4; i += stride) {
            const float4 v = p4[i];
            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
            mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
        for (size_t i = n4 * 4 + t; i < num_voxels; i += stride) {
            mn = fminf(mn, p[i]);
            mx = fmaxf(mx, p[i]);
        }
    } else {
        for (size_t i = t; i < num_voxels; i += stride) {
            mn = fminf(mn, p[i]);
            mx = fmaxf(mx, p[i]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (mn <= mx) {  // false only when every value seen was NaN / nothing was seen
            atomicMin(&keys[0], float_to_key(mn));
            atomicMax(&keys[1], float_to_key(mx));
        }
    }
}

__global__ void minmax_init_kernel(uint32_t* keys) {
    keys[0] = 0xFFFFFFFFu;
    keys[1] = 0u;
}

hipError_t launch_minmax(const float* const* d_members, int cs, size_t num_voxels, uint32_t* d_keys, hipStream_t s) {
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, d_keys);
    size_t blocks = (num_voxels / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 64) blocks = 64;  // 64 blocks x cs members >> 256 CUs
    hipLaunchKernelGGL(minmax_kernel, dim3(unsigned(blocks), unsigned(cs)), dim3(256), 0, s, d_members, num_voxels,
                       d_keys);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Synthetic "box ensemble".  lambda(x,y,z) = sum over 10 boxes of peak(chebyshev((x,y,z)-(cx,cy,zs/2)) / (size/2)),
// peak(u) = 0 for u >= 1 else 1 - max(0, 2|u|-1)^2  (generate_synth_box_ensembles.py:57-61,70-102, with g = xs/8
// for our grids; the original uses xs=ys=128, zs=32, g=16).  Sample of member c at a voxel:
// lambda*s1[c] + (1-lambda)*N(0,1) with s1 = 2*linspace(0,1,cs)-1 (:113-136).  The normal deviate comes from a
// counter-based hash of (seed, global voxel, member), so a z-slab generated on one rank equals the same slab of
// the whole grid generated elsewhere.
__device__ __forceinline__ float peak_fun(float u) {
    if (u >= 1.0f) return 0.0f;
    const float t = fmaxf(0.0f, fabsf(u) * 2.0f - 1.0f);
    return 1.0f - t * t;
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_box_kernel(float* __restrict__ out, int xs, int ys, int zs_local,
                                                        int z_begin, int zs_global, int c, int cs, uint64_t seed) {
    const size_t n = size_t(xs) * ys * zs_local;
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = int(i % size_t(xs));
    const int y = int((i / size_t(xs)) % size_t(ys));
    const int z = int(i / (size_t(xs) * ys)) + z_begin;
    const float g = float(xs) / 8.0f;
    const float cz = float(zs_global / 2);
    // (cx, cy, size) in units of g
    const float boxes[10][3] = {{1.0f, 1.0f, 2.0f}, {7.0f, 7.0f, 2.0f}, {2.5f, 0.5f, 1.0f}, {2.5f, 1.5f, 1.0f},
                                {5.5f, 6.5f, 1.0f}, {5.5f, 7.5f, 1.0f}, {0.5f, 2.5f, 1.0f}, {1.5f, 2.5f, 1.0f},
                                {6.5f, 5.5f, 1.0f}, {7.5f, 5.5f, 1.0f}};
    float lambda = 0.0f;
#pragma unroll
    for (int b = 0; b < 10; b++) {
        const float dx = fabsf(float(x) - boxes[b][0] * g);
        const float dy = fabsf(float(y) - boxes[b][1] * g);
        const float dz = fabsf(float(z) - cz);
        const float dist = fmaxf(dx, fmaxf(dy, dz)) / (boxes[b][2] * g * 0.5f);
        lambda += peak_fun(dist);
    }
    lambda = fminf(lambda, 1.0f);
    const size_t gvoxel = (size_t(z) * ys + y) * size_t(xs) + x;
    const uint64_t h = mix64(seed ^ mix64(gvoxel * uint64_t(cs) + uint64_t(c)));
    const float u1 = (float(uint32_t(h >> 40)) + 0.5f) * (1.0f / 16777216.0f);          // (0,1)
    const float u2 = (float(uint32_t(h >> 8) & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    const float normal = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    const float s1 = cs > 1 ? 2.0f * (float(c) / float(cs - 1)) - 1.0f : -1.0f;
    out[i] = lambda * s1 + (1.0f - lambda) * normal;
}

hipError_t launch_synth_box_member(float* d_out, int xs, int ys, int zs_local, int z_begin, int zs_global, int c,
                                   int cs, uint64_t seed, hipStream_t s) {
    const size_t n = size_t(xs) * ys * zs_local;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(synth_box_kernel, dim3(unsigned(blocks)), dim3(256), 0, s, d_out, xs, ys, zs_local, z_begin,
                       zs_global, c, cs, seed);
    return hipGetLastError();
}

}  // namespace crf
