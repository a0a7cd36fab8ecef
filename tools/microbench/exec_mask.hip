// Does a VALU instruction cost less when only a quarter / half of the wave's lanes are active?  (pearson_split_kernel runs
// G relay stages in which only 64 / G lanes produce a value that is kept.)   hipcc --offload-arch=gfx950 -O3 exec_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ACTIVE>  // lanes 0 .. ACTIVE-1 of each wave run the chain
__global__ __launch_bounds__(256) void chain(float* out, int iters, float seed) {
    const int lane = threadIdx.x & 63;
    float a = seed + lane, b = seed * 0.5f, c = 1.0f, d = 2.0f;
    if (lane < ACTIVE) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 64; u++) {
                a = a * 1.0001f + b;
                c = c * 0.9999f + d;
                b = b * 1.0002f + a;
                d = d * 0.9998f + c;
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

template <int ACTIVE>
float run(float* d_out, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<ACTIVE>, dim3(blocks), dim3(256), 0, 0, d_out, 4, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<ACTIVE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 400;  // 8 blocks of 4 waves per CU: 8 waves per SIMD
    float* d_out;
    hipMalloc(&d_out, size_t(blocks) * 256 * sizeof(float));
    printf("4 independent fp32 mul+add chains, %d x 256 instructions per wave, %d waves per SIMD\n", iters, 8);
    printf("active lanes 64: %.3f ms\n", run<64>(d_out, blocks, iters));
    printf("active lanes 32: %.3f ms\n", run<32>(d_out, blocks, iters));
    printf("active lanes 16: %.3f ms\n", run<16>(d_out, blocks, iters));
    printf("active lanes  8: %.3f ms\n", run<8>(d_out, blocks, iters));
    return 0;
}
