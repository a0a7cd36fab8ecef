// Development microbenchmark: issue rate of the VALU instructions the sort / k-select kernels are built from, on gfx950.
// 8 independent chains per lane, N iterations; reports wave-instructions per cycle per SIMD (clock taken as 2.4 GHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHAINS 8
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed) {
    double a[CHAINS];
    uint32_t u[CHAINS];
    float f[CHAINS];
    for (int c = 0; c < CHAINS; c++) {
        a[c] = seed + c + threadIdx.x;
        u[c] = uint32_t(threadIdx.x * 7 + c);
        f[c] = float(seed) + c;
    }
    double b = seed * 0.5 + 3.0;
    uint32_t ub = uint32_t(seed) + 5u;
    float fb = float(seed) + 2.0f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
            if (OP == 2) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[c]) : "v"(ub));
            if (OP == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fb));
            if (OP == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
            if (OP == 5) asm volatile("v_max_f64 %0, |%0|, |%1|" : "+v"(a[c]) : "v"(b));
            if (OP == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[c]), "+v"(b), "+v"(u[c]) : "v"(ub) : "vcc");
            if (OP == 7) asm volatile("v_min_f32 %0, %0, %1" : "+v"(f[c]) : "v"(fb));
            if (OP == 8) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(ub) : "vcc");
            if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
        }
    }
    double s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c] + double(u[c]) + double(f[c]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, int instr_per_op) {
    const int blocks = 256 * 8, iters = 20000;
    double* out;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, 256>>>(out, 100, 1.0);
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, 256>>>(out, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions: blocks * 4 waves * iters * CHAINS * instr_per_op; SIMDs: 1024
    const double winstr = double(blocks) * 4 * iters * CHAINS * instr_per_op;
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD\n", name, ms, cycles * 1024 / winstr);
    hipFree(out);
}

int main() {
    run<3>("v_add_f32", 1);
    run<7>("v_min_f32", 1);
    run<2>("v_min_u32", 1);
    run<8>("v_cmp_lt_u32 + v_cndmask", 2);
    run<1>("v_add_f64", 1);
    run<4>("v_fma_f64", 1);
    run<0>("v_min_f64", 1);
    run<5>("v_max_f64 |a|,|b|", 1);
    run<6>("v_cmp_lt_f64 + v_cndmask", 2);
    run<9>("v_pk_add_f32", 1);
    return 0;
}
