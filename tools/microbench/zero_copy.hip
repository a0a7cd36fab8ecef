// Development microbenchmark for the host boundary of crf_compute (calculateCpu's float* buffer): how fast can 4 bytes
// per voxel leave the GPU -- DMA engine copies into pinned memory vs. a kernel storing straight into device-mapped
// pinned host memory ("zero copy") -- and how fast can host threads move a pinned staging buffer into a pageable one.
//   hipcc --offload-arch=gfx950 -O3 -o zero_copy zero_copy.hip -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = src[i] * 1.0001f;
    if (MODE == 0) dst[i] = v;
    if (MODE == 1) __builtin_nontemporal_store(v, dst + i);
}

template <int MODE>
__global__ __launch_bounds__(256) void store4_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 v = src[i];
    v.x *= 1.0001f;
    if (MODE == 0) dst[i] = v;
    if (MODE == 1) {
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t n = size_t(256) * 256 * 256;
    const size_t bytes = n * sizeof(float);
    float* d_src;
    CK(hipMalloc(&d_src, bytes));
    CK(hipMemset(d_src, 0, bytes));
    float *h_def, *h_nc, *h_wc;
    CK(hipHostMalloc(reinterpret_cast<void**>(&h_def), bytes, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void**>(&h_nc), bytes, hipHostMallocNonCoherent));
    CK(hipHostMalloc(reinterpret_cast<void**>(&h_wc), bytes, hipHostMallocWriteCombined));
    memset(h_def, 0, bytes);
    memset(h_nc, 0, bytes);
    memset(h_wc, 0, bytes);
    hipStream_t s, s2;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const char* sd = getenv("HSA_ENABLE_SDMA");
    printf("HSA_ENABLE_SDMA=%s hardware_concurrency=%u\n", sd ? sd : "(unset)", std::thread::hardware_concurrency());

    auto time_it = [&](const char* what, auto&& fn, int reps = 10) {
        for (int i = 0; i < 3; i++) fn();
        CK(hipDeviceSynchronize());
        std::vector<double> t;
        for (int i = 0; i < reps; i++) {
            const double t0 = now();
            fn();
            CK(hipDeviceSynchronize());
            t.push_back(now() - t0);
        }
        std::sort(t.begin(), t.end());
        printf("%-64s median %.3f ms  min %.3f ms  (%.1f GB/s)\n", what, t[t.size() / 2] * 1e3, t[0] * 1e3,
               bytes / t[t.size() / 2] / 1e9);
        fflush(stdout);
    };
    struct {
        const char* name;
        float* p;
    } hosts[] = {{"default(coherent)", h_def}, {"non-coherent", h_nc}, {"write-combined", h_wc}};
    for (auto& h : hosts) {
        char label[128];
        snprintf(label, sizeof label, "hipMemcpyAsync D2H, one piece -> %s", h.name);
        time_it(label, [&] { CK(hipMemcpyAsync(h.p, d_src, bytes, hipMemcpyDeviceToHost, s)); });
        for (int pieces : {8, 32}) {
            snprintf(label, sizeof label, "hipMemcpyAsync D2H, %d pieces -> %s", pieces, h.name);
            time_it(label, [&] {
                const size_t per = n / pieces;
                for (int j = 0; j < pieces; j++)
                    CK(hipMemcpyAsync(h.p + j * per, d_src + j * per, per * sizeof(float), hipMemcpyDeviceToHost, s));
            });
        }
        snprintf(label, sizeof label, "kernel dword store -> %s", h.name);
        time_it(label, [&] { store_kernel<0><<<dim3(unsigned(n / 256)), 256, 0, s>>>(d_src, h.p, n); });
        snprintf(label, sizeof label, "kernel dword NT store -> %s", h.name);
        time_it(label, [&] { store_kernel<1><<<dim3(unsigned(n / 256)), 256, 0, s>>>(d_src, h.p, n); });
        snprintf(label, sizeof label, "kernel dwordx4 store -> %s", h.name);
        time_it(label, [&] {
            store4_kernel<0><<<dim3(unsigned(n / 1024)), 256, 0, s>>>(reinterpret_cast<const float4*>(d_src),
                                                                     reinterpret_cast<float4*>(h.p), n / 4);
        });
        for (int pieces : {16, 64}) {
            snprintf(label, sizeof label, "kernel dword NT store, %d launches -> %s", pieces, h.name);
            time_it(label, [&] {
                const size_t per = n / pieces;
                for (int j = 0; j < pieces; j++)
                    store_kernel<1><<<dim3(unsigned(per / 256)), 256, 0, s>>>(d_src + j * per, h.p + j * per, per);
            });
        }
    }
    // kernel (device -> device, 0.65 ms-like) on stream s while DMA pieces run on s2: do they overlap?
    {
        float* d_dst;
        CK(hipMalloc(&d_dst, bytes));
        time_it("device->device kernel alone", [&] { store_kernel<1><<<dim3(unsigned(n / 256)), 256, 0, s>>>(d_src, d_dst, n); });
        time_it("device->device kernel x10 (s) || DMA D2H 8 pieces (s2)", [&] {
            for (int i = 0; i < 10; i++) store_kernel<1><<<dim3(unsigned(n / 256)), 256, 0, s>>>(d_src, d_dst, n);
            const size_t per = n / 8;
            for (int j = 0; j < 8; j++)
                CK(hipMemcpyAsync(h_def + j * per, d_src + j * per, per * sizeof(float), hipMemcpyDeviceToHost, s2));
        });
        time_it("device->device kernel x10 alone", [&] {
            for (int i = 0; i < 10; i++) store_kernel<1><<<dim3(unsigned(n / 256)), 256, 0, s>>>(d_src, d_dst, n);
        });
        CK(hipFree(d_dst));
    }
    // host side: pinned staging -> pageable destination with T threads (resident and fresh destination)
    for (int threads : {1, 2, 4, 8, 16}) {
        for (int fresh = 0; fresh < 2; fresh++) {
            std::vector<double> t;
            float* dst = static_cast<float*>(malloc(bytes));
            memset(dst, 1, bytes);
            for (int rep = 0; rep < 7; rep++) {
                if (fresh) {
                    free(dst);
                    dst = static_cast<float*>(malloc(bytes));
                }
                const double t0 = now();
                std::vector<std::thread> pool;
                for (int w = 0; w < threads; w++)
                    pool.emplace_back([&, w] {
                        const size_t per = bytes / threads;
                        memcpy(reinterpret_cast<char*>(dst) + w * per, reinterpret_cast<const char*>(h_def) + w * per, per);
                    });
                for (auto& th : pool) th.join();
                t.push_back(now() - t0);
            }
            free(dst);
            std::sort(t.begin(), t.end());
            printf("host memcpy pinned -> %s pageable, %2d threads (spawned per call)   median %.3f ms  (%.1f GB/s)\n",
                   fresh ? "fresh   " : "resident", threads, t[t.size() / 2] * 1e3, bytes / t[t.size() / 2] / 1e9);
        }
    }
    return 0;
}
