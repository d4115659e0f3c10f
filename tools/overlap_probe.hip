// Micro-benchmark (not part of the product): does a chain of short dependent
// kernels on one stream keep its pace while one long kernel runs on another
// stream?  Varies the long kernel's footprint and what it does.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ int g_prio;
__global__ void short_kernel(float *x, int iters) {
    if (iters < 0) { iters = -iters; __builtin_amdgcn_s_setprio(3); }
    float v = x[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    x[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

// mode 0: sleep loop, 1: VALU loop, 2: MFMA loop
__global__ void stream_kernel(const float4 *src, float4 *dst, long long n, int reps) {
    // memory-heavy long kernel: persistent blocks stream src -> dst `reps` times
    float4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
             i += (long long)gridDim.x * blockDim.x) {
            float4 v = src[i];
            acc.x += v.x; acc.y += v.y;
            dst[i] = v;
        }
    if (acc.x == 123.f) dst[0] = acc;
}

__global__ void long_kernel(float *x, long long cycles, int mode) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0) sm[0] = 1.f;
    const long long t0 = wall_clock64();
    f32x16 acc = {0};
    float v = x[threadIdx.x];
    while (wall_clock64() - t0 < cycles) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else if (mode == 1) {
            for (int i = 0; i < 64; ++i) v = v * 1.0001f + 0.5f;
        } else {
            for (int i = 0; i < 16; ++i)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 1.0f, acc, 0, 0, 0);
        }
    }
    if (v == 123.f || acc[0] == 77.f) x[0] = v + acc[1];
}

int main(int argc, char **argv) {
    float *x, *y;
    hipMalloc(&x, 1 << 24); hipMalloc(&y, 1 << 24);
    hipMemset(x, 0, 1 << 24); hipMemset(y, 0, 1 << 24);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    int least, greatest;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&b, hipStreamNonBlocking, least);
    const int N = 400;
    auto chain = [&](int blocks, int threads) {
        for (int i = 0; i < N; ++i)
            hipLaunchKernelGGL(short_kernel, dim3(blocks), dim3(threads), 0, a, x, 2000);
    };
    auto timeit = [&](const char *name, int lb, int lt, int lds, int mode, int cb, int ct) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (lb > 0)  // wall_clock64 ticks at 100 MHz: 3 ms
            hipLaunchKernelGGL(long_kernel, dim3(lb), dim3(lt), lds, b, y, 300000LL, mode);
        chain(cb, ct);
        hipStreamSynchronize(a);
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("%-46s chain %7.2f ms (%.1f us/kernel)  all %7.2f ms\n", name,
               std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::milli>(t2 - t0).count());
    };
    for (int rep = 0; rep < 2; ++rep) {
        timeit("chain alone (256x256)", 0, 0, 0, 0, 256, 256);
        timeit("+ long 256x64 sleep", 256, 64, 0, 0, 256, 256);
        timeit("+ long 512x256 lds33K sleep", 512, 256, 33 * 1024, 0, 256, 256);
        timeit("+ long 512x256 lds33K valu", 512, 256, 33 * 1024, 1, 256, 256);
        timeit("+ long 512x256 lds33K mfma", 512, 256, 33 * 1024, 2, 256, 256);
        timeit("+ long 1024x256 lds33K mfma", 1024, 256, 33 * 1024, 2, 256, 256);
        timeit("+ long 256x256 lds33K mfma", 256, 256, 33 * 1024, 2, 256, 256);
        timeit("+ long 2048x256 lds33K mfma (oversubscribed)", 2048, 256, 33 * 1024, 2, 256, 256);
        timeit("chain 256x512 alone", 0, 0, 0, 0, 256, 512);
        timeit("chain 256x512 + long 512x256 mfma", 512, 256, 33 * 1024, 2, 256, 512);
    }
    // ---- the product's pattern: long kernel released mid-chain by an event
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    hipEvent_t tev[N + 1];
    for (int i = 0; i <= N; ++i) hipEventCreate(&tev[i]);
    float4 *big_src, *big_dst;
    const long long nbig = (512LL << 20) / 16;
    hipMalloc(&big_src, nbig * 16); hipMalloc(&big_dst, nbig * 16);
    hipMemset(big_src, 0, nbig * 16);
    for (int variant = 0; variant < 8; ++variant) {
        const int iters = variant >= 6 ? -2000 : 2000;  // >= 6: chain waves at s_setprio 3
        hipStream_t cs = (variant & 1) ? (hipStream_t)0 : a;
        const int mode = (variant & 2) ? 2 : ((variant & 4) ? 3 : 0);
        if (variant >= 6) printf("(chain at setprio 3) ");
        hipDeviceSynchronize();
        for (int i = 0; i < N; ++i) {
            hipEventRecord(tev[i], cs);
            hipLaunchKernelGGL(short_kernel, dim3(256), dim3(256), 0, cs, x, iters);
            if (i == 100) {
                hipEventRecord(ev, cs);
                hipStreamWaitEvent(b, ev, 0);
                if (mode == 3)
                    hipLaunchKernelGGL(stream_kernel, dim3(512), dim3(256), 0, b,
                                       big_src, big_dst, nbig, 8);
                else
                    hipLaunchKernelGGL(long_kernel, dim3(512), dim3(256), 33 * 1024, b, y,
                                       300000LL, mode);
            }
        }
        hipEventRecord(tev[N], cs);
        hipDeviceSynchronize();
        float worst = 0.f, total = 0.f; int wi = 0;
        for (int i = 0; i < N; ++i) {
            float ms; hipEventElapsedTime(&ms, tev[i], tev[i + 1]);
            total += ms;
            if (ms > worst) { worst = ms; wi = i; }
        }
        printf("mid-chain release: chain on %s, long=%s: total %.2f ms, worst period %.3f ms at kernel %d\n",
               (variant & 1) ? "null stream" : "own stream", mode == 3 ? "memory stream" : mode ? "mfma" : "sleep", total, worst, wi);
    }
    return 0;
}
