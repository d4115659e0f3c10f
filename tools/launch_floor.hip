// Micro-benchmark (not part of the product): period of a chain of dependent
// kernels on one stream as a function of what each kernel does.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel(float *x) { if (x == nullptr) x[0] = 0; }
__global__ void touch_kernel(float *x, int n) {
    // one coalesced read-modify-write of n floats per block (L2-warm after warm-up)
    for (int i = threadIdx.x; i < n; i += blockDim.x) x[blockIdx.x * n + i] += 1.f;
}
template <class F> void timeit(const char *name, int N, F launch) {
    for (int i = 0; i < 50; ++i) launch();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) launch();
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    printf("%-44s %.2f us per launch\n", name,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
}
int main() {
    float *x; hipMalloc(&x, 256u << 20); hipMemset(x, 0, 256u << 20);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 4000;
    timeit("empty 1 block x 64", N, [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, x); });
    timeit("empty 256 blocks x 256", N, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, x); });
    timeit("empty 256 blocks x 512, 100 KB LDS", N, [&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 100 * 1024, s, x); });
    timeit("touch 256 blocks x 512, 4 KB each", N, [&] { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(512), 0, s, x, 1024); });
    timeit("touch 256 blocks x 512, 40 KB each", N, [&] { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(512), 0, s, x, 10240); });
    timeit("touch 1024 blocks x 256, 4 KB each", N, [&] { hipLaunchKernelGGL(touch_kernel, dim3(1024), dim3(256), 0, s, x, 1024); });
    return 0;
}
