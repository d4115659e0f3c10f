// Micro-benchmark (not part of the product): do two independent chains of
// short dependent kernels, each on its own stream and each filling HALF the
// CUs, run side by side?  Prints the time of one chain alone and of both.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

__global__ __launch_bounds__(512) void step_kernel(float *x, long long cycles) {
    __shared__ float lds[16 * 1024];   // 64 KiB like the cell kernels
    lds[threadIdx.x] = x[blockIdx.x * 512 + threadIdx.x];
    __syncthreads();
    const long long t0 = wall_clock64();
    float v = lds[(threadIdx.x * 7) & 511];
    while (wall_clock64() - t0 < cycles) v = v * 1.0001f + 0.5f;
    x[blockIdx.x * 512 + threadIdx.x] = v;
}

static void chain(hipStream_t s, float *x, int blocks, int n, long long cycles) {
    for (int i = 0; i < n; ++i)
        hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(512), 0, s, x, cycles);
}

int main(int argc, char **argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 128;
    const double us = argc > 2 ? atof(argv[2]) : 8.0;
    const int n = 2000;
    const long long cycles = (long long)(us * 100.0);   // wall_clock64: 100 MHz
    float *x, *y;
    hipMalloc(&x, 512 * 512 * 4); hipMalloc(&y, 512 * 512 * 4);
    hipMemset(x, 0, 512 * 512 * 4); hipMemset(y, 0, 512 * 512 * 4);
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto d) { return std::chrono::duration<double, std::milli>(d).count(); };
    chain(a, x, blocks, 200, cycles); chain(b, y, blocks, 200, cycles);
    hipDeviceSynchronize();
    auto t0 = now();
    chain(a, x, blocks, n, cycles);
    hipStreamSynchronize(a);
    const double one = ms(now() - t0);
    t0 = now();
    std::thread ta([&] { chain(a, x, blocks, n, cycles); });
    std::thread tb([&] { chain(b, y, blocks, n, cycles); });
    ta.join(); tb.join();
    hipDeviceSynchronize();
    const double two = ms(now() - t0);
    t0 = now();
    for (int i = 0; i < n; ++i) {   // interleaved from one host thread
        hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(512), 0, a, x, cycles);
        hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(512), 0, b, y, cycles);
    }
    hipDeviceSynchronize();
    const double inter = ms(now() - t0);
    // the same single chain replayed from a hipGraph
    hipGraph_t graph; hipGraphExec_t exec;
    hipStreamBeginCapture(a, hipStreamCaptureModeGlobal);
    chain(a, x, blocks, n, cycles);
    hipStreamEndCapture(a, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, a); hipStreamSynchronize(a);
    t0 = now();
    hipGraphLaunch(exec, a); hipStreamSynchronize(a);
    const double gr = ms(now() - t0);
    printf("one chain as a hipGraph: %.2f ms (%.2f us/launch)\n", gr, gr * 1e3 / n);
    printf("blocks %d, %.1f us kernels, %d per chain: one chain %.2f ms (%.2f us/launch); "
           "two chains, two host threads %.2f ms; interleaved from one thread %.2f ms\n",
           blocks, us, n, one, one * 1e3 / n, two, inter);
    return 0;
}
