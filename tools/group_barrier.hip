// Micro-benchmark (not part of the product): cost of a barrier among the 16
// blocks of one XCD-local group versus a kernel boundary, with a real data
// hand-off (each block writes 4 KB, then reads its 15 peers' 4 KB).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__device__ __forceinline__ bool group_barrier(unsigned *cnt, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { ok = false; break; }  // bounded: never hang
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void persistent(float *buf, unsigned *cnt, int phases,
                                                  int group_size, int *err) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int per_xcd = gridDim.x >> 3;
    const int gpx = per_xcd / group_size;            // groups per XCD
    const int g = xcd * gpx + i / group_size, m = i % group_size;
    float *mine = buf + ((size_t)g * group_size + m) * 1024;
    float acc = 0.f;
    for (int p = 1; p <= phases; ++p) {
        for (int k = threadIdx.x; k < 1024; k += blockDim.x) mine[k] = acc + p;
        if (!group_barrier(cnt + g * 32, (unsigned)(p * group_size))) { if (threadIdx.x == 0) *err = 1; return; }
        for (int q = 0; q < group_size; ++q)
            acc += buf[((size_t)g * group_size + q) * 1024 + threadIdx.x];
        // second barrier: nobody overwrites before everyone has read
        if (!group_barrier(cnt + g * 32 + 16, (unsigned)(p * group_size))) { if (threadIdx.x == 0) *err = 1; return; }
    }
    if (acc == 12345.f) buf[0] = acc;
}

__global__ __launch_bounds__(512) void phase_kernel(float *buf, int p, int group_size, int write) {
    const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
    const int gpx = (gridDim.x >> 3) / group_size;
    const int g = xcd * gpx + i / group_size, m = i % group_size;
    float *mine = buf + ((size_t)g * group_size + m) * 1024;
    if (write) { for (int k = threadIdx.x; k < 1024; k += blockDim.x) mine[k] = p; return; }
    float acc = 0.f;
    for (int q = 0; q < group_size; ++q) acc += buf[((size_t)g * group_size + q) * 1024 + threadIdx.x];
    if (acc == 12345.f) buf[0] = acc;
}

int main() {
    float *buf; unsigned *cnt; int *err;
    hipMalloc(&buf, 256 * 4096); hipMalloc(&cnt, 4096 * 4); hipMalloc(&err, 4);
    hipMemset(err, 0, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int phases = 2000;
    for (int gs : {16, 32, 4}) {
        hipMemset(cnt, 0, 4096 * 4);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(persistent, dim3(256), dim3(512), 100 * 1024, s, buf, cnt, phases, gs, err);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
        printf("persistent, groups of %2d blocks: %.2f us per phase (2 barriers + 4 KB hand-off) err=%d\n", gs,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / phases, e);
    }
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int p = 0; p < phases; ++p) {
        hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(512), 100 * 1024, s, buf, p, 16, 1);
        hipLaunchKernelGGL(phase_kernel, dim3(256), dim3(512), 100 * 1024, s, buf, p, 16, 0);
    }
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    printf("two launches per phase: %.2f us per phase\n",
           std::chrono::duration<double, std::micro>(t1 - t0).count() / phases);
    return 0;
}
