// Micro-benchmark (not part of the product): does other work issue under an fp32
// MFMA stream on gfx950 -- from the partner wave of the same SIMD, or from the
// same wave?  The role-split persistent kernel (persist2_fwd.hip) and VERDICT
// r2's ping-pong proposal both rest on "yes".
//
// One block per CU, 512 threads: waves 0-3 (one per SIMD) run the MFMA loop,
// waves 4-7 the other loop (`kind`: 0 v_fma_f32, 1 v_exp_f32, 2 ds_read_b128 +
// one v_add per word, 3 global loads that hit L2 + adds, 4 / 5 the same loads
// with no vector ALU instruction at all).  Each wave stamps its own loop with
// s_memtime-based wall_clock64 (100 MHz); the host prints the slowest wave of
// each role for: MFMA alone, other alone, both at once, and (mode 3) the same
// wave alternating one MFMA with independent instructions of the other kind.
//
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
// result of this round: profiles/r03_mfma_valu_overlap.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ITERS = 4000;
constexpr int MFMA_PER_ITER = 4;     // independent accumulators: no dependent stalls
constexpr int OTHER_PER_ITER = 16;

template <int KIND>
__device__ __forceinline__ void other_body(float (&v)[OTHER_PER_ITER], float *lds,
                                           const float *g, int lane) {
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER; ++i)
            asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(1.0001f));
    } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER; ++i)
            asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (KIND == 2) {
        // (inline asm: a volatile read through the generic pointer compiles to
        // flat_load; `lds` is the kernel's only shared array, at LDS offset 0)
        f32x4 x[OTHER_PER_ITER / 4];
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i)
            asm volatile("ds_read_b128 %0, %1"
                         : "=v"(x[i]) : "v"(((lane + 64 * i) & 255) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i) {
            asm volatile("" : "+v"(x[i]));
            v[4 * i] += x[i][0]; v[4 * i + 1] += x[i][1];
            v[4 * i + 2] += x[i][2]; v[4 * i + 3] += x[i][3];
        }
    } else if (KIND == 4) {   // LDS reads only: no vector ALU instruction at all
        f32x4 x[OTHER_PER_ITER / 4];
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i)
            asm volatile("ds_read_b128 %0, %1"
                         : "=v"(x[i]) : "v"(((lane + 64 * i) & 255) * 16));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i) asm volatile("" : "+v"(x[i]));
    } else if (KIND == 5) {   // global loads only (L2 hits), no vector ALU instruction
        f32x4 x[OTHER_PER_ITER / 4];
        const float *p = g + lane * 4;
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2"
                         : "=v"(x[i]) : "v"(p), "n"(1024 * i));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i) asm volatile("" : "+v"(x[i]));
    } else {
#pragma unroll
        for (int i = 0; i < OTHER_PER_ITER / 4; ++i) {
            f32x4 x = __builtin_nontemporal_load((const f32x4 *)(g + ((lane + 64 * i) & 1023) * 4));
            v[4 * i] += x[0]; v[4 * i + 1] += x[1];
            v[4 * i + 2] += x[2]; v[4 * i + 3] += x[3];
        }
    }
}

// MT 0: v_mfma_f32_32x32x2_f32 (64 cycles), 1: v_mfma_f32_16x16x4_f32 (32 cycles,
// on the first four accumulator registers), 2: v_mfma_f32_32x32x16_bf16 (32 cycles)
template <int MT>
__device__ __forceinline__ void one_mfma(f32x16 &acc, float a, float b) {
    if (MT == 0) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else if (MT == 1) {
        f32x4 c = {acc[0], acc[1], acc[2], acc[3]};
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        acc[0] = c[0]; acc[1] = c[1]; acc[2] = c[2]; acc[3] = c[3];
    } else {
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (__bf16)a; bv[i] = (__bf16)b; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    }
}

template <int KIND, int MT>
__global__ __launch_bounds__(512) void overlap_kernel(int mode, float *sink,
                                                      const float *g,
                                                      unsigned long long *stamps) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) lds[i] = g[i];
    __syncthreads();
    const bool mfma_role = wave < 4;
    f32x16 acc[MFMA_PER_ITER];
    for (int i = 0; i < MFMA_PER_ITER; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float v[OTHER_PER_ITER];
    for (int i = 0; i < OTHER_PER_ITER; ++i) v[i] = 0.001f * (lane + i);
    const float a = 1.0f + lane * 1e-6f, b = 0.5f;
    const unsigned long long t0 = wall_clock64();
    if (mode == 3) {
        if (mfma_role)
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int i = 0; i < MFMA_PER_ITER; ++i)
                    one_mfma<MT>(acc[i], a, b);
                other_body<KIND>(v, lds, g, lane);
            }
    } else if (mfma_role) {
        if (mode == 0 || mode == 2 || mode == 4)
            for (int it = 0; it < ITERS; ++it) {
#pragma unroll
                for (int i = 0; i < MFMA_PER_ITER; ++i)
                    one_mfma<MT>(acc[i], a, b);
            }
    } else {
        if (mode == 4) __builtin_amdgcn_s_setprio(3);   // the other wave outranks the MFMA wave
        if (mode == 1 || mode == 2 || mode == 4)
            for (int it = 0; it < ITERS; ++it) other_body<KIND>(v, lds, g, lane);
    }
    const unsigned long long t1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < MFMA_PER_ITER; ++i) s += acc[i][0] + acc[i][7];
    for (int i = 0; i < OTHER_PER_ITER; ++i) s += v[i];
    if (s == 123.456f) sink[tid] = s;   // keep the work
    if (lane == 0) stamps[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int MT>
static void run(const char *name, float *sink, float *g, unsigned long long *stamps) {
    std::vector<unsigned long long> h(256 * 8);
    double r[5][2];
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {   // second launch is the warm one
            hipLaunchKernelGGL((overlap_kernel<KIND, MT>), dim3(256), dim3(512), 0, 0, mode,
                               sink, g, stamps);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long m = 0, o = 0;
        for (int b = 0; b < 256; ++b)
            for (int w = 0; w < 8; ++w)
                (w < 4 ? m : o) = std::max(w < 4 ? m : o, h[b * 8 + w]);
        r[mode][0] = m / 100.0; r[mode][1] = o / 100.0;   // us
    }
    printf("%-12s  mfma alone %7.1f us | other alone %7.1f us | both: mfma %7.1f "
           "other %7.1f | both, other at s_setprio 3: mfma %7.1f other %7.1f | same wave "
           "interleaved %7.1f  (sum of alones %7.1f)\n",
           name, r[0][0], r[1][1], r[2][0], r[2][1], r[4][0], r[4][1], r[3][0],
           r[0][0] + r[1][1]);
}

int main() {
    float *sink, *g;
    unsigned long long *stamps;
    hipMalloc(&sink, 512 * 4); hipMalloc(&g, 4096 * 4);
    hipMalloc(&stamps, 256 * 8 * 8);
    hipMemset(g, 0, 4096 * 4);
    printf("%d iterations of %d MFMAs and %d other instructions per iteration\n",
           ITERS, MFMA_PER_ITER, OTHER_PER_ITER);
    printf("-- v_mfma_f32_32x32x2_f32\n");
    run<0, 0>("v_fma_f32", sink, g, stamps);
    run<1, 0>("v_exp_f32", sink, g, stamps);
    run<2, 0>("ds_read_b128", sink, g, stamps);
    run<3, 0>("global_load", sink, g, stamps);
    run<4, 0>("ds_read only", sink, g, stamps);
    run<5, 0>("gload only", sink, g, stamps);
    printf("-- v_mfma_f32_16x16x4_f32\n");
    run<0, 1>("v_fma_f32", sink, g, stamps);
    run<1, 1>("v_exp_f32", sink, g, stamps);
    run<2, 1>("ds_read_b128", sink, g, stamps);
    printf("-- v_mfma_f32_32x32x16_bf16 (control)\n");
    run<0, 2>("v_fma_f32", sink, g, stamps);
    run<1, 2>("v_exp_f32", sink, g, stamps);
    run<2, 2>("ds_read_b128", sink, g, stamps);
    run<4, 2>("ds_read only", sink, g, stamps);
    run<5, 2>("gload only", sink, g, stamps);
    return 0;
}
