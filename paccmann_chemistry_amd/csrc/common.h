// Shared declarations of libpaccmann_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/paccmann_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define PCM_WAVE 64

// ---- one host thread at a time inside the library ------------------------------
// Every entry point that enqueues work takes this lock for the time it spends on
// the HOST (microseconds: the kernels run asynchronously on the caller's
// stream): the launch-duration probe, the mixed-precision switch, the per-device
// side streams and the lazily created kernel attributes are process-wide state,
// and callers do enter from two threads (autograd runs backward on its own
// device thread).  Recursive: entry points call each other.
#include <mutex>
std::recursive_mutex &pcm_api_mutex();
#define PCM_API_GUARD \
    std::lock_guard<std::recursive_mutex> pcm_api_guard_(pcm_api_mutex())

// ---- error plumbing -----------------------------------------------------------
void pcm_set_error(const char *fmt, ...);
int pcm_check_launch(const char *what);

#define PCM_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            pcm_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)

static inline int64_t pcm_round_up(int64_t v, int64_t m) {
    return (v + m - 1) / m * m;
}

// ---- generic grouped fp32 MFMA GEMM (gemm.hip) ------------------------------
// C[m][n] (+)= alpha * sum_k A[m][k] * B[k][n] + bias[n]
// A[m][k] at A + m*a_rs + k*a_cs; B[k][n] at B + k*b_rs + n*b_cs.
// Optional second K segment: for k >= K1 the operands come from A2 / B2 at
// index k-K1 (K1 must be a multiple of 16); K1 == 0 disables it.
struct GemmProb {
    const float *A;
    const float *B;
    const float *A2;
    const float *B2;
    float *C;
    const float *bias;
    const float *alpha_ptr;  // device scalar (nullable -> 1)
    // optional by-product of the K-major kernel (pcm_gemm_takes_colsum):
    // colsum1/2[m] (+)= sum_k A[k][m], following `accumulate`
    float *colsum1, *colsum2;
    int cs_accumulate;       // -1: as `accumulate`
    long long a_rs, a_cs, b_rs, b_cs, ldc;
    int M, N, K, K1;
    int accumulate;
};

#define PCM_GEMM_MAX_PROBS 8
struct GemmBatch {
    GemmProb p[PCM_GEMM_MAX_PROBS];
    int count;
};

// Rows of a time-major [steps][batch] reduction that are known to be zero: in
// packed mode step t only has its first active rows filled, the rest of its
// batch block holds zeros.  cum[t] = number of 16-row chunks with at least one
// active row in steps [0, t); chunk j of step t is active iff j < cum[t+1] -
// cum[t].  The K-major kernel then walks the active chunks only (every
// problem of the batch must have K == steps * cps * 16).
#define PCM_SKIP_MAX_STEPS 256
struct GemmRowSkip {
    int steps;
    int cps;                                   // chunks per step (batch / 16)
    unsigned short cum[PCM_SKIP_MAX_STEPS + 1];
};
// false when the shape does not fit the descriptor (the caller then passes none)
bool pcm_gemm_row_skip(const int32_t *active_rows, int steps, int batch,
                       GemmRowSkip *out);

// Bytes of split-K slab scratch pcm_gemm_launch may need for this batch.
size_t pcm_gemm_scratch_bytes(const GemmBatch &batch);
// Launch a grouped GEMM.  scratch may be NULL (then no split-K is used).
// skip (nullable) is honoured by the K-major kernel only; the generic kernels
// multiply the zero rows.
int pcm_gemm_launch(const GemmBatch &batch, void *scratch,
                    size_t scratch_bytes, hipStream_t stream,
                    const GemmRowSkip *skip = nullptr);

// true if pcm_gemm_launch will run this batch on the K-major kernel, the only
// one that honours GemmProb::colsum1/2
bool pcm_gemm_takes_colsum(const GemmBatch &batch, bool have_scratch);

extern int g_pcm_gemm_lds_pad;

static inline GemmProb pcm_gemm_prob(const float *A, long long a_rs,
                                     long long a_cs, const float *B,
                                     long long b_rs, long long b_cs, float *C,
                                     long long ldc, int M, int N, int K) {
    GemmProb p;
    p.A = A; p.B = B; p.A2 = nullptr; p.B2 = nullptr; p.C = C;
    p.bias = nullptr; p.alpha_ptr = nullptr;
    p.colsum1 = nullptr; p.colsum2 = nullptr; p.cs_accumulate = -1;
    p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.K1 = 0; p.accumulate = 0;
    return p;
}

// column sums: out[n] (+)= alpha * sum_m X[m*ldx + n]  (deterministic, 2 stage)
size_t pcm_colsum_scratch_bytes(int M, int N);
int pcm_colsum_launch(const float *X, long long ldx, int M, int N, float *out,
                      const float *alpha_ptr, int accumulate, void *scratch,
                      size_t scratch_bytes, hipStream_t stream);

// ---- launch-duration probe (bench.py roofline) ---------------------------------
// When enabled for one kernel id, the launches of that kernel go through
// hipExtLaunchKernelGGL with a (start, stop) event pair bound to the kernel's own
// dispatch: the elapsed time is the kernel's begin -> end on the device (the
// timestamps a rocprofv3 kernel trace reports), with no marker packets on the
// stream.  pcm_probe_read() sums the elapsed times.
#define PCM_K_CELL_FWD 1     // 11: layer 0 only, 12: layers >= 1 only
#define PCM_K_STACK_FWD 2
#define PCM_K_GATE_BWD 3
#define PCM_K_STACK_BWD 4
#define PCM_K_CELL_BWD 5     // 51: layer 0 only, 52: layers >= 1 only
#define PCM_K_DW_GEMM 6      // K-major weight-gradient GEMM (main kernel)
extern int g_pcm_probe_id;
extern hipEvent_t g_pcm_probe_start, g_pcm_probe_stop;  // set while a probed
                                                        // launch is enqueued
void pcm_probe_begin_(void);
void pcm_probe_end_(void);
static inline bool pcm_probe_match(int id, int layer) {
    return g_pcm_probe_id == id ||
           (layer >= 0 && g_pcm_probe_id == id * 10 + (layer == 0 ? 1 : 2));
}
static inline void pcm_probe_begin(int id, int layer = -1) {
    if (g_pcm_probe_id && pcm_probe_match(id, layer)) pcm_probe_begin_();
}
static inline void pcm_probe_end(int id, int layer = -1) {
    if (g_pcm_probe_id && pcm_probe_match(id, layer)) pcm_probe_end_();
}
// every kernel a probe can name is launched through this
#define PCM_LAUNCH(kernel, grid, block, shmem, stream, ...)                     \
    do {                                                                        \
        if (g_pcm_probe_start)                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream,           \
                                  g_pcm_probe_start, g_pcm_probe_stop, 0,       \
                                  __VA_ARGS__);                                 \
        else                                                                    \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__); \
    } while (0)

// ---- beam search step (head.hip) -------------------------------------------------
struct BeamStepArgs {
    const float *logits;   // [slots][B][V]: row k * B + b
    double *scores;        // [B][width] running scores, in / out
    int32_t *hist_tok;     // [G][B][width] token of each surviving beam
    int32_t *hist_parent;  // [G][B][width] slot its sequence continues
    int64_t *next_tok;     // [width * B] next input token of slot k: k * B + b
    int B, V, width;
    int used;        // beams expanded at this step (<= slots, <= beams alive)
    int top;         // top tokens kept per beam (<= V)
    int nb_next;     // beams alive after the step = min(width, used * top)
    int slots_next;  // state slots fed at the next step
    int step;
    float temperature;
};
int pcm_launch_beam_step(const BeamStepArgs &a, hipStream_t stream);
int pcm_launch_beam_backtrack(const int32_t *hist_tok,
                              const int32_t *hist_parent, const double *scores,
                              int B, int width, int nb, int G, int64_t *best_out,
                              int64_t *beams_out, double *scores_out,
                              hipStream_t stream);

// ---- mixed-precision mode (pcm_set_matmul_precision) ------------------------------
// 0: every product on fp32 MFMAs (the default and the only mode parity is claimed
// on); 1: the recurrent cells' products (forward and reverse) and the weight-
// gradient GEMM round their operands to bf16 in registers and run on
// v_mfma_f32_32x32x16_bf16 / 16x16x32_bf16 with fp32 accumulation; state, gates,
// stack, losses and the optimizer stay fp32.
extern int g_pcm_matmul_bf16;

// ---- device helpers -------------------------------------------------------------
// eight fp32 values -> one bf16 MFMA operand fragment (v_cvt_pk_bf16_f32, RNE)
__device__ __forceinline__ bf16x8 pcm_pack_bf16(f32x4 a, f32x4 b) {
    f32x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_convertvector(v, bf16x8);
}
// n / d for 0 <= n < 2^20 through a float reciprocal (3 VALU instead of ~35):
// n + 0.5 is never within 0.5 of a multiple of d, far above fp32 rounding.
__device__ __forceinline__ int pcm_fdiv(int n, float inv_d) {
    return (int)(((float)n + 0.5f) * inv_d);
}
__device__ __forceinline__ float pcm_sigmoid(float x) {
    return 1.0f / (1.0f + expf(-x));
}

// Counter-based uniform in [0,1): splitmix64 finaliser of (seed, counter).
__host__ __device__ __forceinline__ uint64_t pcm_mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ float pcm_uniform(uint64_t seed,
                                                      uint64_t counter) {
    uint64_t h = pcm_mix64(seed ^ pcm_mix64(counter));
    return (float)(h >> 40) * (1.0f / 16777216.0f);
}
// Inter-layer dropout multiplier for element (layer, t, b, j).
__device__ __forceinline__ float pcm_dropout_scale(uint64_t seed, int layer,
                                                   int t, int b, int j,
                                                   float p, float inv_keep) {
    uint64_t ctr = ((uint64_t)(uint32_t)layer << 56) ^
                   ((uint64_t)(uint32_t)t << 40) ^
                   ((uint64_t)(uint32_t)b << 16) ^ (uint64_t)(uint32_t)j;
    return pcm_uniform(seed, ctr) < p ? 0.0f : inv_keep;
}
