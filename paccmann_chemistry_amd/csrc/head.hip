// Decoder head and latent kernels: fused softmax / cross-entropy / argmax over
// every (step, row), reparameterisation, KL divergence, token choice for
// generation (greedy / temperature sampling) and the Adam update.
#include <mutex>

#include "common.h"

namespace {

int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// active rows per step: cnt[t] = #{b : lengths[b] > t}  (B when lengths == null)
__global__ __launch_bounds__(256) void step_count_kernel(const int32_t *lengths,
                                                         const float *step_weight,
                                                         int T, int B,
                                                         float *inv_cnt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    if (step_weight) {   // caller-supplied divisor (data parallel: global counts)
        inv_cnt[t] = step_weight[t];
        return;
    }
    int c = B;
    if (lengths) {
        c = 0;
        for (int b = 0; b < B; ++b) c += lengths[b] > t ? 1 : 0;
    }
    inv_cnt[t] = c > 0 ? 1.0f / (float)c : 0.0f;
}

// one wave per (t, b) row
__global__ __launch_bounds__(256) void softmax_ce_kernel(
    const float *logits, const int64_t *targets, const int32_t *lengths, int T,
    int B, int V, const float *inv_cnt, float *rowloss, int64_t *argmax,
    float *dlogits, int32_t *status) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long rows = (long long)T * B;
    for (long long row = (long long)blockIdx.x * 4 + wave; row < rows;
         row += (long long)gridDim.x * 4) {
        const int t = (int)(row / B), b = (int)(row - (long long)t * B);
        const bool active = !lengths || t < lengths[b];
        const float *x = logits + row * V;
        if (!active) {
            if (lane == 0) {
                rowloss[row] = 0.f;
                if (argmax) argmax[(long long)b * T + t] = 0;
            }
            if (dlogits)
                for (int v = lane; v < V; v += 64) dlogits[row * V + v] = 0.f;
            continue;
        }
        float m = -INFINITY;
        int mi = 0x7fffffff;
        for (int v = lane; v < V; v += 64) {
            const float xv = x[v];
            if (xv > m) { m = xv; mi = v; }  // first maximum within the lane
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(m, off, 64);
            const int oi = __shfl_xor(mi, off, 64);
            if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        float s = 0.f;
        for (int v = lane; v < V; v += 64) s += expf(x[v] - m);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        const float lse = m + logf(s);
        long long tg = targets[row];
        if (tg < 0 || tg >= V) {
            tg = tg < 0 ? 0 : V - 1;
            if (status && lane == 0) atomicOr(status, PCM_STATUS_BAD_TOKEN);
        }
        const float w = inv_cnt[t];
        if (lane == 0) {
            rowloss[row] = (lse - x[tg]) * w;
            if (argmax) argmax[(long long)b * T + t] = mi;
        }
        if (dlogits)
            for (int v = lane; v < V; v += 64)
                dlogits[row * V + v] =
                    (expf(x[v] - lse) - (v == (int)tg ? 1.f : 0.f)) * w;
    }
}

// ---- output layer fused with the cross entropy ----------------------------------
// logits = x . W^T + b on fp32 MFMAs and, without a round trip of the logits
// through memory, the row's softmax / cross entropy / first-maximum argmax and
// d loss / d logits (vae.py:342-348, 386-388).  One wave owns 64 rows (two 32-row
// MFMA tiles) x all V <= 32 NT columns over the whole K, so a logit is one
// in-order accumulation chain; operand fragments are 16-byte global loads (lane
// half lk holds k = 8 g + 4 lk + e of k group g for both operands), the weights
// (V x H, L2-resident) are shared by every wave.  The finished tile is parked in
// LDS: lane r reduces row r, then the wave writes logits and dlogits as one
// contiguous, lane-linear range.
struct OutputCeArgs {
    const float *x;          // [rows][ldx]
    long long ldx;
    const float *w;          // [V][H]
    const float *bias;       // [V]
    const int64_t *targets;  // [rows]
    const int32_t *lengths;  // [B] or null
    const float *inv_cnt;    // [T]
    float *logits;           // [rows][V] or null
    float *rowloss;          // [rows]
    int64_t *argmax;         // [B][T] or null
    float *dlogits;          // [rows][V] or null
    int32_t *status;
    long long rows;
    int T, B, V, H;
    float inv_v;
};

constexpr int OC_ROWS = 64;   // rows per wave
// operands may sit at any 4-byte boundary (parameters that live inside a flat
// optimizer buffer): global_load_dwordx4 takes dword-aligned addresses
typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

template <int NT>
__global__ __launch_bounds__(256) void output_ce_kernel(OutputCeArgs a) {
    constexpr int LD = NT * 32 + 1;
    extern __shared__ float oc_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    const long long row0 = ((long long)blockIdx.x * 4 + wave) * OC_ROWS;
    if (row0 >= a.rows) return;   // (no block-wide barrier below)
    float *tile = oc_lds + wave * (OC_ROWS * LD + 3 * OC_ROWS);
    float *lse_s = tile + OC_ROWS * LD;
    float *w_s = lse_s + OC_ROWS;
    int *tg_s = (int *)(w_s + OC_ROWS);

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float *xa[2], *wb[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long r = row0 + 32 * i + lrow;
        if (r > a.rows - 1) r = a.rows - 1;   // re-read the last row, masked below
        xa[i] = a.x + r * a.ldx + 4 * lk;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
        wb[j] = a.w + (long long)min(32 * j + lrow, a.V - 1) * a.H + 4 * lk;
    // fragments of the next k group are requested before this group's MFMAs
    // (the last iteration re-reads its own group)
    f32x4 fa[2], fb[NT];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[i] = *(const f32x4a4 *)(xa[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) fb[j] = *(const f32x4a4 *)(wb[j]);
    for (int kb = 0; kb < a.H; kb += 8) {
        const int kn = min(kb + 8, a.H - 8);
        f32x4 na[2], nb[NT];
#pragma unroll
        for (int i = 0; i < 2; ++i) na[i] = *(const f32x4a4 *)(xa[i] + kn);
#pragma unroll
        for (int j = 0; j < NT; ++j) nb[j] = *(const f32x4a4 *)(wb[j] + kn);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = na[i];
#pragma unroll
        for (int j = 0; j < NT; ++j) fb[j] = nb[j];
    }
    // park logits (+ bias) in LDS: lane holds column lrow of tile j, rows
    // (r & 3) + 8 (r >> 2) + 4 lk of tile i
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = 32 * j + lrow;
        const float bv = col < a.V ? a.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
                tile[row * LD + col] = acc[i][j][r] + bv;
            }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // lane r: softmax statistics of row r
    const long long row = row0 + lane;
    const int V = a.V;
    if (row < a.rows) {
        const int t = (int)(row / a.B), b = (int)(row - (long long)t * a.B);
        const bool active = !a.lengths || t < a.lengths[b];
        const float *x = tile + lane * LD;
        float m = -INFINITY;
        int mi = 0;
        for (int v = 0; v < V; ++v) {
            const float xv = x[v];
            if (xv > m) { m = xv; mi = v; }   // first maximum
        }
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += expf(x[v] - m);
        const float lse = m + logf(s);
        long long tg = a.targets[row];
        if (active && (tg < 0 || tg >= V)) {
            tg = tg < 0 ? 0 : V - 1;
            if (a.status) atomicOr(a.status, PCM_STATUS_BAD_TOKEN);
        }
        if (!active) tg = 0;
        const float w = active ? a.inv_cnt[t] : 0.f;
        a.rowloss[row] = active ? (lse - x[tg]) * w : 0.f;
        if (a.argmax) a.argmax[(long long)b * a.T + t] = active ? mi : 0;
        lse_s[lane] = lse; w_s[lane] = w; tg_s[lane] = active ? (int)tg : -1;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // logits / dlogits of the wave's rows: one contiguous range, lane-linear
    const long long left = a.rows - row0;
    const int nrows = left < OC_ROWS ? (int)left : OC_ROWS;
    const int n = nrows * V;
    float *lo = a.logits ? a.logits + row0 * V : nullptr;
    float *dl = a.dlogits ? a.dlogits + row0 * V : nullptr;
    for (int idx = lane; idx < n; idx += 64) {
        const int r = pcm_fdiv(idx, a.inv_v), c = idx - r * V;
        const float xv = tile[r * LD + c];
        if (lo) lo[idx] = xv;
        if (dl) {
            const int tg = tg_s[r];
            dl[idx] = tg < 0 ? 0.f
                             : (expf(xv - lse_s[r]) - (c == tg ? 1.f : 0.f)) *
                                   w_s[r];
        }
    }
}

// deterministic single-block sum of n floats, scaled
__global__ __launch_bounds__(1024) void sum_kernel(const float *x, long long n,
                                                   float scale, float *out) {
    __shared__ float red[1024];
    float s = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) s += x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

__global__ __launch_bounds__(256) void reparam_fwd_kernel(
    const float *mu, const float *logvar, const float *eps, float *z,
    long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (long long)gridDim.x * 256)
        z[i] = __fadd_rn(__fmul_rn(eps[i], expf(0.5f * logvar[i])), mu[i]);
}

__global__ __launch_bounds__(256) void reparam_bwd_kernel(
    const float *dz, const float *logvar, const float *eps, float *dmu,
    float *dlogvar, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (long long)gridDim.x * 256) {
        const float g = dz[i];
        if (dmu) dmu[i] = g;
        if (dlogvar) dlogvar[i] = g * eps[i] * 0.5f * expf(0.5f * logvar[i]);
    }
}

constexpr int KL_BLOCK = 1024;  // elements per partial

__global__ __launch_bounds__(256) void kl_partial_kernel(const float *mu,
                                                         const float *logvar,
                                                         long long n,
                                                         float *partial) {
    __shared__ float red[256];
    const long long base = (long long)blockIdx.x * KL_BLOCK;
    float s = 0.f;
    for (int i = threadIdx.x; i < KL_BLOCK; i += 256) {
        const long long e = base + i;
        if (e < n) {
            const float m = mu[e], lv = logvar[e];
            s += 1.f + lv - m * m - expf(lv);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float *dkl,
                                                     const float *mu,
                                                     const float *logvar,
                                                     float *dmu, float *dlogvar,
                                                     long long n) {
    const float g = dkl[0];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (long long)gridDim.x * 256) {
        if (dmu) dmu[i] = g * mu[i];
        if (dlogvar) dlogvar[i] = g * 0.5f * (expf(logvar[i]) - 1.f);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(
    float *param, const float *grad, float *m, float *v, long long n, float lr,
    float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
    float grad_scale) {
    const float step_size = lr / bc1;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (long long)gridDim.x * 256) {
        const float g = grad[i] * grad_scale;
        // torch: exp_avg.lerp_(grad, 1-beta1); exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
        const float mi = m[i] + (g - m[i]) * (1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * g * g;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param[i] -= step_size * (mi / denom);
    }
}

}  // namespace

// token choice for generation: one thread per row ----------------------------
__global__ __launch_bounds__(256) void choose_token_kernel(
    const float *logits, int B, int V, int search, float inv_temp,
    uint64_t seed, int step, int64_t *next_tok, int64_t *tokens_out,
    long long out_stride, int out_col) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float *x = logits + (long long)b * V;
    float m = -INFINITY;
    int mi = 0;
    for (int v = 0; v < V; ++v) {
        const float xv = x[v];
        if (xv > m) { m = xv; mi = v; }
    }
    int pick = mi;
    if (search == PCM_SEARCH_SAMPLING) {
        // softmax(logits / T), inverse-CDF draw
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += expf((x[v] - m) * inv_temp);
        const float u = pcm_uniform(seed, ((uint64_t)(uint32_t)step << 32) |
                                              (uint32_t)b) * s;
        float c = 0.f;
        pick = V - 1;
        for (int v = 0; v < V; ++v) {
            c += expf((x[v] - m) * inv_temp);
            if (c > u) { pick = v; break; }
        }
    }
    next_tok[b] = pick;
    tokens_out[(long long)b * out_stride + out_col] = pick;
}

// Output layer + token choice of one decode step in one launch (vae.py:455-464:
// output_layer, search.step): a block takes 16 batch rows; its 8 waves share
// the (column tile, K part) items of logits = h . Wout^T on fp32 MFMA with
// operands read straight from L2 (h was just written, Wout is 120 KB), the
// partial tiles meet in LDS, and the row's V logits are searched by 32 threads
// (greedy: the same token as choose_token_kernel; sampling: the same draw, the
// cumulative sums associated per thread and then across threads).  Replaces a 35 us generic GEMM
// + a 21 us one-thread-per-row scan (batch 4096) on the decode critical path.
struct LogitsChooseArgs {
    const float *h;        // [B, ldh] top-layer hidden
    const float *w;        // [V, H]
    const float *bias;     // [V] or null
    float *logits_out;     // [B, V] or null
    int64_t *next_tok;     // [B]
    int64_t *tokens_out;   // [B, out_stride]
    long long ldh, out_stride;
    int B, V, H, NT, ksplit, search, step, out_col;
    float inv_temp;
    uint64_t seed;
};
constexpr int LC_MAX_ITEMS = 8;
constexpr int LC_ROWS = 16;  // batch rows per block
constexpr int LC_LD = 17;    // padded row of a 16 x 16 tile in LDS

// 16 rows per block (256 blocks at 4096 rows; 32-row blocks left half the CUs
// idle and every wave with a chain of 128 dependent 32x32x2 MFMAs: 17.5 us per
// step), 8 waves = (16-column tile, K part) items on 16x16x4 MFMAs, then 32
// threads per row for the bias / arg-max (first maximum in vocabulary order).
__global__ __launch_bounds__(512) void logits_choose_kernel(LogitsChooseArgs a) {
    __shared__ float part[LC_MAX_ITEMS][LC_ROWS][LC_LD];
    __shared__ float rowmax[LC_ROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * LC_ROWS;
    const int items = a.NT * a.ksplit, klen = a.H / a.ksplit;
    const float *hrow = a.h + (long long)min(b0 + r16, a.B - 1) * a.ldh + 4 * g;
    for (int item = wave; item < items; item += 8) {
        const int tile = item % a.NT, kp = item / a.NT;
        const float *wrow =
            a.w + (long long)min(tile * 16 + r16, a.V - 1) * a.H + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int kend = (kp + 1) * klen;
        int k = kp * klen;
        // 8 fragment pairs (128 of K) in flight per round, then the remainder
        for (; k + 128 <= kend; k += 128) {
            f32x4 fa[8], fb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fa[i] = *(const f32x4 *)(hrow + k + 16 * i);
                fb[i] = *(const f32x4a4 *)(wrow + k + 16 * i);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][e], fb[i][e],
                                                               acc, 0, 0, 0);
        }
        for (; k < kend; k += 16) {
            const f32x4 fa = *(const f32x4 *)(hrow + k);
            const f32x4 fb = *(const f32x4a4 *)(wrow + k);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], acc, 0,
                                                           0, 0);
        }
        // D[i][j]: lane holds rows i = 4 g + e, column j = r16
#pragma unroll
        for (int e = 0; e < 4; ++e) part[item][4 * g + e][r16] = acc[e];
    }
    __syncthreads();
    // final logits (K parts summed in order, bias last as the GEMM does) land
    // in the first K part's tiles
    for (int idx = tid; idx < LC_ROWS * a.V; idx += 512) {
        const int r = idx / a.V, v = idx - r * a.V;
        const int tile = v >> 4, c = v & 15;
        float x = part[tile][r][c];
        for (int kp = 1; kp < a.ksplit; ++kp) x += part[kp * a.NT + tile][r][c];
        if (a.bias) x += a.bias[v];
        part[tile][r][c] = x;
        if (a.logits_out && b0 + r < a.B)
            a.logits_out[(long long)(b0 + r) * a.V + v] = x;
    }
    __syncthreads();
    auto lg = [&](int r, int v) -> float & { return part[v >> 4][r][v & 15]; };
    // arg-max: 32 threads per row, each over v = sub, sub + 32, ...; ties go to
    // the smaller index, as a scan in vocabulary order would leave them
    const int row = tid >> 5, sub = tid & 31;
    float m = -INFINITY;
    int pick = 0x7fffffff;
    for (int v = sub; v < a.V; v += 32) {
        const float xv = lg(row, v);
        if (xv > m) { m = xv; pick = v; }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {
        const float om = __shfl_xor(m, d);
        const int op = __shfl_xor(pick, d);
        if (om > m || (om == m && op < pick)) { m = om; pick = op; }
    }
    if (pick == 0x7fffffff) pick = 0;   // nothing compared greater (NaN row)
    if (sub == 0) rowmax[row] = m;
    const int b = b0 + row;
    if (a.search == PCM_SEARCH_SAMPLING) {
        // softmax(logits / T), inverse-CDF draw: the exponentials are taken by
        // all threads
        __syncthreads();
        for (int idx = tid; idx < LC_ROWS * a.V; idx += 512) {
            const int r = idx / a.V, v = idx - r * a.V;
            float &x = lg(r, v);
            x = expf((x - rowmax[r]) * a.inv_temp);
        }
        __syncthreads();
        // inverse CDF over the row's 32 threads: thread `sub` owns the nper
        // consecutive entries from sub * nper on; local running sums, an
        // exclusive scan of the thread totals, the first entry whose
        // cumulative sum exceeds u * total (one thread per row walking the
        // vocabulary twice took 9 of the launch's 19 us)
        const int nper = (a.V + 31) >> 5;        // <= 4 (V <= 128)
        float cum[4];
        float run = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = sub * nper + i;
            if (i < nper && v < a.V) run += lg(row, v);
            cum[i] = run;
        }
        float incl = run;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const float up = __shfl_up(incl, d, 32);
            if (sub >= d) incl += up;
        }
        const float total = __shfl(incl, 31, 32);
        float base = __shfl_up(incl, 1, 32);     // exclusive prefix
        if (sub == 0) base = 0.f;
        const float u =
            pcm_uniform(a.seed, ((uint64_t)(uint32_t)a.step << 32) |
                                    (uint32_t)b) * total;
        pick = 0x7fffffff;
#pragma unroll
        for (int i = 3; i >= 0; --i) {
            const int v = sub * nper + i;
            if (i < nper && v < a.V && base + cum[i] > u) pick = v;
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) pick = min(pick, __shfl_xor(pick, d));
        if (pick == 0x7fffffff) pick = a.V - 1;
    }
    if (sub != 0 || b >= a.B) return;
    a.next_tok[b] = pick;
    a.tokens_out[(long long)b * a.out_stride + a.out_col] = pick;
}

__global__ __launch_bounds__(256) void broadcast_token_kernel(
    const int64_t *src, int B, int64_t *next_tok, int64_t *tokens_out,
    long long out_stride, int out_col) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const int64_t t = src[0];
    if (next_tok) next_tok[b] = t;
    if (tokens_out) tokens_out[(long long)b * out_stride + out_col] = t;
}

int pcm_launch_choose(const float *logits, int B, int V, int search,
                      float temperature, uint64_t seed, int step,
                      int64_t *next_tok, int64_t *tokens_out,
                      long long out_stride, int out_col, hipStream_t stream) {
    hipLaunchKernelGGL(choose_token_kernel, dim3((B + 255) / 256), dim3(256), 0,
                       stream, logits, B, V, search, 1.0f / temperature, seed,
                       step, next_tok, tokens_out, out_stride, out_col);
    return 0;
}

// 1 = launched; 0 = shape not covered (caller falls back to GEMM + choose)
int pcm_launch_logits_choose(const float *h, long long ldh, const float *w,
                             const float *bias, int B, int V, int H, int search,
                             float temperature, uint64_t seed, int step,
                             int64_t *next_tok, int64_t *tokens_out,
                             long long out_stride, int out_col,
                             float *logits_out, hipStream_t stream) {
    const int NT = (V + 15) / 16;
    if (NT > LC_MAX_ITEMS || (H % 16) != 0 || (ldh % 4) != 0 ||
        ((uintptr_t)h & 15))
        return 0;
    LogitsChooseArgs a;
    a.h = h; a.w = w; a.bias = bias; a.logits_out = logits_out;
    a.next_tok = next_tok; a.tokens_out = tokens_out;
    a.ldh = ldh; a.out_stride = out_stride;
    a.B = B; a.V = V; a.H = H; a.NT = NT;
    a.ksplit = NT >= 8 ? 1 : 8 / NT;
    while (a.ksplit > 1 && (H % (16 * a.ksplit)) != 0) a.ksplit >>= 1;
    a.search = search; a.step = step; a.out_col = out_col;
    a.inv_temp = 1.0f / temperature; a.seed = seed;
    hipLaunchKernelGGL(logits_choose_kernel, dim3((B + LC_ROWS - 1) / LC_ROWS),
                       dim3(512), 0, stream, a);
    return 1;
}

int pcm_launch_broadcast_token(const int64_t *src, int B, int64_t *next_tok,
                               int64_t *tokens_out, long long out_stride,
                               int out_col, hipStream_t stream) {
    hipLaunchKernelGGL(broadcast_token_kernel, dim3((B + 255) / 256), dim3(256),
                       0, stream, src, B, next_tok, tokens_out, out_stride,
                       out_col);
    return 0;
}

// ------------------------------------------------------------------------------
// BeamSearch.step on the device (utils/search.py:169-201, 247-275)
// ------------------------------------------------------------------------------
// One block per batch row, one wave per state slot in use.  Wave k takes
// softmax(logits[k][b] / T) (search.py:262), its `top` most probable tokens in
// descending order (np.argsort(-p)[:top], search.py:189) and scores them as
// score_k + log(p + eps) -- p + eps in fp32 as the reference's tensor
// arithmetic does, the logarithm and the running score in double like its
// Python floats (search.py:193).  Wave 0 then keeps the nb_next best
// candidates in (score descending, candidate order) -- Python's stable
// sorted(..., reverse=True)[:beam_width] (search.py:197-201) -- and records
// token and parent slot of each survivor; the sequences themselves are
// rebuilt once at the end (beam_backtrack_kernel).  Like the reference the
// recurrent state of a slot is NOT moved to follow its beam's parent
// (vae.py:471-483: hidden / stack stay with their slot).
__global__ __launch_bounds__(1024) void beam_step_kernel(BeamStepArgs a) {
    extern __shared__ __attribute__((aligned(8))) unsigned char beam_sm[];
    const int ncand = a.used * a.top;
    double *cscore = (double *)beam_sm;                   // [ncand]
    int *ctok = (int *)(cscore + ncand);                  // [ncand]
    float *prob = (float *)(ctok + ncand);                // [used][V]
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int k = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (k < a.used) {
        const float *x = a.logits + ((long long)k * a.B + b) * a.V;
        float *pr = prob + (long long)k * a.V;
        float m = -INFINITY;
        for (int v = lane; v < a.V; v += 64) m = fmaxf(m, x[v] / a.temperature);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            m = fmaxf(m, __shfl_xor(m, off, 64));
        float sum = 0.f;
        for (int v = lane; v < a.V; v += 64) {
            const float e = expf(x[v] / a.temperature - m);
            pr[v] = e;
            sum += e;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        for (int v = lane; v < a.V; v += 64) pr[v] = pr[v] / sum;
        const double base = a.step == 0 ? 0.0 : a.scores[(long long)b * a.width + k];
        for (int r = 0; r < a.top; ++r) {
            // most probable token not taken yet; the lowest index wins a tie
            float best = -1.f;
            int bi = 0x7fffffff;
            for (int v = lane; v < a.V; v += 64)
                if (pr[v] > best) { best = pr[v]; bi = v; }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64);
                const int oi = __shfl_xor(bi, off, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) {
                const float q = best + 2.220446049250313e-16f;
                cscore[k * a.top + r] = base + log((double)q);
                ctok[k * a.top + r] = bi;
            }
            if ((bi & 63) == lane) pr[bi] = -1.f;   // taken
        }
    }
    __syncthreads();
    if (k != 0) return;
    for (int j = 0; j < a.nb_next; ++j) {
        double best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < ncand; c += 64) {
            const double sc = cscore[c];
            if (sc > best) { best = sc; bi = c; }   // first maximum of the lane
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) {
            const long long at = ((long long)a.step * a.B + b) * a.width + j;
            a.scores[(long long)b * a.width + j] = best;
            a.hist_tok[at] = ctok[bi];
            a.hist_parent[at] = bi / a.top;
            if (j < a.slots_next) a.next_tok[(long long)j * a.B + b] = ctok[bi];
        }
        if ((bi & 63) == lane) cscore[bi] = -INFINITY;
        __builtin_amdgcn_wave_barrier();
    }
}

// Rebuild every beam's token sequence from the per-step (token, parent slot)
// records; beams past `nb` (fewer candidates than beam_width) are filled with -1.
__global__ __launch_bounds__(256) void beam_backtrack_kernel(
    const int32_t *hist_tok, const int32_t *hist_parent, const double *scores,
    int B, int width, int nb, int G, int64_t *best_out, int64_t *beams_out,
    double *scores_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * width) return;
    const int b = idx / width, j = idx - b * width;
    if (scores_out)
        scores_out[idx] = j < nb ? scores[idx] : -INFINITY;
    int slot = j;
    for (int t = G - 1; t >= 0; --t) {
        long long tok = -1;
        if (j < nb) {
            const long long at = ((long long)t * B + b) * width + slot;
            tok = hist_tok[at];
            slot = hist_parent[at];
        }
        if (beams_out) beams_out[((long long)b * width + j) * G + t] = tok;
        if (j == 0 && best_out) best_out[(long long)b * G + t] = tok;
    }
}

int pcm_launch_beam_step(const BeamStepArgs &a, hipStream_t stream) {
    const size_t lds = (size_t)a.used * a.top * (sizeof(double) + sizeof(int)) +
                       (size_t)a.used * a.V * sizeof(float);
    PCM_REQUIRE(a.used >= 1 && a.used <= 16 && lds <= 64 * 1024, PCM_EINVAL,
                "beam step: %d slots x %d top tokens x vocab %d exceed the "
                "kernel's limits (16 slots, 64 KiB LDS)", a.used, a.top, a.V);
    hipLaunchKernelGGL(beam_step_kernel, dim3(a.B), dim3(64 * a.used), lds,
                       stream, a);
    return 0;
}

int pcm_launch_beam_backtrack(const int32_t *hist_tok,
                              const int32_t *hist_parent, const double *scores,
                              int B, int width, int nb, int G, int64_t *best_out,
                              int64_t *beams_out, double *scores_out,
                              hipStream_t stream) {
    hipLaunchKernelGGL(beam_backtrack_kernel, dim3((B * width + 255) / 256),
                       dim3(256), 0, stream, hist_tok, hist_parent, scores, B,
                       width, nb, G, best_out, beams_out, scores_out);
    return 0;
}

// encoder / decoder / target sequences of one batch (see pcm_prepare_batch)
__global__ __launch_bounds__(256) void prepare_batch_kernel(
    const int64_t *flat, const int64_t *offsets, int B, int T,
    const int64_t *step_base, float input_keep, uint64_t seed,
    const uint8_t *keep_mask, int64_t dropout_index, int64_t *enc, int64_t *dec,
    int64_t *tgt) {
    const long long total = (long long)T * B;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int t = (int)(idx / B), b = (int)(idx - (long long)t * B);
        const long long o = offsets[b];
        const int len = (int)(offsets[b + 1] - o);
        const bool live = t < len;
        if (step_base && !live) continue;   // packed data holds live tokens only
        const int64_t tok = live ? flat[o + t] : 0;
        const int64_t nxt = t + 1 < len ? flat[o + t + 1] : 0;
        // dropout unit: the token (packed) or the whole time row (padded)
        const uint64_t unit = step_base ? (uint64_t)(o + t) : (uint64_t)t;
        bool keep = true;
        if (keep_mask) keep = keep_mask[unit] != 0;
        else if (input_keep < 1.f) keep = pcm_uniform(seed, unit) < input_keep;
        const long long at = step_base ? step_base[t] + b : idx;
        enc[at] = tok;
        dec[at] = keep ? tok : dropout_index;
        tgt[at] = nxt;
    }
}

// Shared body of pcm_softmax_ce / pcm_softmax_ce_static.  scratch: T*B row
// losses followed by T reciprocal counts; every word the kernels read is written
// by this call first, so the buffer needs no clearing between calls.
static int softmax_ce_enqueue(const float *logits, const int64_t *targets,
                              const int32_t *lengths, const float *step_weight,
                              int32_t steps, int32_t batch, int32_t vocab,
                              float *loss, int64_t *argmax, float *dlogits,
                              float *scratch, int32_t *status, hipStream_t st) {
    const long long rows = (long long)steps * batch;
    float *rowloss = scratch;
    float *inv_cnt = scratch + rows;
    hipLaunchKernelGGL(step_count_kernel, dim3((steps + 255) / 256), dim3(256),
                       0, st, lengths, step_weight, steps, batch, inv_cnt);
    const int blocks = (int)((rows + 3) / 4 > 8192 ? 8192 : (rows + 3) / 4);
    hipLaunchKernelGGL(softmax_ce_kernel, dim3(blocks), dim3(256), 0, st, logits,
                       targets, lengths, steps, batch, vocab, inv_cnt, rowloss,
                       argmax, dlogits, status);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, rowloss, rows,
                       1.0f, loss);
    return pcm_check_launch("softmax_ce");
}

// ---- scratch of the entry point that allocates nothing -----------------------------
// One array per GPU, part of the code object (each device the library runs on
// holds its own copy until the process ends).  In host code the array's NAME is
// only a handle: the device address is looked up once per device with
// hipGetSymbolAddress and only that pointer is ever handed to a kernel.
static __device__ __align__(256) float g_ce_scratch[PCM_CE_STATIC_MAX_FLOATS];
constexpr int CE_MAX_DEVICES = 64;
static float *g_ce_scratch_ptr[CE_MAX_DEVICES] = {};
static std::mutex g_ce_scratch_mutex;

// device address of the CURRENT device's copy (looked up on first use there)
static int ce_static_scratch(float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "softmax_ce_static: %s",
                hipGetErrorString(e));
    PCM_REQUIRE(dev >= 0 && dev < CE_MAX_DEVICES, PCM_EDEVICE,
                "softmax_ce_static: device %d outside 0..%d", dev,
                CE_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_ce_scratch_mutex);
    if (!g_ce_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_ce_scratch));
        PCM_REQUIRE(e == hipSuccess && p, PCM_EDEVICE,
                    "softmax_ce_static: hipGetSymbolAddress: %s",
                    hipGetErrorString(e));
        g_ce_scratch_ptr[dev] = (float *)p;
    }
    *out = g_ce_scratch_ptr[dev];
    return PCM_OK;
}

// called by pcm_device_check(): resolve the current device's address eagerly,
// outside any stream capture
int pcm_ce_static_prepare(void) {
    float *unused = nullptr;
    return ce_static_scratch(&unused);
}

extern "C" {

int pcm_softmax_ce(const float *logits, const int64_t *targets,
                   const int32_t *lengths, const float *step_weight,
                   int32_t steps, int32_t batch,
                   int32_t vocab, float *loss, int64_t *argmax, float *dlogits,
                   float *scratch, int32_t *status, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(logits && targets && loss && scratch, PCM_EINVAL,
                "softmax_ce: null pointer");
    PCM_REQUIRE(steps > 0 && batch > 0 && vocab > 0, PCM_EINVAL,
                "softmax_ce: bad shape T=%d B=%d V=%d", steps, batch, vocab);
    return softmax_ce_enqueue(logits, targets, lengths, step_weight, steps, batch,
                              vocab, loss, argmax, dlogits, scratch, status,
                              (hipStream_t)stream);
}

int pcm_output_ce(const float *x, int64_t ldx, const float *w,
                  const float *bias, const int64_t *targets,
                  const int32_t *lengths, const float *step_weight,
                  int32_t steps, int32_t batch, int32_t vocab, int32_t hidden,
                  float *logits, float *loss, int64_t *argmax, float *dlogits,
                  float *scratch, int32_t *status, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(x && w && bias && targets && loss && scratch, PCM_EINVAL,
                "output_ce: null pointer");
    PCM_REQUIRE(steps > 0 && batch > 0 && vocab > 0 && hidden > 0, PCM_EINVAL,
                "output_ce: bad shape T=%d B=%d V=%d H=%d", steps, batch, vocab,
                hidden);
    PCM_REQUIRE(ldx >= hidden, PCM_EINVAL, "output_ce: ldx %lld < H %d",
                (long long)ldx, hidden);
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)steps * batch;
    const bool fused = vocab <= 128 && hidden % 8 == 0;
    if (!fused) {
        // shapes outside the fused kernel: output layer, then the cross entropy
        PCM_REQUIRE(logits, PCM_EINVAL,
                    "output_ce: V=%d H=%d ldx=%lld needs a logits buffer (the "
                    "fused kernel takes V <= 128 and H %% 8 == 0)", vocab, hidden,
                    (long long)ldx);
        PCM_REQUIRE(rows <= 0x7fffffff, PCM_EINVAL, "output_ce: too many rows");
        int rc = pcm_linear_forward(x, ldx, w, bias, logits, vocab, (int32_t)rows,
                                    vocab, hidden, stream);
        if (rc) return rc;
        return softmax_ce_enqueue(logits, targets, lengths, step_weight, steps,
                                  batch, vocab, loss, argmax, dlogits, scratch,
                                  status, st);
    }
    float *rowloss = scratch;
    float *inv_cnt = scratch + rows;
    hipLaunchKernelGGL(step_count_kernel, dim3((steps + 255) / 256), dim3(256),
                       0, st, lengths, step_weight, steps, batch, inv_cnt);
    OutputCeArgs a;
    a.x = x; a.ldx = ldx; a.w = w; a.bias = bias; a.targets = targets;
    a.lengths = lengths; a.inv_cnt = inv_cnt; a.logits = logits;
    a.rowloss = rowloss; a.argmax = argmax; a.dlogits = dlogits;
    a.status = status; a.rows = rows; a.T = steps; a.B = batch; a.V = vocab;
    a.H = hidden; a.inv_v = 1.0f / (float)vocab;
    const int nt = (vocab + 31) / 32;
    const unsigned blocks = (unsigned)((rows + 4 * OC_ROWS - 1) / (4 * OC_ROWS));
    const size_t lds = 4 * (size_t)(OC_ROWS * (nt * 32 + 1) + 3 * OC_ROWS) *
                       sizeof(float);
    switch (nt) {
    case 1:
        hipLaunchKernelGGL(output_ce_kernel<1>, dim3(blocks), dim3(256), lds, st, a);
        break;
    case 2:
        hipLaunchKernelGGL(output_ce_kernel<2>, dim3(blocks), dim3(256), lds, st, a);
        break;
    case 3:
        hipLaunchKernelGGL(output_ce_kernel<3>, dim3(blocks), dim3(256), lds, st, a);
        break;
    default:
        hipLaunchKernelGGL(output_ce_kernel<4>, dim3(blocks), dim3(256), lds, st, a);
        break;
    }
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, rowloss, rows,
                       1.0f, loss);
    return pcm_check_launch("output_ce");
}

int pcm_softmax_ce_static(const float *logits, const int64_t *targets,
                          const int32_t *lengths, const float *step_weight,
                          int32_t steps, int32_t batch, int32_t vocab,
                          float *loss, int64_t *argmax, float *dlogits,
                          int32_t *status, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(logits && targets && loss, PCM_EINVAL,
                "softmax_ce_static: null pointer");
    PCM_REQUIRE(steps > 0 && batch > 0 && vocab > 0, PCM_EINVAL,
                "softmax_ce_static: bad shape T=%d B=%d V=%d", steps, batch,
                vocab);
    const long long need = (long long)steps * batch + steps;
    PCM_REQUIRE(need <= (long long)PCM_CE_STATIC_MAX_FLOATS, PCM_EINVAL,
                "softmax_ce_static: T*B + T = %lld floats of scratch exceed the "
                "static maximum of %d (use pcm_softmax_ce with a caller "
                "workspace)", need, (int)PCM_CE_STATIC_MAX_FLOATS);
    float *scratch = nullptr;
    int rc = ce_static_scratch(&scratch);
    if (rc) return rc;
    return softmax_ce_enqueue(logits, targets, lengths, step_weight, steps, batch,
                              vocab, loss, argmax, dlogits, scratch, status,
                              (hipStream_t)stream);
}

int pcm_search_step(const float *logits, int64_t rows, int32_t vocab,
                    int32_t search, float temperature, uint64_t seed,
                    int32_t step, int64_t *tokens_out, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(logits && tokens_out && rows > 0 && vocab > 0, PCM_EINVAL,
                "search_step: bad args");
    PCM_REQUIRE(rows < (1ll << 31), PCM_EINVAL, "search_step: too many rows");
    PCM_REQUIRE(search == PCM_SEARCH_GREEDY || search == PCM_SEARCH_SAMPLING,
                PCM_EINVAL, "search_step: unknown search %d", search);
    PCM_REQUIRE(search != PCM_SEARCH_SAMPLING || temperature > 0.f, PCM_EINVAL,
                "search_step: temperature must be positive");
    pcm_launch_choose(logits, (int)rows, vocab, search, temperature, seed, step,
                      tokens_out, tokens_out, 1, 0, (hipStream_t)stream);
    return pcm_check_launch("search_step");
}

int pcm_reparam_forward(const float *mu, const float *logvar, const float *eps,
                        float *z, int64_t n, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(mu && logvar && eps && z && n > 0, PCM_EINVAL,
                "reparam_forward: bad args");
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, mu, logvar, eps, z, (long long)n);
    return pcm_check_launch("reparam_forward");
}

int pcm_reparam_backward(const float *dz, const float *logvar,
                         const float *eps, float *dmu, float *dlogvar,
                         int64_t n, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(dz && logvar && eps && n > 0, PCM_EINVAL,
                "reparam_backward: bad args");
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, dz, logvar, eps, dmu, dlogvar,
                       (long long)n);
    return pcm_check_launch("reparam_backward");
}

size_t pcm_kl_scratch_floats(int64_t n) {
    return (size_t)((n + KL_BLOCK - 1) / KL_BLOCK);
}

int pcm_kl_forward(const float *mu, const float *logvar, float *kl,
                   float *scratch, int64_t n, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(mu && logvar && kl && scratch && n > 0, PCM_EINVAL,
                "kl_forward: bad args");
    const int parts = (int)pcm_kl_scratch_floats(n);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(kl_partial_kernel, dim3(parts), dim3(256), 0, st, mu,
                       logvar, (long long)n, scratch);
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, st, scratch,
                       (long long)parts, -0.5f, kl);
    return pcm_check_launch("kl_forward");
}

int pcm_kl_backward(const float *dkl, const float *mu, const float *logvar,
                    float *dmu, float *dlogvar, int64_t n, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(dkl && mu && logvar && n > 0, PCM_EINVAL,
                "kl_backward: bad args");
    hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, dkl, mu, logvar, dmu, dlogvar,
                       (long long)n);
    return pcm_check_launch("kl_backward");
}

int pcm_prepare_batch(const int64_t *flat, const int64_t *offsets,
                      int32_t batch, int32_t steps, int32_t packed,
                      const int64_t *step_base, float input_keep,
                      uint64_t seed, const uint8_t *keep_mask,
                      int64_t dropout_index, int64_t *enc, int64_t *dec,
                      int64_t *tgt, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(flat && offsets && enc && dec && tgt, PCM_EINVAL,
                "prepare_batch: null pointer");
    PCM_REQUIRE(batch > 0 && steps > 0, PCM_EINVAL,
                "prepare_batch: bad shape B=%d T=%d", batch, steps);
    PCM_REQUIRE(!packed || step_base, PCM_EINVAL,
                "prepare_batch: packed layout needs step_base");
    const long long total = (long long)steps * batch;
    hipLaunchKernelGGL(prepare_batch_kernel, dim3(grid_for(total)), dim3(256), 0,
                       (hipStream_t)stream, flat, offsets, batch, steps,
                       packed ? step_base : nullptr, input_keep, seed, keep_mask,
                       dropout_index, enc, dec, tgt);
    return pcm_check_launch("prepare_batch");
}

int pcm_adam_step(float *param, const float *grad, float *exp_avg,
                  float *exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t step, float grad_scale,
                  void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1,
                PCM_EINVAL, "adam_step: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0,
                       (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    return pcm_check_launch("adam_step");
}

}  // extern "C"
