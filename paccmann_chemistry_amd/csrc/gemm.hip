// Generic grouped fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain
// numerics) with arbitrary operand strides, an optional second K segment,
// deterministic split-K (slabs + ordered reduce), bias / alpha epilogue.
//
// Used for every time-parallel product of the path: token gate table,
// logits, mu/logvar, latent->hidden, all weight gradients (sum over T*B rows)
// and, in the reverse recurrence, the per-step dgates.W products.
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

#include "common.h"

namespace {

constexpr int BK = 16;

struct SlabInfo {
    long long off[PCM_GEMM_MAX_PROBS];  // float offset of each problem's slabs
    long long cs[PCM_GEMM_MAX_PROBS];   // ... of its column-sum partials [splits][M]
};

// Work-group tile = (64*TM) x (64*TN); 4 waves as 2x2, each TM x TN MFMA tiles.
// One tile of problem pz, K split sk of `splits`, at (m0, n0).
template <int TM, int TN>
__device__ __forceinline__ void gemm_tile(const GemmBatch &batch, int pz, int sk,
                                          int m0, int n0, int splits,
                                          float *slab, const SlabInfo &slabs,
                                          float *As_, float *Bs_) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int NA = 4 * TM, NB = 4 * TN;  // elements per thread per chunk
    float (*As)[BK * LDA] = (float (*)[BK * LDA])As_;
    float (*Bs)[BK * LDB] = (float (*)[BK * LDB])Bs_;
    const GemmProb &p = batch.p[pz];

    const int nchunks = (p.K + BK - 1) / BK;
    const int cps = (nchunks + splits - 1) / splits;
    const int c0 = sk * cps;
    const int c1 = min(nchunks, c0 + cps);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const bool a_mfast = (p.a_rs == 1 && p.a_cs != 1);
    const bool b_nfast = (p.b_cs == 1);

    float ra[NA], rb[NB];

    auto load_chunk = [&](int c) {
        const int kb = c * BK;
        const bool seg2 = (p.K1 > 0 && kb >= p.K1);
        const float *Ab = seg2 ? p.A2 : p.A;
        const float *Bb = seg2 ? p.B2 : p.B;
        const int kshift = seg2 ? p.K1 : 0;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            int m, k;
            if (a_mfast) { m = e % BM; k = e / BM; }
            else { k = e % BK; m = e / BK; }
            const int gm = m0 + m, gk = kb + k;
            float v = 0.0f;
            if (gm < p.M && gk < p.K)
                v = Ab[(long long)gm * p.a_rs + (long long)(gk - kshift) * p.a_cs];
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + 256 * i;
            int n, k;
            if (b_nfast) { n = e % BN; k = e / BN; }
            else { k = e % BK; n = e / BK; }
            const int gn = n0 + n, gk = kb + k;
            float v = 0.0f;
            if (gn < p.N && gk < p.K)
                v = Bb[(long long)(gk - kshift) * p.b_rs + (long long)gn * p.b_cs];
            rb[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int e = tid + 256 * i;
            int m, k;
            if (a_mfast) { m = e % BM; k = e / BM; }
            else { k = e % BK; m = e / BK; }
            As[buf][k * LDA + m] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int e = tid + 256 * i;
            int n, k;
            if (b_nfast) { n = e % BN; k = e / BN; }
            else { k = e % BK; n = e / BK; }
            Bs[buf][k * LDB + n] = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int lrow = lane & 31, lk = lane >> 5;
    if (c0 < c1) load_chunk(c0);
    int buf = 0;
    for (int c = c0; c < c1; ++c) {
        store_chunk(buf);
        __syncthreads();
        if (c + 1 < c1) load_chunk(c + 1);
        const float *as = &As[buf][0];
        const float *bs = &Bs[buf][0];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = as[(kk + lk) * LDA + (wm * TM + i) * 32 + lrow];
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = bs[(kk + lk) * LDB + (wn * TN + j) * 32 + lrow];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }

    // epilogue: lane holds column (lane&31), rows (r&3)+8*(r>>2)+4*(lane>>5)
    const float alpha = p.alpha_ptr ? *p.alpha_ptr : 1.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int gn = n0 + (wn * TN + j) * 32 + lrow;
            if (gn >= p.N) continue;
            const float bias = (p.bias && splits == 1) ? p.bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) +
                               4 * lk;
                if (gm >= p.M) continue;
                if (splits == 1) {
                    float *c = p.C + (long long)gm * p.ldc + gn;
                    float v = alpha * acc[i][j][r] + bias;
                    if (p.accumulate) v += *c;
                    *c = v;
                } else {
                    slab[slabs.off[pz] +
                         ((long long)sk * p.M + gm) * p.N + gn] = acc[i][j][r];
                }
            }
        }
    }
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmBatch batch, int splits,
                                                   float *slab,
                                                   SlabInfo slabs) {
    constexpr int BM = 64 * TM, BN = 64 * TN;
    __shared__ float As[2 * BK * (BM + 1)];
    __shared__ float Bs[2 * BK * (BN + 1)];
    const int pz = blockIdx.z / splits;
    const int sk = blockIdx.z - pz * splits;
    const GemmProb &p = batch.p[pz];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= p.M || n0 >= p.N) return;
    gemm_tile<TM, TN>(batch, pz, sk, m0, n0, splits, slab, slabs, As, Bs);
}

// ---- K-major x K-major products (weight gradients) --------------------------------
// C[M,N] (+)= sum_r A[r][m] * B[r][n] with both operands contiguous along their
// output index (a_rs == 1, b_cs == 1), M and N multiples of 4, K of 16 (tiles
// that hang over M or N re-read the last valid columns and mask the store):
// 16-byte global loads straight into a [k][m] LDS layout whose row stride (160
// floats) puts the two k rows a wave reads on disjoint banks; register prefetch
// two chunks ahead, LDS double buffer, one barrier per chunk.  Blocks are dealt
// so that an XCD (blockIdx % 8) works on its own K splits across all tiles: every
// operand byte is fetched from HBM by one XCD only and the tiles marching
// through the same K range share it in that XCD's L2.
constexpr int TN_LD = 160;
struct TileMap {
    int first[PCM_GEMM_MAX_PROBS + 1];  // first tile of each problem
    int tiles_n[PCM_GEMM_MAX_PROBS];
};

// FIXED_MODE 0: every problem of the batch is wide (the wave mapping folds at
// compile time: fewer registers, 3 waves per SIMD); -1: chosen per problem.
// BF16: operands rounded to bf16 in registers, one v_mfma_f32_32x32x16_bf16 per
// sub-tile and chunk (lane half lk holds k = lk, lk + 2, ... of the chunk for
// both operands); the column sums stay fp32 sums of the unrounded values.
// SKIP: the reduction walks the active 16-row chunks of a GemmRowSkip only.
struct NoSkip {};
template <int FIXED_MODE, bool BF16 = false, bool SKIP = false>
__global__ __launch_bounds__(256, FIXED_MODE == 0 ? 3 : 2) void gemm_tn_kernel(GemmBatch batch,
                                                         int splits, float *slab,
                                                         SlabInfo slabs,
                                                         TileMap tm,
                                                         std::conditional_t<SKIP, GemmRowSkip, NoSkip> rs) {
    __shared__ __attribute__((aligned(16))) float As[2][BK * TN_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * TN_LD];
    const int ntiles = tm.first[batch.count];
    int sk, t;
    if ((splits & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int q = j / ntiles;
        sk = xcd + 8 * q;
        t = j - q * ntiles;
    } else {
        sk = blockIdx.x / ntiles;
        t = blockIdx.x - sk * ntiles;
    }
    int pz = 0;
    while (t >= tm.first[pz + 1]) ++pz;
    const GemmProb &p = batch.p[pz];
    const int tl = t - tm.first[pz];
    const int tmi = tl / tm.tiles_n[pz];
    const int m0 = tmi * 128, n0 = (tl - tmi * tm.tiles_n[pz]) * 128;

    int nchunks = p.K / BK;
    if constexpr (SKIP) nchunks = rs.cum[rs.steps];
    const int cps = (nchunks + splits - 1) / splits;
    const int c0 = sk * cps;
    const int c1 = min(nchunks, c0 + cps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 31, lk = lane >> 5;
    // wave -> 32 x 32 sub-tiles.  mode 0: 2 x 2 quadrant of the 128 x 128 tile;
    // narrow problems keep all four SIMDs on valid data: mode 1 (N <= 64) gives
    // each wave one 32-row strip x 2 column tiles, mode 2 (M <= 64) one 32-column
    // strip x 2 row tiles (acc[i][j] is then indexed [0][j] / [i][0]).
    const int mode =
        FIXED_MODE >= 0 ? FIXED_MODE : (p.N <= 64 ? 1 : (p.M <= 64 ? 2 : 0));
    const int wm = mode == 0 ? wave >> 1 : 0, wn = mode == 0 ? wave & 1 : 0;
    const int rbase = mode == 0 ? wm * 64 : (mode == 1 ? wave * 32 : 0);
    const int cbase = mode == 0 ? wn * 64 : (mode == 2 ? wave * 32 : 0);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float cs0 = 0.f, cs1 = 0.f;  // column sums of A (this lane's k parity)
    // (wave-uniform) the first tile column's left waves hold every A column of the tile
    const bool want_cs = (p.colsum1 || p.colsum2) && n0 == 0 &&
                         (mode == 0 ? wn == 0 : (mode == 1 || wave == 0));
    if (c0 < c1) {
        // thread -> (k row, 4-wide column group) of the 16 x 128 operand chunks
        const int k_a = tid >> 5, q_a = tid & 31;  // + 8 rows for the second load
        const float *ga =
            p.A + (long long)k_a * p.a_cs + min(m0 + 4 * q_a, p.M - 4);
        const float *gb =
            p.B + (long long)k_a * p.b_rs + min(n0 + 4 * q_a, p.N - 4);
        const long long a8 = 8 * p.a_cs, b8 = 8 * p.b_rs;
        const int lofs = k_a * TN_LD + 4 * q_a;
        const int last = c1 - 1;
        f32x4 ra[2][2], rb[2][2];
        // SKIP: load_chunk is called with c0, c0 + 1, ... in order; (w_t, w_j)
        // is the step and the chunk inside it of the next active chunk
        int w_t = 0, w_j = 0, w_n = 0, w_c = c0;
        if constexpr (SKIP) {
            int lo = 0, hi = rs.steps;       // last t with cum[t] <= c0
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if ((int)rs.cum[mid] <= c0) lo = mid; else hi = mid;
            }
            w_t = lo;
            w_j = c0 - (int)rs.cum[lo];
            w_n = (int)rs.cum[lo + 1] - (int)rs.cum[lo];
        }
        auto load_chunk = [&](int c, f32x4 (&qa)[2], f32x4 (&qb)[2]) {
            c = min(c, last);  // the tail re-reads the last chunk (never stored)
            if constexpr (SKIP) {
                c = w_t * rs.cps + w_j;
                if (w_c < last) {
                    ++w_c;
                    if (++w_j == w_n) {   // counts never grow: no empty step
                        ++w_t; w_j = 0;   // before the last active chunk
                        w_n = (int)rs.cum[w_t + 1] - (int)rs.cum[w_t];
                    }
                }
            }
            const float *pa = ga + (long long)c * BK * p.a_cs;
            const float *pb = gb + (long long)c * BK * p.b_rs;
            qa[0] = *(const f32x4 *)pa;
            qa[1] = *(const f32x4 *)(pa + a8);
            qb[0] = *(const f32x4 *)pb;
            qb[1] = *(const f32x4 *)(pb + b8);
        };
        auto store_chunk = [&](int buf, const f32x4 (&qa)[2],
                               const f32x4 (&qb)[2]) {
            *(f32x4 *)(&As[buf][lofs]) = qa[0];
            *(f32x4 *)(&As[buf][lofs + 8 * TN_LD]) = qa[1];
            *(f32x4 *)(&Bs[buf][lofs]) = qb[0];
            *(f32x4 *)(&Bs[buf][lofs + 8 * TN_LD]) = qb[1];
        };
        // CS: this wave's column sums are used (first tile column, left waves).
        // An fp32 MFMA and a vector add of the same SIMD do not overlap
        // (tools/mfma_valu_overlap.hip): the other waves must not pay for them.
        auto compute = [&](int buf, auto cs_tag) {
            constexpr bool CS = decltype(cs_tag)::value;
            const float *as = &As[buf][lk * TN_LD + rbase + lrow];
            const float *bs = &Bs[buf][lk * TN_LD + cbase + lrow];
            if constexpr (BF16) {
                f32x8 a0, a1, b0, b1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    a0[i] = as[2 * i * TN_LD];
                    b0[i] = bs[2 * i * TN_LD];
                    if (CS) cs0 += a0[i];
                    if (mode != 1) {
                        a1[i] = as[2 * i * TN_LD + 32];
                        if (CS) cs1 += a1[i];
                    }
                    if (mode != 2) b1[i] = bs[2 * i * TN_LD + 32];
                }
                const bf16x8 pa0 = __builtin_convertvector(a0, bf16x8);
                const bf16x8 pb0 = __builtin_convertvector(b0, bf16x8);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb0, acc[0][0], 0, 0, 0);
                if (mode != 2) {
                    const bf16x8 pb1 = __builtin_convertvector(b1, bf16x8);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa0, pb1, acc[0][1], 0, 0, 0);
                    if (mode == 0) {
                        const bf16x8 pa1 = __builtin_convertvector(a1, bf16x8);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb0, acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb1, acc[1][1], 0, 0, 0);
                    }
                } else {
                    const bf16x8 pa1 = __builtin_convertvector(a1, bf16x8);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa1, pb0, acc[1][0], 0, 0, 0);
                }
            } else if (mode == 0) {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    const float a0 = as[kk * TN_LD], a1 = as[kk * TN_LD + 32];
                    const float b0 = bs[kk * TN_LD], b1 = bs[kk * TN_LD + 32];
                    if (CS) { cs0 += a0; cs1 += a1; }
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
            } else if (mode == 1) {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    const float a0 = as[kk * TN_LD];
                    const float b0 = bs[kk * TN_LD], b1 = bs[kk * TN_LD + 32];
                    if (CS) cs0 += a0;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 2) {
                    const float a0 = as[kk * TN_LD], a1 = as[kk * TN_LD + 32];
                    const float b0 = bs[kk * TN_LD];
                    if (CS) { cs0 += a0; cs1 += a1; }
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                }
            }
        };
        load_chunk(c0, ra[0], rb[0]);
        load_chunk(c0 + 1, ra[1], rb[1]);
        // pairs of chunks so the register sets alternate without copies; the
        // loads are unconditional, hence exact vmcnt waits
        auto k_loop = [&](auto cs_tag) {
            int c = c0;
            for (; c + 1 < c1; c += 2) {
                store_chunk(0, ra[0], rb[0]);
                load_chunk(c + 2, ra[0], rb[0]);
                __syncthreads();
                compute(0, cs_tag);
                store_chunk(1, ra[1], rb[1]);
                load_chunk(c + 3, ra[1], rb[1]);
                __syncthreads();
                compute(1, cs_tag);
            }
            if (c < c1) {
                store_chunk(0, ra[0], rb[0]);
                __syncthreads();
                compute(0, cs_tag);
            }
        };
        if (want_cs) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    }

    if (want_cs) {
        cs0 += __shfl_xor(cs0, 32);
        cs1 += __shfl_xor(cs1, 32);
        if (lk == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (mode == 1 && i == 1) continue;
                const int gm = m0 + rbase + i * 32 + lrow;
                const float v = i ? cs1 : cs0;
                if (gm >= p.M) continue;
                if (splits == 1) {
                    const int ca =
                        p.cs_accumulate < 0 ? p.accumulate : p.cs_accumulate;
                    if (p.colsum1) p.colsum1[gm] = ca ? p.colsum1[gm] + v : v;
                    if (p.colsum2) p.colsum2[gm] = ca ? p.colsum2[gm] + v : v;
                } else {
                    slab[slabs.cs[pz] + (long long)sk * p.M + gm] = v;
                }
            }
        }
    }
    const float alpha = p.alpha_ptr ? *p.alpha_ptr : 1.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if ((mode == 1 && i == 1) || (mode == 2 && j == 1)) continue;
            const int gn = n0 + cbase + j * 32 + lrow;
            if (gn >= p.N) continue;
            const float bias = (p.bias && splits == 1) ? p.bias[gn] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = m0 + rbase + i * 32 + (r & 3) + 8 * (r >> 2) +
                               4 * lk;
                if (gm >= p.M) continue;
                if (splits == 1) {
                    float *c = p.C + (long long)gm * p.ldc + gn;
                    float v = alpha * acc[i][j][r] + bias;
                    if (p.accumulate) v += *c;
                    *c = v;
                } else {
                    slab[slabs.off[pz] +
                         ((long long)sk * p.M + gm) * p.N + gn] = acc[i][j][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmBatch batch,
                                                            int splits,
                                                            const float *slab,
                                                            SlabInfo slabs) {
    const GemmProb &p = batch.p[blockIdx.y];
    const long long total = (long long)p.M * p.N;
    const float alpha = p.alpha_ptr ? *p.alpha_ptr : 1.0f;
    const float *base = slab + slabs.off[blockIdx.y];
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
         e < total; e += (long long)gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int k = 0; k < splits; ++k) s += base[(long long)k * total + e];
        const int m = (int)(e / p.N), n = (int)(e - (long long)m * p.N);
        float v = alpha * s + (p.bias ? p.bias[n] : 0.0f);
        float *c = p.C + (long long)m * p.ldc + n;
        if (p.accumulate) v += *c;
        *c = v;
    }
    if ((p.colsum1 || p.colsum2) && slabs.cs[blockIdx.y] >= 0) {
        const float *cb = slab + slabs.cs[blockIdx.y];
        for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < p.M;
             m += gridDim.x * blockDim.x) {
            float s = 0.0f;
            for (int k = 0; k < splits; ++k) s += cb[(long long)k * p.M + m];
            const int ca = p.cs_accumulate < 0 ? p.accumulate : p.cs_accumulate;
            if (p.colsum1) p.colsum1[m] = ca ? p.colsum1[m] + s : s;
            if (p.colsum2) p.colsum2[m] = ca ? p.colsum2[m] + s : s;
        }
    }
}

struct Plan {
    int big;     // 1: 128x128 tiles, 0: 64x64
    int splits;
    int gx, gy;
};

}  // namespace
int g_pcm_gemm_lds_pad = 0;   // extra dynamic LDS per block (caps blocks per CU)
namespace {

Plan make_plan(const GemmBatch &batch, bool allow_split) {
    Plan pl;
    long long tiles64 = 0, tiles128 = 0;
    int maxM = 0, maxN = 0, minK = 1 << 30;
    for (int i = 0; i < batch.count; ++i) {
        const GemmProb &p = batch.p[i];
        tiles64 += (long long)((p.M + 63) / 64) * ((p.N + 63) / 64);
        tiles128 += (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
        maxM = std::max(maxM, p.M);
        maxN = std::max(maxN, p.N);
        minK = std::min(minK, p.K);
    }
    pl.big = tiles128 >= 128 ? 1 : 0;
    const long long tiles = pl.big ? tiles128 : tiles64;
    const int bm = pl.big ? 128 : 64;
    pl.gx = (maxN + bm - 1) / bm;
    pl.gy = (maxM + bm - 1) / bm;
    int splits = 1;
    if (allow_split && tiles < 512) {
        const int chunks = (minK + BK - 1) / BK;
        splits = (int)((768 + tiles - 1) / tiles);
        splits = std::min(splits, chunks / 16);  // >= 256 k per split
        splits = std::max(1, std::min(splits, 64));
        // second K segment requires chunk-aligned splits only (always true)
    }
    pl.splits = splits;
    return pl;
}

// every problem K-major x K-major with whole 128 x 128 tiles and 16-byte loads?
bool tn_eligible(const GemmBatch &batch) {
    for (int i = 0; i < batch.count; ++i) {
        const GemmProb &p = batch.p[i];
        if (p.a_rs != 1 || p.b_cs != 1 || p.K1 != 0) return false;
        if ((p.M & 3) || (p.N & 3) || (p.K & 15) || p.M < 4 || p.N < 4)
            return false;
        if ((p.a_cs & 3) || (p.b_rs & 3)) return false;
        if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
    }
    return true;
}

// skinny products over a long reduction are worth the 128 x 128 kernel too
bool tn_long_k(const GemmBatch &batch) {
    for (int i = 0; i < batch.count; ++i)
        if (batch.p[i].K < 2048) return false;
    return true;
}

// Split count of a K-major launch: a multiple of 8 (XCD-private K ranges) that
// fills whole rounds of co-resident blocks (3 per CU for wide batches, 2 for
// narrow ones), keeps >= 256 rows per split and fits the slab scratch.
int tn_splits(const GemmBatch &batch, long long tiles, int minK, bool wide,
              size_t scratch_bytes) {
    static const int forced = [] {
        const char *e = getenv("PCM_TN_SPLITS");
        return e ? atoi(e) : 0;
    }();
    const int chunks = minK / BK;
    size_t per_split = 0;
    for (int i = 0; i < batch.count; ++i)
        per_split += (size_t)batch.p[i].M * (batch.p[i].N + 1);
    const long long conc = 256LL * (wide ? 3 : 2);
    int best = 0;
    double best_eff = -1.0;
    for (int sp = 8; sp <= 64; sp += 8) {
        if (chunks / sp < 16) break;
        if (per_split * sp * sizeof(float) > scratch_bytes) break;
        const long long blocks = tiles * sp;
        const long long rounds = (blocks + conc - 1) / conc;
        const double eff = (double)blocks / (double)(rounds * conc);
        if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
    }
    if (forced > 0) best = forced;
    if (best == 0) best = std::max(1, std::min(8, chunks / 16));
    return best;
}

}  // namespace

bool pcm_gemm_takes_colsum(const GemmBatch &batch, bool have_scratch) {
    Plan pl = make_plan(batch, have_scratch);
    return tn_eligible(batch) && (pl.big || tn_long_k(batch));
}

size_t pcm_gemm_scratch_bytes(const GemmBatch &batch) {
    Plan pl = make_plan(batch, true);
    if (tn_eligible(batch)) pl.splits = std::max(pl.splits, 64);
    if (pl.splits == 1) return 0;
    size_t total = 0;
    for (int i = 0; i < batch.count; ++i)
        total += (size_t)pl.splits * batch.p[i].M * batch.p[i].N;
    return total * sizeof(float);
}

bool pcm_gemm_row_skip(const int32_t *active_rows, int steps, int batch,
                       GemmRowSkip *out) {
    if (!active_rows || steps < 1 || steps > PCM_SKIP_MAX_STEPS || (batch % BK))
        return false;
    if ((long long)steps * (batch / BK) > 65535) return false;
    out->steps = steps;
    out->cps = batch / BK;
    int cum = 0, prev = batch;
    for (int t = 0; t < steps; ++t) {
        const int rows = active_rows[t];
        if (rows < 0 || rows > prev) return false;   // must not grow
        prev = rows;
        out->cum[t] = (unsigned short)cum;
        cum += (rows + BK - 1) / BK;
    }
    for (int t = steps; t <= PCM_SKIP_MAX_STEPS; ++t)
        out->cum[t] = (unsigned short)cum;
    // worth it from a tenth of the rows on; an empty reduction keeps the plain
    // path (which writes the zeros / keeps the accumulators)
    return cum > 0 && cum * 10 <= steps * (batch / BK) * 9;
}

int pcm_gemm_launch(const GemmBatch &batch, void *scratch, size_t scratch_bytes,
                    hipStream_t stream, const GemmRowSkip *skip) {
    PCM_REQUIRE(batch.count >= 1 && batch.count <= PCM_GEMM_MAX_PROBS,
                PCM_EINVAL, "gemm: bad problem count %d", batch.count);
    for (int i = 0; i < batch.count; ++i) {
        const GemmProb &p = batch.p[i];
        PCM_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.A && p.B && p.C,
                    PCM_EINVAL, "gemm: empty problem %d (%d,%d,%d)", i, p.M,
                    p.N, p.K);
        PCM_REQUIRE(p.K1 % BK == 0 && p.K1 >= 0 && p.K1 < p.K, PCM_EINVAL,
                    "gemm: K1=%d must be a multiple of %d below K", p.K1, BK);
        PCM_REQUIRE(p.K1 == 0 || (p.A2 && p.B2), PCM_EINVAL,
                    "gemm: second K segment without operands");
    }
    Plan pl = make_plan(batch, scratch != nullptr);
    const bool use_tn = tn_eligible(batch) && (pl.big || tn_long_k(batch));
    if (skip && use_tn)
        for (int i = 0; i < batch.count; ++i)
            PCM_REQUIRE(batch.p[i].K == skip->steps * skip->cps * BK, PCM_EINVAL,
                        "gemm: row-skip descriptor covers %d rows, problem %d "
                        "has K=%d", skip->steps * skip->cps * BK, i,
                        batch.p[i].K);
    if (!use_tn) skip = nullptr;
    TileMap tmap;
    long long tn_tiles = 0;
    if (use_tn) {
        int minK = 1 << 30;
        for (int i = 0; i < PCM_GEMM_MAX_PROBS; ++i) {
            tmap.first[i] = (int)tn_tiles;
            tmap.tiles_n[i] = 1;
            if (i >= batch.count) continue;
            tmap.tiles_n[i] = (batch.p[i].N + 127) / 128;
            tn_tiles += (long long)((batch.p[i].M + 127) / 128) * tmap.tiles_n[i];
            minK = std::min(minK, batch.p[i].K);
        }
        tmap.first[PCM_GEMM_MAX_PROBS] = (int)tn_tiles;
        for (int i = batch.count; i <= PCM_GEMM_MAX_PROBS; ++i)
            tmap.first[i] = (int)tn_tiles;
        bool wide_batch = true;
        for (int i = 0; i < batch.count; ++i)
            wide_batch = wide_batch && batch.p[i].M > 64 && batch.p[i].N > 64;
        if (skip) minK = (int)skip->cum[skip->steps] * BK;
        pl.splits = scratch ? tn_splits(batch, tn_tiles, minK, wide_batch,
                                        scratch_bytes)
                            : 1;
    }
    SlabInfo slabs;
    size_t need = 0;
    for (int i = 0; i < PCM_GEMM_MAX_PROBS; ++i) {
        slabs.off[i] = 0;
        slabs.cs[i] = -1;
    }
    if (pl.splits > 1) {
        for (int i = 0; i < batch.count; ++i) {
            slabs.off[i] = (long long)need;
            need += (size_t)pl.splits * batch.p[i].M * batch.p[i].N;
            if (use_tn && (batch.p[i].colsum1 || batch.p[i].colsum2)) {
                slabs.cs[i] = (long long)need;
                need += (size_t)pl.splits * batch.p[i].M;
            }
        }
        if (need * sizeof(float) > scratch_bytes) {
            pl.splits = 1;  // fall back to a single pass, still correct
        }
    }
    dim3 grid(pl.gx, pl.gy, batch.count * pl.splits);
    float *slab = (float *)scratch;
    if (use_tn) {
        bool wide = true;
        for (int i = 0; i < batch.count; ++i)
            wide = wide && batch.p[i].M > 64 && batch.p[i].N > 64;
        const dim3 tn_grid((unsigned)(tn_tiles * pl.splits));
        // the weight-gradient probe (id 6) times the wide launches only (a
        // cleared start event makes pcm_probe_end_ drop the pair)
        if (!wide && g_pcm_probe_id == PCM_K_DW_GEMM) g_pcm_probe_start = nullptr;
        const NoSkip ns;
        if (skip && wide && g_pcm_matmul_bf16)
            PCM_LAUNCH((gemm_tn_kernel<0, true, true>), tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, *skip);
        else if (skip && wide)
            PCM_LAUNCH((gemm_tn_kernel<0, false, true>), tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, *skip);
        else if (skip)
            PCM_LAUNCH((gemm_tn_kernel<-1, false, true>), tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, *skip);
        else if (wide && g_pcm_matmul_bf16)
            PCM_LAUNCH((gemm_tn_kernel<0, true>), tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, ns);
        else if (wide)
            PCM_LAUNCH(gemm_tn_kernel<0>, tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, ns);
        else
            PCM_LAUNCH(gemm_tn_kernel<-1>, tn_grid, dim3(256),
                               g_pcm_gemm_lds_pad, stream, batch, pl.splits,
                               slab, slabs, tmap, ns);
    } else if (pl.big)
        hipLaunchKernelGGL((gemm_kernel<2, 2>), grid, dim3(256), g_pcm_gemm_lds_pad,
                           stream,
                           batch, pl.splits, slab, slabs);
    else
        hipLaunchKernelGGL((gemm_kernel<1, 1>), grid, dim3(256), g_pcm_gemm_lds_pad,
                           stream,
                           batch, pl.splits, slab, slabs);
    if (pl.splits > 1) {
        long long maxel = 0;
        for (int i = 0; i < batch.count; ++i)
            maxel = std::max(maxel, (long long)batch.p[i].M * batch.p[i].N);
        int bx = (int)std::min<long long>((maxel + 255) / 256, 1024);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(bx, batch.count),
                           dim3(256), 0, stream, batch, pl.splits, slab, slabs);
    }
    return pcm_check_launch("gemm");
}

// ---- column sums ---------------------------------------------------------------
namespace {
constexpr int CS_ROWS = 256;  // rows per partial block

// block: 64 columns x 4 row groups; every thread sums 64 rows of its column
// (coalesced 256-B row segments, 8 independent loads in flight), then the 4
// row groups are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(
    const float *X, long long ldx, int M, int N, float *partial) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    const int r0 = blockIdx.y * CS_ROWS + grp * (CS_ROWS / 4);
    const int r1 = min(M, r0 + CS_ROWS / 4);
    float s = 0.0f;
    if (n < N) {
        int r = r0;
        for (; r + 8 <= r1; r += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = X[(long long)(r + i) * ldx + n];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; r < r1; ++r) s += X[(long long)r * ldx + n];
    }
    red[grp][col] = s;
    __syncthreads();
    if (grp == 0 && n < N)
        partial[(long long)blockIdx.y * N + n] =
            ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
}

// block: 64 columns x 4 part groups; group g sums its contiguous quarter of the
// partials in order with 8 loads in flight (one thread per column walking all
// 240 partials of a T*B = 61440-row reduction one dependent load at a time took
// 55 us), then the groups are combined in order.
__global__ __launch_bounds__(256) void colsum_final_kernel(
    const float *partial, int parts, int N, float *out, const float *alpha_ptr,
    int accumulate) {
    __shared__ float red[4][64];
    const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    const int per = (parts + 3) >> 2;
    const int k0 = grp * per, k1 = min(parts, k0 + per);
    float s = 0.0f;
    if (n < N) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = partial[(long long)(k + i) * N + n];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; k < k1; ++k) s += partial[(long long)k * N + n];
    }
    red[grp][col] = s;
    __syncthreads();
    if (grp != 0 || n >= N) return;
    s = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
    const float alpha = alpha_ptr ? *alpha_ptr : 1.0f;
    float v = alpha * s;
    if (accumulate) v += out[n];
    out[n] = v;
}
}  // namespace

size_t pcm_colsum_scratch_bytes(int M, int N) {
    const int parts = (M + CS_ROWS - 1) / CS_ROWS;
    return (size_t)parts * N * sizeof(float);
}

int pcm_colsum_launch(const float *X, long long ldx, int M, int N, float *out,
                      const float *alpha_ptr, int accumulate, void *scratch,
                      size_t scratch_bytes, hipStream_t stream) {
    PCM_REQUIRE(M > 0 && N > 0 && X && out, PCM_EINVAL, "colsum: bad args");
    PCM_REQUIRE(scratch && scratch_bytes >= pcm_colsum_scratch_bytes(M, N),
                PCM_ENOMEM, "colsum: scratch too small");
    const int parts = (M + CS_ROWS - 1) / CS_ROWS;
    dim3 g1((N + 63) / 64, parts);
    hipLaunchKernelGGL(colsum_partial_kernel, g1, dim3(256), 0, stream, X, ldx,
                       M, N, (float *)scratch);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 63) / 64), dim3(256), 0,
                       stream, (const float *)scratch, parts, N, out, alpha_ptr,
                       accumulate);
    return pcm_check_launch("colsum");
}
