// Reverse-recurrence kernels of the StackGRU (what autograd replays for
// training.py:153 over stack_rnn.py:131-232).
//
//   gate_bwd   elementwise: dL/dh_t + saved (r,z,n,ghn,h_{t-1}) -> gate
//              pre-activation gradients (written over the saved gates, they
//              feed both the per-step dgates.W products and the time-parallel
//              weight-gradient GEMMs) and the direct dL/dh_{t-1} = dL/dh_t * z
//   stack_bwd  adjoint of the push/pop/no-op update, control softmax and push
//              tanh; adds (dc.Wc + du.Ws) into dL/dh_{t-1} of the top layer and
//              takes that layer's gate gradient of step t-1 right there (its
//              last operand), saving a launch per step
#include <stdlib.h>
#include <algorithm>

#include "stackgru.h"

namespace {

// gate derivatives from already loaded (r, z, n, ghn, h_{t-1})
__device__ __forceinline__ void gate_grad_vals(float d, float r, float z, float n,
                                               float ghn, float hp, float *g,
                                               int H, int j, float *dh_out) {
    const float dn = d * (1.f - z);
    const float dz = d * (hp - n);
    const float dpn = dn * (1.f - n * n);
    const float dpz = dz * z * (1.f - z);
    const float dr = dpn * ghn;
    const float dpr = dr * r * (1.f - r);
    g[j] = dpr;
    g[H + j] = dpz;
    g[2 * H + j] = dpn;       // d(W_in.x + b_in)
    g[3 * H + j] = dpn * r;   // d(W_hn.h + b_hn)
    *dh_out = d * z;
}

__device__ __forceinline__ void gate_grad(float d, float *g, int H, int j,
                                          float hp, float *dh_out) {
    gate_grad_vals(d, g[j], g[H + j], g[2 * H + j], g[3 * H + j], hp, g, H, j,
                   dh_out);
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(GateBwdArgs a) {
    // grid: x = blocks of 256 hidden units, y = batch row.  Launched for the
    // top layer of the first step walked (and of every step of a model without
    // a stack), and for every layer on the generic path; elsewhere the gate
    // gradient rides in the stack adjoint / the reverse cell's epilogue.
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    float *dh = a.dh + (long long)b * a.Hp + j;
    const float d = *dh;
    float *g = a.gates + (long long)b * 4 * a.H;
    if (a.lengths && a.t >= a.lengths[b]) {
        // frozen row: h_t = h_{t-1}; its output carries no gradient
        g[j] = 0.f; g[a.H + j] = 0.f; g[2 * a.H + j] = 0.f;
        g[3 * a.H + j] = 0.f;
        return;
    }
    const float dout = a.dout ? a.dout[(long long)b * a.ld_dout + j] : 0.f;
    gate_grad(d + dout, g, a.H, j, a.hprev[(long long)b * a.Hp + j], dh);
}

// One block of 512 threads per BROWS batch rows.  S_{t-1} and dL/dS'_t of the
// rows are staged in LDS (one HBM read each), the three control-gradient sums
// are block-reduced, and the block finishes with dL/dh_top += dc.Wc + du.Ws.
constexpr int BTHREADS = 512;
#ifndef FIN_UNROLL
#define FIN_UNROLL 16
#endif

template <int BROWS>
__global__ __launch_bounds__(BTHREADS) void stack_bwd_kernel(StackBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int W = a.W, D = a.D, DW = D * W;
    const int DWp = (DW + 3) & ~3;
    const float inv_DW = 1.0f / (float)DW, inv_W = 1.0f / (float)W;
    const float inv_AVW = 1.0f / (float)a.AVW;
    const float inv_AV3 = 1.0f / (float)max(a.AVW - 3, 1);
    const float inv_nq = 1.0f / (float)max(DW >> 2, 1);
    float *Srow = sm;                       // [BROWS][DWp]  S_{t-1}
    float *Grow = Srow + BROWS * DWp;       // [BROWS][DWp]  dL/dS'_t (+ dtop)
    float *avs = Grow + BROWS * DWp;        // [BROWS][AVW]  a, v
    float *dcu = avs + BROWS * a.AVW;       // [BROWS][AVW]  dc, du
    float *red = dcu + BROWS * a.AVW;       // [BROWS][3][8] wave partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * BROWS;

    // fused top-layer gate gradient of step t-1: its operands are cold HBM
    // lines, the first column's are fetched with the staging loads below
    struct GatePre { float d, r, z, n, g, hp, dout; };
    auto gate_load = [&](int r, int j) -> GatePre {
        GatePre q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int b = b0 + r;
        if (b >= a.gate_B || j >= a.H) return q;
        q.d = a.g_dh[(long long)b * a.Hp + j];
        if (a.gate_mode == 1) {
            const float *g = a.g_gates + (long long)b * 4 * a.H;
            q.r = g[j]; q.z = g[a.H + j]; q.n = g[2 * a.H + j];
            q.g = g[3 * a.H + j];
            q.hp = a.g_hprev[(long long)b * a.Hp + j];
            if (a.g_dout) q.dout = a.g_dout[(long long)b * a.g_ld_dout + j];
        }
        return q;
    };
    GatePre gpre[BROWS];
    if (a.gate_mode) {
#pragma unroll
        for (int r = 0; r < BROWS; ++r) gpre[r] = gate_load(r, tid);
    }
    // dL/dh_{t-1} += dc.Wc + du.Ws (the stack-control path of this step), then
    // the gate derivatives in place; needs dcu[] complete and synchronised
    auto gate_part = [&]() {
        const int H = a.H, NO = 3 + W;
        for (int j = tid; j < H; j += BTHREADS) {
            float s[BROWS];
#pragma unroll
            for (int r = 0; r < BROWS; ++r) s[r] = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const float w = a.ctrl_w[(long long)o * H + j];
#pragma unroll
                for (int r = 0; r < BROWS; ++r)
                    s[r] = fmaf(dcu[r * a.AVW + o], w, s[r]);
            }
#pragma unroll 16
            for (int o = 3; o < NO; ++o) {
                const float w = a.push_w[(long long)(o - 3) * H + j];
#pragma unroll
                for (int r = 0; r < BROWS; ++r)
                    s[r] = fmaf(dcu[r * a.AVW + o], w, s[r]);
            }
#pragma unroll
            for (int r = 0; r < BROWS; ++r) {
                const int b = b0 + r;
                if (b >= a.gate_B) continue;
                const GatePre q = j == tid ? gpre[r] : gate_load(r, j);
                float *dh = a.g_dh + (long long)b * a.Hp + j;
                const float d = q.d + s[r];
                if (a.gate_mode == 2) { *dh = d; continue; }
                float *g = a.g_gates + (long long)b * 4 * H;
                if (a.lengths && a.t - 1 >= a.lengths[b]) {
                    // frozen row: h_t = h_{t-1}; its output carries no gradient
                    g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f;
                    g[3 * H + j] = 0.f;
                    *dh = d;
                    continue;
                }
                gate_grad_vals(d + q.dout, q.r, q.z, q.n, q.g, q.hp, g, H, j, dh);
            }
        }
    };
    // the stack-top gradient (the reverse cell may leave it as K-part partials,
    // summed here in order): fetched with the first loads, used after staging
    auto dtop_sum = [&](int r, int w) -> float {
        const int b = b0 + r;
        if (b >= a.B || w >= W) return 0.f;
        const float *src = a.dtop + (long long)b * a.Wp + w;
        // all partials in flight at once (a rolled loop would wait for each)
        constexpr int PMAX = 16;
        float part[PMAX];
#pragma unroll
        for (int p = 0; p < PMAX; ++p)
            part[p] = p < a.dtop_parts ? src[p * a.dtop_stride] : 0.f;
        float v = part[0];
#pragma unroll
        for (int p = 1; p < PMAX; ++p) v += part[p];
        for (int p = PMAX; p < a.dtop_parts; ++p) v += src[p * a.dtop_stride];
        return v;
    };
    float dtop_pre = 0.f;
    if (tid < BROWS * W) {
        const int r = pcm_fdiv(tid, inv_W);
        dtop_pre = dtop_sum(r, tid - r * W);
    }
    if (b0 >= a.B) {
        // rows that finish at step t-1 (shrinking active batch): no stack step
        // at t, hence no control path -- the gate gradient only
        for (int idx = tid; idx < BROWS * a.AVW; idx += BTHREADS) dcu[idx] = 0.f;
        __syncthreads();
        gate_part();
        return;
    }

    if ((DW & 3) == 0) {
        const int nq = DW >> 2;
        for (int idx = tid; idx < BROWS * nq; idx += BTHREADS) {
            const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
            const int b = b0 + r;
            f32x4 sv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
            if (b < a.B) {
                sv = *(const f32x4 *)(a.S_prev + (long long)b * DW + q * 4);
                if (a.dS_next)
                    gv = *(const f32x4 *)(a.dS_next + (long long)b * DW + q * 4);
            }
            *(f32x4 *)(Srow + r * DWp + q * 4) = sv;
            *(f32x4 *)(Grow + r * DWp + q * 4) = gv;
        }
    } else {
        for (int idx = tid; idx < BROWS * DW; idx += BTHREADS) {
            const int r = pcm_fdiv(idx, inv_DW), e = idx - r * DW;
            const int b = b0 + r;
            const bool ok = b < a.B;
            Srow[r * DWp + e] = ok ? a.S_prev[(long long)b * DW + e] : 0.f;
            Grow[r * DWp + e] =
                (ok && a.dS_next) ? a.dS_next[(long long)b * DW + e] : 0.f;
        }
    }
    for (int idx = tid; idx < BROWS * a.AVW; idx += BTHREADS) {
        const int r = pcm_fdiv(idx, inv_AVW), o = idx - r * a.AVW;
        const int b = b0 + r;
        avs[idx] = b < a.B ? a.av[(long long)b * a.AVW + o] : 0.f;
    }
    __syncthreads();
    // the stack top fed layer 0: its gradient joins row 0 of G
    for (int idx = tid; idx < BROWS * W; idx += BTHREADS) {
        const int r = pcm_fdiv(idx, inv_W), w = idx - r * W;
        const int b = b0 + r;
        if (b < a.B && !(a.lengths && a.t >= a.lengths[b]))
            Grow[r * DWp + w] += idx == tid ? dtop_pre : dtop_sum(r, w);
    }
    __syncthreads();

    // elementwise adjoint + partial sums of the control gradients
    float s_push[BROWS], s_pop[BROWS], s_noop[BROWS];
#pragma unroll
    for (int r = 0; r < BROWS; ++r) { s_push[r] = 0.f; s_pop[r] = 0.f; s_noop[r] = 0.f; }
#pragma unroll
    for (int r = 0; r < BROWS; ++r) {
        const int b = b0 + r;
        if (b >= a.B) continue;
        const bool frozen = a.lengths && a.t >= a.lengths[b];
        const float *Sp = Srow + r * DWp;
        const float *G = Grow + r * DWp;
        const float *c = avs + r * a.AVW;
        const float a0 = c[0], a1 = c[1], a2 = c[2];
        float *dSp = a.dS_prev + (long long)b * DW;
        for (int e = tid; e < DW; e += BTHREADS) {
            const float g = G[e];
            if (frozen) { dSp[e] = g; continue; }
            const int d = pcm_fdiv(e, inv_W), w = e - d * W;
            const float s = Sp[e];
            const float up = d == 0 ? c[3 + w] : Sp[e - W];
            const float dn = d == D - 1 ? 0.f : Sp[e + W];
            s_push[r] = fmaf(g, up, s_push[r]);
            s_pop[r] = fmaf(g, dn, s_pop[r]);
            s_noop[r] = fmaf(g, s, s_noop[r]);
            // S[d] feeds S'[d] (no-op), S'[d+1] (push) and S'[d-1] (pop)
            float v = a2 * g;
            if (d + 1 < D) v += a0 * G[e + W];
            if (d >= 1) v += a1 * G[e - W];
            dSp[e] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < BROWS; ++r) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s_push[r] += __shfl_xor(s_push[r], off, 64);
            s_pop[r] += __shfl_xor(s_pop[r], off, 64);
            s_noop[r] += __shfl_xor(s_noop[r], off, 64);
        }
        if (lane == 0) {
            red[(r * 3 + 0) * 8 + wave] = s_push[r];
            red[(r * 3 + 1) * 8 + wave] = s_pop[r];
            red[(r * 3 + 2) * 8 + wave] = s_noop[r];
        }
    }
    __syncthreads();
    if (tid < BROWS) {
        const int r = tid;
        const int b = b0 + r;
        const bool frozen = b < a.B && a.lengths && a.t >= a.lengths[b];
        float sp = 0.f, so = 0.f, sn = 0.f;
        for (int w8 = 0; w8 < 8; ++w8) {
            sp += red[(r * 3 + 0) * 8 + w8];
            so += red[(r * 3 + 1) * 8 + w8];
            sn += red[(r * 3 + 2) * 8 + w8];
        }
        const float *c = avs + r * a.AVW;
        // softmax backward: dc_i = a_i * (da_i - sum_j a_j da_j)
        const float dot = c[0] * sp + c[1] * so + c[2] * sn;
        float *out = dcu + r * a.AVW;
        out[0] = frozen ? 0.f : c[0] * (sp - dot);
        out[1] = frozen ? 0.f : c[1] * (so - dot);
        out[2] = frozen ? 0.f : c[2] * (sn - dot);
    }
    // dv[w] = a_push * G[0][w]; du = dv * (1 - v^2); zero the padding
    for (int idx = tid; idx < BROWS * (a.AVW - 3); idx += BTHREADS) {
        const int r = pcm_fdiv(idx, inv_AV3), w = idx - r * (a.AVW - 3);
        const int b = b0 + r;
        const bool frozen = b < a.B && a.lengths && a.t >= a.lengths[b];
        float val = 0.f;
        if (w < W && !frozen) {
            const float *c = avs + r * a.AVW;
            const float v = c[3 + w];
            val = c[0] * Grow[r * DWp + w] * (1.f - v * v);
        }
        dcu[r * a.AVW + 3 + w] = val;
    }
    __syncthreads();
    // keep dc/du for the time-parallel dWc / dWs products
    for (int idx = tid; idx < BROWS * a.AVW; idx += BTHREADS) {
        const int r = pcm_fdiv(idx, inv_AVW), o = idx - r * a.AVW;
        const int b = b0 + r;
        if (b < a.B) a.av[(long long)b * a.AVW + o] = dcu[idx];
    }
    if (a.gate_mode) gate_part();
}

// ---- the stack adjoint in two parts ---------------------------------------------
// stack_bwd_kernel sits between the layer-0 reverse cell (whose stack-top
// gradient it consumes) and the top cell of the step below (whose gate gradient
// it produces), but only rows 0 and 1 of its 30 KB-per-row elementwise work
// depend on that stack-top gradient: dL/dS'_t = G + [dtop in row 0], and both the
// adjoint and the control sums are linear in it.  The bulk (this function; one
// batch row per block, called from the extra blocks of the top layer's reverse
// cell launch) runs on G alone, beside the cells; stack_bwd_finish_kernel adds
// the dtop terms.  sm: 2 * DWp + 32 floats of LDS.
__device__ __forceinline__ void stack_bwd_bulk(const StackBulkArgs &a, float *sm,
                                               int b, int tid) {
    const int W = a.W, D = a.D, DW = D * W;
    const int DWp = (DW + 3) & ~3;
    const float inv_W = 1.0f / (float)W;
    float *Srow = sm, *Grow = sm + DWp, *red = Grow + DWp;   // red: [3][8]
    const int lane = tid & 63, wave = tid >> 6;
    const float *c = a.av + (long long)b * a.AVW;
    const float a0 = c[0], a1 = c[1], a2 = c[2];
    const bool frozen = a.lengths && a.t >= a.lengths[b];
    const float *Sg = a.S_prev + (long long)b * DW;
    const float *Gg = a.dS_next ? a.dS_next + (long long)b * DW : nullptr;
    if ((DW & 3) == 0) {
        for (int q = tid; q < (DW >> 2); q += BTHREADS) {
            *(f32x4 *)(Srow + q * 4) = *(const f32x4 *)(Sg + q * 4);
            *(f32x4 *)(Grow + q * 4) =
                Gg ? *(const f32x4 *)(Gg + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        for (int e = tid; e < DW; e += BTHREADS) {
            Srow[e] = Sg[e];
            Grow[e] = Gg ? Gg[e] : 0.f;
        }
    }
    __syncthreads();
    float s_push = 0.f, s_pop = 0.f, s_noop = 0.f;
    float *dSp = a.dS_prev + (long long)b * DW;
    for (int e = tid; e < DW; e += BTHREADS) {
        const float g = Grow[e];
        if (frozen) { dSp[e] = g; continue; }
        const int d = pcm_fdiv(e, inv_W), w = e - d * W;
        const float sv = Srow[e];
        const float up = d == 0 ? c[3 + w] : Srow[e - W];
        const float dn = d == D - 1 ? 0.f : Srow[e + W];
        s_push = fmaf(g, up, s_push);
        s_pop = fmaf(g, dn, s_pop);
        s_noop = fmaf(g, sv, s_noop);
        // S[d] feeds S'[d] (no-op), S'[d+1] (push) and S'[d-1] (pop)
        float v = a2 * g;
        if (d + 1 < D) v += a0 * Grow[e + W];
        if (d >= 1) v += a1 * Grow[e - W];
        dSp[e] = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s_push += __shfl_xor(s_push, off, 64);
        s_pop += __shfl_xor(s_pop, off, 64);
        s_noop += __shfl_xor(s_noop, off, 64);
    }
    if (lane == 0) {
        red[0 * 8 + wave] = s_push;
        red[1 * 8 + wave] = s_pop;
        red[2 * 8 + wave] = s_noop;
    }
    __syncthreads();
    if (tid < 3) {
        float v = 0.f;
        for (int w8 = 0; w8 < 8; ++w8) v += red[tid * 8 + w8];
        a.psum[(long long)b * 4 + tid] = v;
    }
}

// R batch rows per block.  Takes the bulk's control sums and dL/dS_{t-1}, adds the
// stack-top gradient's share (dL/dS'_t[0] += dtop: the no-op term of row 0, the
// pop term of row 1, three W-long dot products; wave r serves row r), then
// softmax / tanh backward, dc / du, and -- as stack_bwd_kernel -- the top layer's
// gate gradient of step t - 1: dL/dh += dc.Wc + du.Ws.  That product reads the
// whole 108 KB [Wc ; Ws] per block whatever the number of rows (55 MB per launch
// through the L2s at one row per block); R rows share every weight a thread
// loads -- and measured 2-3 x slower (launcher), so R = 1 is what runs.
template <int R>
__global__ __launch_bounds__(BTHREADS) void stack_bwd_finish_kernel(StackBwdArgs a) {
    static_assert(R <= BTHREADS / 64, "one wave per row");
    __shared__ float dcu[R][64];
    const int W = a.W, D = a.D, DW = D * W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * R;
    struct GatePre { float d, r, z, n, g, hp, dout; };
    auto gate_load = [&](int b, int j) -> GatePre {
        GatePre q = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (b >= a.gate_B || j >= a.H) return q;
        q.d = a.g_dh[(long long)b * a.Hp + j];
        if (a.gate_mode == 1) {
            const float *g = a.g_gates + (long long)b * 4 * a.H;
            q.r = g[j]; q.z = g[a.H + j]; q.n = g[2 * a.H + j];
            q.g = g[3 * a.H + j];
            q.hp = a.g_hprev[(long long)b * a.Hp + j];
            if (a.g_dout) q.dout = a.g_dout[(long long)b * a.g_ld_dout + j];
        }
        return q;
    };
    for (int idx = tid; idx < R * 64; idx += BTHREADS) dcu[idx >> 6][idx & 63] = 0.f;

    // ---- wave r: the stack-top terms of row b0 + r ---------------------------------
    // (the column of [Wc ; Ws] is fetched at the point of use: requested up front,
    // 64 loads per thread in front of everything else, the launch measured 9.2 us
    // against 8.0)
    const int b = b0 + wave;
    const bool row = wave < R && b < a.B;
    const bool frozen = row && a.lengths && a.t >= a.lengths[b];
    const bool col = row && lane < W;     // this lane holds a stack-top column
    const bool live = col && !frozen;
    constexpr int PMAX = 16;
    float part[PMAX], s0 = 0.f, s1 = 0.f, c1 = 0.f, c2 = 0.f, o0 = 0.f, o1 = 0.f,
                      ps = 0.f, g0 = 0.f, av_l = 0.f;
    const float *src = a.dtop + (long long)b * a.Wp + lane;
    float *d0 = a.dS_prev + (long long)b * DW + lane;
#pragma unroll
    for (int p = 0; p < PMAX; ++p)
        part[p] = (live && p < a.dtop_parts) ? src[p * a.dtop_stride] : 0.f;
    if (row && lane < a.AVW) av_l = a.av[(long long)b * a.AVW + lane];
    if (live) {
        s0 = a.S_prev[(long long)b * DW + lane];
        s1 = D > 1 ? a.S_prev[(long long)b * DW + W + lane] : 0.f;
        c1 = a.av[(long long)b * a.AVW + 1];
        c2 = a.av[(long long)b * a.AVW + 2];
        o0 = *d0;
        o1 = D > 1 ? d0[W] : 0.f;
    }
    if (col) g0 = a.dS_next ? a.dS_next[(long long)b * DW + lane] : 0.f;
    if (row && lane < 3) ps = a.psum[(long long)b * 4 + lane];
    asm volatile("" ::: "memory");
    // the gate operands of this thread's column, all R rows (cold lines:
    // requested last, used last)
    GatePre gpre[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        gpre[r] = GatePre{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (a.gate_mode) gpre[r] = gate_load(b0 + r, tid);
    }
    if (wave < R) {
        float p_push = 0.f, p_pop = 0.f, p_noop = 0.f;
        // v of this lane's column: av[3 + lane] sits in lane 3 + lane
        const float vv = __shfl(av_l, min(lane + 3, 63), 64);
        if (live) {
            float dt = part[0];
#pragma unroll
            for (int p = 1; p < PMAX; ++p) dt += part[p];
            for (int p = PMAX; p < a.dtop_parts; ++p) dt += src[p * a.dtop_stride];
            p_push = dt * vv; p_pop = dt * s1; p_noop = dt * s0;
            *d0 = fmaf(c2, dt, o0);                 // no-op: S[0] feeds S'[0]
            if (D > 1) d0[W] = fmaf(c1, dt, o1);    // pop:   S[1] feeds S'[0]
            g0 += dt;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            p_push += __shfl_xor(p_push, off, 64);
            p_pop += __shfl_xor(p_pop, off, 64);
            p_noop += __shfl_xor(p_noop, off, 64);
        }
        // lanes 0..2 hold the bulk's push / pop / no-op sums; softmax backward:
        // dc_i = a_i * (da_i - sum_j a_j da_j)
        const float da = ps + (lane == 0 ? p_push : (lane == 1 ? p_pop : p_noop));
        const float ada = lane < 3 ? av_l * da : 0.f;
        const float dot = __shfl(ada, 0, 64) + __shfl(ada, 1, 64) + __shfl(ada, 2, 64);
        const float a_push = __shfl(av_l, 0, 64);
        float out = 0.f;
        if (row && !frozen) {
            if (lane < 3) out = av_l * (da - dot);
            // dv[w] = a_push * G[0][w]; du = dv * (1 - v^2): column w -> slot 3 + w
        }
        // move column results to their slots: slot o = 3 + w comes from lane o - 3
        const float du = (live) ? a_push * g0 * (1.f - vv * vv) : 0.f;
        const float du_s = __shfl(du, max(lane - 3, 0), 64);
        if (row && !frozen && lane >= 3 && lane < 3 + W) out = du_s;
        dcu[wave][lane] = out;
        // keep dc / du for the time-parallel dWc / dWs products
        if (row && lane < a.AVW) a.av[(long long)b * a.AVW + lane] = out;
    }
    __syncthreads();
    if (!a.gate_mode) return;

    // ---- all threads: dL/dh_{t-1} += dc.Wc + du.Ws, gate derivatives in place --------
    const int H = a.H, NO = 3 + W;
    for (int j = tid; j < H; j += BTHREADS) {
        float s[R];
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] = 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            const float w = a.ctrl_w[(long long)o * H + j];
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] = fmaf(dcu[r][o], w, s[r]);
        }
        // (the loop as hipcc pipelines it at R = 1 -- 16 loads, then waits that
        // count down -- is the fastest form found: with R > 1 the rows' LDS reads
        // make it wait for every load on its own (16.4 us at R = 2), and batches
        // written out by hand read 10 us at R = 1 and 11 - 18 us at R = 2 / 4)
#pragma unroll FIN_UNROLL
        for (int o = 3; o < NO; ++o) {
            const float w = a.push_w[(long long)(o - 3) * H + j];
#pragma unroll
            for (int r = 0; r < R; ++r) s[r] = fmaf(dcu[r][o], w, s[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int bb = b0 + r;
            if (bb >= a.gate_B) continue;
            const GatePre q = j == tid ? gpre[r] : gate_load(bb, j);
            float *dh = a.g_dh + (long long)bb * a.Hp + j;
            const float d = q.d + s[r];
            if (a.gate_mode == 2) { *dh = d; continue; }
            float *g = a.g_gates + (long long)bb * 4 * H;
            if (a.lengths && a.t - 1 >= a.lengths[bb]) {
                // frozen row: h_t = h_{t-1}; its output carries no gradient
                g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f;
                g[3 * H + j] = 0.f;
                *dh = d;
                continue;
            }
            gate_grad_vals(d + q.dout, q.r, q.z, q.n, q.g, q.hp, g, H, j, dh);
        }
    }
}


// ------------------------------------------------------------------------------
// fused reverse GRU cell: dgates . W on fp32 MFMA, LDS-staged 64-wide K chunks,
// intra-block split-K over the 4 waves, accumulate / dropout-mask epilogue
// ------------------------------------------------------------------------------
constexpr int BCK = 64;         // K per chunk
constexpr int BCM = 32;         // batch rows per block
// output columns per block: BCN = 64 (2 MFMA tiles, 24 KiB stages) or 32 (one
// tile, 16 KiB stages) -- the narrow variant doubles the block count when the
// wide one leaves CUs idle (layer 0: 9 x 16 wide tiles on 256 CUs)

// Wave-specialised like the forward cell: waves 4-7 stream the operand chunks
// into LDS with LDS-DMA loads (global_load_lds_dwordx4; unpadded 256-byte rows,
// 16-byte granules XOR-swizzled with (row & 15) on the source address; two
// stages, one chunk in flight across the raw barriers, hand-counted vmcnt),
// waves 0-3 read fragments and issue the MFMAs.  The consumer waves issue no
// global loads inside the loop, so they fetch everything the epilogue needs
// (running gradients, the lower layer's saved gates: cold HBM lines) at kernel
// start and only wait for it after the last chunk.
__device__ __forceinline__ void glds6(const char *const (&base)[6],
                                      const uint32_t (&off)[6],
                                      uint32_t lds_byte) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %10, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %11, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %12, %6\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %13, %7\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_byte), "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3]),
          "s"(base[4]), "s"(base[5]), "v"(off[0]), "v"(off[1]), "v"(off[2]),
          "v"(off[3]), "v"(off[4]), "v"(off[5])
        : "memory", "scc");
}

__device__ __forceinline__ void glds4(const char *const (&base)[4],
                                      const uint32_t (&off)[4],
                                      uint32_t lds_byte) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %5\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_byte), "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3]),
          "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3])
        : "memory", "scc");
}

__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// XPART (narrow tiles, layer 0): the grid holds the hidden-side column tiles
// only, and the input-side product -- the stack-top gradient, Kxp columns, which
// as tiles of its own would put a second block on 1/8 of the CUs and double the
// launch's MFMA time there -- is spread over them instead: block (column tile
// ct) adds, after its own K loop, the product of K part ct / ncx and input-side
// column tile ct % ncx (ncx = Kxp / 32) in a second accumulator and stores it
// as partial number ct / ncx; the stack adjoint sums the partials in order.
// RIDER: blocks past the cell's own tiles run the bulk stack adjoint of the step
// (stack_bwd_bulk, one batch row each) in the launch's shadow: two cell blocks and
// one rider block fit a CU (48 VGPRs, 3 x 48 KiB of LDS).
template <int BCN, int BNSTAGE, bool XPART = false, bool BF16 = false,
          bool RIDER = false>
__global__ __launch_bounds__(512) void gru_cell_bwd_kernel(CellBwdArgs a) {
    static_assert(!XPART || BCN == 32, "XPART rides on the narrow tile");
    constexpr int BSTAGE = (BCM + BCN) * BCK;    // floats per LDS stage
    constexpr int BCELL_LDS = BNSTAGE * BSTAGE;  // 48 / 32 KiB
    constexpr int NTL = XPART ? 2 : BCN / 32;    // accumulator tiles per block
    constexpr int NQ = NTL * 2;                  // epilogue outputs per thread
    constexpr int NI = (BCM + BCN) / 16;         // DMA instructions per loader wave
    static_assert(BCELL_LDS >= 4 * NTL * 1024, "reduction buffer must fit");
    __shared__ __attribute__((aligned(16))) float lds[BCELL_LDS];
    const int tid = threadIdx.x;
    if constexpr (RIDER) {
        if ((int)blockIdx.x >= a.ncell) {
            // rider_per rows per block, one after the other
            const int r0 = ((int)blockIdx.x - a.ncell) * a.rider_per;
            for (int r = r0; r < min(r0 + a.rider_per, a.rider.B); ++r) {
                if (r > r0) __syncthreads();
                stack_bwd_bulk(a.rider, lds, r, tid);
            }
            return;
        }
    }
    const int lane = tid & 63;
    // wave index as a scalar: role branches and LDS-DMA bases stay on the SALU
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile map: blocks are dealt round-robin over the 8 XCDs, so
    // blockIdx % 8 names the blocks that share an L2.  Each XCD gets one
    // rectangular region of the tile grid (1/4 of the column tiles x 1/2 of
    // the batch tiles): its weight rows plus dgates rows fit the 4 MiB L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int sc_ = slot % a.reg_cols, sr_ = slot / a.reg_cols;
    const int ct = (xcd & 3) * a.reg_cols + sc_;          // column tile
    const int bt = (xcd >> 2) * a.reg_rows + sr_;         // batch tile
    if (ct >= (XPART ? a.H : a.NT) / BCN || bt * BCM >= a.B) return;
    const int n0 = ct * BCN;   // first output column (row of wpackT)
    const int b0 = bt * BCM;
    const int H = a.H, K3 = 3 * H;
    const bool hidden_side = n0 < H;   // tile of dh[l] (uses dnh) or of dx (dni)
    const int nchunks = K3 / BCK;
    // XPART: xcpp extra chunks, K part xkp of input-side column tile xct
    const int ncx = XPART ? (a.NT - H) / 32 : 1;
    const int xct = ct % ncx, xkp = ct / ncx;
    const int xcpp = XPART ? nchunks / a.xparts : 0;
    const int ntot = nchunks + xcpp;
    const int last = ntot - 1;
    const int lrow = lane & 31, lh = lane >> 5;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // Epilogue operands of this thread's 4 outputs (element e = tid + 512 q ->
    // tile e >> 10, row (e & 1023) >> 5, column e & 31), fetched before anything
    // else: they are cold HBM lines (running gradients, the lower layer's saved
    // gates) and with in-order vmcnt they are the oldest loads of either role,
    // so their latency is paid once under the first chunk's DMA instead of at
    // the end of the kernel.
    float pre_d[NQ], pre_r[NQ], pre_z[NQ], pre_n[NQ], pre_g[NQ], pre_h[NQ];
    int pre_len[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = min(b0 + (tid >> 5) + 16 * q, a.B - 1);
        pre_len[q] = a.lengths ? a.lengths[b] : 0x7fffffff;
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 512 * q;
        const int tl = e >> 10, ee = e & 1023;
        const int b = min(b0 + (ee >> 5), a.B - 1);
        const int n = n0 + tl * 32 + (ee & 31);
        pre_d[q] = pre_r[q] = pre_z[q] = pre_n[q] = pre_g[q] = pre_h[q] = 0.f;
        if (XPART && tl == 1) continue;   // a partial product: nothing to add to
        if (hidden_side) {
            pre_d[q] = a.dh[(long long)b * a.Hp + n];
        } else {
            const int j = min(n - H, a.dx_cols - 1);
            if (a.gates_low || a.dx_accumulate)
                pre_d[q] = a.dx[(long long)b * a.ldx + j];
            if (a.gates_low) {
                const float *g = a.gates_low + (long long)b * 4 * H;
                pre_r[q] = g[j]; pre_z[q] = g[H + j];
                pre_n[q] = g[2 * H + j]; pre_g[q] = g[3 * H + j];
                pre_h[q] = a.hprev_low[(long long)b * a.Hp + j];
            }
        }
    }

    if (wave >= 4) {
        // ---------------- loader waves ----------------
        // wave lw issues instructions 6*lw .. 6*lw+5 of a chunk; instruction q
        // fills stage rows 4q .. 4q+3 (rows 0-31: dgates, 32-95: weights);
        // lane -> (row 4q + lane/16, granule slot lane%16)
        const int lw = wave - 4;
        const char *gb = (const char *)a.dgates, *wb = (const char *)a.wpackT;
        uint32_t off[NI], offx[NI];
        bool is_a[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int q = lw * NI + i;
            const int row = 4 * q + (lane >> 4);
            const int g = (lane & 15) ^ (row & 15);   // source granule (swizzle)
            is_a[i] = q < BCM / 4;
            if (is_a[i]) {
                // rows past the batch re-read the last row (never stored)
                const int b = min(b0 + row, a.B - 1);
                off[i] = (uint32_t)(b * 4 * H + g * 4) * 4u;
                offx[i] = off[i];
            } else {
                off[i] = (uint32_t)((n0 + row - BCM) * K3 + g * 4) * 4u;
                offx[i] =
                    (uint32_t)((H + xct * 32 + row - BCM) * K3 + g * 4) * 4u;
            }
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds);
        auto issue = [&](int cc) {
            const int c = min(cc, last);         // the tail re-reads the last chunk
            const bool extra = XPART && c >= nchunks;
            const int kb = (extra ? xkp * xcpp + c - nchunks : c) * BCK;
            // dgates columns: [0,2H) shared; third segment dni at 2H, dnh at 3H
            const int acol = (kb >= 2 * H && hidden_side && !extra) ? kb + H : kb;
            const char *base[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i)
                base[i] = is_a[i] ? gb + acol * 4 : wb + kb * 4;
            const uint32_t dst =
                lds0 + ((cc % BNSTAGE) * BSTAGE + lw * NI * 256) * 4u;
            if constexpr (NI == 6) glds6(base, off, dst);
            else if (extra) glds4(base, offx, dst);
            else glds4(base, off, dst);
        };
        if constexpr (BNSTAGE >= 3) {
            // One barrier per chunk: at barrier k every consumer has waited
            // for its fragment reads of chunk k - 1 (lgkmcnt(0) in front of the
            // barrier), so that chunk's stage is free for chunk k + 2 while
            // chunk k is read and chunk k + 1 is in flight.
            issue(0); issue(1);
            for (int k = 0; k < ntot; ++k) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI) : "memory");
                raw_barrier();   // chunk k landed, stage of chunk k - 1 free
                issue(k + 2);
            }
        } else {
#pragma unroll
            for (int c = 0; c < BNSTAGE; ++c) issue(c);
            for (int k = 0; k < ntot; ++k) {
                // chunk k landed (the newest chunk may be in flight)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (BNSTAGE - 1))
                             : "memory");
                raw_barrier();   // A_k: the consumers may read chunk k
                raw_barrier();   // B_k: they hold its fragments, the stage is free
                issue(k + BNSTAGE);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---------------- consumer waves ----------------
        // this wave's 16-wide k slice; lane half lh supplies k = 8s + 4*lh + e,
        // i.e. granule 4*wave + lh + 2s of its row, stored at slot ^ (row & 15)
        int slot[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            slot[s] = lrow * BCK + (((4 * wave + lh + 2 * s) ^ (lrow & 15)) << 2);
        // two barriers per chunk, one fragment set: a simpler, smaller kernel
        // measured faster than the deeper pipeline (four stages, fragment
        // double buffer) it replaced -- the MFMA loop is at its issue bound
        auto chunk = [&](int c, f32x16 &acc_main) {
            f32x4 fa[2], f0[2], f1[2];
            raw_barrier();   // A_c
            const float *As = lds + (c % BNSTAGE) * BSTAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *(const f32x4 *)(As + slot[s]);
                f0[s] = *(const f32x4 *)(As + BCM * BCK + slot[s]);
                if constexpr (BCN == 64)
                    f1[s] = *(const f32x4 *)(As + (BCM + 32) * BCK + slot[s]);
            }
            if constexpr (BNSTAGE < 3)
                raw_barrier();   // B_c (waits for the fragment reads first)
            if constexpr (BF16) {
                // the lane's eight k of the chunk slice as one bf16 fragment
                const bf16x8 ba = pcm_pack_bf16(fa[0], fa[1]);
                acc_main = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(f0[0], f0[1]), acc_main, 0, 0, 0);
                if constexpr (BCN == 64)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        ba, pcm_pack_bf16(f1[0], f1[1]), acc1, 0, 0, 0);
                return;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc_main = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[s][e], f0[s][e], acc_main, 0, 0, 0);
                    if constexpr (BCN == 64)
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            fa[s][e], f1[s][e], acc1, 0, 0, 0);
                }
        };
        // two straight loops so that each accumulator is updated in place
        for (int c = 0; c < nchunks; ++c) chunk(c, acc0);
        if constexpr (XPART)
            for (int c = nchunks; c < ntot; ++c) chunk(c, acc1);
    }

    __syncthreads();
    float *red = lds;  // [wave][tile][32][32]
    if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float *base = red + (wave * NTL) * 1024 + row * 32 + lrow;
            base[0] = acc0[r];
            if constexpr (NTL == 2) base[1024] = acc1[r];
        }
    }
    __syncthreads();

#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = tid + 512 * q;       // 0..32*BCN-1
        const int tl = e >> 10, ee = e & 1023;
        const int row = ee >> 5, col = ee & 31;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v += red[(w * NTL + tl) * 1024 + ee];
        const int b = b0 + row, n = n0 + tl * 32 + col;
        if (b >= a.B) continue;
        if (XPART && tl == 1) {
            // partial xkp of the input-side gradient (stack top: no dropout,
            // no layer below)
            const int j = xct * 32 + col;
            if (j < a.dx_cols)
                a.dx[(long long)xkp * a.xpart_stride + (long long)b * a.ldx + j] = v;
            continue;
        }
        if (hidden_side) {
            a.dh[(long long)b * a.Hp + n] = pre_d[q] + v;
        } else {
            const int j = n - H;
            if (j >= a.dx_cols) continue;
            if (a.drop_p > 0.f)
                v *= pcm_dropout_scale(a.seed, a.layer, a.t, b + a.b_off, j, a.drop_p,
                                       a.inv_keep);
            float *dst = a.dx + (long long)b * a.ldx + j;
            if (a.gates_low) {
                // v completes dL/dh_t of the layer below: take its gate
                // derivatives right here (saves a launch per step)
                const float d = pre_d[q] + v;
                float *g = a.gates_low + (long long)b * 4 * H;
                if (a.t >= pre_len[q & 1]) {
                    g[j] = 0.f; g[H + j] = 0.f; g[2 * H + j] = 0.f;
                    g[3 * H + j] = 0.f;
                    *dst = d;
                } else {
                    gate_grad_vals(d, pre_r[q], pre_z[q], pre_n[q], pre_g[q],
                                   pre_h[q], g, H, j, dst);
                }
            } else {
                *dst = a.dx_accumulate ? pre_d[q] + v : v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void packT_kernel(PackTArgs a) {
    // wpackT[n][k], k in [0,3H): n < H -> W_hh[k][n]; n >= H -> W_ih[k][x_off+n-H]
    const int K3 = 3 * a.H;
    const long long total = (long long)(a.H + a.Kxp) * K3;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        // idx enumerates (k, n) with n fastest so the READS are coalesced
        const int n = (int)(idx % (a.H + a.Kxp));
        const int k = (int)(idx / (a.H + a.Kxp));
        float v = 0.f;
        if (n < a.H) v = a.w_hh[(long long)k * a.H + n];
        else if (n - a.H < a.Kx)
            v = a.w_ih[(long long)k * a.ld_ih + a.x_off + (n - a.H)];
        a.wpackT[(long long)n * K3 + k] = v;
    }
}

__global__ __launch_bounds__(256) void dropout_add_kernel(
    float *dst, long long ldd, const float *src, long long lds_, int B, int H,
    uint64_t seed, int layer, int t, float p, float inv_keep) {
    const long long total = (long long)B * H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / H), j = (int)(idx - (long long)b * H);
        dst[(long long)b * ldd + j] +=
            src[(long long)b * lds_ + j] *
            pcm_dropout_scale(seed, layer, t, b, j, p, inv_keep);
    }
}

__global__ __launch_bounds__(256) void onehot_kernel(const int64_t *tokens,
                                                     long long rows, int V,
                                                     float *out) {
    const long long total = rows * V;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const long long r = idx / V;
        const int v = (int)(idx - r * V);
        long long tk = tokens[r];
        tk = tk < 0 ? 0 : (tk >= V ? V - 1 : tk);
        out[idx] = (tk == v) ? 1.f : 0.f;
    }
}

// [rows][128]: columns 0-63 the stack top fed to layer 0 (Wp <= 64 wide, zero
// beyond), columns 64-127 the one-hot token -- the joint right-hand operand of
// the stack-top and token-table weight gradients
__global__ __launch_bounds__(256) void xcat_kernel(const float *xtop, int Wp,
                                                   const int64_t *tokens,
                                                   long long rows, int V,
                                                   float *out) {
    const long long total = rows * 32;   // float4 groups
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const long long r = idx >> 5;
        const int c = (int)(idx & 31) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (c + i < Wp) v[i] = xtop[r * Wp + c + i];
        } else {
            long long tk = tokens[r];
            tk = tk < 0 ? 0 : (tk >= V ? V - 1 : tk);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                v[i] = (tk == c - 64 + i) ? 1.f : 0.f;
        }
        *(f32x4 *)(out + r * 128 + c) = v;
    }
}

int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int pcm_launch_gate_bwd(const GateBwdArgs &a, hipStream_t stream) {
    PCM_LAUNCH(gate_bwd_kernel, dim3((a.H + 255) / 256, a.B), dim3(256),
                       0, stream, a);
    return 0;
}

// Narrow (32-column) tiles with a third LDS stage are the default: half the MFMA
// time per block and two or more blocks per CU measured faster than the wide
// (64-column, two-stage) variant at every batch tried (ms per training step,
// wide -> narrow: batch 512 29.4 -> 28.7, 768 43.9 -> 40.0, 1024 49.5 -> 48.3,
// 2048 88.3 -> 86.8).  PCM_BWD_TILE=64 selects the wide one.
static bool bwd_narrow(const CellBwdArgs &a) {
    static const int force = [] {
        const char *e = getenv("PCM_BWD_TILE");
        return e ? atoi(e) : 0;
    }();
    (void)a;
    return force != 64;
}

// Number of K-part partials a layer-0 launch leaves its input-side gradient
// in (1 = the plain result): the XPART variant, for narrow tiles whose
// input-side columns divide the hidden-side column tiles evenly.
int pcm_cell_bwd_xparts(const CellBwdArgs &a) {
    static const bool on = [] {
        const char *e = getenv("PCM_BWD_XPART");
        return !e || atoi(e) != 0;
    }();
    const int kx = a.NT - a.H;
    if (!on || !bwd_narrow(a) || a.gates_low || a.dx_accumulate || kx <= 0 ||
        a.drop_p > 0.f || a.H % kx != 0)
        return 1;
    const int parts = a.H / kx;              // = (H/32) / (kx/32)
    return (3 * a.H / BCK) % parts == 0 ? parts : 1;
}

bool pcm_cell_bwd_takes_rider(const CellBwdArgs &a, int D, int W) {
    const int DWp = (D * W + 3) & ~3;
    // (the narrow tile's three stages: 3 * (32 + 32) * 64 floats)
    return bwd_narrow(a) && 2 * DWp + 32 <= 3 * (BCM + 32) * BCK;
}

int pcm_launch_cell_bwd(const CellBwdArgs &a, hipStream_t stream) {
    const int nbt = (a.B + BCM - 1) / BCM;
    const bool narrow = bwd_narrow(a);
    CellBwdArgs b = a;
    b.xparts = pcm_cell_bwd_xparts(a);
    const bool xpart = b.xparts > 1;
    const int nct = (xpart ? a.H : a.NT) / (narrow ? 32 : 64);
    b.reg_cols = (nct + 3) / 4;
    b.reg_rows = (nbt + 1) / 2;
    dim3 grid(8 * b.reg_cols * b.reg_rows);
    if (a.rider_rows > 0) {
        PCM_REQUIRE(pcm_cell_bwd_takes_rider(a, a.rider.D, a.rider.W), PCM_EINVAL,
                    "cell_bwd: this launch cannot carry rider blocks");
        static const int per = [] {
            const char *e = getenv("PCM_STACK_RIDER_PER");
            return e ? std::max(1, atoi(e)) : 1;
        }();
        b.ncell = (int)grid.x;
        b.rider_per = per;
        const dim3 rgrid(grid.x + (a.rider_rows + per - 1) / per);
        if (g_pcm_matmul_bf16) {
            if (xpart)
                PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, true, true, true>), rgrid,
                           dim3(512), 0, stream, b);
            else
                PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, false, true, true>), rgrid,
                           dim3(512), 0, stream, b);
        } else if (xpart) {
            PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, true, false, true>), rgrid,
                       dim3(512), 0, stream, b);
        } else {
            PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, false, false, true>), rgrid,
                       dim3(512), 0, stream, b);
        }
        return 0;
    }
    if (g_pcm_matmul_bf16) {
        if (xpart)
            PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, true, true>), grid, dim3(512),
                       0, stream, b);
        else if (narrow)
            PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, false, true>), grid,
                       dim3(512), 0, stream, b);
        else
            PCM_LAUNCH((gru_cell_bwd_kernel<64, 2, false, true>), grid,
                       dim3(512), 0, stream, b);
        return 0;
    }
    if (xpart)
        PCM_LAUNCH((gru_cell_bwd_kernel<32, 3, true>), grid, dim3(512), 0,
                           stream, b);
    else if (narrow)
        PCM_LAUNCH((gru_cell_bwd_kernel<32, 3>), grid, dim3(512), 0,
                           stream, b);
    else
        PCM_LAUNCH((gru_cell_bwd_kernel<64, 2>), grid, dim3(512), 0,
                           stream, b);
    return 0;
}

int pcm_launch_packT(const PackTArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(packT_kernel,
                       dim3(grid_for((long long)(a.H + a.Kxp) * 3 * a.H)),
                       dim3(256), 0, stream, a);
    return 0;
}

// LDS bytes one batch row of the stack adjoint stages (saved stack, running
// stack gradient, controls and their gradients, wave partials)
static size_t stack_bwd_row_bytes(int D, int W, int AVW) {
    const int DWp = (D * W + 3) & ~3;
    return (size_t)(2 * DWp + 2 * AVW + 3 * 8) * sizeof(float);
}

int pcm_stack_bwd_check(int D, int W, int AVW) {
    PCM_REQUIRE(stack_bwd_row_bytes(D, W, AVW) <= PCM_STACK_LDS_BYTES, PCM_EINVAL,
                "stack step (backward): stack %dx%d exceeds the %d KiB LDS row "
                "cache (needs (2*D*W + 2*W + 30) * 4 bytes per row)", D, W,
                PCM_STACK_LDS_BYTES / 1024);
    return PCM_OK;
}

int pcm_launch_stack_bwd(const StackBwdArgs &a, hipStream_t stream) {
    const size_t row_bytes = stack_bwd_row_bytes(a.D, a.W, a.AVW);
    int rc = pcm_stack_bwd_check(a.D, a.W, a.AVW);
    if (rc) return rc;
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                                        dev);
        return n;
    }();
    // one row per block while that is at most two blocks per CU (batch 512:
    // 8.2 us against 9.5 for two rows), else two
    const int rows = a.gate_mode ? (a.gate_B > a.B ? a.gate_B : a.B) : a.B;
    if (rows <= 2 * cus || 2 * row_bytes > PCM_STACK_LDS_BYTES)
        PCM_LAUNCH(stack_bwd_kernel<1>, dim3(rows), dim3(BTHREADS),
                           row_bytes, stream, a);
    else
        PCM_LAUNCH(stack_bwd_kernel<2>, dim3((rows + 1) / 2),
                           dim3(BTHREADS), 2 * row_bytes, stream, a);
    return 0;
}

int pcm_launch_stack_bwd_finish(const StackBwdArgs &a, hipStream_t stream) {
    PCM_REQUIRE(a.psum && a.W <= 61 && a.AVW <= 64, PCM_EINVAL,
                "stack_bwd_finish: needs the bulk's sums and W <= 61");
    const int rows = a.gate_mode ? (a.gate_B > a.B ? a.gate_B : a.B) : a.B;
    static const int per = [] {
        // measured at batch 512 (us per launch): 1 row 7.0, 2 rows 16.4, 4 rows
        // 17.1, 8 rows 21.0 -- as with the fused kernel, one row per block
        const char *e = getenv("PCM_STACK_FINISH_ROWS");
        return e ? atoi(e) : 1;
    }();
    if (per >= 8)
        PCM_LAUNCH(stack_bwd_finish_kernel<8>, dim3((rows + 7) / 8), dim3(BTHREADS),
                   0, stream, a);
    else if (per >= 4)
        PCM_LAUNCH(stack_bwd_finish_kernel<4>, dim3((rows + 3) / 4), dim3(BTHREADS),
                   0, stream, a);
    else if (per >= 2)
        PCM_LAUNCH(stack_bwd_finish_kernel<2>, dim3((rows + 1) / 2), dim3(BTHREADS),
                   0, stream, a);
    else
        PCM_LAUNCH(stack_bwd_finish_kernel<1>, dim3(rows), dim3(BTHREADS), 0, stream,
                   a);
    return 0;
}

int pcm_launch_dropout_add(float *dst, long long ldd, const float *src,
                           long long lds_, int B, int H, uint64_t seed,
                           int layer, int t, float p, hipStream_t stream) {
    hipLaunchKernelGGL(dropout_add_kernel, dim3(grid_for((long long)B * H)),
                       dim3(256), 0, stream, dst, ldd, src, lds_, B, H, seed,
                       layer, t, p, 1.0f / (1.0f - p));
    return 0;
}

int pcm_launch_xcat(const float *xtop, int Wp, const int64_t *tokens,
                    long long rows, int V, float *out, hipStream_t stream) {
    hipLaunchKernelGGL(xcat_kernel, dim3(grid_for(rows * 32)), dim3(256), 0,
                       stream, xtop, Wp, tokens, rows, V, out);
    return 0;
}

int pcm_launch_onehot(const int64_t *tokens, long long rows, int V, float *out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(onehot_kernel, dim3(grid_for(rows * V)), dim3(256), 0,
                       stream, tokens, rows, V, out);
    return 0;
}
