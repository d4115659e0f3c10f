// Reverse-recurrence kernels of the StackGRU (what autograd replays for
// training.py:153 over stack_rnn.py:131-232).
//
//   gate_bwd   elementwise: dL/dh_t + saved (r,z,n,ghn,h_{t-1}) -> gate
//              pre-activation gradients (written over the saved gates, they
//              feed both the per-step dgates.W products and the time-parallel
//              weight-gradient GEMMs) and the direct dL/dh_{t-1} = dL/dh_t * z
//   stack_bwd  adjoint of the push/pop/no-op update, control softmax and push
//              tanh; adds (dc.Wc + du.Ws) into dL/dh_{t-1} of the top layer
#include "stackgru.h"

namespace {

__global__ __launch_bounds__(256) void gate_bwd_kernel(GateBwdArgs a) {
    const long long total = (long long)a.B * a.H;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int b = (int)(idx / a.H), j = (int)(idx - (long long)b * a.H);
        float *g = a.gates + (long long)b * 4 * a.H;
        float *dh = a.dh + (long long)b * a.Hp + j;
        float d = *dh;
        if (a.lengths && a.t >= a.lengths[b]) {
            // frozen row: h_t = h_{t-1}; its (zeroed) output carries no gradient
            g[j] = 0.f; g[a.H + j] = 0.f; g[2 * a.H + j] = 0.f;
            g[3 * a.H + j] = 0.f;
            continue;
        }
        if (a.dout) d += a.dout[(long long)b * a.ld_dout + j];
        const float r = g[j], z = g[a.H + j], n = g[2 * a.H + j],
                    ghn = g[3 * a.H + j];
        const float hp = a.hprev[(long long)b * a.Hp + j];
        const float dn = d * (1.f - z);
        const float dz = d * (hp - n);
        const float dpn = dn * (1.f - n * n);
        const float dpz = dz * z * (1.f - z);
        const float dr = dpn * ghn;
        const float dpr = dr * r * (1.f - r);
        g[j] = dpr;
        g[a.H + j] = dpz;
        g[2 * a.H + j] = dpn;       // d(W_in.x + b_in)
        g[3 * a.H + j] = dpn * r;   // d(W_hn.h + b_hn)
        *dh = d * z;
    }
}

// One wave per batch row, 4 rows per block.
constexpr int BROWS = 4;

__global__ __launch_bounds__(256) void stack_bwd_kernel(StackBwdArgs a) {
    extern __shared__ float sm[];
    // per row: dcu[AVW] (dc0..2, du[W]); a_v[AVW]
    float *dcu = sm;                       // [BROWS][AVW]
    float *avs = sm + BROWS * a.AVW;       // [BROWS][AVW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x * BROWS + wave;
    const int W = a.W, D = a.D, DW = D * W, H = a.H;
    const bool valid = b < a.B;
    const bool frozen = valid && a.lengths && a.t >= a.lengths[b];

    if (valid) {
        for (int o = lane; o < a.AVW; o += 64)
            avs[wave * a.AVW + o] = a.av[(long long)b * a.AVW + o];
    }
    __syncthreads();
    if (valid) {
        const float *Sp = a.S_prev + (long long)b * DW;
        const float *Gn = a.dS_next ? a.dS_next + (long long)b * DW : nullptr;
        const float *dtop = a.dtop + (long long)b * a.Wp;
        float *dSp = a.dS_prev + (long long)b * DW;
        const float *c = avs + wave * a.AVW;
        const float a0 = c[0], a1 = c[1], a2 = c[2];
        if (frozen) {
            for (int e = lane; e < DW; e += 64) dSp[e] = Gn ? Gn[e] : 0.f;
            for (int o = lane; o < a.AVW; o += 64) {
                dcu[wave * a.AVW + o] = 0.f;
                a.av[(long long)b * a.AVW + o] = 0.f;
            }
        } else {
            // G = dL/dS'_t with the stack-top gradient added to row 0
            float s_push = 0.f, s_pop = 0.f, s_noop = 0.f;
            for (int e = lane; e < DW; e += 64) {
                const int d = e / W, w = e - d * W;
                const float g = (Gn ? Gn[e] : 0.f) + (d == 0 ? dtop[w] : 0.f);
                const float s = Sp[e];
                const float up = d == 0 ? c[3 + w] : Sp[e - W];
                const float dn = d == D - 1 ? 0.f : Sp[e + W];
                s_push = fmaf(g, up, s_push);
                s_pop = fmaf(g, dn, s_pop);
                s_noop = fmaf(g, s, s_noop);
                // adjoint of the shifts: S[d] feeds S'[d] (no-op), S'[d+1]
                // (push) and S'[d-1] (pop)
                float v = a2 * g;
                if (d + 1 < D) v += a0 * (Gn ? Gn[e + W] : 0.f);
                if (d >= 1)
                    v += a1 * ((Gn ? Gn[e - W] : 0.f) +
                               (d == 1 ? dtop[w] : 0.f));
                dSp[e] = v;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                s_push += __shfl_xor(s_push, off, 64);
                s_pop += __shfl_xor(s_pop, off, 64);
                s_noop += __shfl_xor(s_noop, off, 64);
            }
            // softmax backward: dc_i = a_i * (da_i - sum_j a_j da_j)
            const float dot = a0 * s_push + a1 * s_pop + a2 * s_noop;
            float *out = dcu + wave * a.AVW;
            if (lane == 0) {
                out[0] = a0 * (s_push - dot);
                out[1] = a1 * (s_pop - dot);
                out[2] = a2 * (s_noop - dot);
            }
            // dv[w] = a_push * G[0][w]; du = dv * (1 - v^2)
            for (int w = lane; w < W; w += 64) {
                const float g0 = (Gn ? Gn[w] : 0.f) + dtop[w];
                const float v = c[3 + w];
                out[3 + w] = a0 * g0 * (1.f - v * v);
            }
            for (int o = 3 + W + lane; o < a.AVW; o += 64) out[o] = 0.f;
        }
    }
    __syncthreads();
    if (valid && !frozen) {
        float *out = dcu + wave * a.AVW;
        // keep dc/du for the time-parallel dWc / dWs products
        for (int o = lane; o < a.AVW; o += 64)
            a.av[(long long)b * a.AVW + o] = out[o];
        // dL/dh_top += dc.Wc + du.Ws
        float *dh = a.dh_top + (long long)b * a.Hp;
        for (int j = lane; j < H; j += 64) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 3; ++o)
                s = fmaf(out[o], a.ctrl_w[(long long)o * H + j], s);
            for (int w = 0; w < W; ++w)
                s = fmaf(out[3 + w], a.push_w[(long long)w * H + j], s);
            dh[j] += s;
        }
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

int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int pcm_launch_gate_bwd(const GateBwdArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for((long long)a.B * a.H)),
                       dim3(256), 0, stream, a);
    return 0;
}

int pcm_launch_stack_bwd(const StackBwdArgs &a, hipStream_t stream) {
    const size_t lds = (size_t)2 * BROWS * a.AVW * sizeof(float);
    hipLaunchKernelGGL(stack_bwd_kernel, dim3((a.B + BROWS - 1) / BROWS),
                       dim3(256), lds, stream, a);
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

int pcm_launch_onehot(const int64_t *tokens, long long rows, int V, float *out,
                      hipStream_t stream) {
    hipLaunchKernelGGL(onehot_kernel, dim3(grid_for(rows * V)), dim3(256), 0,
                       stream, tokens, rows, V, out);
    return 0;
}
