// Forward kernels of the StackGRU recurrence (stack_rnn.py:131-232).
//
// Per time step:
//   stack_step_fwd  a = softmax(Wc.h_top+bc), v = tanh(Ws.h_top+bs), the fused
//                   push/pop/no-op update of the [D,W] stack (1 read + 1 write
//                   of the stack, HBM-bound) and the stack top for layer 0
//   gru_cell_fwd    one GRU layer: [x | h] . Wpack^T on fp32 MFMA (32x32x2),
//                   LDS-staged 64-wide K chunks, intra-block split-K over the
//                   4 waves, gate math + token-gate gather in the epilogue
#include <stdlib.h>

#include "stackgru.h"

namespace {

// ------------------------------------------------------------------------------
// fused GRU cell
// ------------------------------------------------------------------------------
constexpr int CK = 64;        // K per chunk
constexpr int CM = 32;        // batch rows per block
constexpr int CN = 32;        // hidden units per block (x3 gates)

// Wave-specialised: waves 0-3 ("consumers", one per SIMD) only read operand
// fragments from LDS and issue MFMAs, each owning a 16-wide k slice of every
// 64-wide chunk (intra-block split-K, LDS reduce at the end); waves 4-7
// ("loaders", the SIMD partners) stream the chunks with LDS-DMA loads
// (global_load_lds_dwordx4) that land in LDS directly -- no staging registers,
// no ds_write pass, no vector address arithmetic.  An LDS-DMA wave instruction
// writes 1 KiB lane-linear, so rows are unpadded (256 B) and bank conflicts are
// avoided by an XOR swizzle of the 16-byte granule index with (row & 15),
// applied on the SOURCE address by the loaders and on the LDS address by the
// consumers.  The loaders keep one chunk in flight across the raw s_barriers
// (counted vmcnt, never __syncthreads inside the loop, which would drain the
// DMA queue).
constexpr int GSTAGE = (CM + 3 * CN) * CK;   // floats per stage (32 KiB)

// Eight LDS-DMA loads of one loader wave: lane offsets off[i] (bytes) from the
// uniform base, destinations lds_byte + 1 KiB * i.  Inline asm on purpose: hipcc
// drains the DMA queue (vmcnt(0)) between builtin calls whose LDS destinations
// it cannot tell apart; the waits are counted by hand in the caller.  M0 (the
// DMA's LDS base) is saved and restored inside the statement.
__device__ __forceinline__ void glds8(const char *base, const uint32_t (&off)[8],
                                      uint32_t lds_byte) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %10, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(base), "s"(lds_byte), "v"(off[0]), "v"(off[1]), "v"(off[2]),
          "v"(off[3]), "v"(off[4]), "v"(off[5]), "v"(off[6]), "v"(off[7])
        : "memory", "scc");
}

__device__ __forceinline__ void raw_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Gate math of one (batch row, hidden unit) from its four raw sums
// s = {r, z, n (input side), n (hidden side)}: token-gate gather, biases,
// sigma / tanh, (h - n) z + n, length freeze; then (cell_fwd_store) gate save,
// new hidden state, dropout of the next layer's input.
struct CellFwdOut {
    float r, z, n, ghn, hn;
};

__device__ __forceinline__ CellFwdOut cell_fwd_math(const CellFwdArgs &a, int b,
                                                    int j, const float (&s)[4]) {
    const int Hn = a.Hn;
    float tr = 0.f, tz = 0.f, tn = 0.f;
    if (a.tokg) {
        long long tk = a.tok[b];
        if (tk < 0 || tk >= a.V) {
            tk = tk < 0 ? 0 : a.V - 1;
            if (a.status) atomicOr(a.status, PCM_STATUS_BAD_TOKEN);
        }
        const float *tg = a.tokg + tk * (3LL * Hn);
        tr = tg[j]; tz = tg[Hn + j]; tn = tg[2 * Hn + j];
    }
    CellFwdOut o;
    o.r = pcm_sigmoid(s[0] + a.bpack[j] + tr);
    o.z = pcm_sigmoid(s[1] + a.bpack[Hn + j] + tz);
    o.ghn = s[3] + a.bpack[3 * Hn + j];
    o.n = tanhf(s[2] + a.bpack[2 * Hn + j] + tn + o.r * o.ghn);
    const float hp = a.hprev[(long long)b * a.Hp + j];
    // ATen GRU cell: (h - n) * z + n
    o.hn = __fadd_rn(__fmul_rn(hp - o.n, o.z), o.n);
    if (a.lengths && a.t >= a.lengths[b]) o.hn = hp;
    return o;
}

__device__ __forceinline__ void cell_fwd_store(const CellFwdArgs &a, int b, int j,
                                               const CellFwdOut &o) {
    const int H = a.H;
    a.hnew[(long long)b * a.Hp + j] = o.hn;
    if (a.gates) {
        float *g = a.gates + (long long)b * 4 * H;
        // read again only by the reverse pass: streaming stores
        __builtin_nontemporal_store(o.r, g + j);
        __builtin_nontemporal_store(o.z, g + H + j);
        __builtin_nontemporal_store(o.n, g + 2 * H + j);
        __builtin_nontemporal_store(o.ghn, g + 3 * H + j);
    }
    if (a.xdrop)
        a.xdrop[(long long)b * a.Hp + j] =
            o.hn * pcm_dropout_scale(a.seed, a.next_layer, a.t, b + a.b_off, j,
                                     a.drop_p, a.inv_keep);
}

__device__ __forceinline__ void cell_fwd_element(const CellFwdArgs &a, int b,
                                                 int j, const float (&s)[4]) {
    cell_fwd_store(a, b, j, cell_fwd_math(a, b, j, s));
}

// Cross-wave reduction of the four K-slice partials through LDS, then the GRU
// cell epilogue for the 32 x 32 tile (two outputs per thread).
// hz: the hoisted hidden-side sums (r, z, n) of this thread's two outputs
// (CellFwdArgs::hraw; zeros when the launch computed them itself).  raw_out: an
// extra block of a layer-0 launch (CellFwdArgs::hz_raw) -- the reduced r, z,
// n(hidden) sums are stored as they are and nothing else happens.
__device__ __forceinline__ void cell_fwd_tail(const CellFwdArgs &a, float *lds,
                                              const f32x16 &acc_r,
                                              const f32x16 &acc_z,
                                              const f32x16 &acc_ni,
                                              const f32x16 &acc_nh, int tid,
                                              int wave, int lrow, int lh,
                                              int j0, int b0,
                                              const f32x4 (&wf)[2],
                                              const float (&hz)[2][3],
                                              bool raw_out = false) {
    // cross-wave reduction of the 4 partial tiles through LDS
    __syncthreads();
    float *red = lds;  // [wave][tile][32][32]
    if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            float *base = red + (wave * 4) * 1024 + row * 32 + lrow;
            base[0 * 1024] = acc_r[r];
            base[1 * 1024] = acc_z[r];
            base[2 * 1024] = acc_ni[r];
            base[3 * 1024] = acc_nh[r];
        }
    }
    __syncthreads();

    const int H = a.H;
    CellFwdOut out[2];
    bool live[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 512 * q;
        const int row = e >> 5, col = e & 31;
        const int b = b0 + row, j = j0 + col;
        float s[4];
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * 4 + tl) * 1024 + e];
            s[tl] = v;
        }
        live[q] = b < a.B && j < H;
        out[q].hn = 0.f;
        if (raw_out) {
            if (live[q]) {
                float *dst = a.hz_raw + (long long)b * 3 * a.Hn + j;
                dst[0] = s[0]; dst[a.Hn] = s[1]; dst[2 * a.Hn] = s[3];
            }
            continue;
        }
        s[0] += hz[q][0]; s[1] += hz[q][1]; s[3] += hz[q][2];
        if (live[q]) out[q] = cell_fwd_math(a, b, j, s);
    }
    if (raw_out) return;
    f32x4 part = {0.f, 0.f, 0.f, 0.f};
    if (a.cs_part) {
        // Top layer with a stack: this tile's share of the next stack step's
        // 3 + W control / push pre-activations, [32 rows x 32 k] . [32 k x 64
        // outputs] as eight 16 x 16 tiles, one per wave (loaders included: row
        // half w & 1, output tile w >> 1; 16x16x4 MFMAs, weight fragments in wf
        // since kernel start), stored as partial number `column tile`, which
        // the stack kernel sums in order -- it then no longer streams the whole
        // [Wc ; Ws] per batch row.  Done BEFORE the epilogue's stores, between
        // raw barriers (LDS traffic only), so nothing here waits for a global
        // store; four stores per lane (two waves with 32 x 32 tiles and sixteen
        // stores per lane cost the launch 1.2 us, all of it in those stores).
        constexpr int HT = 36;     // row stride of the tile (conflict-free b128)
        raw_barrier();             // every thread has read its sums from `red`
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 512 * q;
            lds[(e >> 5) * HT + (e & 31)] = out[q].hn;
        }
        raw_barrier();
        const int r16 = tid & 15, g = (tid >> 4) & 3;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const f32x4 fa = *(const f32x4 *)(lds + (16 * (wave & 1) + r16) * HT +
                                              16 * jj + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                part = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], wf[jj][e], part,
                                                            0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 512 * q;
        if (live[q]) cell_fwd_store(a, b0 + (e >> 5), j0 + (e & 31), out[q]);
    }
    if (a.cs_part) {
        // D[i][j]: lane (r16, g) holds rows i = 4 g + e of the half, column r16
        const int r16 = tid & 15, g = (tid >> 4) & 3;
        const int parts = a.Hn / CN;
        float *dst = a.cs_part + ((long long)b0 * parts + j0 / CN) * 64 +
                     16 * (wave >> 1) + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * (wave & 1) + 4 * g + e;
            if (b0 + row < a.B) dst[(long long)row * parts * 64] = part[e];
        }
    }
}

// weight fragments of cell_fwd_tail's control partials: lane (r16, g) of wave w
// holds, for jj = 0, 1, the four k = j0 + 16 jj + 4 g + 0..3 of output
// 16 (w >> 1) + r16 (rows 0-2: controls, 3 .. 2 + W: push, beyond: zero)
typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void cell_fwd_ctrl_frags(const CellFwdArgs &a,
                                                    int wave, int lane, int j0,
                                                    f32x4 (&wf)[2]) {
    wf[0] = wf[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!a.cs_part) return;
    const int o = 16 * (wave >> 1) + (lane & 15), g = lane >> 4;
    if (o >= 3 + a.cs_W) return;
    const float *wrow = (o < 3 ? a.cs_ctrl_w + (long long)o * a.H
                               : a.cs_push_w + (long long)(o - 3) * a.H) + j0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
        wf[jj] = *(const f32x4a4 *)(wrow + 16 * jj + 4 * g);
}

typedef float f32x4u4 __attribute__((ext_vector_type(4), aligned(4)));

// Stack step fused into layer 0's cell (CellFwdArgs::xt_*): the block's 32 rows'
// control / push pre-activations are the bias plus the partial sums the top
// cell of the step before left (one [row][64] per column tile, summed in
// order), then softmax / tanh and the NEW stack top
//   top = no_op * S[0] + push * v + pop * S[1]      (stack_rnn.py:229-231 order)
// which goes straight into the activation rows of stage 0 (the input-side chunk,
// granule-swizzled like a DMA'd chunk).  16 threads per row, four outputs each.
// Column tile 0 records controls + push value and the stack top.  Every wave
// calls this once, before its loop (two raw barriers inside).
__device__ __forceinline__ void cell_fwd_xtop_prologue(const CellFwdArgs &a,
                                                       float *lds, int tid,
                                                       int b0, int ct) {
    // (all eight waves, one row per 16 threads: the consumer waves alone, two
    // rows per thread, measured 2.4 us slower per launch)
    const int r = tid >> 4, og = tid & 15;
    const int W = a.xt_W, D = a.xt_D;
    const bool valid = b0 + r < a.B;
    const int b = min(b0 + r, a.B - 1);
    // the two stack rows the top depends on, this thread's four columns
    const float *S0 = a.xt_S + (long long)b * D * W;
    float s0[4], s1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int w = 4 * og + i;
        s0[i] = w < W ? S0[w] : 0.f;
        s1[i] = (w < W && D > 1) ? S0[W + w] : 0.f;
    }
    // (all partial loads in flight at once: a rolled loop would pay one L2
    // round trip per partial -- measured 7 us for 16 of them)
    const float *pp = a.xt_part + (long long)b * a.xt_parts * 64 + 4 * og;
    f32x4 pv[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
        pv[p] = p < a.xt_parts ? *(const f32x4 *)(pp + p * 64)
                               : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < 16; ++p) sum += pv[p];
    for (int p = 16; p < a.xt_parts; ++p) sum += *(const f32x4 *)(pp + p * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int o = 4 * og + i;
        sum[i] += o < 3 ? a.xt_ctrl_b[o] : (o < 3 + W ? a.xt_push_b[o - 3] : 0.f);
    }
    float *pre = lds;   // [32][64]: stage 0's activation rows, plain layout
    *(f32x4 *)(pre + r * 64 + 4 * og) = sum;
    raw_barrier();
    const float *c = pre + r * 64;
    const float m = fmaxf(c[0], fmaxf(c[1], c[2]));
    const float e0 = expf(c[0] - m), e1 = expf(c[1] - m), e2 = expf(c[2] - m);
    const float es = e0 + e1 + e2;
    const float a_push = e0 / es, a_pop = e1 / es, a_noop = e2 / es;
    const bool frozen = valid && a.lengths && a.t >= a.lengths[b];
    f32x4 top, rec;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int w = 4 * og + i, o = w;
        const float v = w < W ? tanhf(c[3 + w]) : 0.f;
        float nv = __fadd_rn(__fadd_rn(__fmul_rn(a_noop, s0[i]),
                                       __fmul_rn(a_push, v)),
                             __fmul_rn(a_pop, s1[i]));
        if (frozen) nv = s0[i];
        top[i] = w < W ? nv : 0.f;
        // record o (column tile 0 only): the three controls, then tanh of the
        // push pre-activations
        rec[i] = 0.f;
        if (ct == 0)
            rec[i] = o == 0 ? a_push
                            : (o == 1 ? a_pop
                                      : (o == 2 ? a_noop
                                                : (o < 3 + W ? tanhf(c[o])
                                                             : 0.f)));
    }
    raw_barrier();   // every thread has read the pre-activations
    *(f32x4 *)(lds + r * 64 + ((og ^ (r & 15)) << 2)) = top;
    if (ct == 0 && valid) {
        *(f32x4 *)(a.xt_xtop + (long long)b * 64 + 4 * og) = top;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * og + i < a.xt_AVW)
                a.xt_av[(long long)b * a.xt_AVW + 4 * og + i] = rec[i];
    }
}

// Deferred stack update (CellFwdArgs::su_*): rows [row0, row0 + su_rows) of
//   S_new = no_op * S + push * [v ; S[:-1]] + pop * [S[1:] ; 0]
// with the controls / push value the fused layer-0 cell recorded; the arithmetic
// (and its order) is stack_step_fwd_kernel's.
__device__ __forceinline__ void cell_fwd_stack_update(const CellFwdArgs &a,
                                                      int row0, int tid) {
    const int W = a.su_W, D = a.su_D, DW = D * W;
    for (int rr = 0; rr < a.su_rows; ++rr) {
        const int b = row0 + rr;
        if (b >= a.B) return;
        const float *Sp = a.su_S + (long long)b * DW;
        float *Sn = a.su_Snew + (long long)b * DW;
        const float *c = a.su_av + (long long)b * a.su_AVW;
        const bool frozen = a.lengths && a.t >= a.lengths[b];
        const float c0 = c[0], c1 = c[1], c2 = c[2];
        if ((DW & 3) == 0) {
            for (int q = tid; q < (DW >> 2); q += 512) {
                const int e0 = q * 4;
                const f32x4 sv = *(const f32x4 *)(Sp + e0);
                f32x4 nv = sv;
                if (!frozen) {
                    f32x4u4 up4 = {0.f, 0.f, 0.f, 0.f}, dn4 = up4;
                    if (e0 >= W) up4 = *(const f32x4u4 *)(Sp + e0 - W);
                    if (e0 + 3 + W < DW) dn4 = *(const f32x4u4 *)(Sp + e0 + W);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = e0 + i;
                        const float up =
                            e < W ? c[3 + e] : (e0 >= W ? up4[i] : Sp[e - W]);
                        const float dn =
                            e + W >= DW ? 0.f
                                        : (e0 + 3 + W < DW ? dn4[i] : Sp[e + W]);
                        nv[i] = __fadd_rn(__fadd_rn(__fmul_rn(c2, sv[i]),
                                                    __fmul_rn(c0, up)),
                                          __fmul_rn(c1, dn));
                    }
                }
                *(f32x4 *)(Sn + e0) = nv;
            }
        } else {
            for (int e = tid; e < DW; e += 512) {
                const float sv = Sp[e];
                float nv = sv;
                if (!frozen) {
                    const float up = e < W ? c[3 + e] : Sp[e - W];
                    const float dn = e + W >= DW ? 0.f : Sp[e + W];
                    nv = __fadd_rn(__fadd_rn(__fmul_rn(c2, sv), __fmul_rn(c0, up)),
                                   __fmul_rn(c1, dn));
                }
                Sn[e] = nv;
            }
        }
    }
}

// Two 32 KiB stages, two raw barriers per chunk (A: chunk landed, B: fragments
// read, stage free), one fragment set, accumulators updated in place in two
// straight loops (input part, hidden part).  Deeper pipelines (3-4 stages, one
// barrier per chunk, fragment double buffering) were measured and are slower:
// the MFMA loop already sits at its issue bound, and at 64 KiB LDS / 86 VGPRs
// two blocks fit a CU, so with more blocks than CUs (large batches) one block's
// prologue and epilogue run under the other's main loop.
constexpr int CNSTAGE = 2;
static_assert(CNSTAGE * GSTAGE >= 4 * 4 * 1024, "reduction buffer must fit");

// XT: with the stack step's prologue (CellFwdArgs::xt_*, opt-in) -- a separate
// instantiation, so that the default kernel does not carry its registers.
template <bool BF16, bool XT>
__global__ __launch_bounds__(512, 4) void gru_cell_fwd_kernel(CellFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[CNSTAGE * GSTAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.su_S && (int)blockIdx.y >= a.su_nbt) {
        // extra blocks: the step's stack update, off the critical path
        const int id = ((int)blockIdx.y - a.su_nbt) * gridDim.x + blockIdx.x;
        cell_fwd_stack_update(a, id * a.su_rows, tid);
        return;
    }
    const int j0 = blockIdx.x * CN;
    // role H (block-uniform; layer-0 launches only): an extra block computing the
    // hidden-side sums of the layer above for row tile blockIdx.y - hz_nbt
    const bool role_h = XT && a.hz_raw && (int)blockIdx.y >= a.hz_nbt;
    const int b0 = (role_h ? (int)blockIdx.y - a.hz_nbt : (int)blockIdx.y) * CM;
    const int nx_chunks = role_h ? 0 : a.nx_chunks;
    const int nchunks = nx_chunks + (role_h ? a.Hp / CK : a.nh_chunks);
    const int last = nchunks - 1;
    const int lrow = lane & 31, lh = lane >> 5;

    f32x16 acc_r, acc_z, acc_ni, acc_nh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_r[r] = 0.f; acc_z[r] = 0.f; acc_ni[r] = 0.f; acc_nh[r] = 0.f;
    }
    f32x4 wf[2];
    cell_fwd_ctrl_frags(a, wave, lane, j0, wf);
    // the hoisted hidden-side sums of this thread's two outputs, requested now
    float hz[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    if (a.hraw) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 512 * q;
            const int b = min(b0 + (e >> 5), a.B - 1), j = min(j0 + (e & 31), a.Hn - 1);
            const float *src = a.hraw + (long long)b * 3 * a.Hn + j;
            hz[q][0] = src[0]; hz[q][1] = src[a.Hn]; hz[q][2] = src[2 * a.Hn];
        }
    }
    if (wave >= 4) {
        // loaders: same lane -> (row, swizzled granule) map as above
        const int lw = wave - 4;
        const char *xb = (const char *)a.xin;
        const char *hb = (const char *)(role_h ? a.hz_hprev : a.hprev);
        // role H: the weights of the layer above, from the first chunk of their
        // hidden side on
        const char *wb = role_h ? (const char *)a.hz_wpack + a.hz_kx * (CK * 4)
                                : (const char *)a.wpack;
        const int Kp = role_h ? a.hz_Kp : a.Kp;
        uint32_t off0[8], off1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 4 * (lw * 8 + i) + (lane >> 4);
            const int g = (lane & 15) ^ (row & 15);
            if (lw == 0) {
                const int b = min(b0 + row, a.B - 1);
                off0[i] = (uint32_t)(b * a.ldx + g * 4) * 4u;
                off1[i] = (uint32_t)(b * a.Hp + g * 4) * 4u;
            } else {
                const int wr = row - CM, gate = wr >> 5, jj = wr & 31;
                off0[i] = (uint32_t)((gate * a.Hn + j0 + jj) * Kp + g * 4) * 4u;
                off1[i] = 0;
            }
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds);
        auto issue = [&](int cc) {
            const int c = min(cc, last);
            const uint32_t dst =
                lds0 + ((cc % CNSTAGE) * GSTAGE + lw * 8 * 256) * 4u;
            if (lw == 0) {
                if (c < nx_chunks) {
                    // (fused stack step: the prologue writes these rows)
                    if (!XT) glds8(xb + c * (CK * 4), off0, dst);
                } else {
                    glds8(hb + (c - nx_chunks) * (CK * 4), off1, dst);
                }
            } else {
                glds8(wb + c * (CK * 4), off0, dst);
            }
        };
        issue(0); issue(1);
        if constexpr (XT)
            if (!role_h) cell_fwd_xtop_prologue(a, lds, tid, b0, blockIdx.x);
        for (int k = 0; k < nchunks; ++k) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // chunk k landed
            raw_barrier();   // A_k: the consumers may read chunk k
            raw_barrier();   // B_k: they hold its fragments, the stage is free
            issue(k + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if constexpr (XT)
            if (!role_h) cell_fwd_xtop_prologue(a, lds, tid, b0, blockIdx.x);
        int slot[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            slot[s] = lrow * CK + (((4 * wave + lh + 2 * s) ^ (lrow & 15)) << 2);
        // two straight loops (input part, hidden part) so that every accumulator
        // is updated in place -- a branch inside one loop makes hipcc ping-pong
        // the accumulators between two register sets
        auto chunk = [&](int c, f32x16 &acc_n) {
            f32x4 fa[2], fr[2], fz[2], fn[2];
            raw_barrier();   // A_c
            const float *As = lds + (c % CNSTAGE) * GSTAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *(const f32x4 *)(As + slot[s]);
                fr[s] = *(const f32x4 *)(As + (CM + 0 * CN) * CK + slot[s]);
                fz[s] = *(const f32x4 *)(As + (CM + 1 * CN) * CK + slot[s]);
                fn[s] = *(const f32x4 *)(As + (CM + 2 * CN) * CK + slot[s]);
            }
            raw_barrier();   // B_c (waits for the fragment reads first)
            if constexpr (BF16) {
                // the lane's eight k of the chunk slice as ONE bf16 fragment
                // (any lane -> k map serves as long as both operands share it)
                const bf16x8 ba = pcm_pack_bf16(fa[0], fa[1]);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fr[0], fr[1]), acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fz[0], fz[1]), acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fn[0], fn[1]), acc_n, 0, 0, 0);
                return;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[s][e], fr[s][e], acc_r, 0, 0, 0);
                    acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[s][e], fz[s][e], acc_z, 0, 0, 0);
                    acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[s][e], fn[s][e], acc_n, 0, 0, 0);
                }
        };
        for (int c = 0; c < nx_chunks; ++c) chunk(c, acc_ni);
        for (int c = nx_chunks; c < nchunks; ++c) chunk(c, acc_nh);
    }
    cell_fwd_tail(a, lds, acc_r, acc_z, acc_ni, acc_nh, tid, wave, lrow, lh, j0,
                  b0, wf, hz, role_h);
}


// ------------------------------------------------------------------------------
// narrow fused GRU cell: 32 batch rows x 16 hidden units per block
// ------------------------------------------------------------------------------
// Half the work of the 32 x 32 tile per block, for launches that would otherwise
// give each CU at most one block: twice the blocks, two per CU.  Same loader /
// consumer split and LDS-DMA staging; the consumers use 16x16x4 MFMAs -- wave w
// owns the 16-row half (w & 1) of the tile and the half (w >> 1) of every
// 64-wide chunk's K, for all three gates -- so no column of an MFMA tile is
// wasted on the n gate's split accumulation (a 32-wide MFMA tile over 16 hidden
// units would spend a third more MFMA work).  The two K halves meet in LDS, each
// wave finishing two of its four output rows per lane.
constexpr int NCN = 16;                         // hidden units per block
constexpr int NROWS = CM + 3 * NCN;             // stage rows (80)
constexpr int NSTAGE = NROWS * CK;              // floats per stage (20 KiB)
constexpr int NNSTAGE = 2;
constexpr int NNI = NROWS / 16;                 // DMA instructions per loader wave

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds5(const char *const (&base)[5],
                                      const uint32_t (&off)[5],
                                      uint32_t lds_byte) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %3\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %9, %4\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %10, %5\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %11, %6\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_byte), "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3]),
          "s"(base[4]), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]),
          "v"(off[4])
        : "memory", "scc");
}

template <bool BF16>
__global__ __launch_bounds__(512) void gru_cell_fwd16_kernel(CellFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[NNSTAGE * NSTAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * NCN;
    const int b0 = blockIdx.y * CM;
    const int nchunks = a.nx_chunks + a.nh_chunks;
    const int last = nchunks - 1;

    f32x4v acc_r = {0.f, 0.f, 0.f, 0.f}, acc_z = acc_r, acc_ni = acc_r,
           acc_nh = acc_r;
    // control partials (top layer with a stack, see cell_fwd_tail): wave w's
    // weight fragment of output 16 (w >> 1) + r16, k = j0 + 4 g .. + 3
    f32x4v wf = {0.f, 0.f, 0.f, 0.f};
    if (a.cs_part) {
        const int o = 16 * (wave >> 1) + (lane & 15);
        if (o < 3 + a.cs_W)
            wf = *(const f32x4a4 *)((o < 3 ? a.cs_ctrl_w + (long long)o * a.H
                                           : a.cs_push_w + (long long)(o - 3) * a.H) +
                                    j0 + 4 * (lane >> 4));
    }

    if (wave >= 4) {
        // loaders: wave lw issues instructions 5 lw .. 5 lw + 4 of a chunk;
        // instruction q fills stage rows 4q .. 4q+3 (rows 0-31: activations,
        // 32-79: the r, z, n weight rows of the 16 hidden units); lane ->
        // (row 4q + lane / 16, swizzled granule)
        const int lw = wave - 4;
        const char *xb = (const char *)a.xin, *hb = (const char *)a.hprev;
        const char *wb = (const char *)a.wpack;
        uint32_t offx[NNI], offh[NNI];
        bool is_a[NNI];
#pragma unroll
        for (int i = 0; i < NNI; ++i) {
            const int q = lw * NNI + i;
            const int row = 4 * q + (lane >> 4);
            const int g = (lane & 15) ^ (row & 15);
            is_a[i] = q < CM / 4;
            if (is_a[i]) {
                const int b = min(b0 + row, a.B - 1);
                offx[i] = (uint32_t)(b * a.ldx + g * 4) * 4u;
                offh[i] = (uint32_t)(b * a.Hp + g * 4) * 4u;
            } else {
                const int wr = row - CM, gate = wr >> 4, jj = wr & 15;
                offx[i] = (uint32_t)((gate * a.Hn + j0 + jj) * a.Kp + g * 4) * 4u;
                offh[i] = offx[i];
            }
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds);
        auto issue = [&](int cc) {
            const int c = min(cc, last);
            const bool xpart = c < a.nx_chunks;
            const char *abase = xpart ? xb + c * (CK * 4)
                                      : hb + (c - a.nx_chunks) * (CK * 4);
            const char *base[NNI];
            uint32_t off[NNI];
#pragma unroll
            for (int i = 0; i < NNI; ++i) {
                base[i] = is_a[i] ? abase : wb + c * (CK * 4);
                off[i] = xpart ? offx[i] : offh[i];
            }
            glds5(base, off,
                  lds0 + ((cc % NNSTAGE) * NSTAGE + lw * NNI * 256) * 4u);
        };
        issue(0); issue(1);
        for (int k = 0; k < nchunks; ++k) {
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // chunk k landed
            raw_barrier();   // A_k: the consumers may read chunk k
            raw_barrier();   // B_k: they hold its fragments, the stage is free
            issue(k + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // consumers: lane (r16, g) of wave (rh, kh) reads, for s = 0, 1, the
        // 16-byte granule 8 kh + 4 s + g of its row: four consecutive k, one
        // per MFMA e, whose four lane groups together cover 16 of K
        const int r16 = lane & 15, g = lane >> 4;
        const int rh = wave & 1, kh = wave >> 1;
        int slot_a[2], slot_w[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int gran = 8 * kh + 4 * s + g;
            const int ra = rh * 16 + r16;           // activation row
            slot_a[s] = ra * CK + ((gran ^ (ra & 15)) << 2);
            // weight rows CM + 16 gate + r16: (row & 15) == r16 for every gate
            slot_w[s] = (CM + r16) * CK + ((gran ^ r16) << 2);
        }
        auto chunk = [&](int c, f32x4v &acc_n) {
            f32x4v fa[2], fr[2], fz[2], fn[2];
            raw_barrier();   // A_c
            const float *As = lds + (c % NNSTAGE) * NSTAGE;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                fa[s] = *(const f32x4v *)(As + slot_a[s]);
                fr[s] = *(const f32x4v *)(As + slot_w[s]);
                fz[s] = *(const f32x4v *)(As + 1 * NCN * CK + slot_w[s]);
                fn[s] = *(const f32x4v *)(As + 2 * NCN * CK + slot_w[s]);
            }
            raw_barrier();   // B_c (waits for the fragment reads first)
            if constexpr (BF16) {
                const bf16x8 ba = pcm_pack_bf16(fa[0], fa[1]);
                acc_r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    ba, pcm_pack_bf16(fr[0], fr[1]), acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    ba, pcm_pack_bf16(fz[0], fz[1]), acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    ba, pcm_pack_bf16(fn[0], fn[1]), acc_n, 0, 0, 0);
                return;
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc_r = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fa[s][e], fr[s][e], acc_r, 0, 0, 0);
                    acc_z = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fa[s][e], fz[s][e], acc_z, 0, 0, 0);
                    acc_n = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fa[s][e], fn[s][e], acc_n, 0, 0, 0);
                }
        };
        for (int c = 0; c < a.nx_chunks; ++c) chunk(c, acc_ni);
        for (int c = a.nx_chunks; c < nchunks; ++c) chunk(c, acc_nh);
    }

    // The two K halves of a 16-row half meet in LDS: lane element e (row
    // 4 g + e of the half, column r16) is finished by wave kh == (e >> 1), so
    // each consumer wave parks the two rows it does not finish and all four
    // run the gate math, two elements per lane.
    __syncthreads();
    float *red = lds;   // [wave][pair p][tile][lane]
    if (wave < 4) {
        const int kh = wave >> 1;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = 2 * (1 - kh) + p;   // rows finished by the other wave
            float *dst = red + ((wave * 2 + p) * 4) * 64 + lane;
            dst[0 * 64] = acc_r[e]; dst[1 * 64] = acc_z[e];
            dst[2 * 64] = acc_ni[e]; dst[3 * 64] = acc_nh[e];
        }
    }
    __syncthreads();
    constexpr int HT = 20;                 // row stride of the hidden tile
    float *tile = lds + 8 * 4 * 64;        // behind the reduction buffer
    CellFwdOut out[2];
    bool live[2] = {false, false};
    int ob[2] = {0, 0};
    if (wave < 4) {
        const int r16 = lane & 15, g = lane >> 4;
        const int rh = wave & 1, kh = wave >> 1;
        const int other = wave ^ 2;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int e = 2 * kh + p;          // rows this wave finishes
            const float *src = red + ((other * 2 + p) * 4) * 64 + lane;
            // partial of the K half 0 first, then half 1 (a fixed order)
            float s[4], mine[4] = {acc_r[e], acc_z[e], acc_ni[e], acc_nh[e]};
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
                s[tl] = kh == 0 ? mine[tl] + src[tl * 64]
                                : src[tl * 64] + mine[tl];
            const int b = b0 + rh * 16 + 4 * g + e, j = j0 + r16;
            ob[p] = b;
            live[p] = b < a.B && j < a.H;
            out[p].hn = 0.f;
            if (live[p]) out[p] = cell_fwd_math(a, b, j, s);
            if (a.cs_part) tile[(rh * 16 + 4 * g + e) * HT + r16] = out[p].hn;
        }
    }
    f32x4v part = {0.f, 0.f, 0.f, 0.f};
    if (a.cs_part) {
        // [32 rows x 16 k] . [16 k x 64 outputs] as eight 16 x 16 tiles, one per
        // wave (row half w & 1, output tile w >> 1), before the global stores
        __syncthreads();
        const int r16 = lane & 15, g = lane >> 4;
        const f32x4v fa =
            *(const f32x4v *)(tile + (16 * (wave & 1) + r16) * HT + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            part = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], wf[e], part, 0, 0,
                                                        0);
    }
    if (wave < 4) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
            if (live[p]) cell_fwd_store(a, ob[p], j0 + (lane & 15), out[p]);
    }
    if (a.cs_part) {
        // D[i][j]: lane (r16, g) holds rows i = 4 g + e of the half, column r16
        const int r16 = lane & 15, g = lane >> 4;
        const int parts = a.Hn / NCN;
        float *dst = a.cs_part + ((long long)b0 * parts + blockIdx.x) * 64 +
                     16 * (wave >> 1) + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 16 * (wave & 1) + 4 * g + e;
            if (b0 + row < a.B) dst[(long long)row * parts * 64] = part[e];
        }
    }
}

// ------------------------------------------------------------------------------
// large-batch fused GRU cell: 64 batch rows x 64 hidden units per block
// ------------------------------------------------------------------------------
// With thousands of rows every CU gets several blocks, so what counts is MFMA
// time over block time.  Here consumer wave (rh, ch) owns the 32-row x 32-unit
// quadrant (rh, ch) of the tile over the WHOLE K -- no split-K, no cross-wave
// reduction, gate math straight from the accumulators --, and one block does
// four times the MFMA work of the 32 x 32 tile per prologue / epilogue and
// stages half the bytes per FLOP.  K chunks are 32 wide (two 32 KiB stages, so
// two blocks share a CU and fill each other's barrier waits): a DMA instruction
// fills 8 rows of 128 bytes, granule g of row r sits in slot g ^ ((r >> 1) & 7)
// (rows r, r + 1 are half a bank sweep apart), conflict-free for the 16-lane
// passes of the ds_read_b128 fragment reads.
constexpr int LM = 64, LN = 64, LK = 32;
constexpr int LROWS = LM + 3 * LN;     // stage rows (256)
constexpr int LSTAGE = LROWS * LK;     // floats per stage (32 KiB)

template <bool BF16>
__global__ __launch_bounds__(512, 4) void gru_cell_fwd64_kernel(CellFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float lds[2 * LSTAGE];
    const int tid = threadIdx.x;
    if (a.su_S && (int)blockIdx.y >= a.su_nbt) {
        // extra blocks: the step's stack update, off the critical path
        const int id = ((int)blockIdx.y - a.su_nbt) * gridDim.x + blockIdx.x;
        cell_fwd_stack_update(a, id * a.su_rows, tid);
        return;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = blockIdx.x * LN;
    const int b0 = blockIdx.y * LM;
    const int nx = 2 * a.nx_chunks;               // 32-wide chunks
    const int nchunks = nx + 2 * a.nh_chunks;
    const int last = nchunks - 1;
    const int lrow = lane & 31, lh = lane >> 5;

    if (wave >= 4) {
        // loaders: wave lw fills stage rows 64 lw .. 64 lw + 63 (lw 0: the
        // activation rows; 1-3: the r, z, n weight rows of the 64 hidden units)
        // with 8 DMA instructions of 8 rows each
        const int lw = wave - 4;
        const char *xb = (const char *)a.xin, *hb = (const char *)a.hprev;
        const char *wb = (const char *)a.wpack;
        uint32_t off0[8], off1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * (lw * 8 + i) + (lane >> 3);
            const int g = (lane & 7) ^ ((row >> 1) & 7);
            if (lw == 0) {
                const int b = min(b0 + row, a.B - 1);
                off0[i] = (uint32_t)(b * a.ldx + g * 4) * 4u;
                off1[i] = (uint32_t)(b * a.Hp + g * 4) * 4u;
            } else {
                const int jj = row & 63;
                off0[i] =
                    (uint32_t)(((lw - 1) * a.Hn + j0 + jj) * a.Kp + g * 4) * 4u;
                off1[i] = 0;
            }
        }
        const uint32_t lds0 = __builtin_amdgcn_readfirstlane(
            (uint32_t)(size_t)(__attribute__((address_space(3))) void *)lds);
        auto issue = [&](int cc) {
            const int c = min(cc, last);
            const uint32_t dst = lds0 + ((cc & 1) * LSTAGE + lw * 64 * LK) * 4u;
            if (lw == 0) {
                if (c < nx)
                    glds8(xb + c * (LK * 4), off0, dst);
                else
                    glds8(hb + (c - nx) * (LK * 4), off1, dst);
            } else {
                glds8(wb + c * (LK * 4), off0, dst);
            }
        };
        issue(0); issue(1);
        for (int k = 0; k < nchunks; ++k) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // chunk k landed
            raw_barrier();   // A_k: the consumers may read chunk k
            raw_barrier();   // B_k: they hold its fragments, the stage is free
            issue(k + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // C: the tail's re-issued chunks have landed, the consumers may reuse
        // the stages for the control partials
        if (a.cs_part) raw_barrier();
        return;
    }

    f32x16 acc_r, acc_z, acc_ni, acc_nh;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_r[r] = 0.f; acc_z[r] = 0.f; acc_ni[r] = 0.f; acc_nh[r] = 0.f;
    }
    const int rh = wave & 1, ch = wave >> 1;
    if (a.tokg) {
        // layer 0: the token's gate-table rows (the embedding side of the input
        // product) are the accumulators' start values -- 48 gathers per lane in
        // flight while the loaders fetch the first chunks, instead of 2 x 24
        // between the epilogue's math and stores (4.5 us of a 68 us launch)
        const int jt = j0 + 32 * ch + lrow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = min(b0 + 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * lh,
                              a.B - 1);
            long long tk = a.tok[b];
            if (tk < 0 || tk >= a.V) {
                tk = tk < 0 ? 0 : a.V - 1;
                if (a.status) atomicOr(a.status, PCM_STATUS_BAD_TOKEN);
            }
            const float *tg = a.tokg + tk * (3LL * a.Hn) + jt;
            acc_r[r] = tg[0]; acc_z[r] = tg[a.Hn]; acc_ni[r] = tg[2 * a.Hn];
        }
    }
    // k group q (8 wide) of a chunk = granules 4 q + lh + 2 s (s = 0, 1) of the
    // lane's activation row and of its weight row of every gate; every such
    // row r has ((r >> 1) & 7) == ((lrow >> 1) & 7)
    const int sw = (lrow >> 1) & 7;
    int slot[4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            slot[2 * q + s] = ((4 * q + lh + 2 * s) ^ sw) << 2;
    const int row_a = (32 * rh + lrow) * LK;
    const int row_w = (LM + 32 * ch + lrow) * LK;
    auto chunk = [&](int c, f32x16 &acc_n) {
        f32x4 fa[4], fr[4], fz[4], fn[4];
        raw_barrier();   // A_c
        const float *As = lds + (c & 1) * LSTAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *(const f32x4 *)(As + row_a + slot[i]);
            fr[i] = *(const f32x4 *)(As + row_w + slot[i]);
            fz[i] = *(const f32x4 *)(As + LN * LK + row_w + slot[i]);
            fn[i] = *(const f32x4 *)(As + 2 * LN * LK + row_w + slot[i]);
        }
        raw_barrier();   // B_c (waits for the fragment reads first)
        if constexpr (BF16) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bf16x8 ba = pcm_pack_bf16(fa[2 * q], fa[2 * q + 1]);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fr[2 * q], fr[2 * q + 1]), acc_r, 0, 0, 0);
                acc_z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fz[2 * q], fz[2 * q + 1]), acc_z, 0, 0, 0);
                acc_n = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    ba, pcm_pack_bf16(fn[2 * q], fn[2 * q + 1]), acc_n, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[i][e], fr[i][e], acc_r, 0, 0, 0);
                    acc_z = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[i][e], fz[i][e], acc_z, 0, 0, 0);
                    acc_n = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        fa[i][e], fn[i][e], acc_n, 0, 0, 0);
                }
        }
    };
    for (int c = 0; c < nx; ++c) chunk(c, acc_ni);
    for (int c = nx; c < nchunks; ++c) chunk(c, acc_nh);

    // gate math straight from the accumulators: lane = hidden unit, register r =
    // batch row (r & 3) + 8 (r >> 2) + 4 lh of the quadrant
    const int j = j0 + 32 * ch + lrow;
    if (j >= a.H) return;   // (never with cs_part: the launcher wants H == Hn)
    // Top layer with a stack (cs_part): the block's 64 x 64 tile of new hidden
    // states is parked in LDS (row stride HT) and multiplied with the [64 k x 64
    // outputs] block of [Wc ; Ws]: this column tile's share of the NEXT stack
    // step's 3 + W control / push pre-activations (partial number blockIdx.x of
    // Hn / 64, summed in order by the stack kernel, which then no longer
    // streams [Wc ; Ws] and the hidden rows: 13 of its 31 us at 4096 rows).
    // Wave (rh, ch): rows 32 rh.., outputs 32 ch.., 32x32x2 MFMAs with
    // instruction i of lane half lh taking k = 32 lh + i.
    constexpr int HT = 68;
    const bool ctrl = a.cs_part != nullptr;
    f32x4 wv[8];
    CellFwdArgs an = a;    // (the token gates are in the sums already)
    an.tokg = nullptr;
    // in groups of eight: all of a group's operand loads are requested before
    // its first store (an element at a time, every element's loads queue behind
    // the stores of the one before: 16 memory round trips per lane)
#pragma unroll
    for (int g8 = 0; g8 < 2; ++g8) {
        CellFwdOut out[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * g8 + i;
            const int b = b0 + 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float s[4] = {acc_r[r], acc_z[r], acc_ni[r], acc_nh[r]};
            out[i].hn = 0.f;
            if (b < a.B) out[i] = cell_fwd_math(an, b, j, s);
        }
        if (ctrl && g8 == 1) {
            // (requested once the accumulators are dead: 32 registers)
            const int o = 32 * ch + lrow;
#pragma unroll
            for (int q = 0; q < 8; ++q) wv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (o < 3 + a.cs_W) {
                const float *wrow =
                    (o < 3 ? a.cs_ctrl_w + (long long)o * a.H
                           : a.cs_push_w + (long long)(o - 3) * a.H) +
                    j0 + 32 * lh;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    wv[q] = *(const f32x4a4 *)(wrow + 4 * q);
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * g8 + i;
            const int b = b0 + 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (b < a.B) cell_fwd_store(a, b, j, out[i]);
        }
        if (ctrl) {
            if (g8 == 0) raw_barrier();   // C (see the loaders)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = 8 * g8 + i;
                const int row = 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * lh;
                lds[row * HT + 32 * ch + lrow] = out[i].hn;
            }
        }
    }
    if (!ctrl) return;
    raw_barrier();   // D: the tile is complete (consumer waves only by now)
    f32x16 pacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
    const float *trow = lds + (32 * rh + lrow) * HT + 32 * lh;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const f32x4 fa = *(const f32x4 *)(trow + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            pacc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], wv[q][e], pacc, 0,
                                                        0, 0);
    }
    const int parts = a.Hn / LN;
    float *dst = a.cs_part + ((long long)b0 * parts + blockIdx.x) * 64 + 32 * ch +
                 lrow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (b0 + row < a.B) dst[(long long)row * parts * 64] = pacc[r];
    }
}

// ------------------------------------------------------------------------------
// stack controls + fused stack update
// ------------------------------------------------------------------------------
// One block of 512 threads per SROWS batch rows.  The rows' previous stacks are
// staged into LDS while the 3+W control / push dot products run (weights
// stream from L2, 8 waves x 7 outputs, lanes along K), so the update itself
// reads HBM once and writes once (20 KB / row, the algorithmic minimum).
constexpr int STHREADS = 512;
constexpr int SOUT = 7;  // outputs per wave (8 waves x 7 = 56 >= 3 + W)

// LEAN: no register prefetch of the weights / previous stack rows.  The
// prefetching variant needs 200 VGPRs (one block per CU): best when the grid is
// a single round of blocks (batch 512 = 256 blocks, the step is latency-bound);
// with more rows than that the lean one keeps several blocks per CU in flight.
// (80 VGPRs, three blocks per CU: 72.9 -> 39.1 us per step at batch 4096; four
// rows per block or a deeper unroll of the dot loop measured no better).
// SROWS rows per block: 1 while that still is at most two blocks per CU (batch
// 512: 512 lean blocks, 10.7 us against 11.2 for 256 prefetching ones), else 2.
template <bool LEAN, int SROWS>
__global__ __launch_bounds__(STHREADS, LEAN ? 4 : 2) void stack_step_fwd_kernel(
    StackFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int H = a.H, W = a.W, D = a.D, DW = D * W, NO = 3 + W;
    const int DWp = (DW + 3) & ~3;
    const float inv_DW = 1.0f / (float)DW, inv_W = 1.0f / (float)W;
    const float inv_H = 1.0f / (float)H, inv_AVW = 1.0f / (float)a.AVW;
    const float inv_nq = 1.0f / (float)max(DW >> 2, 1);
    float *Srow = sm;                          // [SROWS][DWp]
    float *hrow = Srow + SROWS * DWp;          // [SROWS][max(H, 512)]
    float *cu = hrow + SROWS * max(H, 512);    // [SROWS][AVW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * SROWS;

    // partial sums left by the previous step's top cell stand in for the dot
    // products (and for the hidden rows they would read)
    const bool parts = a.cs_part != nullptr;
    // control / push weights of this wave's outputs go to registers FIRST so
    // their L2 round trip overlaps the staging loads below (H <= 512 path)
    constexpr int KS = 8;
    const bool wregs = !parts && !LEAN && H <= 64 * KS && NO <= 8 * SOUT;
    float wr[SOUT][KS];
    if (wregs) {
#pragma unroll
        for (int i = 0; i < SOUT; ++i) {
            const int o = min(wave * SOUT + i, NO - 1);
            const float *wrow = o < 3 ? a.ctrl_w + (long long)o * H
                                      : a.push_w + (long long)(o - 3) * H;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                const int k = lane + 64 * kk;
                wr[i][kk] = k < H ? wrow[k] : 0.f;
            }
        }
    }

    if (parts) {
        // wave g sums partials [g * per, (g + 1) * per) of every output in
        // order (loads coalesced over the outputs, in flight together with
        // the stack loads below) and parks the group sum where the hidden rows
        // would go: [row][group][output]
        const int g = tid >> 6, o = tid & 63;
        const int per = (a.cs_parts + 7) >> 3;
        const int p0 = g * per, p1 = min(p0 + per, a.cs_parts);
#pragma unroll
        for (int r = 0; r < SROWS; ++r) {
            const int b = min(b0 + r, a.B - 1);
            const float *src = a.cs_part + (long long)b * a.cs_parts * 64 + o;
            // (loads of a group in flight together, added in order)
            float v = 0.f;
            int p = p0;
            for (; p + 4 <= p1; p += 4) {
                const float t0 = src[p * 64], t1 = src[(p + 1) * 64];
                const float t2 = src[(p + 2) * 64], t3 = src[(p + 3) * 64];
                v += t0; v += t1; v += t2; v += t3;
            }
            if (p + 2 <= p1) {
                const float t0 = src[p * 64], t1 = src[(p + 1) * 64];
                v += t0; v += t1;
                p += 2;
            }
            if (p < p1) v += src[p * 64];
            hrow[(r * 8 + g) * 64 + o] = v;
        }
    }
    // top-layer hidden rows -> LDS (needed first, by the dot products)
    if (!parts)
    for (int idx = tid; idx < SROWS * H; idx += STHREADS) {
        const int r = pcm_fdiv(idx, inv_H), k = idx - r * H;
        const int b = b0 + r;
        hrow[idx] = b < a.B ? a.htop[(long long)b * a.Hp + k] : 0.f;
    }
    // previous stacks: the loads are issued now, into registers, and only
    // parked in LDS after the dot products, which hide their latency
    // (vectorised when the row size allows it)
    constexpr int SQ = 4;  // float4 per thread held in flight
    const bool vec = (DW & 3) == 0;
    const int nq = DW >> 2;
    const bool sregs = !LEAN && vec && SROWS * nq <= SQ * STHREADS;
    f32x4 sq[SQ];
    if (sregs) {
#pragma unroll
        for (int i = 0; i < SQ; ++i) {
            const int idx = min(tid + STHREADS * i, SROWS * nq - 1);
            const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
            const int b = min(b0 + r, a.B - 1);
            sq[i] = *(const f32x4 *)(a.S_prev + (long long)b * DW + q * 4);
        }
    } else if (vec) {
        for (int idx = tid; idx < SROWS * nq; idx += STHREADS) {
            const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
            const int b = b0 + r;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (b < a.B)
                v = *(const f32x4 *)(a.S_prev + (long long)b * DW + q * 4);
            *(f32x4 *)(Srow + r * DWp + q * 4) = v;
        }
    } else {
        for (int idx = tid; idx < SROWS * DW; idx += STHREADS) {
            const int r = pcm_fdiv(idx, inv_DW), e = idx - r * DW;
            const int b = b0 + r;
            Srow[r * DWp + e] = b < a.B ? a.S_prev[(long long)b * DW + e] : 0.f;
        }
    }
    __syncthreads();

    // dot products: wave handles SOUT outputs per pass, lanes stride K
    if (parts) {
        // the eight group sums of every (row, output), in group order
        for (int idx = tid; idx < SROWS * 64; idx += STHREADS) {
            const int r = idx >> 6, o = idx & 63;
            if (o >= NO) continue;
            float v = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) v += hrow[(r * 8 + g) * 64 + o];
            cu[r * a.AVW + o] = v + (o < 3 ? a.ctrl_b[o] : a.push_b[o - 3]);
        }
    } else if (wregs) {
        float p[SOUT][SROWS];
#pragma unroll
        for (int i = 0; i < SOUT; ++i)
#pragma unroll
            for (int r = 0; r < SROWS; ++r) p[i][r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int k = lane + 64 * kk;
            if (k < H) {
#pragma unroll
                for (int r = 0; r < SROWS; ++r) {
                    const float hv = hrow[r * H + k];
#pragma unroll
                    for (int i = 0; i < SOUT; ++i)
                        p[i][r] = fmaf(wr[i][kk], hv, p[i][r]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SOUT; ++i)
#pragma unroll
            for (int r = 0; r < SROWS; ++r) {
                float v = p[i][r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                p[i][r] = v;
            }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < SOUT; ++i) {
                const int o = wave * SOUT + i;
                if (o < NO) {
                    const float bias = o < 3 ? a.ctrl_b[o] : a.push_b[o - 3];
#pragma unroll
                    for (int r = 0; r < SROWS; ++r)
                        cu[r * a.AVW + o] = p[i][r] + bias;
                }
            }
        }
    } else if (LEAN && (H & 255) == 0 && H <= 1024 && NO <= 8 * SOUT) {
        // 16-byte loads, four consecutive k per lane: a quarter of the load
        // instructions of the lane-strided loop below, all in flight at once
        // (the dot products are what this kernel spends most time in: every
        // block streams the whole [Wc ; Ws] from L2)
        float p[SOUT][SROWS];
        const float *wrow[SOUT];
#pragma unroll
        for (int i = 0; i < SOUT; ++i) {
            const int o = min(wave * SOUT + i, NO - 1);
            wrow[i] = (o < 3 ? a.ctrl_w + (long long)o * H
                             : a.push_w + (long long)(o - 3) * H) + 4 * lane;
#pragma unroll
            for (int r = 0; r < SROWS; ++r) p[i][r] = 0.f;
        }
#pragma unroll
        for (int k0 = 0; k0 < 1024; k0 += 256) {
            if (k0 < H) {
                f32x4 w[SOUT];
#pragma unroll
                for (int i = 0; i < SOUT; ++i)
                    w[i] = *(const f32x4 *)(wrow[i] + k0);
#pragma unroll
                for (int r = 0; r < SROWS; ++r) {
                    const f32x4 hv =
                        *(const f32x4 *)(hrow + r * H + k0 + 4 * lane);
#pragma unroll
                    for (int i = 0; i < SOUT; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            p[i][r] = fmaf(w[i][e], hv[e], p[i][r]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < SOUT; ++i)
#pragma unroll
            for (int r = 0; r < SROWS; ++r) {
                float v = p[i][r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                p[i][r] = v;
            }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < SOUT; ++i) {
                const int o = wave * SOUT + i;
                if (o < NO) {
                    const float bias = o < 3 ? a.ctrl_b[o] : a.push_b[o - 3];
#pragma unroll
                    for (int r = 0; r < SROWS; ++r)
                        cu[r * a.AVW + o] = p[i][r] + bias;
                }
            }
        }
    } else
    for (int obase = wave * SOUT; obase < NO; obase += 8 * SOUT) {
        float p[SOUT][SROWS];
        const float *wrow[SOUT];
#pragma unroll
        for (int i = 0; i < SOUT; ++i) {
            const int o = min(obase + i, NO - 1);
            wrow[i] = o < 3 ? a.ctrl_w + (long long)o * H
                            : a.push_w + (long long)(o - 3) * H;
#pragma unroll
            for (int r = 0; r < SROWS; ++r) p[i][r] = 0.f;
        }
#pragma unroll 2
        for (int k = lane; k < H; k += 64) {
            float w[SOUT];
#pragma unroll
            for (int i = 0; i < SOUT; ++i) w[i] = wrow[i][k];
#pragma unroll
            for (int r = 0; r < SROWS; ++r) {
                const float hv = hrow[r * H + k];
#pragma unroll
                for (int i = 0; i < SOUT; ++i) p[i][r] = fmaf(w[i], hv, p[i][r]);
            }
        }
#pragma unroll
        for (int i = 0; i < SOUT; ++i)
#pragma unroll
            for (int r = 0; r < SROWS; ++r) {
                float v = p[i][r];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                p[i][r] = v;
            }
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < SOUT; ++i) {
                const int o = obase + i;
                if (o < NO) {
                    const float bias = o < 3 ? a.ctrl_b[o] : a.push_b[o - 3];
#pragma unroll
                    for (int r = 0; r < SROWS; ++r)
                        cu[r * a.AVW + o] = p[i][r] + bias;
                }
            }
        }
    }
    if (sregs) {
#pragma unroll
        for (int i = 0; i < SQ; ++i) {
            const int idx = tid + STHREADS * i;
            if (idx < SROWS * nq) {
                const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
                *(f32x4 *)(Srow + r * DWp + q * 4) = sq[i];
            }
        }
    }
    __syncthreads();

    // softmax over the 3 controls (one thread per row), tanh of the push value
    if (tid < SROWS) {
        float *c = cu + tid * a.AVW;
        const float m = fmaxf(c[0], fmaxf(c[1], c[2]));
        const float e0 = expf(c[0] - m), e1 = expf(c[1] - m), e2 = expf(c[2] - m);
        const float s = e0 + e1 + e2;
        c[0] = e0 / s; c[1] = e1 / s; c[2] = e2 / s;
    }
    for (int idx = tid; idx < SROWS * W; idx += STHREADS) {
        const int r = pcm_fdiv(idx, inv_W), w = idx - r * W;
        cu[r * a.AVW + 3 + w] = tanhf(cu[r * a.AVW + 3 + w]);
    }
    __syncthreads();

    // record a, v (needed by the reverse pass) and update the stack
    for (int idx = tid; idx < SROWS * a.AVW; idx += STHREADS) {
        const int r = pcm_fdiv(idx, inv_AVW), o = idx - r * a.AVW;
        const int b = b0 + r;
        if (b < a.B) a.av[(long long)b * a.AVW + o] = o < NO ? cu[idx] : 0.f;
    }
    auto updated = [&](int r, int e, bool frozen) -> float {
        const int d = pcm_fdiv(e, inv_W), w = e - d * W;
        const float *Sp = Srow + r * DWp;
        const float sv = Sp[e];
        float nv = sv;
        if (!frozen) {
            const float *c = cu + r * a.AVW;
            const float up = d == 0 ? c[3 + w] : Sp[e - W];
            const float dn = d == D - 1 ? 0.f : Sp[e + W];
            // reference order: no_op*S, + push*up, + pop*down (stack_rnn.py:229-231)
            nv = __fadd_rn(__fadd_rn(__fmul_rn(c[2], sv), __fmul_rn(c[0], up)),
                           __fmul_rn(c[1], dn));
        }
        return nv;
    };
    if (vec) {
        for (int idx = tid; idx < SROWS * nq; idx += STHREADS) {
            const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
            const int b = b0 + r;
            if (b >= a.B) continue;
            const bool frozen = a.lengths && a.t >= a.lengths[b];
            float *xt = a.xtop + (long long)b * a.Wp;
            f32x4 nv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = q * 4 + i;
                nv[i] = updated(r, e, frozen);
                if (e < W) xt[e] = nv[i];   // row 0 = the new stack top
            }
            *(f32x4 *)(a.S_new + (long long)b * DW + q * 4) = nv;
        }
    } else {
        for (int idx = tid; idx < SROWS * DW; idx += STHREADS) {
            const int r = pcm_fdiv(idx, inv_DW), e = idx - r * DW;
            const int b = b0 + r;
            if (b >= a.B) continue;
            const bool frozen = a.lengths && a.t >= a.lengths[b];
            const float nv = updated(r, e, frozen);
            a.S_new[(long long)b * DW + e] = nv;
            if (e < W) a.xtop[(long long)b * a.Wp + e] = nv;
        }
    }
    const int padw = a.Wp - W;
    const float inv_padw = 1.0f / (float)max(padw, 1);
    for (int idx = tid; idx < SROWS * padw; idx += STHREADS) {
        const int r = pcm_fdiv(idx, inv_padw), w = W + idx - r * padw;
        const int b = b0 + r;
        if (b < a.B) a.xtop[(long long)b * a.Wp + w] = 0.f;
    }
}


// Stack step for large batches: 16 batch rows per block.  The 3 + W control /
// push pre-activations of the 16 rows are one [16 x H] . [H x 64] product on
// 16x16x4 MFMAs (wave w: output column tile w & 3, K half w >> 2, operands
// straight from L2 as in logits_choose_kernel), so the weights are read once
// per 16 rows instead of once per row or two; the stack update then streams
// S_prev -> S_new from global memory (the three taps of an element hit the same
// lines in L1), with no LDS staging of the stacks and nothing but loads,
// multiply-adds and stores on the way.

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

// MROWS rows per block.  The MFMA tile holds 16, but 8 rows per block (half the
// tile empty, twice the blocks: two per CU at 4096 rows) keep more of the update's
// loads in flight: encoder-only 165.8 k -> 168.2 k molecules/s at batch 4096,
// greedy decode 186.9 k -> 189.7 k; 4 rows per block: 161.2 k (the weights are
// read once per block).  PCM_STACK_MROWS = 4 / 8 / 16 forces a value.
template <int MROWS>
__global__ __launch_bounds__(512) void stack_step_fwd_mfma_kernel(StackFwdArgs a) {
    __shared__ float part[2][MROWS][65];   // [K half][row][output]
    __shared__ float cu[MROWS][64];        // a[3], v[W] per row
    const int H = a.H, W = a.W, D = a.D, DW = D * W, NO = 3 + W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * MROWS;
    if (a.cs_part) {
        // the previous step's top cell left partial sums of the pre-activations
        // (one per column tile of its launch): sum them in order
        for (int idx = tid; idx < MROWS * 64; idx += 512) {
            const int r = idx >> 6, o = idx & 63;
            const float *src =
                a.cs_part + (long long)min(b0 + r, a.B - 1) * a.cs_parts * 64 + o;
            float v = 0.f;
            if (o < NO) {
                float pv[16];
                int p = 0;
                for (; p + 16 <= a.cs_parts; p += 16) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) pv[i] = src[(p + i) * 64];
#pragma unroll
                    for (int i = 0; i < 16; ++i) v += pv[i];
                }
                for (; p + 8 <= a.cs_parts; p += 8) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) pv[i] = src[(p + i) * 64];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v += pv[i];
                }
                for (; p < a.cs_parts; ++p) v += src[p * 64];
                v += o < 3 ? a.ctrl_b[o] : a.push_b[o - 3];
                if (o >= 3) v = tanhf(v);
            }
            cu[r][o] = v;
        }
    } else {
        {
        const int r16 = lane & 15, g = lane >> 4;
        const int ct = wave & 3, kh = wave >> 2;
        const int o = min(ct * 16 + r16, NO - 1);
        const float *wrow = (o < 3 ? a.ctrl_w + (long long)o * H
                                   : a.push_w + (long long)(o - 3) * H) + 4 * g;
        const float *hrow =
            a.htop + (long long)min(b0 + r16, a.B - 1) * a.Hp + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int klen = H / 2, kend = (kh + 1) * klen;
        int k = kh * klen;
        for (; k + 128 <= kend; k += 128) {
            f32x4 fa[8], fb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                fa[i] = *(const f32x4 *)(hrow + k + 16 * i);
                fb[i] = *(const f32x4 *)(wrow + k + 16 * i);
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
            const f32x4 fb = *(const f32x4 *)(wrow + k);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], acc, 0,
                                                           0, 0);
        }
        // D[i][j]: lane holds rows i = 4 g + e, column j = r16
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g + e < MROWS) part[kh][4 * g + e][ct * 16 + r16] = acc[e];
    }
    __syncthreads();
    for (int idx = tid; idx < MROWS * 64; idx += 512) {
        const int r = idx >> 6, o = idx & 63;
        float v = 0.f;
        if (o < NO) {
            v = part[0][r][o] + part[1][r][o];
            v += o < 3 ? a.ctrl_b[o] : a.push_b[o - 3];
            if (o >= 3) v = tanhf(v);
        }
        cu[r][o] = v;
    }
    }
    __syncthreads();
    if (tid < MROWS) {
        float *c = cu[tid];
        const float m = fmaxf(c[0], fmaxf(c[1], c[2]));
        const float e0 = expf(c[0] - m), e1 = expf(c[1] - m), e2 = expf(c[2] - m);
        const float s = e0 + e1 + e2;
        c[0] = e0 / s; c[1] = e1 / s; c[2] = e2 / s;
    }
    __syncthreads();
    // record a, v (needed by the reverse pass)
    for (int idx = tid; idx < MROWS * a.AVW; idx += 512) {
        const int r = idx / a.AVW, o = idx - r * a.AVW;
        if (b0 + r < a.B)
            a.av[(long long)(b0 + r) * a.AVW + o] = o < NO ? cu[r][o] : 0.f;
    }
    // the update, four consecutive elements per thread (DW % 4 == 0); top_only:
    // the quads that hold row 0 only (the rest rides in the top cell's launch)
    const int nq = a.top_only ? (W + 3) >> 2 : DW >> 2;
    const float inv_nq = 1.0f / (float)nq;
#pragma unroll 4
    for (int idx = tid; idx < MROWS * nq; idx += 512) {
        const int r = pcm_fdiv(idx, inv_nq), q = idx - r * nq;
        const int b = b0 + r;
        if (b >= a.B) continue;
        const float *Sp = a.S_prev + (long long)b * DW;
        const f32x4 sv = *(const f32x4 *)(Sp + q * 4);
        f32x4 nv = sv;
        if (!(a.lengths && a.t >= a.lengths[b])) {
            const float *c = cu[r];
            const int e0 = q * 4;
            // rows above (d - 1; the pushed value for d == 0) and below (d + 1;
            // zero under the last row): element-wise, a group of four may
            // straddle a row boundary
            f32x4u up4 = {0.f, 0.f, 0.f, 0.f}, dn4 = up4;
            if (e0 >= W) up4 = *(const f32x4u *)(Sp + e0 - W);
            if (e0 + 3 + W < DW) dn4 = *(const f32x4u *)(Sp + e0 + W);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = e0 + i;
                const float up = e < W ? c[3 + e] : (e0 >= W ? up4[i] : Sp[e - W]);
                const float dn = e + W >= DW ? 0.f
                                 : (e0 + 3 + W < DW ? dn4[i] : Sp[e + W]);
                // reference order: no_op*S, + push*up, + pop*down
                nv[i] = __fadd_rn(__fadd_rn(__fmul_rn(c[2], sv[i]),
                                            __fmul_rn(c[0], up)),
                                  __fmul_rn(c[1], dn));
            }
        }
        if (!a.top_only) *(f32x4 *)(a.S_new + (long long)b * DW + q * 4) = nv;
        float *xt = a.xtop + (long long)b * a.Wp;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (q * 4 + i < W) xt[q * 4 + i] = nv[i];   // row 0 = the new top
    }
    const int padw = a.Wp - W;
    for (int idx = tid; idx < MROWS * padw; idx += 512) {
        const int r = idx / padw, w = W + idx - r * padw;
        if (b0 + r < a.B) a.xtop[(long long)(b0 + r) * a.Wp + w] = 0.f;
    }
    (void)D;
}

// ------------------------------------------------------------------------------
// weight packing, state init, copies
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    // wpack[(g*Hn + j)*Kp + k]
    const long long total = 3LL * a.Hn * a.Kp;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int k = (int)(idx % a.Kp);
        const int row = (int)(idx / a.Kp);
        const int g = row / a.Hn, j = row - g * a.Hn;
        float v = 0.f;
        if (j < a.H) {
            if (k < a.Kxp) {
                if (k < a.Kx)
                    v = a.w_ih[(long long)(g * a.H + j) * a.ld_ih + a.x_off + k];
            } else {
                const int kk = k - a.Kxp;
                if (kk < a.H) v = a.w_hh[(long long)(g * a.H + j) * a.H + kk];
            }
        }
        a.wpack[idx] = v;
    }
    // bpack[q*Hn + j]: q=0 r, 1 z, 2 n (input side), 3 n (hidden side)
    if (blockIdx.x == 0) {
        for (int idx = threadIdx.x; idx < 4 * a.Hn; idx += 256) {
            const int q = idx / a.Hn, j = idx - q * a.Hn;
            float v = 0.f;
            if (j < a.H) {
                const float bi_r = a.fold_bih ? 0.f : a.b_ih[j];
                const float bi_z = a.fold_bih ? 0.f : a.b_ih[a.H + j];
                const float bi_n = a.fold_bih ? 0.f : a.b_ih[2 * a.H + j];
                if (q == 0) v = bi_r + a.b_hh[j];
                else if (q == 1) v = bi_z + a.b_hh[a.H + j];
                else if (q == 2) v = bi_n;
                else v = a.b_hh[2 * a.H + j];
            }
            a.bpack[idx] = v;
        }
    }
}

// dst[r][c] (+)= src[r][c] for a [rows, cols] block
__global__ __launch_bounds__(256) void add_block_kernel(const float *src,
                                                        long long lds_,
                                                        float *dst,
                                                        long long ldd, int rows,
                                                        int cols,
                                                        int accumulate) {
    const long long total = (long long)rows * cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx - (long long)r * cols);
        const float v = src[r * lds_ + c];
        float *d = dst + r * ldd + c;
        *d = accumulate ? *d + v : v;
    }
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float *src,
                                                        long long lds_,
                                                        float *dst,
                                                        long long ldd, int rows,
                                                        int cols, int dst_cols) {
    // copies [rows, cols]; columns cols..dst_cols-1 of dst are zero-filled
    const long long total = (long long)rows * dst_cols;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * 256) {
        const int c = (int)(idx % dst_cols);
        const long long r = idx / dst_cols;
        float v = 0.f;
        if (c < cols && src) v = src[r * lds_ + c];
        dst[r * ldd + c] = v;
    }
}

__global__ __launch_bounds__(256) void fill_kernel(float *dst, float value,
                                                   long long n) {
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < n;
         idx += (long long)gridDim.x * 256)
        dst[idx] = value;
}

// buf is [slots][B][width]; row (t + slot_off, b) is zeroed for t >= lengths[b]
// (one block per (t, b): most exit at once)
__global__ __launch_bounds__(256) void zero_past_length_kernel(
    float *buf, int width, int B, int slot_off, const int32_t *lengths) {
    const int t = blockIdx.y, b = blockIdx.x;
    if (t < lengths[b]) return;
    float *row = buf + ((long long)(t + slot_off) * B + b) * width;
    for (int j = threadIdx.x; j < width; j += 256) row[j] = 0.f;
}

// out[b][0..cols) = buf[slot(b)][b][0..cols), slot(b) = min(lengths[b], T)
// (or its parity when the state ping-pongs between two slots)
__global__ __launch_bounds__(256) void gather_at_length_kernel(
    const float *buf, long long slot_floats, int width, int T, int parity,
    const int32_t *lengths, float *out, int cols) {
    const int b = blockIdx.x;
    int slot = min(max(lengths[b], 0), T);
    if (parity) slot &= 1;
    const float *row = buf + slot * slot_floats + (long long)b * width;
    for (int j = threadIdx.x; j < cols; j += 256)
        out[(long long)b * cols + j] = row[j];
}

int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

int pcm_launch_zero_past_length(float *buf, int width, int T, int B,
                                int slot_off, const int32_t *lengths,
                                hipStream_t stream) {
    hipLaunchKernelGGL(zero_past_length_kernel, dim3(B, T), dim3(256), 0, stream,
                       buf, width, B, slot_off, lengths);
    return 0;
}

int pcm_launch_gather_at_length(const float *buf, long long slot_floats,
                                int width, int T, int parity,
                                const int32_t *lengths, float *out, int cols,
                                int B, hipStream_t stream) {
    hipLaunchKernelGGL(gather_at_length_kernel, dim3(B), dim3(256), 0, stream,
                       buf, slot_floats, width, T, parity, lengths, out, cols);
    return 0;
}

// tile of the forward cell for this launch: 64 (64 x 64, thousands of rows),
// 16 (32 rows x 16 units) or 32 (32 x 32)
static int cell_fwd_tile(const CellFwdArgs &a) {
    static const int force = [] {
        const char *e = getenv("PCM_FWD_TILE");   // 16 / 32 / 64: force a tile
        return e ? atoi(e) : 0;
    }();
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                                        dev);
        return n;
    }();
    const int nbt = (a.B + CM - 1) / CM;
    // thousands of rows: 64 x 64 tiles (see the kernel)
    if ((force ? force == 64 : a.B >= 2048) && (a.Hn % LN) == 0) return 64;
    // Narrow tiles when they balance better over the CUs: a wide block is two
    // units of work, a narrow one 1.06 (measured: 16.5 against 15.5 us at one
    // wide block per CU); compare the busiest CU's load.  Batch 512 -> wide
    // (256 blocks), 768 -> narrow (43.4 -> 41.5 ms per step), <= 256 -> narrow.
    const int wide_tiles = (a.Hn / CN) * nbt;
    const float load_wide = 2.0f * ((wide_tiles + cus - 1) / cus);
    const float load_narrow = 1.06f * ((2 * wide_tiles + cus - 1) / cus);
    const bool narrow = force ? force == 16 : load_narrow < load_wide;
    return narrow ? 16 : 32;
}

int pcm_cell_fwd_tile(const CellFwdArgs &a) { return cell_fwd_tile(a); }

int pcm_cell_fwd_ctrl_partials(const CellFwdArgs &a) {
    // every forward tile's tail computes them: one partial per column tile (0:
    // this launch writes none)
    if (!a.cs_part || a.H != a.Hn || 3 + a.cs_W > 64) return 0;
    const int tile = cell_fwd_tile(a);
    if (tile == 32 && (a.H % CN) == 0) return a.Hn / CN;
    if (tile == 64 && (a.H % LN) == 0) return a.Hn / LN;
    if (tile == 16 && (a.H % NCN) == 0) return a.Hn / NCN;
    return 0;
}

int pcm_launch_cell_fwd(const CellFwdArgs &a, hipStream_t stream) {
    const int nbt = (a.B + CM - 1) / CM;
    const int tile = cell_fwd_tile(a);
    PCM_REQUIRE(tile == 32 || (!a.xt_part && (tile == 64 || !a.su_S)), PCM_EINVAL,
                "cell_fwd: the fused stack step needs the 32 x 32 tile (the "
                "update alone may ride with the 64 x 64 tile)");
    PCM_REQUIRE(!a.cs_part || pcm_cell_fwd_ctrl_partials(a) > 0, PCM_EINVAL,
                "cell_fwd: control partials asked of a launch that cannot "
                "write them (tile %d, H %d, Hn %d)", tile, a.H, a.Hn);
    if (tile == 64) {
        dim3 grid(a.Hn / LN, (a.B + LM - 1) / LM);
        if (a.su_S) {
            PCM_REQUIRE(a.su_nbt == (int)grid.y && a.su_rows > 0, PCM_EINVAL,
                        "cell_fwd: bad stack-update arguments");
            const int ublocks = (a.B + a.su_rows - 1) / a.su_rows;
            grid.y += (ublocks + grid.x - 1) / grid.x;
        }
        if (g_pcm_matmul_bf16)
            PCM_LAUNCH(gru_cell_fwd64_kernel<true>, grid, dim3(512), 0, stream, a);
        else
            PCM_LAUNCH(gru_cell_fwd64_kernel<false>, grid, dim3(512), 0, stream, a);
    } else if (tile == 16) {
        dim3 grid(a.Hn / NCN, nbt);
        if (g_pcm_matmul_bf16)
            PCM_LAUNCH(gru_cell_fwd16_kernel<true>, grid, dim3(512), 0, stream, a);
        else
            PCM_LAUNCH(gru_cell_fwd16_kernel<false>, grid, dim3(512), 0, stream, a);
    } else {
        dim3 grid(a.Hn / CN, nbt);
        if (a.su_S) {
            // extra block rows: the deferred stack update, su_rows rows per block
            PCM_REQUIRE(a.su_nbt == nbt && a.su_rows > 0, PCM_EINVAL,
                        "cell_fwd: bad stack-update arguments");
            const int ublocks = (a.B + a.su_rows - 1) / a.su_rows;
            grid.y += (ublocks + grid.x - 1) / grid.x;
        }
        PCM_REQUIRE(!a.hz_raw || (a.xt_part && !a.su_S && a.hz_nbt == nbt),
                    PCM_EINVAL, "cell_fwd: the hoisted hidden-side product rides "
                    "in the fused layer-0 launch only");
        if (a.hz_raw) grid.y += nbt;
        if (a.xt_part) {
            if (g_pcm_matmul_bf16)
                PCM_LAUNCH((gru_cell_fwd_kernel<true, true>), grid, dim3(512), 0,
                           stream, a);
            else
                PCM_LAUNCH((gru_cell_fwd_kernel<false, true>), grid, dim3(512),
                           0, stream, a);
        } else if (g_pcm_matmul_bf16) {
            PCM_LAUNCH((gru_cell_fwd_kernel<true, false>), grid, dim3(512), 0,
                       stream, a);
        } else {
            PCM_LAUNCH((gru_cell_fwd_kernel<false, false>), grid, dim3(512), 0,
                       stream, a);
        }
        return 0;
    }
    return 0;
}

// LDS bytes one batch row of the forward stack step stages (previous stack,
// top-layer hidden row, controls)
static size_t stack_fwd_row_bytes(int D, int W, int H, int AVW) {
    const int DWp = (D * W + 3) & ~3;
    return (size_t)(DWp + (H > 512 ? H : 512) + AVW) * sizeof(float);
}

int pcm_stack_fwd_check(int D, int W, int H, int AVW) {
    PCM_REQUIRE(stack_fwd_row_bytes(D, W, H, AVW) <= PCM_STACK_LDS_BYTES,
                PCM_EINVAL,
                "stack step: stack %dx%d + hidden %d exceed the %d KiB LDS row "
                "cache (needs (D*W + max(H, 512) + W + 3) * 4 bytes per row)", D, W, H,
                PCM_STACK_LDS_BYTES / 1024);
    return PCM_OK;
}

static bool stack_fwd_mfma_shape(const StackFwdArgs &a, int mfma_rows) {
    return a.B >= mfma_rows && (a.H % 32) == 0 && (a.Hp % 4) == 0 &&
           3 + a.W <= 64 && ((a.D * a.W) & 3) == 0 && a.AVW <= 64;
}
static int stack_fwd_mfma_rows() {
    // large batches: 16 rows per block with the dot products on MFMA
    static const int mfma_rows = [] {
        const char *e = getenv("PCM_STACK_MFMA");   // rows from which it is used
        return e ? atoi(e) : 4096;
    }();
    return mfma_rows;
}
bool pcm_stack_fwd_is_mfma(const StackFwdArgs &a) {
    return stack_fwd_mfma_shape(a, stack_fwd_mfma_rows());
}

int pcm_launch_stack_fwd(const StackFwdArgs &a, hipStream_t stream) {
    const size_t row_bytes = stack_fwd_row_bytes(a.D, a.W, a.H, a.AVW);
    int rc = pcm_stack_fwd_check(a.D, a.W, a.H, a.AVW);
    if (rc) return rc;
    static const int force = [] {
        const char *e = getenv("PCM_STACK_LEAN");  // 0 / 1: force a variant
        return e ? atoi(e) : -1;
    }();
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                                        dev);
        return n;
    }();
    const int mfma_rows = stack_fwd_mfma_rows();
    PCM_REQUIRE(!a.top_only || stack_fwd_mfma_shape(a, mfma_rows), PCM_EINVAL,
                "stack step: top_only needs the large-batch kernel");
    if (stack_fwd_mfma_shape(a, mfma_rows)) {
        static const int mrows = [] {
            const char *e = getenv("PCM_STACK_MROWS");
            const int v = e ? atoi(e) : 8;
            return (v == 16 || v == 4) ? v : 8;
        }();
        if (mrows == 4)
            PCM_LAUNCH(stack_step_fwd_mfma_kernel<4>, dim3((a.B + 3) / 4),
                       dim3(512), 0, stream, a);
        else if (mrows == 8)
            PCM_LAUNCH(stack_step_fwd_mfma_kernel<8>, dim3((a.B + 7) / 8),
                       dim3(512), 0, stream, a);
        else
            PCM_LAUNCH(stack_step_fwd_mfma_kernel<16>, dim3((a.B + 15) / 16),
                       dim3(512), 0, stream, a);
        return 0;
    }
    // (two rows per block only while both fit the LDS row cache)
    const int rows =
        (a.B <= 2 * cus || 2 * row_bytes > PCM_STACK_LDS_BYTES) ? 1 : 2;
    const dim3 grid((a.B + rows - 1) / rows);
    const bool lean = force >= 0 ? force == 1 : (int)grid.x > cus;
    const size_t lds = rows * row_bytes;
    if (rows == 1 && lean)
        PCM_LAUNCH((stack_step_fwd_kernel<true, 1>), grid,
                           dim3(STHREADS), lds, stream, a);
    else if (rows == 1)
        PCM_LAUNCH((stack_step_fwd_kernel<false, 1>), grid,
                           dim3(STHREADS), lds, stream, a);
    else if (lean)
        PCM_LAUNCH((stack_step_fwd_kernel<true, 2>), grid,
                           dim3(STHREADS), lds, stream, a);
    else
        PCM_LAUNCH((stack_step_fwd_kernel<false, 2>), grid,
                           dim3(STHREADS), lds, stream, a);
    return 0;
}

int pcm_launch_pack(const PackArgs &a, hipStream_t stream) {
    hipLaunchKernelGGL(pack_kernel, dim3(grid_for(3LL * a.Hn * a.Kp)),
                       dim3(256), 0, stream, a);
    return 0;
}

int pcm_launch_copy_rows(const float *src, long long lds_, float *dst,
                         long long ldd, int rows, int cols,
                         hipStream_t stream) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for((long long)rows * cols)),
                       dim3(256), 0, stream, src, lds_, dst, ldd, rows, cols,
                       cols);
    return 0;
}

int pcm_launch_add_block(const float *src, long long lds_, float *dst,
                         long long ldd, int rows, int cols, int accumulate,
                         hipStream_t stream) {
    hipLaunchKernelGGL(add_block_kernel, dim3(grid_for((long long)rows * cols)),
                       dim3(256), 0, stream, src, lds_, dst, ldd, rows, cols,
                       accumulate);
    return 0;
}

int pcm_launch_fill(float *dst, float value, long long n, hipStream_t stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, stream, dst,
                       value, n);
    return 0;
}

int pcm_launch_init_state(const SGLayout &lay, float *saved, const float *h0,
                          long long h0_layer_stride, const float *stack0,
                          hipStream_t stream) {
    for (int l = 0; l < lay.L; ++l) {
        const float *src = h0 ? h0 + (long long)l * h0_layer_stride : nullptr;
        float *dst = saved + lay.off_hs[l];
        hipLaunchKernelGGL(copy_rows_kernel,
                           dim3(grid_for((long long)lay.B * lay.Hp)), dim3(256),
                           0, stream, src, (long long)lay.H, dst,
                           (long long)lay.Hp, lay.B, lay.H, lay.Hp);
    }
    if (lay.use_stack) {
        float *dst = saved + lay.s_index(0);
        hipLaunchKernelGGL(copy_rows_kernel,
                           dim3(grid_for((long long)lay.B * lay.DW)), dim3(256),
                           0, stream, stack0, (long long)lay.DW, dst,
                           (long long)lay.DW, lay.B, lay.DW, lay.DW);
    }
    return 0;
}
