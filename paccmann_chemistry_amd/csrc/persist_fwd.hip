// Persistent, weight-stationary forward recurrence of the StackGRU
// (stack_rnn.py:131-232 looped as vae.py:111-113 / 337-341): ONE launch walks
// all T steps of a batch of up to 512 rows.
//
// Decomposition (H = 512, L = 2, stack width <= 64: the BASELINE shape):
//   - the batch is cut into groups of 64 rows; group g is served by the 32
//     blocks with blockIdx % 8 == g (observed to share an XCD: an L2 locality
//     choice, never a correctness assumption);
//   - block ci of a group owns hidden units [16 ci, 16 ci + 16) of BOTH layers
//     for all 64 rows of the group: its 48 gate columns x K of W_ih / W_hh live
//     in REGISTERS for the whole sequence (4 waves x (9 + 16) chunks x 3 gates x
//     f32x4 = 300 registers per lane, one wave per SIMD), so no weight byte is
//     re-read after the prologue;
//   - a cell is [64 x K] . [K x 48] on 16x16x4 fp32 MFMAs, K split four ways
//     over the waves (wave w owns the 16-wide k slice w of every 64-wide chunk),
//     activations loaded straight from global memory into MFMA operand
//     registers (16 B per lane, several chunks in flight), the four partial
//     tiles meet in LDS, then the usual gate math;
//   - the stack step of a row (controls, push value, [D, W] update) is done by
//     the block that owns rows {2 ci, 2 ci + 1} of the group.
//
// Blocks of a group exchange three things per step -- the stack tops, h0_t and
// h1_t -- through global memory with the placement-independent hand-off of the
// programming guide (Guideline 16, R1, counter form): every handed-off byte is
// stored write-through (sc1, 16 B), every storing wave drains (vmcnt(0)), the
// block barriers, ONE lane adds to the group's counter; consumers poll that
// counter relaxed and read the bytes with sc1 loads only.  Each step writes
// fresh addresses (the per-step slots of the saved workspace).  Counters are
// zeroed by a memset ahead of the launch; every spin is bounded.
#include <stdio.h>
#include <stdlib.h>

#include "stackgru.h"

namespace {

constexpr int PH = 512;          // hidden size (= Hp = Hn)
constexpr int PHC = PH / 64;     // 64-wide chunks of a hidden vector
constexpr int PROWS = 64;        // rows of a group
constexpr int PMEM = 32;         // blocks of a group
constexpr int PCOLS = 16;        // hidden units of a block
constexpr int PTHREADS = 256;    // 4 waves, one per SIMD
constexpr int PWP = 64;          // padded stack width
constexpr int NCH0 = 1 + PHC;    // chunks of layer 0: stack top, h0
constexpr int NCH1 = 2 * PHC;    // chunks of layer 1: h0 (input), h1
constexpr int RED_LD = 20;       // padded row of a 16-wide partial tile in LDS
constexpr int RED_FLOATS = 4 * 3 * PROWS * RED_LD;   // [wave][gate][row][RED_LD]
constexpr int HID_FLOATS = 3 * PROWS * RED_LD;       // [gate][row][RED_LD]
// LDS besides the block's two stack rows (double buffered): partial tiles,
// hidden-side sums of both layers, control partials, h_{t-1}, biases, misc
constexpr int LDS_FIXED_FLOATS = RED_FLOATS + 2 * HID_FLOATS + 4 * PHC * 2 * 16 +
                                 4 * PROWS * 2 +
                                 2 * PROWS * PCOLS + 2 * 4 * PCOLS + 2 * 64 + 16;
constexpr int LDS_MAX_FLOATS = 160 * 1024 / 4;
constexpr int SYNC_STRIDE = 32;  // words between counters (128 B lines)
constexpr unsigned SPIN_LIMIT = 1u << 22;   // polls before a wait gives up

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __amdgpu_buffer_rsrc_t i32x4s;   // 128-bit buffer descriptor
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SC1 16   // buffer-op aux bit: system-coherent (write-through / L1 bypass)

__device__ __forceinline__ i32x4s make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 load_sc1(i32x4s rsrc, unsigned voff,
                                          unsigned soff) {
    return __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, SC1));
}
__device__ __forceinline__ void store_sc1(f32x4 v, i32x4s rsrc, unsigned voff,
                                          unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                           voff, soff, SC1);
}

// ---- group hand-off: arrive / wait on a monotonic counter ---------------------------
// arrive: called by every thread of the block after its sc1 stores
__device__ __forceinline__ void group_arrive(gu32 *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}
// wait: one lane polls relaxed; returns false when the spin limit was hit or
// another wait already failed (the error word), uniformly for the block
__device__ __forceinline__ bool group_wait(gu32 *counter, unsigned target,
                                           gu32 *error, int *lds_flag) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT ||
                ((spins & 1023) == 0 &&
                 __hip_atomic_load(error, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        *lds_flag = ok;
    }
    __syncthreads();
    const bool ok = *lds_flag != 0;
    // (no instruction: keeps the compiler from hoisting loads above the poll)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return ok;
}

struct PersistFwdArgs {
    // regions of the saved workspace (float pointers) and their byte sizes
    float *hs[2];          // [(T+1)][B][PH]
    float *gates[2];       // [T][B][4 PH] or null (inference)
    float *xdrop;          // [T][B][PH] dropped input of layer 1, or null
    float *S;              // training: [(T+1)][B][DW], else [2][B][DW]
    float *av;             // training: [T][B][AVW], else [B][AVW]
    float *xtop;           // [T][B][PWP] (per step in both modes)
    float *cpre;           // [T][B][64] stack control / push pre-activations
    const float *tokg;     // [V][3 PH]
    const float *wpack[2]; // [3 PH][Kp]
    const float *bpack[2]; // [4 PH]
    const float *ctrl_w, *ctrl_b, *push_w, *push_b;
    const int64_t *tokens; // [T][B]
    int32_t *status;
    unsigned *sync;        // [8][4][SYNC_STRIDE] counters, then the error word
    int B, T, V, D, W, AVW, DW, training;
    float drop_p, inv_keep;
    uint64_t seed;
    unsigned long long *dbg;   // diagnostics build only: phase clocks of block 0
    int variant;               // diagnostics: bit 0 plain loads, bit 1 no MFMA
};

#ifdef PCM_PERSIST_STAMPS
#define STAMP(i)                                                        \
    do {                                                                \
        if (A.dbg && blockIdx.x == 0 && threadIdx.x == 0) {             \
            const unsigned long long now_ = wall_clock64();             \
            A.dbg[i] += now_ - last_stamp;                              \
            last_stamp = now_;                                          \
        }                                                               \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// One K chunk (64 wide) of a cell for this wave: 4 row tiles x 3 gates x 4 k-steps.
// a[mt]: the wave's 16-wide k slice of rows 16 mt + (lane & 15), four consecutive
// k per lane; w[gate]: the matching weights of hidden unit (lane & 15).
__device__ __forceinline__ void chunk_mfma(const f32x4 (&a)[4],
                                           const f32x4 (&w)[3],
                                           f32x4 (&acc_r)[4], f32x4 (&acc_z)[4],
                                           f32x4 (&acc_n)[4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            acc_r[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][ks], w[0][ks],
                                                            acc_r[mt], 0, 0, 0);
            acc_z[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][ks], w[1][ks],
                                                            acc_z[mt], 0, 0, 0);
            acc_n[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][ks], w[2][ks],
                                                            acc_n[mt], 0, 0, 0);
        }
}

// issue the sc1 loads of one chunk's activation fragments
__device__ __forceinline__ void chunk_load(f32x4 (&a)[4], i32x4s rsrc,
                                           unsigned voff, unsigned soff,
                                           unsigned row_tile_bytes) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        a[mt] = load_sc1(rsrc, voff + mt * row_tile_bytes, soff);
}

// NCH chunks of one operand ([64 rows][ld floats], chunk c at column 64 c) times
// the register-resident weights w[C0 + c]; PF chunks of loads kept in flight.
// CTRL: the same fragments also feed the block's two stack control / push rows
// (cwl: their k slices in LDS, [chunk][output][lane group] x f32x4 from this
// lane's base; VALU dot products accumulated per lane in cdot[row tile][output]).
template <int NCH, int C0, int NW, bool CTRL>
__device__ __forceinline__ void gemm_side(const f32x4 (&w)[NW][3], i32x4s rsrc,
                                          unsigned voff, unsigned row_tile_bytes,
                                          f32x4 (&acc_r)[4], f32x4 (&acc_z)[4],
                                          f32x4 (&acc_n)[4],
                                          const float *cwl,
                                          float (&cdot)[4][2]) {
    constexpr int PF = NCH < 3 ? NCH : 3;
    f32x4 a[PF][4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        chunk_load(a[p], rsrc, voff, p * 256u, row_tile_bytes);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 cur[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) cur[mt] = a[c % PF][mt];
        if (c + PF < NCH)
            chunk_load(a[c % PF], rsrc, voff, (c + PF) * 256u, row_tile_bytes);
        chunk_mfma(cur, w[C0 + c], acc_r, acc_z, acc_n);
        if (CTRL) {
            // the control / push weights of this (wave, lane group) k slice:
            // 16 lanes share an address (LDS broadcast)
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const f32x4 cwv = *(const f32x4 *)(cwl + (c * 2 + o) * 16);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        cdot[mt][o] = fmaf(cur[mt][e], cwv[e], cdot[mt][o]);
            }
        }
    }
}

// partial tiles of this wave -> LDS: red[(wave * 3 + gate) * PROWS + row][unit]
__device__ __forceinline__ void park(float *red, int wave, int gate, int lane,
                                     const f32x4 (&acc)[4]) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            red[((wave * 3 + gate) * PROWS + 16 * mt + 4 * g + e) * RED_LD + r16] =
                acc[mt][e];
}
// sum of the four waves' partials for (row, units 4 q .. 4 q + 3) of one gate
__device__ __forceinline__ f32x4 gather(const float *red, int gate, int row,
                                        int q) {
    f32x4 s = *(const f32x4 *)(red + ((0 * 3 + gate) * PROWS + row) * RED_LD + 4 * q);
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const f32x4 v =
            *(const f32x4 *)(red + ((w * 3 + gate) * PROWS + row) * RED_LD + 4 * q);
        s += v;
    }
    return s;
}

__device__ __forceinline__ void zero4(f32x4 (&acc)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// hidden-side sums of a cell (r, z, n_h): parked partials -> hid[gate][row][unit]
__device__ __forceinline__ void reduce_hidden(const float *red, float *hid) {
    for (int idx = threadIdx.x; idx < 3 * PROWS * 4; idx += PTHREADS) {
        const int gate = idx / (PROWS * 4), rem = idx - gate * PROWS * 4;
        const int row = rem >> 2, q = rem & 3;
        *(f32x4 *)(hid + (gate * PROWS + row) * RED_LD + 4 * q) =
            gather(red, gate, row, q);
    }
}

// ---- the stack step of the block's two rows ---------------------------------------------
// a = softmax(c[0:3] + bc), v = tanh(c[3:] + bs) from the pre-activations the group
// has just handed over (cpre), then S' = a_noop S + a_push up + a_pop down in the
// reference's add order (stack_rnn.py:188-232).  The rows' stacks stay in LDS for
// the whole sequence (Sold -> Snew); the new stack goes to the saved workspace,
// its top row -- layer 0's input -- to every block of the group (sc1).
__device__ __forceinline__ void stack_phase(const PersistFwdArgs &A, float *cu,
                                            const float *Sold, float *Snew, int t,
                                            int row0, int nrows, i32x4s rs_cpre,
                                            i32x4s rs_xtop) {
    const int tid = threadIdx.x;
    const int W = A.W, D = A.D, DW = A.DW, NO = 3 + W, AVW = A.AVW;
    const int DWp = (DW + 3) & ~3;
    if (nrows <= 0) return;
    if (tid < 32) {
        const int r = tid >> 4, q4 = tid & 15;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r < nrows)
            v = load_sc1(rs_cpre, (unsigned)(((row0 + r) * 64 + 4 * q4) * 4), 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = 4 * q4 + e;
            const float bias = o < 3 ? A.ctrl_b[o] : (o < NO ? A.push_b[o - 3] : 0.f);
            cu[64 * r + o] = o < 3 ? v[e] + bias : tanhf(v[e] + bias);
        }
    }
    __syncthreads();
    if (tid < 2) {
        float *c = cu + 64 * tid;
        const float m = fmaxf(c[0], fmaxf(c[1], c[2]));
        const float e0 = expf(c[0] - m), e1 = expf(c[1] - m), e2 = expf(c[2] - m);
        const float s = e0 + e1 + e2;
        c[0] = e0 / s; c[1] = e1 / s; c[2] = e2 / s;
    }
    __syncthreads();
    if (A.training) {   // a, v for the reverse pass (own rows: plain stores)
        float *av = A.av + (size_t)t * A.B * AVW;
        for (int idx = tid; idx < nrows * AVW; idx += PTHREADS) {
            const int r = idx / AVW, o = idx - r * AVW;
            av[(size_t)(row0 + r) * AVW + o] = o < NO ? cu[64 * r + o] : 0.f;
        }
    }
    const bool to_global = A.training || t == A.T - 1;
    float *Sn = A.S + (size_t)(A.training ? t + 1 : ((t + 1) & 1)) * A.B * DW;
    const float inv_W = 1.0f / (float)W;
    const bool vec = (DW & 3) == 0;
    const int nq = (DW + 3) >> 2;
    for (int idx = tid; idx < nrows * nq; idx += PTHREADS) {
        const int r = idx >= nq ? 1 : 0, q = idx - r * nq;
        const float *Sr = Sold + r * DWp;
        const float *c = cu + 64 * r;
        f32x4 nv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * q + i;
            float x = 0.f;
            if (e < DW) {
                const int d = pcm_fdiv(e, inv_W), w = e - d * W;
                const float sv = Sr[e];
                const float up = d == 0 ? c[3 + w] : Sr[e - W];
                const float dn = d == D - 1 ? 0.f : Sr[e + W];
                x = __fadd_rn(__fadd_rn(__fmul_rn(c[2], sv), __fmul_rn(c[0], up)),
                              __fmul_rn(c[1], dn));
            }
            nv[i] = x;
        }
        *(f32x4 *)(Snew + r * DWp + 4 * q) = nv;
        if (to_global) {
            float *dst = Sn + (size_t)(row0 + r) * DW;
            if (vec) {
                *(f32x4 *)(dst + 4 * q) = nv;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (4 * q + i < DW) dst[4 * q + i] = nv[i];
            }
        }
    }
    __syncthreads();
    // stack top (row 0 of the new stack), zero padded to PWP: 2 x 16 lanes x 16 B
    if (tid < 32) {
        const int r = tid >> 4, q4 = tid & 15;
        if (r < nrows) {
            f32x4 top;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * q4 + i;
                top[i] = e < W ? Snew[r * DWp + e] : 0.f;
            }
            store_sc1(top, rs_xtop, (unsigned)(((row0 + r) * PWP + 4 * q4) * 4), 0);
        }
    }
}

// ---- gate math of one cell for this block's [64 rows x 16 units] ------------------------
// thread (row = tid / 4, q = tid % 4) finishes units 4 q .. 4 q + 3 of its row:
// input-side sums from the parked partials, hidden-side sums from hid, biases
// and h_{t-1} from LDS, token gates prefetched by the caller (layer 0)
__device__ __forceinline__ void cell_epilogue(
    const PersistFwdArgs &A, const float *red, const float *hid, const float *bias,
    float *hprev, int layer, int t, int row0, int nrows, int j0, f32x4 tr,
    f32x4 tz, f32x4 tn, i32x4s rs_hnew, i32x4s rs_xdrop) {
    const int tid = threadIdx.x;
    const int row = tid >> 2, q = tid & 3;
    if (row >= nrows) return;
    const int b = row0 + row, j = j0 + 4 * q;
    const f32x4 sr = gather(red, 0, row, q) +
                     *(const f32x4 *)(hid + (0 * PROWS + row) * RED_LD + 4 * q);
    const f32x4 sz = gather(red, 1, row, q) +
                     *(const f32x4 *)(hid + (1 * PROWS + row) * RED_LD + 4 * q);
    const f32x4 sni = gather(red, 2, row, q);
    const f32x4 snh = *(const f32x4 *)(hid + (2 * PROWS + row) * RED_LD + 4 * q);
    const f32x4 b_r = *(const f32x4 *)(bias + 0 * PCOLS + 4 * q);
    const f32x4 b_z = *(const f32x4 *)(bias + 1 * PCOLS + 4 * q);
    const f32x4 b_ni = *(const f32x4 *)(bias + 2 * PCOLS + 4 * q);
    const f32x4 b_nh = *(const f32x4 *)(bias + 3 * PCOLS + 4 * q);
    const f32x4 hp = *(const f32x4 *)(hprev + row * PCOLS + 4 * q);
    f32x4 r_, z_, n_, ghn, hn;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        r_[e] = pcm_sigmoid(sr[e] + b_r[e] + tr[e]);
        z_[e] = pcm_sigmoid(sz[e] + b_z[e] + tz[e]);
        ghn[e] = snh[e] + b_nh[e];
        n_[e] = tanhf(sni[e] + b_ni[e] + tn[e] + r_[e] * ghn[e]);
        // ATen GRU cell: (h - n) * z + n
        hn[e] = __fadd_rn(__fmul_rn(hp[e] - n_[e], z_[e]), n_[e]);
    }
    *(f32x4 *)(hprev + row * PCOLS + 4 * q) = hn;
    store_sc1(hn, rs_hnew, (unsigned)((b * PH + j) * 4), 0);
    if (A.gates[layer]) {
        float *g = A.gates[layer] + ((size_t)t * A.B + b) * 4 * PH + j;
        *(f32x4 *)(g) = r_;
        *(f32x4 *)(g + PH) = z_;
        *(f32x4 *)(g + 2 * PH) = n_;
        *(f32x4 *)(g + 3 * PH) = ghn;
    }
    if (layer == 0 && A.xdrop) {
        f32x4 xd;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            xd[e] = hn[e] * pcm_dropout_scale(A.seed, 1, t, b, j + e, A.drop_p,
                                              A.inv_keep);
        store_sc1(xd, rs_xdrop, (unsigned)((b * PH + j) * 4), 0);
    }
}

__global__ __launch_bounds__(PTHREADS, 1) void stackgru_persist_fwd_kernel(
    PersistFwdArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int DWp = (A.DW + 3) & ~3;
    float *red = lds;                          // [4][3][PROWS][RED_LD]
    float *hid = red + RED_FLOATS;             // [2 layers][3][PROWS][RED_LD]
    float *cwlds = hid + 2 * HID_FLOATS;       // [4 waves][PHC][2][4 groups][4]
    float *cred = cwlds + 4 * PHC * 2 * 16;    // [4][PROWS][2]
    float *hprev = cred + 4 * PROWS * 2;       // [2 layers][PROWS][PCOLS]
    float *bias = hprev + 2 * PROWS * PCOLS;   // [2 layers][4][PCOLS]
    float *cu = bias + 2 * 4 * PCOLS;          // [2][64]
    int *flag = (int *)(cu + 2 * 64);          // [16]
    float *Sbuf = cu + 2 * 64 + 16;            // [2 slots][2 rows][DWp]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.x & 7, ci = blockIdx.x >> 3;
    const int grow0 = group * PROWS;
    const int nrows = min(PROWS, A.B - grow0);
    if (nrows <= 0 || ci >= PMEM) return;   // a group without rows has no hand-offs
    const int j0 = ci * PCOLS;
    const int r16 = lane & 15, g = lane >> 4;
    gu32 *cnt_c = (gu32 *)(A.sync + (group * 4 + 0) * SYNC_STRIDE);
    gu32 *cnt_x = (gu32 *)(A.sync + (group * 4 + 1) * SYNC_STRIDE);
    gu32 *cnt_0 = (gu32 *)(A.sync + (group * 4 + 2) * SYNC_STRIDE);
    gu32 *cnt_1 = (gu32 *)(A.sync + (group * 4 + 3) * SYNC_STRIDE);
    gu32 *error = (gu32 *)(A.sync + 8 * 4 * SYNC_STRIDE);
    const int NO = 3 + A.W;
    const int srow0 = grow0 + 2 * ci;                 // stack rows of this block
    const int snrows = max(0, min(2, grow0 + nrows - srow0));

    // ---- weights of this block's 48 gate columns -> registers ---------------------
    f32x4 w0[NCH0][3], w1[NCH1][3];
    {
        const int Kp0 = NCH0 * 64, Kp1 = NCH1 * 64;
#pragma unroll
        for (int c = 0; c < NCH0; ++c)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                w0[c][gate] = *(const f32x4 *)(A.wpack[0] +
                    (size_t)(gate * PH + j0 + r16) * Kp0 + 64 * c + 16 * wave + 4 * g);
#pragma unroll
        for (int c = 0; c < NCH1; ++c)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                w1[c][gate] = *(const f32x4 *)(A.wpack[1] +
                    (size_t)(gate * PH + j0 + r16) * Kp1 + 64 * c + 16 * wave + 4 * g);
    }
    // the block's two stack control / push rows (outputs 2 ci, 2 ci + 1) -> LDS,
    // laid out as the lanes of each wave consume them: [wave][chunk][o][g][4]
    for (int idx = tid; idx < 4 * PHC * 2 * 16; idx += PTHREADS) {
        const int e = idx & 3, gg = (idx >> 2) & 3, o = (idx >> 4) & 1;
        const int c = (idx >> 5) % PHC, wv = idx / (PHC * 32);
        const int out = 2 * ci + o;
        float v = 0.f;
        if (out < NO) {
            const float *wrow = out < 3 ? A.ctrl_w + (size_t)out * PH
                                        : A.push_w + (size_t)(out - 3) * PH;
            v = wrow[64 * c + 16 * wv + 4 * gg + e];
        }
        cwlds[idx] = v;
    }
    // biases and initial hidden state of the block's units, initial stacks of its rows
    for (int idx = tid; idx < 2 * 4 * PCOLS; idx += PTHREADS) {
        const int l = idx / (4 * PCOLS), rem = idx - l * 4 * PCOLS;
        const int k = rem / PCOLS, u = rem - k * PCOLS;
        bias[idx] = A.bpack[l][k * PH + j0 + u];
    }
    for (int idx = tid; idx < 2 * PROWS * PCOLS; idx += PTHREADS) {
        const int l = idx / (PROWS * PCOLS), rem = idx - l * PROWS * PCOLS;
        const int row = rem / PCOLS, u = rem - row * PCOLS;
        hprev[idx] = row < nrows ? A.hs[l][(size_t)(grow0 + row) * PH + j0 + u] : 0.f;
    }
    for (int idx = tid; idx < 2 * DWp; idx += PTHREADS) {
        const int r = idx >= DWp ? 1 : 0, e = idx - r * DWp;
        Sbuf[idx] = (r < snrows && e < A.DW)
                        ? A.S[(size_t)(srow0 + r) * A.DW + e] : 0.f;
    }
    __syncthreads();

    const unsigned voff_h = (unsigned)(((grow0 + r16) * PH + 16 * wave + 4 * g) * 4);
    const unsigned voff_x = (unsigned)(((grow0 + r16) * PWP + 16 * wave + 4 * g) * 4);
    const unsigned slab_h = (unsigned)A.B * PH * 4, slab_x = (unsigned)A.B * PWP * 4;
    const int erow = tid >> 2, eq = tid & 3;          // epilogue element of this thread
    const int eb = grow0 + min(erow, nrows - 1);
    long long tk_next = A.tokens[eb];                 // token of step 0
    float cdot_none[4][2] = {};
    const float *cwl = cwlds + wave * (PHC * 32) + g * 4;

#ifdef PCM_PERSIST_STAMPS
    unsigned long long last_stamp = wall_clock64();
#endif
    for (int t = 0; t < A.T; ++t) {
        const i32x4s rs_h0p = make_rsrc(A.hs[0] + (size_t)t * A.B * PH, slab_h);
        const i32x4s rs_h1p = make_rsrc(A.hs[1] + (size_t)t * A.B * PH, slab_h);
        const i32x4s rs_h0n = make_rsrc(A.hs[0] + (size_t)(t + 1) * A.B * PH, slab_h);
        const i32x4s rs_h1n = make_rsrc(A.hs[1] + (size_t)(t + 1) * A.B * PH, slab_h);
        const i32x4s rs_xt = make_rsrc(A.xtop + (size_t)t * A.B * PWP, slab_x);
        const i32x4s rs_cp = make_rsrc(A.cpre + (size_t)t * A.B * 64, slab_x);
        const i32x4s rs_xd = make_rsrc(
            A.xdrop ? A.xdrop + (size_t)t * A.B * PH : A.hs[0], slab_h);
        const unsigned target = (unsigned)(t + 1) * PMEM;
        f32x4 acc_r[4], acc_z[4], acc_n[4];

        // ---- layer 1, hidden side (h1_{t-1}) + the stack control products -----------
        if (t > 0 && !group_wait(cnt_1, (unsigned)t * PMEM, error, flag)) return;
        STAMP(0);
        {
            float cdot[4][2] = {};
            zero4(acc_r); zero4(acc_z); zero4(acc_n);
            gemm_side<PHC, PHC, NCH1, true>(w1, rs_h1p, voff_h, 16u * PH * 4, acc_r,
                                            acc_z, acc_n, cwl, cdot);
            STAMP(1);
            park(red, wave, 0, lane, acc_r);
            park(red, wave, 1, lane, acc_z);
            park(red, wave, 2, lane, acc_n);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    float v = cdot[mt][o];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (g == 0) cred[(wave * PROWS + 16 * mt + r16) * 2 + o] = v;
                }
        }
        __syncthreads();
        reduce_hidden(red, hid + HID_FLOATS);
        if (tid < PROWS) {
            float c0 = 0.f, c1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                c0 += cred[(w * PROWS + tid) * 2 + 0];
                c1 += cred[(w * PROWS + tid) * 2 + 1];
            }
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 pair = {__float_as_uint(c0), __float_as_uint(c1)};
            if (tid < nrows)
                __builtin_amdgcn_raw_buffer_store_b64(
                    pair, rs_cp, (unsigned)(((grow0 + tid) * 64 + 2 * ci) * 4), 0, SC1);
        }
        group_arrive(cnt_c);
        STAMP(2);

        // ---- layer 0, hidden side (h0_{t-1}) ------------------------------------------
        zero4(acc_r); zero4(acc_z); zero4(acc_n);
        gemm_side<PHC, 1, NCH0, false>(w0, rs_h0p, voff_h, 16u * PH * 4, acc_r, acc_z,
                                       acc_n, cwl, cdot_none);
        STAMP(3);
        park(red, wave, 0, lane, acc_r);
        park(red, wave, 1, lane, acc_z);
        park(red, wave, 2, lane, acc_n);
        __syncthreads();
        reduce_hidden(red, hid);
        STAMP(4);

        // ---- the stack step of this block's two rows ----------------------------------
        if (!group_wait(cnt_c, target, error, flag)) return;
        STAMP(5);
        stack_phase(A, cu, Sbuf + (t & 1) * 2 * DWp, Sbuf + ((t + 1) & 1) * 2 * DWp, t,
                    srow0, snrows, rs_cp, rs_xt);
        STAMP(6);
        group_arrive(cnt_x);
        STAMP(7);

        // ---- layer 0, input side (the stack tops) and gate math -----------------------
        // (token gates of this step: fetched now, consumed after the x product)
        f32x4 tr, tz, tn;
        {
            long long tk = tk_next;
            if (tk < 0 || tk >= A.V) {
                tk = tk < 0 ? 0 : A.V - 1;
                if (A.status && erow < nrows) atomicOr(A.status, PCM_STATUS_BAD_TOKEN);
            }
            const float *tg = A.tokg + tk * (3LL * PH) + j0 + 4 * eq;
            tr = *(const f32x4 *)(tg);
            tz = *(const f32x4 *)(tg + PH);
            tn = *(const f32x4 *)(tg + 2 * PH);
            if (t + 1 < A.T) tk_next = A.tokens[(size_t)(t + 1) * A.B + eb];
        }
        zero4(acc_r); zero4(acc_z); zero4(acc_n);
        if (!group_wait(cnt_x, target, error, flag)) return;
        STAMP(8);
        gemm_side<1, 0, NCH0, false>(w0, rs_xt, voff_x, 16u * PWP * 4, acc_r, acc_z,
                                     acc_n, cwl, cdot_none);
        park(red, wave, 0, lane, acc_r);
        park(red, wave, 1, lane, acc_z);
        park(red, wave, 2, lane, acc_n);
        __syncthreads();
        STAMP(9);
        cell_epilogue(A, red, hid, bias, hprev, 0, t, grow0, nrows, j0, tr, tz, tn,
                      rs_h0n, rs_xd);
        STAMP(10);
        group_arrive(cnt_0);
        STAMP(11);

        // ---- layer 1, input side (h0_t, dropped in training) and gate math ------------
        zero4(acc_r); zero4(acc_z); zero4(acc_n);
        if (!group_wait(cnt_0, target, error, flag)) return;
        STAMP(12);
        gemm_side<PHC, 0, NCH1, false>(w1, A.xdrop ? rs_xd : rs_h0n, voff_h,
                                       16u * PH * 4, acc_r, acc_z, acc_n, cwl,
                                       cdot_none);
        STAMP(13);
        park(red, wave, 0, lane, acc_r);
        park(red, wave, 1, lane, acc_z);
        park(red, wave, 2, lane, acc_n);
        __syncthreads();
        STAMP(14);
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            cell_epilogue(A, red, hid + HID_FLOATS, bias + 4 * PCOLS,
                          hprev + PROWS * PCOLS, 1, t, grow0, nrows, j0, zero, zero,
                          zero, rs_h1n, rs_xd);
        }
        STAMP(15);
        group_arrive(cnt_1);
        STAMP(16);
    }
}

// a hand-off that timed out leaves its mark in the caller's status word
__global__ void persist_report_kernel(const unsigned *error, int32_t *status) {
    if (*error != 0) atomicOr(status, PCM_STATUS_TIMEOUT);
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------
size_t pcm_persist_fwd_sync_words() { return 8 * 4 * SYNC_STRIDE + SYNC_STRIDE; }

int pcm_persist_fwd_mode() {
    // PCM_PERSIST: 0 never, 1 the first persistent kernel (this file), 2 the
    // role-split one (persist2_fwd.hip); unset: see pcm_persist_fwd_supported
    static const int mode = [] {
        const char *e = getenv("PCM_PERSIST");
        return e ? atoi(e) : -1;
    }();
    return mode;
}

bool pcm_persist_fwd_supported(const SGLayout &lay, const int32_t *lengths) {
    const int mode = pcm_persist_fwd_mode();
    if (mode == 0) return false;
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount,
                                        dev);
        return n;
    }();
    if (lengths || cus < 8 * PMEM) return false;
    if (mode == 2) return pcm_persist2_fwd_shape_ok(lay);
    const bool shape = lay.H == PH && lay.L == 2 && lay.use_stack && lay.W <= PWP &&
                       lay.W >= 1 && 3 + lay.W <= 56 && lay.AVW <= 64 &&
                       lay.B <= 8 * PROWS &&
                       LDS_FIXED_FLOATS + 4 * ((lay.DW + 3) & ~3) <= LDS_MAX_FLOATS;
    if (!shape) return false;
    // Measured on MI355X at batch 512: 48 us per step against 42.7 us
    // for the launch chain, so the chain stays the default; PCM_PERSIST=1 selects
    // this kernel wherever the shape allows.
    return mode == 1;
}

int pcm_launch_persist_fwd(const SGLayout &lay, float *sv,
                           const pcm_stackgru_params *w, const int64_t *tokens,
                           float dropout_p, uint64_t seed, int32_t *status,
                           unsigned *sync, hipStream_t stream) {
    PersistFwdArgs a;
    for (int l = 0; l < 2; ++l) {
        a.hs[l] = sv + lay.off_hs[l];
        a.gates[l] = lay.training ? sv + lay.off_gates[l] : nullptr;
        a.wpack[l] = sv + lay.off_wpack[l];
        a.bpack[l] = sv + lay.off_bpack[l];
    }
    a.xdrop = lay.dropout ? sv + lay.off_xdrop[1] : nullptr;
    a.S = sv + lay.off_S; a.av = sv + lay.off_av; a.xtop = sv + lay.off_xtop;
    a.cpre = sv + lay.off_cpre;
    a.tokg = sv + lay.off_tokg;
    a.ctrl_w = w->ctrl_w; a.ctrl_b = w->ctrl_b;
    a.push_w = w->push_w; a.push_b = w->push_b;
    a.tokens = tokens; a.status = status; a.sync = sync;
    a.B = lay.B; a.T = lay.T; a.V = lay.V; a.D = lay.D; a.W = lay.W;
    a.AVW = lay.AVW; a.DW = lay.DW; a.training = lay.training;
    a.drop_p = dropout_p;
    a.inv_keep = dropout_p < 1.f ? 1.f / (1.f - dropout_p) : 0.f;
    a.seed = seed;
    const size_t lds =
        (size_t)(LDS_FIXED_FLOATS + 4 * ((lay.DW + 3) & ~3)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(
            (const void *)stackgru_persist_fwd_kernel,
            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_FLOATS * 4);
        PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "persist_fwd: LDS attribute: %s",
                    hipGetErrorString(e));
        attr_set = true;
    }
    hipError_t e = hipMemsetAsync(sync, 0, pcm_persist_fwd_sync_words() * 4, stream);
    PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "persist_fwd: memset: %s",
                hipGetErrorString(e));
    a.dbg = nullptr;
    a.variant = 0;
#ifdef PCM_PERSIST_STAMPS
    static unsigned long long *dbg = nullptr;
    static int calls = 0;
    if (!dbg) (void)hipMalloc(&dbg, 32 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg, 0, 32 * sizeof(unsigned long long), stream);
    a.dbg = dbg;
#endif
    hipLaunchKernelGGL(stackgru_persist_fwd_kernel, dim3(8 * PMEM), dim3(PTHREADS),
                       lds, stream, a);
#ifdef PCM_PERSIST_STAMPS
    if (++calls == 6) {   // a warm call: print where block 0's time went
        unsigned long long h[32];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        static const char *name[17] = {
            "wait h1", "cell1 hidden gemm", "cell1 hid reduce+arrive c",
            "cell0 hidden gemm", "cell0 hid reduce", "wait c", "stack",
            "arrive x", "wait x", "cell0 x gemm+park", "cell0 epilogue",
            "arrive 0", "wait 0", "cell1 x gemm", "cell1 park",
            "cell1 epilogue", "arrive 1"};
        double total = 0;
        for (int i = 0; i < 17; ++i) total += (double)h[i];
        for (int i = 0; i < 17; ++i)
            fprintf(stderr, "persist phase %-26s %8.2f us/step\n", name[i],
                    (double)h[i] * 0.01 / lay.T);
        fprintf(stderr, "persist total %8.2f us/step\n", total * 0.01 / lay.T);
    }
#endif
    if (status)
        hipLaunchKernelGGL(persist_report_kernel, dim3(1), dim3(1), 0, stream,
                           sync + 8 * 4 * SYNC_STRIDE, status);
    return 0;
}
