// Persistent, weight-resident forward recurrence of the StackGRU, second design
// (stack_rnn.py:131-232 looped as vae.py:111-113 / 337-341): ONE launch walks all
// T steps of a batch of up to 512 rows, and inside every workgroup the work of a
// step is split between two ROLES of four waves each that run side by side on
// the CU's four SIMDs (one wave of each role per SIMD):
//
//   X, the critical chain of a step:   wait h1_{t-1} -> stack-control products
//        -> stack step of the block's two rows -> layer 0: stack-top product +
//        gate math -> layer 1: input-side product (h0_t) + gate math;
//   Y, everything the chain does not wait for: the two hidden-side products
//        h1_{t-1} . W_hh1 and h0_t . W_hh0 (two thirds of a step's MFMA work),
//        which only need a hidden state that was published a phase earlier and
//        whose sums X picks up from LDS when it reaches the gate math.
//
// So the 10 us of hidden-side MFMAs of a step run UNDER X's hand-off waits, VALU
// phases and stores instead of beside them (the first persistent kernel,
// persist_fwd.hip, ran all of it in one instruction stream: 48 us per step).
//
// Decomposition (H = 512, L = 2, stack width <= 64: the BASELINE shape), as in
// persist_fwd.hip: the batch is cut into groups of 64 rows; group g is served by
// the 32 blocks with blockIdx % 8 == g (observed to share an XCD: an L2
// locality choice, never a correctness assumption); block ci owns hidden units
// [16 ci, 16 ci + 16) of BOTH layers for the 64 rows of its group, and the stack
// rows {2 ci, 2 ci + 1} of the group.  A product is [64 x K] . [K x 48] on
// 16x16x4 fp32 MFMAs, K split four ways over the role's waves; the partial tiles
// meet in LDS.  Weights: X keeps W_ih1 and the stack-top columns of W_ih0 in
// registers (108 per lane), Y keeps W_hh0 (96 per lane) and streams its slice of
// W_hh1 (96 KB per step, L2 hits) straight into MFMA operand registers -- at two
// waves per SIMD a wave has 256 registers, not the 300 all four blocks would take.
//
// Hand-offs between the blocks of a group (stack-control pre-activations, stack
// tops, h0_t, h1_t) follow the programming guide's placement-independent form
// (Guideline 16 R1, counter form): every handed-off byte is an sc1 store, every
// storing wave drains (vmcnt(0)), the role's waves meet, ONE lane adds to the
// group's counter (agent scope); a consumer role polls the counter with ONE lane
// (relaxed agent load = sc1), publishes the result to its other waves through an
// LDS word, and only then does any wave load the bytes, with sc1 loads only.
// Every step writes fresh addresses.  Counters are zeroed by a memset ahead of
// the launch.  Roles synchronise among themselves through LDS counters (no
// s_barrier: the two roles run at different phases); every spin is bounded.
#include <stdio.h>
#include <stdlib.h>

#include "stackgru.h"

namespace {

constexpr int PH = 512;          // hidden size (= Hp = Hn)
constexpr int PHC = PH / 64;     // 64-wide chunks of a hidden vector
constexpr int PROWS = 64;        // rows of a group
constexpr int PMEM = 32;         // blocks of a group
constexpr int PCOLS = 16;        // hidden units of a block
constexpr int RTHREADS = 256;    // threads of a role (4 waves)
constexpr int PTHREADS = 512;    // X + Y
constexpr int PWP = 64;          // padded stack width
constexpr int KP0 = 64 + PH;     // packed K of layer 0: stack top, h0
constexpr int KP1 = 2 * PH;      // packed K of layer 1: h0 (input), h1
constexpr int PARK_FLOATS = 4 * 3 * PROWS * PCOLS;   // [wave][gate][row][unit]
constexpr int HID_FLOATS = 3 * PROWS * PCOLS;        // [gate][row][unit]
constexpr int CW_FLOATS = 4 * PHC * 2 * 16;          // [wave][chunk][o][g][4]
constexpr int CRED_FLOATS = 4 * PROWS * 2;
constexpr int SYNC_WORDS = 32;
// Hand-off tiles are stored fragment-major, the order the MFMA operand loads
// take them in: [row tile][chunk][k slice (= consumer wave)][lane][4 floats], so
// one wave-instruction of a consumer reads 1 KiB of consecutive bytes (row-major
// rows made every such load touch sixteen 128-byte lines, a quarter of each)
constexpr unsigned FRAG_CHUNK_BYTES = 4 * 64 * 16;                 // 4 KiB
constexpr unsigned FRAG_ROWTILE_BYTES = PHC * FRAG_CHUNK_BYTES;    // 32 KiB
constexpr int FRAG_GROUP_FLOATS = PROWS * PH;                      // 128 KiB
constexpr int XFRAG_GROUP_FLOATS = PROWS * PWP;                    // 16 KiB
// LDS besides the block's two stack rows
constexpr int LDS_FIXED_FLOATS = 2 * PARK_FLOATS + 2 * HID_FLOATS + CW_FLOATS +
                                 CRED_FLOATS + 2 * 4 * PCOLS + 2 * 64 + SYNC_WORDS;
constexpr int LDS_MAX_FLOATS = 160 * 1024 / 4;
constexpr int STACK_MAX_QUADS = 9;   // f32x4 per thread a stack step may hold
constexpr int SYNC_STRIDE = 32;      // words between global counters (128 B lines)
constexpr unsigned SPIN_LIMIT = 1u << 22;       // polls of a global counter
constexpr unsigned LDS_SPIN_LIMIT = 1u << 26;   // polls of an LDS word

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __amdgpu_buffer_rsrc_t i32x4s;   // 128-bit buffer descriptor
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define SC1 16   // buffer-op aux bit: system-coherent (write-through / L1 bypass)
__device__ __forceinline__ i32x4s make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 load_sc1(i32x4s rsrc, unsigned voff,
                                          unsigned soff) {
    return __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, SC1));
}
// loads of the fragment-major hand-off tiles (hidden states, stack tops)
__device__ __forceinline__ f32x4 load_tile(i32x4s rsrc, unsigned voff,
                                           unsigned soff) {
    return __builtin_bit_cast(
        f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, SC1));
}

__device__ __forceinline__ void store_sc1(f32x4 v, i32x4s rsrc, unsigned voff,
                                          unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                           voff, soff, SC1);
}

// ---- role-level synchronisation -----------------------------------------------------
// LDS words: [0] X barrier, [1] Y barrier, [2] X go, [3] Y go, [4] hid0 ready,
// [5] hid1 ready, [6] abort, [7] X has finished its layer-1 product of the step
struct Role {
    unsigned *bar, *go, *abort;
    unsigned epoch, seq;
    int w, lane;
};

__device__ __forceinline__ unsigned lds_load(unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// spin until *p >= target; false when the role must give up (another wait of the
// workgroup failed, or this one ran out)
__device__ __forceinline__ bool lds_wait(unsigned *p, unsigned target,
                                         unsigned *abort) {
    unsigned spins = 0;
    while (lds_load(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 255u) == 0) {
            if (lds_load(abort) != 0) return false;
            if (spins > LDS_SPIN_LIMIT) {
                __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                return false;
            }
        }
    }
    return true;
}

// the four waves of a role meet (LDS writes before it are visible after it)
__device__ __forceinline__ bool role_barrier(Role &r) {
    r.epoch += 4;
    if (r.lane == 0)
        __hip_atomic_fetch_add(r.bar, 1u, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    return lds_wait(r.bar, r.epoch, r.abort);
}

// the role waits for a group counter: ONE lane polls (relaxed agent loads), the
// other waves learn the outcome from an LDS word that lane then sets
__device__ __forceinline__ bool role_wait(Role &r, gu32 *counter, unsigned target,
                                          gu32 *error) {
    r.seq += 1;
    if (r.w == 0 && r.lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT ||
                ((spins & 1023) == 0 &&
                 (__hip_atomic_load(error, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                  lds_load(r.abort) != 0))) {
                __hip_atomic_store(error, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(r.abort, 1u, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
        }
        __hip_atomic_store(r.go, r.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (!lds_wait(r.go, r.seq, r.abort)) return false;
    return lds_load(r.abort) == 0;
}

// the role has stored hand-off bytes (sc1): drain every storing wave, meet, ONE
// lane adds to the group's counter
__device__ __forceinline__ bool role_arrive(Role &r, gu32 *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!role_barrier(r)) return false;
    if (r.w == 0 && r.lane == 0)
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    return true;
}

struct Persist2Args {
    float *hs[2];          // [(T+1)][B][PH]
    float *gates[2];       // [T][B][4 PH] or null (inference)
    float *xdrop;          // [T][B][PH] dropped input of layer 1, or null
    float *S;              // training: [(T+1)][B][DW], else [2][B][DW]
    float *av;             // training: [T][B][AVW], else [B][AVW]
    float *xtop;           // [T][B][PWP]
    float *cpre;           // [T][B][64] stack control / push pre-activations
    float *hf[2];          // [(T+1)][groups][FRAG_GROUP_FLOATS] fragment-major h
    float *xdf;            // [T][groups][FRAG_GROUP_FLOATS] dropped h0, or null
    float *xtf;            // [T][groups][XFRAG_GROUP_FLOATS] fragment-major stack tops
    const float *wf;       // [32 blocks][8][4][3][64][4] W_hh1, consumption order
    int groups;
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
};

#ifdef PCM_PERSIST_STAMPS
#define STAMP(i)                                                        \
    do {                                                                \
        if (A.dbg && blockIdx.x == 0 && threadIdx.x == STAMP_THREAD) {  \
            const unsigned long long now_ = wall_clock64();             \
            A.dbg[STAMP_BASE + (i)] += now_ - last_stamp;               \
            last_stamp = now_;                                          \
        }                                                               \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// One K chunk (64 wide) for this wave: 4 row tiles x 3 gates x 4 k-steps.
// a[mt]: the wave's 16-wide k slice of rows 16 mt + (lane & 15), four consecutive
// k per lane; w[gate]: the matching weights of hidden unit (lane & 15).
__device__ __forceinline__ void chunk_mfma(const f32x4 (&a)[4],
                                           const f32x4 (&w)[3],
                                           f32x4 (&acc)[3][4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                acc[gate][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    a[mt][ks], w[gate][ks], acc[gate][mt], 0, 0, 0);
}

__device__ __forceinline__ void chunk_load(f32x4 (&a)[4], i32x4s rsrc,
                                           unsigned voff, unsigned soff,
                                           unsigned row_tile_bytes) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        a[mt] = load_tile(rsrc, voff + mt * row_tile_bytes, soff);
}

__device__ __forceinline__ void zero_acc(f32x4 (&acc)[3][4]) {
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[gate][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// NCH chunks of one operand ([64 rows][ld floats], chunk c at column 64 c) times
// register-resident weights w[C0 + c]; PF chunks of activation loads in flight
template <int NCH, int C0, int NW>
__device__ __forceinline__ void gemm_resident(const f32x4 (&w)[NW][3], i32x4s rsrc,
                                              unsigned voff,
                                              unsigned row_tile_bytes,
                                              f32x4 (&acc)[3][4]) {
    constexpr int PF = NCH < 2 ? NCH : 2;
    f32x4 a[PF][4];
#pragma unroll
    for (int p = 0; p < PF; ++p)
        chunk_load(a[p], rsrc, voff, p * FRAG_CHUNK_BYTES, row_tile_bytes);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 cur[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) cur[mt] = a[c % PF][mt];
        if (c + PF < NCH)
            chunk_load(a[c % PF], rsrc, voff, (c + PF) * FRAG_CHUNK_BYTES,
                       row_tile_bytes);
        chunk_mfma(cur, w[C0 + c], acc);
        __builtin_amdgcn_sched_barrier(0);   // keep the loads PF chunks ahead, no more
    }
}

// Half a group (two row tiles, rows 32 half .. 32 half + 31; voff points at the
// half's first row tile): fewer accumulator / fragment registers per pass, so
// that PF chunks of loads can be kept in flight behind a hand-off
__device__ __forceinline__ void chunk_mfma2(const f32x4 (&a)[2],
                                            const f32x4 (&w)[3],
                                            f32x4 (&acc)[3][2]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                acc[gate][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                    a[mt][ks], w[gate][ks], acc[gate][mt], 0, 0, 0);
}

template <int NCH, int C0, int NW, int PF>
__device__ __forceinline__ void gemm_resident_half(const f32x4 (&w)[NW][3],
                                                   i32x4s rsrc, unsigned voff,
                                                   unsigned row_tile_bytes,
                                                   f32x4 (&acc)[3][2]) {
    f32x4 a[PF][2];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            a[p][mt] = load_tile(rsrc, voff + mt * row_tile_bytes,
                                p * FRAG_CHUNK_BYTES);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 cur[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) cur[mt] = a[c % PF][mt];
        if (c + PF < NCH) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                a[c % PF][mt] =
                    load_tile(rsrc, voff + mt * row_tile_bytes,
                             (c + PF) * FRAG_CHUNK_BYTES);
        }
        chunk_mfma2(cur, w[C0 + c], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void park_half(float *buf, int w, int lane, int half,
                                          const f32x4 (&acc)[3][2]) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                buf[((w * 3 + gate) * PROWS + 32 * half + 16 * mt + 4 * g + e) *
                        PCOLS + r16] = acc[gate][mt][e];
}

// the same with the weights streamed from memory (plain loads of constants),
// packed per block and wave in the order they are consumed: wbase points at this
// lane's f32x4 of chunk 0, gate 0; [chunk][k slice][gate][lane][4]
constexpr int WFRAG_GATE_FLOATS = 64 * 4;
constexpr int WFRAG_CHUNK_FLOATS = 4 * 3 * WFRAG_GATE_FLOATS;
template <int NCH>
__device__ __forceinline__ void gemm_streamed(const float *wbase, i32x4s rsrc,
                                              unsigned voff,
                                              unsigned row_tile_bytes,
                                              f32x4 (&acc)[3][4]) {
    f32x4 a[2][4], w[2][3];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
            w[p][gate] = *(const f32x4 *)(wbase + gate * WFRAG_GATE_FLOATS +
                                          p * WFRAG_CHUNK_FLOATS);
        chunk_load(a[p], rsrc, voff, p * FRAG_CHUNK_BYTES, row_tile_bytes);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        f32x4 cur[4], wc[3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) cur[mt] = a[c & 1][mt];
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) wc[gate] = w[c & 1][gate];
        if (c + 2 < NCH) {
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
                w[c & 1][gate] =
                    *(const f32x4 *)(wbase + gate * WFRAG_GATE_FLOATS +
                                     (c + 2) * WFRAG_CHUNK_FLOATS);
            chunk_load(a[c & 1], rsrc, voff, (c + 2) * FRAG_CHUNK_BYTES,
                       row_tile_bytes);
        }
        chunk_mfma(cur, wc, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// partial tiles of this wave -> LDS: park[((wave * 3 + gate) * PROWS + row) * 16 + unit]
__device__ __forceinline__ void park(float *buf, int w, int lane,
                                     const f32x4 (&acc)[3][4]) {
    const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                buf[((w * 3 + gate) * PROWS + 16 * mt + 4 * g + e) * PCOLS + r16] =
                    acc[gate][mt][e];
}
// sum of the four waves' partials for (row, units 4 q .. 4 q + 3) of one gate
__device__ __forceinline__ f32x4 gather(const float *buf, int gate, int row,
                                        int q) {
    f32x4 s = *(const f32x4 *)(buf + ((0 * 3 + gate) * PROWS + row) * PCOLS + 4 * q);
#pragma unroll
    for (int w = 1; w < 4; ++w)
        s += *(const f32x4 *)(buf + ((w * 3 + gate) * PROWS + row) * PCOLS + 4 * q);
    return s;
}

// ---- gate math of one cell for this block's [64 rows x 16 units] (role X) ---------------
// thread (row = xt / 4, q = xt % 4) finishes units 4 q .. 4 q + 3 of its row:
// input-side sums from X's parked partials, hidden-side sums from Y's hid, the
// previous hidden state from this thread's registers (hp, updated in place)
struct CellOut {
    f32x4 r, z, n, ghn, h, xd;
};
__device__ __forceinline__ void cell_epilogue(
    const Persist2Args &A, const float *xpark, const float *hid, const float *bias,
    f32x4 &hp, int xt, int layer, int t, int row0, int nrows, int ci, f32x4 tr,
    f32x4 tz, f32x4 tn, i32x4s rs_hnew, i32x4s rs_xdrop, CellOut &o) {
    const int j0 = ci * PCOLS;
    const int row = xt >> 2, q = xt & 3;
    if (row >= nrows) return;
    const int b = row0 + row, j = j0 + 4 * q;
    const f32x4 sr = gather(xpark, 0, row, q) +
                     *(const f32x4 *)(hid + (0 * PROWS + row) * PCOLS + 4 * q);
    const f32x4 sz = gather(xpark, 1, row, q) +
                     *(const f32x4 *)(hid + (1 * PROWS + row) * PCOLS + 4 * q);
    const f32x4 sni = gather(xpark, 2, row, q);
    const f32x4 snh = *(const f32x4 *)(hid + (2 * PROWS + row) * PCOLS + 4 * q);
    const f32x4 b_r = *(const f32x4 *)(bias + 0 * PCOLS + 4 * q);
    const f32x4 b_z = *(const f32x4 *)(bias + 1 * PCOLS + 4 * q);
    const f32x4 b_ni = *(const f32x4 *)(bias + 2 * PCOLS + 4 * q);
    const f32x4 b_nh = *(const f32x4 *)(bias + 3 * PCOLS + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o.r[e] = pcm_sigmoid(sr[e] + b_r[e] + tr[e]);
        o.z[e] = pcm_sigmoid(sz[e] + b_z[e] + tz[e]);
        o.ghn[e] = snh[e] + b_nh[e];
        o.n[e] = tanhf(sni[e] + b_ni[e] + tn[e] + o.r[e] * o.ghn[e]);
        // ATen GRU cell: (h - n) * z + n
        o.h[e] = __fadd_rn(__fmul_rn(hp[e] - o.n[e], o.z[e]), o.n[e]);
    }
    hp = o.h;
    // the hand-off bytes: fragment-major, lane q * 16 + (row & 15) of (row tile,
    // chunk ci / 4, k slice ci % 4)
    const unsigned fo = (unsigned)((((((row >> 4) * PHC + (ci >> 2)) * 4 +
                                     (ci & 3)) * 64 + q * 16 + (row & 15)) * 16));
    store_sc1(o.h, rs_hnew, fo, 0);
    if (layer == 0 && A.xdrop) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o.xd[e] = o.h[e] * pcm_dropout_scale(A.seed, 1, t, b, j + e, A.drop_p,
                                                 A.inv_keep);
        store_sc1(o.xd, rs_xdrop, fo, 0);
    }
}
// ... and, behind the group's signal, what only later launches read: row-major
// copies for the reverse pass and the output head, the saved gates
__device__ __forceinline__ void cell_lazy_stores(const Persist2Args &A, int xt,
                                                 int layer, int t, int row0,
                                                 int nrows, int ci,
                                                 const CellOut &o) {
    const int row = xt >> 2, q = xt & 3;
    if (row >= nrows) return;
    const int b = row0 + row, j = ci * PCOLS + 4 * q;
    *(f32x4 *)(A.hs[layer] + ((size_t)(t + 1) * A.B + b) * PH + j) = o.h;
    if (layer == 0 && A.xdrop)
        *(f32x4 *)(A.xdrop + ((size_t)t * A.B + b) * PH + j) = o.xd;
    if (A.gates[layer]) {
        float *gp = A.gates[layer] + ((size_t)t * A.B + b) * 4 * PH + j;
        __builtin_nontemporal_store(o.r, (f32x4 *)(gp));
        __builtin_nontemporal_store(o.z, (f32x4 *)(gp + PH));
        __builtin_nontemporal_store(o.n, (f32x4 *)(gp + 2 * PH));
        __builtin_nontemporal_store(o.ghn, (f32x4 *)(gp + 3 * PH));
    }
}

// ---- role X: the critical chain --------------------------------------------------------
__device__ __forceinline__ void role_x(const Persist2Args &A, float *lds,
                                       unsigned *sw, int xt) {
    const int DWp = (A.DW + 3) & ~3;
    float *xpark = lds;
    float *hid = lds + 2 * PARK_FLOATS;              // [2 layers][3][PROWS][PCOLS]
    float *cwlds = hid + 2 * HID_FLOATS;
    float *cred = cwlds + CW_FLOATS;
    float *bias = cred + CRED_FLOATS;                // [2 layers][4][PCOLS]
    float *cu = bias + 2 * 4 * PCOLS;                // [2][64]
    float *Sbuf = cu + 2 * 64 + SYNC_WORDS;          // [2 rows][DWp]
    const int lane = xt & 63;
    Role R;
    R.bar = sw + 0; R.go = sw + 2; R.abort = sw + 6;
    R.epoch = 0; R.seq = 0;
    R.w = __builtin_amdgcn_readfirstlane(xt >> 6); R.lane = lane;
    unsigned *flag_h0 = sw + 4, *flag_h1 = sw + 5;
    const int wave = R.w;
    const int group = blockIdx.x & 7, ci = blockIdx.x >> 3;
    const int grow0 = group * PROWS;
    const int nrows = min(PROWS, A.B - grow0);
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
    __builtin_amdgcn_s_setprio(2);

    // ---- this role's weights -> registers: stack-top columns of W_ih0, W_ih1 -------
    f32x4 w0x[1][3], w1x[PHC][3];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate)
        w0x[0][gate] = *(const f32x4 *)(A.wpack[0] +
            (size_t)(gate * PH + j0 + r16) * KP0 + 16 * wave + 4 * g);
#pragma unroll
    for (int c = 0; c < PHC; ++c)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
            w1x[c][gate] = *(const f32x4 *)(A.wpack[1] +
                (size_t)(gate * PH + j0 + r16) * KP1 + 64 * c + 16 * wave + 4 * g);
    // the block's two stack control / push rows (outputs 2 ci, 2 ci + 1) -> LDS,
    // laid out as the lanes of each wave consume them: [wave][chunk][o][g][4]
    for (int idx = xt; idx < CW_FLOATS; idx += RTHREADS) {
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
    for (int idx = xt; idx < 2 * 4 * PCOLS; idx += RTHREADS) {
        const int l = idx / (4 * PCOLS), rem = idx - l * 4 * PCOLS;
        const int k = rem / PCOLS, u = rem - k * PCOLS;
        bias[idx] = A.bpack[l][k * PH + j0 + u];
    }
    for (int idx = xt; idx < 2 * DWp; idx += RTHREADS) {
        const int r = idx >= DWp ? 1 : 0, e = idx - r * DWp;
        Sbuf[idx] = (r < snrows && e < A.DW)
                        ? A.S[(size_t)(srow0 + r) * A.DW + e] : 0.f;
    }
    // previous hidden state of this thread's gate-math element, both layers
    const int erow = xt >> 2, eq = xt & 3;
    const int eb = grow0 + min(erow, nrows - 1);
    f32x4 hp0 = *(const f32x4 *)(A.hs[0] + (size_t)eb * PH + j0 + 4 * eq);
    f32x4 hp1 = *(const f32x4 *)(A.hs[1] + (size_t)eb * PH + j0 + 4 * eq);
    long long tk_next = A.tokens[eb];                 // token of step 0
    if (!role_barrier(R)) return;

    // fragment-major tiles: this wave's k slice, this lane
    const unsigned voff_h = (unsigned)((wave * 64 + lane) * 16);
    const unsigned voff_x = voff_h;
    const unsigned slab_h = FRAG_GROUP_FLOATS * 4, slab_x = XFRAG_GROUP_FLOATS * 4;
    const unsigned slab_c = (unsigned)A.B * 64 * 4;
    const float *cwl = cwlds + wave * (PHC * 32) + g * 4;
    const int W = A.W, D = A.D, DW = A.DW, AVW = A.AVW;
    const float inv_W = 1.0f / (float)W;
    const int nq = (DW + 3) >> 2;

#ifdef PCM_PERSIST_STAMPS
    constexpr int STAMP_THREAD = 0, STAMP_BASE = 0;
    unsigned long long last_stamp = wall_clock64();
#endif
    for (int t = 0; t < A.T; ++t) {
        const size_t gslot = (size_t)t * A.groups + group;
        const i32x4s rs_h1p = make_rsrc(A.hf[1] + gslot * FRAG_GROUP_FLOATS, slab_h);
        const i32x4s rs_h0n =
            make_rsrc(A.hf[0] + (gslot + A.groups) * FRAG_GROUP_FLOATS, slab_h);
        const i32x4s rs_h1n =
            make_rsrc(A.hf[1] + (gslot + A.groups) * FRAG_GROUP_FLOATS, slab_h);
        const i32x4s rs_xt = make_rsrc(A.xtf + gslot * XFRAG_GROUP_FLOATS, slab_x);
        const i32x4s rs_cp = make_rsrc(A.cpre + (size_t)t * A.B * 64, slab_c);
        const i32x4s rs_xd = make_rsrc(
            A.xdf ? A.xdf + gslot * FRAG_GROUP_FLOATS : A.hf[0], slab_h);
        const unsigned target = (unsigned)(t + 1) * PMEM;
        // index math of the stack step and of the gate math is cheap; derived from
        // a value the compiler must take as new in every step, it is recomputed
        // per step instead of being hoisted out of the loop and kept in (spilled)
        // registers for the whole sequence
        int xs = xt;
        asm volatile("" : "+v"(xs));

        // ---- X1 / X2: stack control products of the block's two outputs --------------
        if (t > 0 && !role_wait(R, cnt_1, (unsigned)t * PMEM, error)) return;
        STAMP(0);
        {
            float cdot[4][2] = {};
            f32x4 a[2][4];
            // (read per step: hoisted out of the loop, the sixteen weight
            // fragments would sit in registers for the whole sequence)
            const float *cwl_t = cwl;
            asm volatile("" : "+v"(cwl_t));
            chunk_load(a[0], rs_h1p, voff_h, 0, FRAG_ROWTILE_BYTES);
            chunk_load(a[1], rs_h1p, voff_h, FRAG_CHUNK_BYTES, FRAG_ROWTILE_BYTES);
            // a rolled loop over chunk pairs: fully unrolled, hipcc issues all
            // 32 fragment loads up front (128 registers)
#pragma unroll 1
            for (int c = 0; c < PHC; c += 2) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 cur[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) cur[mt] = a[h][mt];
                    if (c + h + 2 < PHC)
                        chunk_load(a[h], rs_h1p, voff_h,
                                   (c + h + 2) * FRAG_CHUNK_BYTES,
                                   FRAG_ROWTILE_BYTES);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const f32x4 cwv =
                            *(const f32x4 *)(cwl_t + ((c + h) * 2 + o) * 16);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                cdot[mt][o] = fmaf(cur[mt][e], cwv[e], cdot[mt][o]);
                    }
                }
            }
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
        if (!role_barrier(R)) return;
        if (xt < PROWS) {
            float c0 = 0.f, c1 = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                c0 += cred[(w * PROWS + xt) * 2 + 0];
                c1 += cred[(w * PROWS + xt) * 2 + 1];
            }
            u32x2 pair = {__float_as_uint(c0), __float_as_uint(c1)};
            if (xt < nrows)
                __builtin_amdgcn_raw_buffer_store_b64(
                    pair, rs_cp, (unsigned)(((grow0 + xt) * 64 + 2 * ci) * 4), 0, SC1);
        }
        if (!role_arrive(R, cnt_c)) return;
        STAMP(1);

        // ---- X3: the stack step of this block's two rows ------------------------------
        if (!role_wait(R, cnt_c, target, error)) return;
        STAMP(2);
        if (snrows > 0) {
            if (xs < 32) {
                const int r = xs >> 4, q4 = xs & 15;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (r < snrows)
                    v = load_sc1(rs_cp, (unsigned)(((srow0 + r) * 64 + 4 * q4) * 4), 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int o = 4 * q4 + e;
                    const float bb = o < 3 ? A.ctrl_b[o]
                                           : (o < NO ? A.push_b[o - 3] : 0.f);
                    cu[64 * r + o] = o < 3 ? v[e] + bb : tanhf(v[e] + bb);
                }
            }
        }
        if (!role_barrier(R)) return;
        if (snrows > 0 && xs < 2) {
            float *c = cu + 64 * xs;
            const float m = fmaxf(c[0], fmaxf(c[1], c[2]));
            const float e0 = expf(c[0] - m), e1 = expf(c[1] - m), e2 = expf(c[2] - m);
            const float s = e0 + e1 + e2;
            c[0] = e0 / s; c[1] = e1 / s; c[2] = e2 / s;
        }
        if (!role_barrier(R)) return;
        {
            // S' = a_noop S + a_push up + a_pop down in the reference's add order
            // (stack_rnn.py:188-232), in place: new values are held in registers
            // until every thread has read its neighbours
            f32x4 nv[STACK_MAX_QUADS];
#pragma unroll
            for (int k = 0; k < STACK_MAX_QUADS; ++k) {
                const int idx = xs + k * RTHREADS;
                nv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (idx < snrows * nq) {
                    const int r = idx >= nq ? 1 : 0, q = idx - r * nq;
                    const float *Sr = Sbuf + r * DWp;
                    const float *c = cu + 64 * r;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int e = 4 * q + i;
                        float x = 0.f;
                        if (e < DW) {
                            const int d = pcm_fdiv(e, inv_W), w = e - d * W;
                            const float sv = Sr[e];
                            const float up = d == 0 ? c[3 + w] : Sr[e - W];
                            const float dn = d == D - 1 ? 0.f : Sr[e + W];
                            x = __fadd_rn(__fadd_rn(__fmul_rn(c[2], sv),
                                                    __fmul_rn(c[0], up)),
                                          __fmul_rn(c[1], dn));
                        }
                        nv[k][i] = x;
                    }
                }
            }
            if (!role_barrier(R)) return;
            const bool to_global = A.training || t == A.T - 1;
            float *Sn = A.S + (size_t)(A.training ? t + 1 : ((t + 1) & 1)) * A.B * DW;
            const bool vec = (DW & 3) == 0;
#pragma unroll
            for (int k = 0; k < STACK_MAX_QUADS; ++k) {
                const int idx = xs + k * RTHREADS;
                if (idx < snrows * nq) {
                    const int r = idx >= nq ? 1 : 0, q = idx - r * nq;
                    *(f32x4 *)(Sbuf + r * DWp + 4 * q) = nv[k];
                    // stack top (row 0 of the new stack), zero padded to PWP:
                    // the hand-off bytes, stored first (fragment-major: group row
                    // 2 ci + r, k slice q / 4, lane (q % 4) * 16 + row % 16)
                    if (4 * q < PWP) {
                        f32x4 top;
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            top[i] = 4 * q + i < W ? nv[k][i] : 0.f;
                        const int grow = 2 * ci + r;
                        store_sc1(top, rs_xt,
                                  (unsigned)(((((grow >> 4) * 4 + (q >> 2)) * 64 +
                                               (q & 3) * 16 + (grow & 15)) * 16)),
                                  0);
                    }
                }
            }
            // the group is told now; what only later launches read (row-major
            // stack tops, controls, the new stack) is stored behind the signal
            if (!role_arrive(R, cnt_x)) return;
            if (A.training) {
#pragma unroll
                for (int k = 0; k < STACK_MAX_QUADS; ++k) {
                    const int idx = xs + k * RTHREADS;
                    if (idx < snrows * nq) {
                        const int r = idx >= nq ? 1 : 0, q = idx - r * nq;
                        if (4 * q < PWP) {
                            f32x4 top;
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                top[i] = 4 * q + i < W ? nv[k][i] : 0.f;
                            *(f32x4 *)(A.xtop + ((size_t)t * A.B + srow0 + r) * PWP +
                                       4 * q) = top;
                        }
                    }
                }
            }
            if (snrows > 0 && A.training) {   // a, v for the reverse pass
                float *av = A.av + (size_t)t * A.B * AVW;
                for (int idx = xs; idx < snrows * AVW; idx += RTHREADS) {
                    const int r = idx / AVW, o = idx - r * AVW;
                    av[(size_t)(srow0 + r) * AVW + o] = o < NO ? cu[64 * r + o] : 0.f;
                }
            }
            if (to_global) {
#pragma unroll
                for (int k = 0; k < STACK_MAX_QUADS; ++k) {
                    const int idx = xs + k * RTHREADS;
                    if (idx < snrows * nq) {
                        const int r = idx >= nq ? 1 : 0, q = idx - r * nq;
                        float *dst = Sn + (size_t)(srow0 + r) * DW;
                        if (vec) {
                            __builtin_nontemporal_store(nv[k], (f32x4 *)(dst + 4 * q));
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (4 * q + i < DW) dst[4 * q + i] = nv[k][i];
                        }
                    }
                }
            }
        }
        STAMP(3);

        // ---- X4: layer 0, input side (the stack tops) and gate math -------------------
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
        f32x4 acc[3][4];
        zero_acc(acc);
        if (!role_wait(R, cnt_x, target, error)) return;
        STAMP(4);
        gemm_resident<1, 0, 1>(w0x, rs_xt, voff_x, FRAG_CHUNK_BYTES, acc);
        park(xpark, wave, lane, acc);
        if (!role_barrier(R)) return;
        if (!lds_wait(flag_h0, 4u * (unsigned)(t + 1), R.abort)) return;
        STAMP(5);
        {
            CellOut co;
            cell_epilogue(A, xpark, hid, bias, hp0, xs, 0, t, grow0, nrows, ci, tr,
                          tz, tn, rs_h0n, rs_xd, co);
            if (!role_arrive(R, cnt_0)) return;
            cell_lazy_stores(A, xs, 0, t, grow0, nrows, ci, co);
        }
        STAMP(6);

        // ---- X5: layer 1, input side (h0_t, dropped in training) and gate math --------
        if (!role_wait(R, cnt_0, target, error)) return;
        STAMP(7);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 acc2[3][2];
#pragma unroll
            for (int gate = 0; gate < 3; ++gate)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc2[gate][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            gemm_resident_half<PHC, 0, PHC, 4>(
                w1x, A.xdrop ? rs_xd : rs_h0n,
                voff_h + half * 2u * FRAG_ROWTILE_BYTES, FRAG_ROWTILE_BYTES, acc2);
            park_half(xpark, wave, lane, half, acc2);
        }
        // Y's h0_t . W_hh0 starts now, not beside this product: the two would
        // share the SIMDs' matrix pipes and this one is on the critical chain
        if (lane == 0)
            __hip_atomic_fetch_add(sw + 7, 1u, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_WORKGROUP);
        STAMP(8);
        if (!role_barrier(R)) return;
        if (!lds_wait(flag_h1, 4u * (unsigned)(t + 1), R.abort)) return;
        STAMP(9);
        {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            CellOut co;
            cell_epilogue(A, xpark, hid + HID_FLOATS, bias + 4 * PCOLS, hp1, xs, 1, t,
                          grow0, nrows, ci, zero, zero, zero, rs_h1n, rs_xd, co);
            if (!role_arrive(R, cnt_1)) return;
            cell_lazy_stores(A, xs, 1, t, grow0, nrows, ci, co);
        }
        STAMP(10);
    }
}

// ---- role Y: the hidden-side products -----------------------------------------------------
__device__ __forceinline__ bool y_finish(Role &R, float *ypark, float *hid_out,
                                         unsigned *flag, int yt, int lane,
                                         const f32x4 (&acc)[3][4]) {
    park(ypark, R.w, lane, acc);
    if (!role_barrier(R)) return false;
    for (int idx = yt; idx < 3 * PROWS * 4; idx += RTHREADS) {
        const int gate = idx / (PROWS * 4), rem = idx - gate * PROWS * 4;
        const int row = rem >> 2, q = rem & 3;
        *(f32x4 *)(hid_out + (gate * PROWS + row) * PCOLS + 4 * q) =
            gather(ypark, gate, row, q);
    }
    // the sums are in LDS: tell X (one add per wave, after the wave's writes)
    if (lane == 0)
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    // ... and the partial tiles may be overwritten once every wave has summed
    return role_barrier(R);
}

__device__ __forceinline__ void role_y(const Persist2Args &A, float *lds,
                                       unsigned *sw, int yt) {
    float *ypark = lds + PARK_FLOATS;
    float *hid = lds + 2 * PARK_FLOATS;
    const int lane = yt & 63;
    Role R;
    R.bar = sw + 1; R.go = sw + 3; R.abort = sw + 6;
    R.epoch = 0; R.seq = 0;
    R.w = __builtin_amdgcn_readfirstlane(yt >> 6); R.lane = lane;
    unsigned *flag_h0 = sw + 4, *flag_h1 = sw + 5;
    const int wave = R.w;
    const int group = blockIdx.x & 7, ci = blockIdx.x >> 3;
    const int j0 = ci * PCOLS;
    const int r16 = lane & 15, g = lane >> 4;
    gu32 *cnt_1 = (gu32 *)(A.sync + (group * 4 + 3) * SYNC_STRIDE);
    gu32 *error = (gu32 *)(A.sync + 8 * 4 * SYNC_STRIDE);

    // W_hh0 of the block's 48 gate columns -> registers
    f32x4 w0h[PHC][3];
#pragma unroll
    for (int c = 0; c < PHC; ++c)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
            w0h[c][gate] = *(const f32x4 *)(A.wpack[0] +
                (size_t)(gate * PH + j0 + r16) * KP0 + 64 + 64 * c + 16 * wave +
                4 * g);
    // W_hh1: streamed every step from its fragment-packed copy
    const float *w1h = A.wf + (size_t)ci * PHC * WFRAG_CHUNK_FLOATS +
                       wave * 3 * WFRAG_GATE_FLOATS + lane * 4;

    const unsigned voff_h = (unsigned)((wave * 64 + lane) * 16);
    const unsigned slab_h = FRAG_GROUP_FLOATS * 4;
#ifdef PCM_PERSIST_STAMPS
    constexpr int STAMP_THREAD = RTHREADS, STAMP_BASE = 16;
    unsigned long long last_stamp = wall_clock64();
#endif
    f32x4 acc[3][4];
    {   // hidden-side sums of layer 0 for step 0 (initial hidden state)
        const i32x4s rs =
            make_rsrc(A.hf[0] + (size_t)group * FRAG_GROUP_FLOATS, slab_h);
        zero_acc(acc);
        gemm_resident<PHC, 0, PHC>(w0h, rs, voff_h, FRAG_ROWTILE_BYTES, acc);
        if (!y_finish(R, ypark, hid, flag_h0, yt, lane, acc)) return;
    }
    for (int t = 0; t < A.T; ++t) {
        // ---- Y1: h1_{t-1} . W_hh1 -> hid1 of step t -------------------------------------
        if (t > 0 && !role_wait(R, cnt_1, (unsigned)t * PMEM, error)) return;
        STAMP(0);
        {
            const i32x4s rs = make_rsrc(
                A.hf[1] + ((size_t)t * A.groups + group) * FRAG_GROUP_FLOATS, slab_h);
            zero_acc(acc);
            gemm_streamed<PHC>(w1h, rs, voff_h, FRAG_ROWTILE_BYTES, acc);
            STAMP(1);
            if (!y_finish(R, ypark, hid + HID_FLOATS, flag_h1, yt, lane, acc)) return;
        }
        STAMP(2);
        // ---- Y2: h0_t . W_hh0 -> hid0 of step t + 1 -------------------------------------
        if (t + 1 < A.T) {
            // (h0_t has been published: X polled the group's counter before
            // its own product, and sets this word after it)
            if (!lds_wait(sw + 7, 4u * (unsigned)(t + 1), R.abort)) return;
            STAMP(3);
            const i32x4s rs = make_rsrc(
                A.hf[0] + ((size_t)(t + 1) * A.groups + group) * FRAG_GROUP_FLOATS,
                slab_h);
            zero_acc(acc);
            gemm_resident<PHC, 0, PHC>(w0h, rs, voff_h, FRAG_ROWTILE_BYTES, acc);
            STAMP(4);
            if (!y_finish(R, ypark, hid, flag_h0, yt, lane, acc)) return;
            STAMP(5);
        }
    }
}

__global__ __launch_bounds__(PTHREADS, 2) void stackgru_persist2_fwd_kernel(
    Persist2Args A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // sync words sit behind cu (see role_x's carve-up)
    unsigned *sw = (unsigned *)(lds + 2 * PARK_FLOATS + 2 * HID_FLOATS + CW_FLOATS +
                                CRED_FLOATS + 2 * 4 * PCOLS + 2 * 64);
    const int tid = threadIdx.x;
    const int group = blockIdx.x & 7, ci = blockIdx.x >> 3;
    if (A.B - group * PROWS <= 0 || ci >= PMEM) return;   // a group without rows
    if (tid < SYNC_WORDS) sw[tid] = 0;
    __syncthreads();   // the only workgroup-wide barrier: before the roles part
    if (tid < RTHREADS) role_x(A, lds, sw, tid);
    else role_y(A, lds, sw, tid - RTHREADS);
}

// Ahead of the launch: the initial hidden states in fragment-major order (slot 0
// of hf) and W_hh1 packed per block / chunk / wave / gate / lane.
__global__ __launch_bounds__(256) void persist2_prepare_kernel(Persist2Args A) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int nh = 2 * A.B * (PH / 4);             // (layer, row, quad of units)
    if (tid < nh) {
        const int l = tid / (A.B * (PH / 4)), rem = tid - l * A.B * (PH / 4);
        const int b = rem / (PH / 4), jq = rem - b * (PH / 4), j = 4 * jq;
        const int group = b / PROWS, row = b - group * PROWS;
        const f32x4 v = *(const f32x4 *)(A.hs[l] + (size_t)b * PH + j);
        const int lane = ((j >> 2) & 3) * 16 + (row & 15);
        *(f32x4 *)(A.hf[l] + (size_t)group * FRAG_GROUP_FLOATS +
                   ((((row >> 4) * PHC + (j >> 6)) * 4 + ((j >> 4) & 3)) * 64 + lane) *
                       4) = v;
        return;
    }
    const int w = tid - nh;                        // (block, chunk, slice, gate, lane)
    if (w < PMEM * PHC * 4 * 3 * 64) {
        const int lane = w & 63, gate = (w >> 6) % 3, rest = (w >> 6) / 3;
        const int ks = rest & 3, c = (rest >> 2) % PHC, ci = (rest >> 2) / PHC;
        const int r16 = lane & 15, g = lane >> 4;
        const f32x4 v = *(const f32x4 *)(A.wpack[1] +
            (size_t)(gate * PH + ci * PCOLS + r16) * KP1 + PH + 64 * c + 16 * ks +
            4 * g);
        *(f32x4 *)((float *)A.wf + (size_t)w * 4) = v;
    }
}

// a hand-off that timed out leaves its mark in the caller's status word
__global__ void persist2_report_kernel(const unsigned *error, int32_t *status) {
    if (*error != 0) atomicOr(status, PCM_STATUS_TIMEOUT);
}

}  // namespace

// ---- host side ----------------------------------------------------------------------------
// floats of the fragment-major regions for `steps` steps of a batch of B rows
void pcm_persist2_fwd_regions(int B, int steps, size_t *hfrag, size_t *xdfrag,
                              size_t *xtfrag, size_t *wfrag) {
    const size_t groups = (size_t)(B + PROWS - 1) / PROWS;
    *hfrag = (size_t)(steps + 1) * groups * FRAG_GROUP_FLOATS;
    *xdfrag = (size_t)steps * groups * FRAG_GROUP_FLOATS;
    *xtfrag = (size_t)steps * groups * XFRAG_GROUP_FLOATS;
    *wfrag = (size_t)PMEM * PHC * WFRAG_CHUNK_FLOATS;
}

bool pcm_persist2_fwd_shape_ok(const SGLayout &lay) {
    const int DWp = (lay.DW + 3) & ~3;
    return lay.H == PH && lay.L == 2 && lay.use_stack && lay.W <= PWP &&
           lay.W >= 1 && 3 + lay.W <= 56 && lay.AVW <= 64 &&
           lay.B <= 8 * PROWS && lay.DW >= PWP &&
           LDS_FIXED_FLOATS + 2 * DWp <= LDS_MAX_FLOATS &&
           2 * ((lay.DW + 3) >> 2) <= STACK_MAX_QUADS * RTHREADS;
}

int pcm_launch_persist2_fwd(const SGLayout &lay, float *sv,
                            const pcm_stackgru_params *w, const int64_t *tokens,
                            float dropout_p, uint64_t seed, int32_t *status,
                            unsigned *sync, hipStream_t stream) {
    Persist2Args a;
    for (int l = 0; l < 2; ++l) {
        a.hs[l] = sv + lay.off_hs[l];
        a.gates[l] = lay.training ? sv + lay.off_gates[l] : nullptr;
        a.wpack[l] = sv + lay.off_wpack[l];
        a.bpack[l] = sv + lay.off_bpack[l];
    }
    a.xdrop = lay.dropout ? sv + lay.off_xdrop[1] : nullptr;
    a.S = sv + lay.off_S; a.av = sv + lay.off_av; a.xtop = sv + lay.off_xtop;
    a.cpre = sv + lay.off_cpre;
    a.groups = (lay.B + PROWS - 1) / PROWS;
    a.hf[0] = sv + lay.off_hfrag[0]; a.hf[1] = sv + lay.off_hfrag[1];
    a.xdf = lay.dropout ? sv + lay.off_xdfrag : nullptr;
    a.xtf = sv + lay.off_xtfrag;
    a.wf = sv + lay.off_wfrag;
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
        (size_t)(LDS_FIXED_FLOATS + 2 * ((lay.DW + 3) & ~3)) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(
            (const void *)stackgru_persist2_fwd_kernel,
            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_FLOATS * 4);
        PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "persist2_fwd: LDS attribute: %s",
                    hipGetErrorString(e));
        attr_set = true;
    }
    hipError_t e = hipMemsetAsync(sync, 0, pcm_persist_fwd_sync_words() * 4, stream);
    PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "persist2_fwd: memset: %s",
                hipGetErrorString(e));
    a.dbg = nullptr;
#ifdef PCM_PERSIST_STAMPS
    static unsigned long long *dbg = nullptr;
    static int calls = 0;
    if (!dbg) (void)hipMalloc(&dbg, 32 * sizeof(unsigned long long));
    (void)hipMemsetAsync(dbg, 0, 32 * sizeof(unsigned long long), stream);
    a.dbg = dbg;
#endif
    {
        const int items = 2 * lay.B * (PH / 4) + PMEM * PHC * 4 * 3 * 64;
        hipLaunchKernelGGL(persist2_prepare_kernel, dim3((items + 255) / 256),
                           dim3(256), 0, stream, a);
    }
    hipLaunchKernelGGL(stackgru_persist2_fwd_kernel, dim3(8 * PMEM), dim3(PTHREADS),
                       lds, stream, a);
#ifdef PCM_PERSIST_STAMPS
    if (++calls == 6) {   // a warm call: print where block 0's time went
        unsigned long long h[32];
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        static const char *xname[11] = {
            "X wait h1", "X ctrl sweep + publish + arrive c", "X wait c",
            "X stack step + arrive x", "X wait x (+ token gates)",
            "X l0 x product + park + wait hid0", "X l0 gate math + arrive 0",
            "X wait h0", "X l1 x product", "X park + wait hid1",
            "X l1 gate math + arrive 1"};
        static const char *yname[6] = {
            "Y wait h1", "Y h1 . W_hh1 (streamed)", "Y park + reduce + signal",
            "Y wait h0", "Y h0 . W_hh0 (resident)", "Y park + reduce + signal"};
        double tx = 0, ty = 0;
        for (int i = 0; i < 11; ++i) tx += (double)h[i];
        for (int i = 0; i < 6; ++i) ty += (double)h[16 + i];
        for (int i = 0; i < 11; ++i)
            fprintf(stderr, "persist2 phase %-36s %8.2f us/step\n", xname[i],
                    (double)h[i] * 0.01 / lay.T);
        fprintf(stderr, "persist2 X total %8.2f us/step\n", tx * 0.01 / lay.T);
        for (int i = 0; i < 6; ++i)
            fprintf(stderr, "persist2 phase %-36s %8.2f us/step\n", yname[i],
                    (double)h[16 + i] * 0.01 / lay.T);
        fprintf(stderr, "persist2 Y total %8.2f us/step\n", ty * 0.01 / lay.T);
    }
#endif
    if (status)
        hipLaunchKernelGGL(persist2_report_kernel, dim3(1), dim3(1), 0, stream,
                           sync + 8 * 4 * SYNC_STRIDE, status);
    return 0;
}
