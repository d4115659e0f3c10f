// Workspace layout and kernel argument blocks of the StackGRU sequence ops.
#pragma once
#include "common.h"

// Layout of the caller-provided `saved` workspace (offsets in floats).
//
// hidden vectors are stored with row stride Hp = round_up(H, 64) and zero
// padding so the fused cell kernel streams K in whole 64-wide chunks without
// bounds checks; the stack top fed to layer 0 likewise with stride Wp.
struct SGLayout {
    int B, V, E, H, L, D, W, use_stack;
    int T, training, dropout;
    int xtop_per_step;
    int Hp, Hn, Wp, AVW, DW;
    int Kx[PCM_MAX_LAYERS];   // width of the non-token GRU input of layer l
    int Kxp[PCM_MAX_LAYERS];  // ... padded to 64
    int Kp[PCM_MAX_LAYERS];   // packed K = Kxp + Hp
    size_t off_hs[PCM_MAX_LAYERS];     // (T+1) x B x Hp ; [0] = initial hidden
    size_t off_xdrop[PCM_MAX_LAYERS];  // T x B x Hp, dropped input of layer l>=1
    size_t off_gates[PCM_MAX_LAYERS];  // T x B x 4H : r, z, n, W_hn.h+b_hn
    size_t off_S;      // training: (T+1) x B x D*W ; else 2 x B x D*W
    size_t off_av;     // training: T x B x AVW ; else B x AVW  (a[3], v[W])
    size_t off_xtop;   // T x B x Wp when xtop_per_step (training, or the
                       // persistent kernel, which hands it over at fresh
                       // addresses), else one B x Wp slot
    size_t off_sync;   // hand-off counters of the persistent kernel (words)
    size_t off_hraw;   // B x 3 x Hn: hoisted hidden-side sums of layer 1 (CellFwdArgs::hz_raw)
    size_t off_cspart; // B x (Hn / 32) x 64: partial control / push sums that
                       // the top layer's forward cell leaves for the next stack step
    size_t off_cpre;   // T x B x 64: stack control / push pre-activations the
                       // persistent kernel hands over (0 floats when that
                       // kernel cannot take this shape)
    // role-split persistent kernel (persist2_fwd.hip) only: hand-off tiles in
    // fragment-major order (hidden states of both layers, dropped layer-1 input,
    // stack tops) and W_hh1 packed in consumption order; 0 floats otherwise
    size_t off_hfrag[2], off_xdfrag, off_xtfrag, off_wfrag;
    size_t off_tokg;   // V x 3*Hn   token gate table (b_ih0 folded in)
    size_t off_wpack[PCM_MAX_LAYERS];  // 3*Hn x Kp[l]
    size_t off_bpack[PCM_MAX_LAYERS];  // 4*Hn
    size_t total;      // floats

    size_t hs_step() const { return (size_t)B * Hp; }
    size_t s_index(int t) const {  // slot holding the stack BEFORE step t
        return off_S + (size_t)(training ? t : (t & 1)) * B * DW;
    }
    size_t av_index(int t) const {
        return off_av + (size_t)(training ? t : 0) * B * AVW;
    }
    size_t xtop_index(int t) const {
        return off_xtop + (size_t)(xtop_per_step ? t : 0) * B * Wp;
    }
};

int pcm_sg_layout(const pcm_dims *dims, int steps, int training, int dropout,
                  SGLayout *out);

// Layout of the backward `scratch` workspace (offsets in floats).
struct SGScratch {
    size_t off_dh;      // L x B x Hp running dL/dh
    size_t off_dS;      // 2 x B x D*W ping-pong dL/dstack
    size_t off_dtop;    // B x Wp
    size_t off_onehot;  // T*B x V  one-hot of the input tokens
    size_t off_dtokg;   // V x 3H   gradient of the token gate table
    size_t off_tmp;     // B x Hp   scratch product (generic reverse path)
    size_t off_psum;    // B x 4    control sums of the bulk stack adjoint (StackBulkArgs)
    size_t off_dwx;     // 3H x Wp  gradient of W_ih0's stack-top columns (padded)
    size_t off_dcw;     // AVW x H  gradient of [ctrl_w ; push_w] (padded rows)
    size_t off_wpackT[PCM_MAX_LAYERS];  // (H + Kxp) x 3H transposed weights
    size_t off_gemm;    // split-K slabs / column-sum partials (main stream)
    size_t off_gemm2;   // same for the weight-gradient side stream
    size_t gemm_bytes;
    size_t total;       // floats
};
int pcm_sg_scratch(const SGLayout &lay, SGScratch *out);

// ---- kernel launchers (stackgru_fwd.hip / stackgru_bwd.hip) ------------------
struct CellFwdArgs {
    const float *xin;   // [B, ldx] non-token input (stack top / lower layer)
    const float *hprev; // [B, Hp]
    const float *wpack; // [3*Hn, Kp]
    const float *bpack; // [4*Hn]
    const float *tokg;  // [V, 3*Hn] or null (layers >= 1)
    const int64_t *tok; // [B] or null
    float *hnew;        // [B, Hp]
    float *gates;       // [B, 4H] or null
    float *xdrop;       // [B, Hp] or null: dropout(hnew) for the next layer
    const int32_t *lengths;
    int32_t *status;
    int ldx, nx_chunks, nh_chunks, Kp;
    int B, H, Hn, Hp, V, t, next_layer;
    int b_off;          // batch index of row 0 (a launch over a slice of the
                        // batch: the dropout hash is a function of the GLOBAL row)
    float drop_p, inv_keep;
    uint64_t seed;
    // top layer with a stack: the block's share of the NEXT stack step's 3 + W
    // control / push pre-activations, [row][column tile][64]
    // (null: not wanted); cs_W = stack width
    const float *cs_ctrl_w, *cs_push_w;
    float *cs_part;
    int cs_W;
    // Layer 0 with the stack step fused in (xt_part != null; the partials are
    // the previous step's cs_part): every block derives the controls, the push
    // value and the NEW stack top of its 32 rows itself and feeds it to the
    // input-side chunk; column tile 0 records controls / push value (xt_av) and
    // the stack top (xt_xtop) for the reverse pass and for the update below.
    const float *xt_part, *xt_ctrl_b, *xt_push_b;
    const float *xt_S;     // [B][DW] stack before the step
    float *xt_av, *xt_xtop;
    int xt_parts, xt_W, xt_D, xt_AVW;
    // Top layer carrying the deferred stack update of the same step as extra
    // blocks (blockIdx.y >= su_nbt, su_rows rows each; su_S == null: none):
    // S_new = update(S, controls, push value) -- off the critical path, the
    // next step's layer 0 only needs it after this launch has ended.
    const float *su_S, *su_av;
    float *su_Snew;
    int su_nbt, su_rows, su_D, su_W, su_AVW;
    // Hidden-side product of the layer ABOVE, hoisted into this (layer 0, fused)
    // launch as extra blocks (blockIdx.y >= hz_nbt; hz_raw == null: none): it
    // only needs that layer's previous hidden state, and this launch's own blocks
    // spend their first microseconds in the stack prologue with the matrix pipes
    // idle.  hz_raw[b][3][Hn]: the raw r, z, n(hidden) sums of h_prev . W_hh^T.
    const float *hz_hprev;   // [B, Hp] previous hidden state of the layer above
    const float *hz_wpack;   // its packed weights [3*Hn, hz_Kp]
    float *hz_raw;
    int hz_nbt, hz_Kp, hz_kx;   // hz_kx: 64-wide chunks of its input side (skipped)
    // ... and the launch of the layer above then runs its input side only
    // (nh_chunks = 0) and adds these sums in the epilogue
    const float *hraw;       // [B][3][Hn] or null
};
int pcm_launch_cell_fwd(const CellFwdArgs &a, hipStream_t stream);
// tile the forward cell launcher picks for `a` (16, 32 or 64); the fused stack
// step (xt_*, su_*) lives in the 32 x 32 kernel only
int pcm_cell_fwd_tile(const CellFwdArgs &a);
// true when pcm_launch_cell_fwd will run `a` on the kernel that writes cs_part
// number of control partials per row this launch would write (0: none)
int pcm_cell_fwd_ctrl_partials(const CellFwdArgs &a);

struct StackFwdArgs {
    const float *htop;   // [B, Hp] top-layer hidden before the step
    const float *S_prev; // [B, D*W]
    float *S_new;        // [B, D*W]
    const float *ctrl_w, *ctrl_b, *push_w, *push_b;
    float *av;           // [B, AVW]
    float *xtop;         // [B, Wp]
    const int32_t *lengths;
    int B, H, Hp, D, W, Wp, AVW, t;
    // partial control / push sums left by the top layer's forward cell of the
    // step before ([row][cs_parts][64], summed in order); null: the dot products
    // are computed here
    const float *cs_part;
    int cs_parts;
    // top_only (large-batch kernel): controls, push value and the new stack TOP
    // only -- the update of the whole stack rides in the top layer's forward-cell
    // launch of the same step (CellFwdArgs::su_*), off the critical chain
    int top_only;
};
int pcm_launch_stack_fwd(const StackFwdArgs &a, hipStream_t stream);
// true if that launch goes to the large-batch kernel, the one that takes top_only
bool pcm_stack_fwd_is_mfma(const StackFwdArgs &a);
// The stack kernels stage whole [D, W] stacks in LDS: PCM_OK if one batch row
// fits the dynamic LDS of a block, else PCM_EINVAL with the limit in the
// message.  Checked once per call, before anything is launched.
#define PCM_STACK_LDS_BYTES (64 * 1024)
int pcm_stack_fwd_check(int D, int W, int H, int AVW);
int pcm_stack_bwd_check(int D, int W, int AVW);

struct PackArgs {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    float *wpack, *bpack;
    int H, Hn, Hp, Kx, Kxp, Kp, ld_ih, x_off, fold_bih;
};
int pcm_launch_pack(const PackArgs &a, hipStream_t stream);

// Persistent forward recurrence (persist_fwd.hip): one launch for all T steps
// when the shape is the tuned one (H = 512, L = 2, stack width <= 64, padded
// batch of 257..512 rows); the launch chain otherwise.
bool pcm_persist_fwd_supported(const SGLayout &lay, const int32_t *lengths);
size_t pcm_persist_fwd_sync_words();
int pcm_persist_fwd_mode();
// persist2_fwd.hip: the role-split persistent kernel (same workspace regions)
bool pcm_persist2_fwd_shape_ok(const SGLayout &lay);
void pcm_persist2_fwd_regions(int B, int steps, size_t *hfrag, size_t *xdfrag,
                              size_t *xtfrag, size_t *wfrag);
int pcm_launch_persist2_fwd(const SGLayout &lay, float *sv,
                            const pcm_stackgru_params *w, const int64_t *tokens,
                            float dropout_p, uint64_t seed, int32_t *status,
                            unsigned *sync, hipStream_t stream);
int pcm_launch_persist_fwd(const SGLayout &lay, float *sv,
                           const pcm_stackgru_params *w, const int64_t *tokens,
                           float dropout_p, uint64_t seed, int32_t *status,
                           unsigned *sync, hipStream_t stream);

// hs[l][0] <- h0 (or 0), with zero padding; S[0] <- stack0 (or 0)
int pcm_launch_init_state(const SGLayout &lay, float *saved, const float *h0,
                          long long h0_layer_stride, const float *stack0,
                          hipStream_t stream);
// packed mode with a shrinking active batch: zero rows (t + slot_off, b),
// t >= lengths[b], of a [slots][B][width] buffer; gather each row's state at
// slot lengths[b] (parity != 0: slot lengths[b] & 1 of a two-slot ping-pong)
int pcm_launch_zero_past_length(float *buf, int width, int T, int B,
                                int slot_off, const int32_t *lengths,
                                hipStream_t stream);
int pcm_launch_gather_at_length(const float *buf, long long slot_floats,
                                int width, int T, int parity,
                                const int32_t *lengths, float *out, int cols,
                                int B, hipStream_t stream);
// h_final / stack_final extraction (strided copy)
int pcm_launch_copy_rows(const float *src, long long lds, float *dst,
                         long long ldd, int rows, int cols,
                         hipStream_t stream);

struct GateBwdArgs {
    float *dh;            // [B, Hp] in: dL/dh_t ; out: dL/dh_{t-1} direct part
    const float *dout;    // [B, ld_dout] extra gradient of h_t or null
    float *gates;         // [B, 4H] in: r,z,n,ghn ; out: dr,dz,dni,dnh
    const float *hprev;   // [B, Hp]
    const int32_t *lengths;
    long long ld_dout;
    int B, H, Hp, t;
};
int pcm_launch_gate_bwd(const GateBwdArgs &a, hipStream_t stream);

// Fused reverse GRU cell (H % 64 == 0): one launch computes
//   dh[l]   += [dr,dz,dnh] . W_hh                (columns 0..H-1 of the tile grid)
//   dx       = [dr,dz,dni] . W_ih[:, x-part]     (columns H.. ) which is added,
//              dropout-masked, to dh[l-1] (l > 0) or stored as dtop (l == 0)
// from K-contiguous transposed weights wpackT[(n), 3H].
// Bulk part of the stack adjoint of a step: everything that does not depend on
// the stack-top gradient the layer-0 reverse cell is about to produce (the
// elementwise adjoint over all D*W elements and the three control sums, 30 KB of
// traffic per row).  It rides in the TOP layer's reverse-cell launch of the same
// step as extra blocks, off the critical chain; stack_bwd_finish_kernel adds the
// stack-top terms (rows 0 and 1 of dL/dS_{t-1}, three W-long dot products).
struct StackBulkArgs {
    const float *dS_next; // [B, D*W] dL/dS'_t without the stack-top gradient
    float *dS_prev;       // [B, D*W] dL/dS_{t-1}, complete except for that gradient
    const float *S_prev;  // [B, D*W]
    const float *av;      // [B, AVW] a, v of step t
    float *psum;          // [B, 4] control sums (push, pop, no-op) of the bulk
    const int32_t *lengths;
    int B, D, W, AVW, t;
};

struct CellBwdArgs {
    const float *dgates;  // [B, 4H]  dr, dz, dni, dnh
    const float *wpackT;  // [H + Kxp, 3H]
    float *dh;            // [B, Hp]  += (this layer)
    float *dx;            // [B, ldx] += (l > 0: dh of the layer below) or = (dtop)
    int ldx, dx_accumulate, dx_cols;
    int B, H, Hp, NT;     // NT = H + Kxp rows of wpackT
    int reg_cols, reg_rows;  // tile region per XCD (set by the launcher)
    int xparts;              // K-part partials of dx (set by the launcher)
    long long xpart_stride;  // floats between partials of dx (>= B_total * ldx)
    int t, layer;
    int b_off;               // batch index of row 0 (dropout hash, see CellFwdArgs)
    float drop_p, inv_keep;  // drop_p > 0: mask dx with the forward's dropout
    uint64_t seed;
    // fused gate gradient of the layer below (l > 0): once dx completes
    // dL/dh_t of layer l-1, its gate derivatives are taken in the epilogue
    float *gates_low;        // [B, 4H] in: r,z,n,ghn ; out: dr,dz,dni,dnh (nullable)
    const float *hprev_low;  // [B, Hp]
    const int32_t *lengths;
    // rider_rows > 0: that many extra blocks (one per batch row) behind the
    // cell's own tiles run the bulk stack adjoint (ncell is set by the launcher)
    int rider_rows, ncell, rider_per;   // rider_per: rows per rider block (launcher)
    StackBulkArgs rider;
};
int pcm_launch_cell_bwd(const CellBwdArgs &a, hipStream_t stream);
// true if that launch can carry rider blocks (narrow tiles, the stack rows fit
// the cell's LDS)
bool pcm_cell_bwd_takes_rider(const CellBwdArgs &a, int D, int W);
// how many partials (stride xpart_stride, to be summed in order) that launch
// leaves dx in; 1 = the plain result
int pcm_cell_bwd_xparts(const CellBwdArgs &a);

struct PackTArgs {
    const float *w_ih, *w_hh;
    float *wpackT;        // [H + Kxp, 3H]
    int H, Kx, Kxp, ld_ih, x_off;
};
int pcm_launch_packT(const PackTArgs &a, hipStream_t stream);

struct StackBwdArgs {
    const float *dS_next; // [B, D*W] dL/dS'_t (null = zeros)
    float *dS_prev;       // [B, D*W] dL/dS_{t-1}
    const float *dtop;    // [B, Wp] gradient of the stack top fed to layer 0,
                          // as dtop_parts partials dtop_stride floats apart
    int dtop_parts;
    long long dtop_stride;
    const float *S_prev;  // [B, D*W]
    float *av;            // [B, AVW] in: a,v ; out: dc, du (kept for the
                          // time-parallel dWc / dWs products)
    const int32_t *lengths;
    int B, H, Hp, D, W, Wp, AVW, t;
    // fused gate gradient of the TOP layer at step t-1 (what gate_bwd does in
    // a launch of its own when there is no stack step above it): gate_mode
    // 0 = off, 1 = full, 2 = control path only (t == 0: the path reaches the
    // initial hidden state).  gate_B rows (>= B with a shrinking batch).
    int gate_mode, gate_B;
    float *g_dh;             // [B, Hp] running dL/dh of the top layer
    const float *g_dout;     // [B, g_ld_dout] output gradient of step t-1 / null
    float *g_gates;          // [B, 4H] saved gates of step t-1 (in place)
    const float *g_hprev;    // [B, Hp] top-layer hidden before step t-1
    const float *ctrl_w, *push_w;
    long long g_ld_dout;
    // pcm_launch_stack_bwd_finish only: the bulk's control sums; dS_next /
    // dS_prev are then read / corrected in rows 0 and 1 only
    const float *psum;
};
int pcm_launch_stack_bwd(const StackBwdArgs &a, hipStream_t stream);
// the critical-chain remainder of the stack adjoint when the bulk rode in the
// top cell's launch (StackBulkArgs)
int pcm_launch_stack_bwd_finish(const StackBwdArgs &a, hipStream_t stream);

// x[b][j] *= dropout multiplier of (seed, layer, t, b, j)
int pcm_launch_dropout_mask_mul(float *x, long long ldx, int B, int H,
                                uint64_t seed, int layer, int t, float p,
                                hipStream_t stream);
// out[row][v] = (tokens[row] == v)
int pcm_launch_xcat(const float *xtop, int Wp, const int64_t *tokens,
                    long long rows, int V, float *out, hipStream_t stream);
int pcm_launch_onehot(const int64_t *tokens, long long rows, int V, float *out,
                      hipStream_t stream);
int pcm_launch_fill(float *dst, float value, long long n, hipStream_t stream);
// dst[r][c] (+)= src[r][c]
int pcm_launch_add_block(const float *src, long long lds, float *dst,
                         long long ldd, int rows, int cols, int accumulate,
                         hipStream_t stream);
