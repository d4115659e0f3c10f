// C ABI drivers: enqueue the kernels of one call on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "stackgru.h"

int pcm_launch_dropout_add(float *dst, long long ldd, const float *src,
                           long long lds_, int B, int H, uint64_t seed,
                           int layer, int t, float p, hipStream_t stream);
int pcm_launch_choose(const float *logits, int B, int V, int search,
                      float temperature, uint64_t seed, int step,
                      int64_t *next_tok, int64_t *tokens_out,
                      long long out_stride, int out_col, hipStream_t stream);
int pcm_launch_logits_choose(const float *h, long long ldh, const float *w,
                             const float *bias, int B, int V, int H, int search,
                             float temperature, uint64_t seed, int step,
                             int64_t *next_tok, int64_t *tokens_out,
                             long long out_stride, int out_col,
                             float *logits_out, hipStream_t stream);
int pcm_ce_static_prepare(void);   // head.hip
int pcm_launch_broadcast_token(const int64_t *src, int B, int64_t *next_tok,
                               int64_t *tokens_out, long long out_stride,
                               int out_col, hipStream_t stream);

std::recursive_mutex &pcm_api_mutex() {
    static std::recursive_mutex m;
    return m;
}

// ---- errors -----------------------------------------------------------------------
static thread_local char g_err[512] = "";

void pcm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pcm_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        pcm_set_error("%s: %s", what, hipGetErrorString(e));
        return PCM_EDEVICE;
    }
    return PCM_OK;
}

// ---- launch-duration probe ----------------------------------------------------------
#include <vector>
int g_pcm_probe_id = 0;
int g_pcm_matmul_bf16 = [] {
    const char *e = getenv("PCM_BF16");
    return (e && atoi(e) == 1) ? 1 : 0;
}();
hipEvent_t g_pcm_probe_start = nullptr, g_pcm_probe_stop = nullptr;
static std::vector<hipEvent_t> g_probe_ev;   // pairs: start, stop
static size_t g_probe_used = 0;              // events handed out
static size_t g_probe_cap = 0;
static double g_probe_ms = 0.0;
static long long g_probe_n = 0;
static long long g_probe_dropped = 0;

static void probe_drain() {
    for (size_t i = 0; i + 1 < g_probe_used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_probe_ev[i + 1]) == hipSuccess &&
            hipEventElapsedTime(&ms, g_probe_ev[i], g_probe_ev[i + 1]) ==
                hipSuccess) {
            g_probe_ms += ms;
            g_probe_n += 1;
        }
    }
    (void)hipGetLastError();   // an unrecorded pair is not a failed call
    g_probe_used = 0;
}

void pcm_probe_begin_(void) {
    if (g_probe_used + 2 > g_probe_cap) { g_probe_dropped++; return; }
    g_pcm_probe_start = g_probe_ev[g_probe_used];
    g_pcm_probe_stop = g_probe_ev[g_probe_used + 1];
}

void pcm_probe_end_(void) {
    if (!g_pcm_probe_start) return;
    g_pcm_probe_start = g_pcm_probe_stop = nullptr;
    g_probe_used += 2;
}

// ---- side stream for the time-parallel weight gradients ---------------------------
// The reverse recurrence is a chain of short, latency-bound launches; the
// weight-gradient GEMMs of steps the chain has already passed run beside it on
// a second stream (created once per device, never destroyed).
#include <stdlib.h>
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev[8] = {};
    int next = 0;
    bool ok = false;
};
static SideStream g_side[16];

static SideStream *side_stream() {
    // Off by default: with the K-major gradient kernel the products are short and
    // MFMA-dense, and fp32 MFMA waves starve the chain's VALU phases when they
    // share a SIMD, so running them beside the chain no longer beats one
    // full-length launch after it (1.66 M vs 1.64 M tokens/s).  PCM_OVERLAP=1
    // turns the side stream back on (chunk schedule: PCM_DW_CUTS).
    static const bool enabled = getenv("PCM_OVERLAP") != nullptr;
    if (!enabled) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream &s = g_side[dev];
    if (!s.ok) {
        // lowest priority: the latency-critical chain on the caller's stream
        // is dispatched first whenever both have work
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&s.stream, hipStreamNonBlocking,
                                        least) != hipSuccess)
            return nullptr;
        for (int i = 0; i < 8; ++i)
            if (hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) !=
                hipSuccess)
                return nullptr;
        s.ok = true;
    }
    return &s;
}

// ---- second stream for the second half of the batch ("two chains") --------------
// The recurrence is a chain of short dependent launches whose fixed costs
// (launch gap, prologue, epilogue) leave the CUs idle between MFMA phases.  Rows
// are independent, so the batch is walked as two half-batch chains on two
// streams: their launches interleave on the chip and one chain's MFMA phase
// fills the other's gaps.  Same kernels, same arithmetic per row.
static SideStream g_chain[16];

static SideStream *chain_stream() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    SideStream &s = g_chain[dev];
    if (!s.ok) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) !=
            hipSuccess)
            return nullptr;
        for (int i = 0; i < 8; ++i)
            if (hipEventCreateWithFlags(&s.ev[i], hipEventDisableTiming) !=
                hipSuccess)
                return nullptr;
        s.ok = true;
    }
    return &s;
}

// Row at which the second chain starts (0 = one chain, the default).
// PCM_CHAINS=2 walks the batch as two half-batch chains.
static int chain_split(const SGLayout &lay) {
    static const int force = [] {
        const char *e = getenv("PCM_CHAINS");
        return e ? atoi(e) : 0;
    }();
    static const bool overlap = getenv("PCM_OVERLAP") != nullptr;
    if (force == 1 || overlap || (lay.H % 64) != 0) return 0;
    // measured on MI355X at batch 512: two half-batch chains take 27.6 ms per
    // training step against 26.9 ms for one chain (co-running kernels stretch
    // each other by more than the hidden gaps), so one chain is the default
    if (force != 2) return 0;
    if (lay.B < 64) return 0;
    return (int)pcm_round_up(lay.B / 2, 32);
}

// make `waiter` wait for everything enqueued on `signal` so far
static void stream_wait(SideStream *s, hipStream_t signal, hipStream_t waiter) {
    hipEvent_t e = s->ev[s->next];
    s->next = (s->next + 1) & 7;
    (void)hipEventRecord(e, signal);
    (void)hipStreamWaitEvent(waiter, e, 0);
}

// ---- layouts ----------------------------------------------------------------------
static size_t take(size_t &off, size_t n) {
    size_t at = off;
    off += (size_t)pcm_round_up((int64_t)n, 64);
    return at;
}

static int check_dims(const pcm_dims *d) {
    PCM_REQUIRE(d, PCM_EINVAL, "dims is null");
    PCM_REQUIRE(d->batch > 0 && d->vocab > 0 && d->emb > 0 && d->hidden > 0,
                PCM_EINVAL, "dims: B=%d V=%d E=%d H=%d must be positive",
                d->batch, d->vocab, d->emb, d->hidden);
    PCM_REQUIRE(d->layers >= 1 && d->layers <= PCM_MAX_LAYERS, PCM_EINVAL,
                "dims: layers=%d outside 1..%d", d->layers, PCM_MAX_LAYERS);
    PCM_REQUIRE(!d->use_stack || (d->stack_depth >= 1 && d->stack_width >= 1),
                PCM_EINVAL, "dims: stack %dx%d must be positive",
                d->stack_depth, d->stack_width);
    return PCM_OK;
}

int pcm_sg_layout(const pcm_dims *d, int steps, int training, int dropout,
                  SGLayout *o) {
    int rc = check_dims(d);
    if (rc) return rc;
    PCM_REQUIRE(steps >= 1, PCM_EINVAL, "steps=%d must be >= 1", steps);
    memset(o, 0, sizeof(*o));
    o->B = d->batch; o->V = d->vocab; o->E = d->emb; o->H = d->hidden;
    o->L = d->layers; o->use_stack = d->use_stack ? 1 : 0;
    o->D = o->use_stack ? d->stack_depth : 0;
    o->W = o->use_stack ? d->stack_width : 0;
    o->T = steps; o->training = training ? 1 : 0;
    o->dropout = (training && dropout) ? 1 : 0;
    o->Hp = (int)pcm_round_up(o->H, 64);
    o->Hn = (int)pcm_round_up(o->H, 32);
    o->Wp = o->use_stack ? (int)pcm_round_up(o->W, 64) : 0;
    o->AVW = o->use_stack ? (int)pcm_round_up(3 + o->W, 4) : 0;
    o->DW = o->D * o->W;
    size_t off = 0;
    const size_t BH = (size_t)o->B * o->Hp;
    // the per-step hand-off regions of the persistent forward kernel exist only
    // where that kernel can run (an eval-mode forward of the launch chain at
    // batch 4096 x 120 steps would otherwise carry 250 MB it never touches)
    const bool persist = o->use_stack && pcm_persist_fwd_supported(*o, nullptr);
    for (int l = 0; l < o->L; ++l) {
        o->Kx[l] = l == 0 ? o->W : o->H;
        o->Kxp[l] = l == 0 ? o->Wp : o->Hp;
        o->Kp[l] = o->Kxp[l] + o->Hp;
        o->off_hs[l] = take(off, (size_t)(steps + 1) * BH);
    }
    for (int l = 1; l < o->L; ++l)
        if (o->dropout) o->off_xdrop[l] = take(off, (size_t)steps * BH);
    if (o->training)
        for (int l = 0; l < o->L; ++l)
            o->off_gates[l] = take(off, (size_t)steps * o->B * 4 * o->H);
    if (o->use_stack) {
        const size_t slots = o->training ? steps + 1 : 2;
        const size_t per = o->training ? steps : 1;
        o->off_S = take(off, slots * o->B * o->DW);
        o->off_av = take(off, per * o->B * o->AVW);
        o->xtop_per_step = (o->training || persist) ? 1 : 0;
        o->off_xtop =
            take(off, (size_t)(o->xtop_per_step ? steps : 1) * o->B * o->Wp);
    }
    o->off_sync = take(off, persist ? pcm_persist_fwd_sync_words() : 0);
    o->off_hraw = take(off, (o->use_stack && o->L >= 2) ? (size_t)o->B * 3 * o->Hn : 0);
    o->off_cspart =
        take(off, o->use_stack ? (size_t)o->B * (o->Hn / 16) * 64 : 0);
    o->off_cpre = take(off, persist ? (size_t)steps * o->B * 64 : 0);
    {
        size_t hf = 0, xdf = 0, xtf = 0, wf = 0;
        if (persist && pcm_persist_fwd_mode() == 2)
            pcm_persist2_fwd_regions(o->B, steps, &hf, &xdf, &xtf, &wf);
        o->off_hfrag[0] = take(off, hf);
        o->off_hfrag[1] = take(off, hf);
        o->off_xdfrag = take(off, o->dropout ? xdf : 0);
        o->off_xtfrag = take(off, xtf);
        o->off_wfrag = take(off, wf);
    }
    o->off_tokg = take(off, (size_t)o->V * 3 * o->Hn);
    for (int l = 0; l < o->L; ++l) {
        o->off_wpack[l] = take(off, (size_t)3 * o->Hn * o->Kp[l]);
        o->off_bpack[l] = take(off, (size_t)4 * o->Hn);
    }
    o->total = off;
    return PCM_OK;
}

static const size_t GEMM_SCRATCH_FLOATS = 40u * 1024u * 1024u;

// compute units of the current device (256 on MI355X), looked up once
static int pcm_cu_count() {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess)
            (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n;
    }();
    return cus;
}


int pcm_sg_scratch(const SGLayout &lay, SGScratch *o) {
    memset(o, 0, sizeof(*o));
    size_t off = 0;
    o->off_dh = take(off, (size_t)lay.L * lay.B * lay.Hp);
    o->off_dS = take(off, (size_t)2 * lay.B * (lay.use_stack ? lay.DW : 1));
    // (up to H / Wp K-part partials of the stack-top gradient)
    o->off_dtop = take(off, (size_t)lay.B * (lay.use_stack
                                ? (size_t)lay.Wp * std::max(1, lay.H / lay.Wp)
                                : 1));
    o->off_onehot =
        take(off, (size_t)lay.T * lay.B * (lay.V > 128 ? lay.V : 128));
    o->off_dtokg = take(off, (size_t)lay.V * 3 * lay.H);
    o->off_tmp = take(off, (size_t)lay.B * lay.Hp);
    o->off_psum = take(off, (size_t)lay.B * 4);
    o->off_dwx = take(off, (size_t)3 * lay.H * (lay.Wp > 128 ? lay.Wp : 128));
    o->off_dcw = take(off, (size_t)(lay.AVW > 0 ? lay.AVW : 4) * (lay.H + 1));
    for (int l = 0; l < lay.L; ++l)
        o->off_wpackT[l] =
            take(off, (size_t)(lay.H + lay.Kxp[l]) * 3 * lay.H);
    o->off_gemm = take(off, GEMM_SCRATCH_FLOATS);
    o->off_gemm2 = take(off, GEMM_SCRATCH_FLOATS);
    o->gemm_bytes = GEMM_SCRATCH_FLOATS * sizeof(float);
    o->total = off;
    return PCM_OK;
}

// One StackGRU step on explicit state buffers (shared by forward / generate).
struct StepBufs {
    const float *h_prev[PCM_MAX_LAYERS];
    float *h_new[PCM_MAX_LAYERS];
    float *gates[PCM_MAX_LAYERS];
    float *xdrop[PCM_MAX_LAYERS];  // dropped input OF layer l (l >= 1)
    const float *S_prev;
    float *S_new;
    float *av;
    float *xtop;
    const int64_t *tok;
};

static int enqueue_step(const SGLayout &lay, const pcm_stackgru_params *w,
                        const float *sv_tokg, float *const *wpack,
                        float *const *bpack, const StepBufs &sb,
                        const int32_t *lengths, int t, float drop_p,
                        uint64_t seed, int32_t *status, hipStream_t st,
                        int rows = -1, int b0 = 0, float *cs_part = nullptr,
                        int *have_part = nullptr, float *hraw = nullptr) {
    // cs_part / have_part: the top layer's forward cell may leave partial sums
    // of the next step's stack controls (see CellFwdArgs); *have_part says, on
    // entry, that the previous step left them for (at least) these rows, and on
    // return whether this step did
    // The step over batch rows [b0, b0 + rows): rows = the length-sorted prefix
    // of the batch still running at this step (packed mode with a shrinking
    // active batch), or one chain's half of it; -1 = everything from b0 on.
    // StepBufs holds whole-batch base pointers, the slice is applied here.
    const int Bt = rows < 0 ? lay.B - b0 : rows;
    if (Bt <= 0) return PCM_OK;
    const size_t oH = (size_t)b0 * lay.Hp, oG = (size_t)b0 * 4 * lay.H;
    const size_t oS = (size_t)b0 * lay.DW, oAV = (size_t)b0 * lay.AVW;
    const size_t oX = (size_t)b0 * lay.Wp;
    const int32_t *len = lengths ? lengths + b0 : nullptr;
    // Stack step fused into the cells (opt-in, PCM_FUSE_STACK=1): with the
    // previous step's control partials at hand, layer 0's cell derives controls,
    // push value and the new stack top itself and the top layer's launch carries
    // the stack update as extra blocks -- two launches per step instead of
    // three.  Measured 26.03 against 26.25 ms per step: the prologue (every
    // block of a row tile pulls the same 16 partials per row: 2.5 us) and the
    // update blocks (+1.9 us on the top cell) cost what the stack launch costs,
    // so the default keeps the three-launch step and its per-kernel pricing.
    // Round 3: measured again with this round's reverse pass, batch 512: 25.37
    // against 25.75 ms per step (three A/B pairs), level at the other batch sizes
    // where it engages and slower at batch 1024 (46.1 against 44.8) -- so it is
    // the default where the cells' grid is at most one block per CU (the
    // benchmarked shape), PCM_FUSE_STACK=0 / 1 force it off / on.
    static const int fuse_mode = [] {
        const char *e = getenv("PCM_FUSE_STACK");
        return e ? (atoi(e) == 1 ? 1 : 0) : -1;
    }();
    bool fuse = false;
    if (fuse_mode != 0 && lay.use_stack && lay.L >= 2 && cs_part && have_part &&
        *have_part == lay.Hn / 32 && lay.Kxp[0] == 64 && lay.Wp == 64 && lay.AVW <= 64 &&
        3 + lay.W <= 64 && lay.H == lay.Hn) {
        CellFwdArgs probe;
        memset(&probe, 0, sizeof(probe));
        probe.B = Bt; probe.Hn = lay.Hn;
        fuse = pcm_cell_fwd_tile(probe) == 32;
        if (fuse_mode < 0 && (long long)((Bt + 31) / 32) * (lay.Hn / 32) > pcm_cu_count())
            fuse = false;
    }
    // Opt-in (PCM_STACK_SPLIT64=rows per rider block), large batches (the 64 x 64
    // forward tile + the MFMA stack step, 4096 rows and more): the stack launch
    // computes controls, push value and the new stack TOP only, the 2 * D * W * 4
    // bytes per row of the update ride as extra blocks in the top layer's
    // launch.  Measured SLOWER at batch 4096 (encoder-only 169.9 k against
    // 173.6 k molecules/s, ragged 213 k against 265 k): two 64 KiB cell blocks
    // fill a CU's LDS, so the riders only start when cell blocks end -- at the
    // launch's tail, on the chain again -- and the stack launch stays.
    static const int split64_rows = [] {
        const char *e = getenv("PCM_STACK_SPLIT64");   // rows per rider block
        return e ? atoi(e) : 0;
    }();
    bool split64 = false;
    if (lay.use_stack && !fuse && split64_rows > 0 && lay.L >= 2) {
        StackFwdArgs ps;
        memset(&ps, 0, sizeof(ps));
        ps.B = Bt; ps.H = lay.H; ps.Hp = lay.Hp; ps.D = lay.D; ps.W = lay.W;
        ps.AVW = lay.AVW;
        CellFwdArgs pc;
        memset(&pc, 0, sizeof(pc));
        pc.B = Bt; pc.Hn = lay.Hn;
        split64 = pcm_stack_fwd_is_mfma(ps) && pcm_cell_fwd_tile(pc) == 64;
    }
    if (lay.use_stack && !fuse) {
        StackFwdArgs s;
        s.top_only = split64 ? 1 : 0;
        s.htop = sb.h_prev[lay.L - 1] + oH;
        s.S_prev = sb.S_prev + oS; s.S_new = sb.S_new + oS;
        s.ctrl_w = w->ctrl_w; s.ctrl_b = w->ctrl_b;
        s.push_w = w->push_w; s.push_b = w->push_b;
        s.av = sb.av + oAV; s.xtop = sb.xtop + oX; s.lengths = len;
        s.B = Bt; s.H = lay.H; s.Hp = lay.Hp; s.D = lay.D; s.W = lay.W;
        s.Wp = lay.Wp; s.AVW = lay.AVW; s.t = t;
        s.cs_part = (cs_part && have_part && *have_part)
                        ? cs_part + (size_t)b0 * (lay.Hn / 16) * 64 : nullptr;
        s.cs_parts = have_part ? *have_part : 0;
        pcm_probe_begin(PCM_K_STACK_FWD);
        int rc = pcm_launch_stack_fwd(s, st);
        pcm_probe_end(PCM_K_STACK_FWD);
        if (rc) return rc;
    }
    for (int l = 0; l < lay.L; ++l) {
        CellFwdArgs c;
        c.xin = l == 0 ? sb.xtop + oX
                       : (lay.dropout ? sb.xdrop[l] : sb.h_new[l - 1]) + oH;
        c.hprev = sb.h_prev[l] + oH;
        c.wpack = wpack[l]; c.bpack = bpack[l];
        c.tokg = l == 0 ? sv_tokg : nullptr;
        c.tok = l == 0 ? sb.tok + b0 : nullptr;
        c.hnew = sb.h_new[l] + oH;
        c.gates = sb.gates[l] ? sb.gates[l] + oG : nullptr;
        c.xdrop = (lay.dropout && l + 1 < lay.L) ? sb.xdrop[l + 1] + oH : nullptr;
        c.lengths = len; c.status = status;
        c.ldx = l == 0 ? lay.Wp : lay.Hp;
        c.nx_chunks = lay.Kxp[l] / 64; c.nh_chunks = lay.Hp / 64;
        c.Kp = lay.Kp[l];
        c.B = Bt; c.H = lay.H; c.Hn = lay.Hn; c.Hp = lay.Hp; c.V = lay.V;
        c.t = t; c.next_layer = l + 1; c.b_off = b0;
        c.drop_p = drop_p; c.inv_keep = drop_p < 1.f ? 1.f / (1.f - drop_p) : 0.f;
        c.seed = seed;
        c.cs_ctrl_w = c.cs_push_w = nullptr; c.cs_part = nullptr; c.cs_W = 0;
        c.xt_part = nullptr; c.su_S = nullptr;
        c.hz_raw = nullptr; c.hraw = nullptr; c.hz_hprev = c.hz_wpack = nullptr;
        c.hz_nbt = c.hz_Kp = c.hz_kx = 0;
        // Opt-in (PCM_FWD_HOIST=1): layer 1's hidden-side product -- it only needs
        // h1_{t-1} -- rides as 256 extra blocks in layer 0's fused launch, whose
        // own blocks spend their first 4 us in the stack prologue with the matrix
        // pipes idle.  Measured at batch 512: the top cell 19.2 -> 14.3 us, but
        // the layer-0 launch 18.2 -> 24.5 us: the co-resident blocks do not hide
        // in the prologue, they stretch it (25.13 against 25.05 ms per step).
        static const bool hoist_on = [] {
            const char *e = getenv("PCM_FWD_HOIST");
            return e && atoi(e) != 0;
        }();
        const bool hoist = fuse && hoist_on && hraw && lay.L >= 2;
        if (hoist && l == 0) {
            c.hz_hprev = sb.h_prev[1] + oH;
            c.hz_wpack = wpack[1];
            c.hz_raw = hraw + (size_t)b0 * 3 * lay.Hn;
            c.hz_nbt = (Bt + 31) / 32;
            c.hz_Kp = lay.Kp[1];
            c.hz_kx = lay.Kxp[1] / 64;
        }
        if (hoist && l == 1) {
            c.hraw = hraw + (size_t)b0 * 3 * lay.Hn;
            c.nh_chunks = 0;
        }
        if (fuse && l == 0) {
            c.xt_part = cs_part + (size_t)b0 * (lay.Hn / 16) * 64;
            c.xt_parts = lay.Hn / 32;
            c.xt_ctrl_b = w->ctrl_b; c.xt_push_b = w->push_b;
            c.xt_S = sb.S_prev + oS;
            c.xt_av = sb.av + oAV; c.xt_xtop = sb.xtop + oX;
            c.xt_W = lay.W; c.xt_D = lay.D; c.xt_AVW = lay.AVW;
        }
        if ((fuse || split64) && l == lay.L - 1) {
            c.su_S = sb.S_prev + oS; c.su_Snew = sb.S_new + oS;
            c.su_av = sb.av + oAV;
            c.su_nbt = fuse ? (Bt + 31) / 32 : (Bt + 63) / 64;
            c.su_rows = fuse ? 2 : split64_rows;
            c.su_D = lay.D; c.su_W = lay.W; c.su_AVW = lay.AVW;
        }
        if (l == lay.L - 1 && have_part) {
            *have_part = 0;
            if (cs_part && lay.use_stack && 3 + lay.W <= 64) {
                c.cs_ctrl_w = w->ctrl_w; c.cs_push_w = w->push_w;
                c.cs_part = cs_part;
                c.cs_W = lay.W;
                // partials per row: one per column tile of THIS launch's tile
                // width; a chain's slice starts at the widest layout's offset
                *have_part = pcm_cell_fwd_ctrl_partials(c);
                c.cs_part = *have_part
                                ? cs_part + (size_t)b0 * (lay.Hn / 16) * 64
                                : nullptr;
            }
        }
        pcm_probe_begin(PCM_K_CELL_FWD, l);
        int rc = pcm_launch_cell_fwd(c, st);
        pcm_probe_end(PCM_K_CELL_FWD, l);
        if (rc) return rc;
    }
    return PCM_OK;
}

// rows [b0, b0 + n) that chain c walks at a step with `total` active rows
// (the batch's length-sorted prefix); split == 0: one chain takes them all
static inline void chain_rows(int c, int split, int total, int *b0, int *n) {
    if (split == 0) { *b0 = 0; *n = total; return; }
    if (c == 0) { *b0 = 0; *n = total < split ? total : split; }
    else { *b0 = split; *n = total > split ? total - split : 0; }
}

// pack weights + token gate table into the workspace
static int enqueue_prepare(const SGLayout &lay, const pcm_stackgru_params *w,
                           float *sv, hipStream_t st) {
    const int ld0 = lay.E + lay.W;
    for (int l = 0; l < lay.L; ++l) {
        PackArgs p;
        p.w_ih = w->w_ih[l]; p.w_hh = w->w_hh[l];
        p.b_ih = w->b_ih[l]; p.b_hh = w->b_hh[l];
        p.wpack = sv + lay.off_wpack[l]; p.bpack = sv + lay.off_bpack[l];
        p.H = lay.H; p.Hn = lay.Hn; p.Hp = lay.Hp;
        p.Kx = lay.Kx[l]; p.Kxp = lay.Kxp[l]; p.Kp = lay.Kp[l];
        p.ld_ih = l == 0 ? ld0 : lay.H;
        p.x_off = l == 0 ? lay.E : 0;
        p.fold_bih = l == 0 ? 1 : 0;
        pcm_launch_pack(p, st);
    }
    // tokg[v][g*Hn + j] = sum_e emb[v][e] * w_ih0[g*H + j][e] + b_ih0[g*H + j]
    GemmBatch gb;
    gb.count = 3;
    for (int g = 0; g < 3; ++g) {
        GemmProb p = pcm_gemm_prob(
            w->emb, lay.E, 1, w->w_ih[0] + (long long)g * lay.H * ld0, 1, ld0,
            sv + lay.off_tokg + (size_t)g * lay.Hn, 3LL * lay.Hn, lay.V, lay.H,
            lay.E);
        p.bias = w->b_ih[0] + (long long)g * lay.H;
        gb.p[g] = p;
    }
    return pcm_gemm_launch(gb, nullptr, 0, st);
}

static int check_params(const SGLayout &lay, const pcm_stackgru_params *w,
                        const char *who) {
    PCM_REQUIRE(w, PCM_EINVAL, "%s: params is null", who);
    PCM_REQUIRE(w->emb, PCM_EINVAL, "%s: embedding is null", who);
    if (lay.use_stack)
        PCM_REQUIRE(w->ctrl_w && w->ctrl_b && w->push_w && w->push_b,
                    PCM_EINVAL, "%s: stack layer weights are null", who);
    for (int l = 0; l < lay.L; ++l)
        PCM_REQUIRE(w->w_ih[l] && w->w_hh[l] && w->b_ih[l] && w->b_hh[l],
                    PCM_EINVAL, "%s: GRU layer %d weights are null", who, l);
    return PCM_OK;
}

static int check_active_rows(const int32_t *active_rows, const int32_t *lengths,
                             int steps, int batch, const char *who) {
    if (!active_rows) return PCM_OK;
    PCM_REQUIRE(lengths, PCM_EINVAL, "%s: active_rows needs lengths", who);
    int prev = batch;
    for (int t = 0; t < steps; ++t) {
        PCM_REQUIRE(active_rows[t] >= 0 && active_rows[t] <= prev, PCM_EINVAL,
                    "%s: active_rows[%d]=%d must lie in [0, %d] (non-increasing, "
                    "length-sorted batch)", who, t, active_rows[t], prev);
        prev = active_rows[t];
    }
    return PCM_OK;
}

extern "C" {

int pcm_version(void) { return 4; }

int pcm_set_matmul_precision(int32_t mode) {
    PCM_API_GUARD;
    PCM_REQUIRE(mode == 0 || mode == 1, PCM_EINVAL,
                "set_matmul_precision: mode %d (0 = fp32, 1 = bf16 operands)",
                (int)mode);
    g_pcm_matmul_bf16 = mode;
    return PCM_OK;
}

int pcm_get_matmul_precision(void) { return g_pcm_matmul_bf16; }

int pcm_probe_enable(int32_t kernel_id, int32_t max_launches) {
    PCM_API_GUARD;
    PCM_REQUIRE((kernel_id >= 0 && kernel_id <= PCM_K_DW_GEMM) ||
                    kernel_id == 11 || kernel_id == 12 || kernel_id == 51 ||
                    kernel_id == 52,
                PCM_EINVAL, "probe: unknown kernel id %d", kernel_id);
    probe_drain();
    g_probe_ms = 0.0; g_probe_n = 0; g_probe_dropped = 0;
    if (kernel_id != 0) {
        const size_t want = (size_t)(max_launches > 0 ? max_launches : 0) * 2;
        while (g_probe_ev.size() < want) {
            hipEvent_t ev;
            hipError_t e = hipEventCreate(&ev);
            PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "probe: %s",
                        hipGetErrorString(e));
            g_probe_ev.push_back(ev);
        }
        g_probe_cap = want;
    }
    g_pcm_probe_id = kernel_id;
    return PCM_OK;
}

// a kernel that does nothing, for pcm_probe_calibrate
__global__ void pcm_empty_kernel(int *sink) {
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *sink = 0;
}

int pcm_probe_calibrate(int32_t launches, double *probe_us, double *chain_us,
                        void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(launches >= 8 && launches <= 4096 && probe_us && chain_us,
                PCM_EINVAL, "probe_calibrate: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int keep_id = g_pcm_probe_id;
    g_pcm_probe_id = 0;
    std::vector<hipEvent_t> ev(2 * (size_t)launches + 2);
    for (auto &e : ev) {
        hipError_t err = hipEventCreate(&e);
        PCM_REQUIRE(err == hipSuccess, PCM_EDEVICE, "probe_calibrate: %s",
                    hipGetErrorString(err));
    }
    auto run = [&](bool bound) {
        for (int i = 0; i < launches; ++i) {
            if (bound)
                hipExtLaunchKernelGGL(pcm_empty_kernel, dim3(256), dim3(256), 0,
                                      st, ev[2 * i], ev[2 * i + 1], 0,
                                      (int *)nullptr);
            else
                hipLaunchKernelGGL(pcm_empty_kernel, dim3(256), dim3(256), 0, st,
                                   (int *)nullptr);
        }
    };
    run(false);                                   // warm the code object
    // (1) the chain's wall clock per dependent launch, no per-launch events
    (void)hipEventRecord(ev[2 * launches], st);
    run(false);
    (void)hipEventRecord(ev[2 * launches + 1], st);
    // (2) the same launches with a (start, stop) pair bound to each
    run(true);
    (void)hipStreamSynchronize(st);
    float wall_ms = 0.f, sum_ms = 0.f;
    (void)hipEventElapsedTime(&wall_ms, ev[2 * launches], ev[2 * launches + 1]);
    for (int i = 0; i < launches; ++i) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        sum_ms += ms;
    }
    for (auto &e : ev) (void)hipEventDestroy(e);
    g_pcm_probe_id = keep_id;
    *chain_us = (double)wall_ms * 1e3 / launches;
    *probe_us = (double)sum_ms * 1e3 / launches;
    return pcm_check_launch("probe_calibrate");
}

int pcm_probe_read(double *total_ms, int64_t *launches, int64_t *dropped) {
    PCM_API_GUARD;
    probe_drain();
    if (total_ms) *total_ms = g_probe_ms;
    if (launches) *launches = g_probe_n;
    if (dropped) *dropped = g_probe_dropped;
    return PCM_OK;
}

const char *pcm_last_error(void) { return g_err; }

int pcm_device_check(void) {
    PCM_API_GUARD;
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "no HIP device: %s",
                hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    PCM_REQUIRE(e == hipSuccess, PCM_EDEVICE, "device properties: %s",
                hipGetErrorString(e));
    PCM_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, PCM_EDEVICE,
                "device %d is %s; this library is built for gfx950 only", dev,
                prop.gcnArchName);
    // per-device set-up that must not happen inside a stream capture: the
    // address of pcm_softmax_ce_static's scratch array on this device
    return pcm_ce_static_prepare();
}

size_t pcm_stackgru_saved_bytes(const pcm_dims *dims, int32_t steps,
                                int32_t training) {
    SGLayout lay;
    if (pcm_sg_layout(dims, steps, training, 1, &lay)) return 0;
    return lay.total * sizeof(float);
}

int pcm_stackgru_out_view(const pcm_dims *dims, int32_t steps, int32_t training,
                          size_t *offset_floats, size_t *row_stride_floats) {
    SGLayout lay;
    int rc = pcm_sg_layout(dims, steps, training, 1, &lay);
    if (rc) return rc;
    PCM_REQUIRE(offset_floats && row_stride_floats, PCM_EINVAL,
                "out_view: null output");
    *offset_floats = lay.off_hs[lay.L - 1] + lay.hs_step();
    *row_stride_floats = lay.Hp;
    return PCM_OK;
}

size_t pcm_stackgru_scratch_bytes(const pcm_dims *dims, int32_t steps) {
    SGLayout lay;
    if (pcm_sg_layout(dims, steps, 1, 1, &lay)) return 0;
    SGScratch sc;
    pcm_sg_scratch(lay, &sc);
    return sc.total * sizeof(float);
}

int pcm_stackgru_forward(const pcm_dims *dims,
                         const pcm_stackgru_params *weights,
                         const int64_t *tokens, int32_t steps,
                         const int32_t *lengths, const int32_t *active_rows,
                         const float *h0, int64_t h0_layer_stride,
                         const float *stack0, float dropout_p, uint64_t seed,
                         int32_t training, void *saved, size_t saved_bytes,
                         float *h_final, float *stack_final, int32_t *status,
                         void *stream) {
    PCM_API_GUARD;
    SGLayout lay;
    // the layout always reserves the dropout slots so that saved_bytes() does
    // not depend on dropout_p; they are only written when dropout is active
    int rc = pcm_sg_layout(dims, steps, training, 1, &lay);
    if (rc) return rc;
    rc = check_params(lay, weights, "stackgru_forward");
    if (rc) return rc;
    PCM_REQUIRE(tokens && saved, PCM_EINVAL, "stackgru_forward: null pointer");
    PCM_REQUIRE(saved_bytes >= lay.total * sizeof(float), PCM_ENOMEM,
                "stackgru_forward: saved workspace %zu < %zu bytes", saved_bytes,
                lay.total * sizeof(float));
    PCM_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, PCM_EINVAL,
                "stackgru_forward: dropout %f outside [0,1)", dropout_p);
    PCM_REQUIRE(((uintptr_t)saved & 15) == 0, PCM_EINVAL,
                "stackgru_forward: saved workspace must be 16-byte aligned");
    rc = check_active_rows(active_rows, lengths, steps, lay.B,
                           "stackgru_forward");
    if (rc) return rc;
    if (lay.use_stack) {
        // an unsupported stack size fails here, before any launch -- for a
        // training-mode forward also against the reverse pass's (tighter) limit
        rc = pcm_stack_fwd_check(lay.D, lay.W, lay.H, lay.AVW);
        if (!rc && training) rc = pcm_stack_bwd_check(lay.D, lay.W, lay.AVW);
        if (rc) return rc;
    }
    const bool drop = training && dropout_p > 0.f && lay.L > 1;
    lay.dropout = drop ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    float *sv = (float *)saved;

    if (active_rows) {
        // rows past their length are not computed at all: whatever the
        // time-parallel products of the reverse pass (and the caller's output
        // layer) read there must be zero, not stale memory
        const int T = steps, B = lay.B;
        for (int l = 0; l < lay.L; ++l) {
            pcm_launch_zero_past_length(sv + lay.off_hs[l], lay.Hp, T, B, 1,
                                        lengths, st);
            if (drop && l >= 1)
                pcm_launch_zero_past_length(sv + lay.off_xdrop[l], lay.Hp, T, B,
                                            0, lengths, st);
            if (training)
                pcm_launch_zero_past_length(sv + lay.off_gates[l], 4 * lay.H, T,
                                            B, 0, lengths, st);
        }
        if (training && lay.use_stack) {
            pcm_launch_zero_past_length(sv + lay.off_av, lay.AVW, T, B, 0,
                                        lengths, st);
            pcm_launch_zero_past_length(sv + lay.off_xtop, lay.Wp, T, B, 0,
                                        lengths, st);
        }
    }
    if (lay.Hp != lay.H) {
        // zero the padding columns of every hidden-state slot once
        for (int l = 0; l < lay.L; ++l)
            pcm_launch_fill(sv + lay.off_hs[l], 0.f,
                            (long long)(steps + 1) * lay.hs_step(), st);
        if (drop)
            for (int l = 1; l < lay.L; ++l)
                pcm_launch_fill(sv + lay.off_xdrop[l], 0.f,
                                (long long)steps * lay.hs_step(), st);
    }
    rc = enqueue_prepare(lay, weights, sv, st);
    if (rc) return rc;
    pcm_launch_init_state(lay, sv, h0, h0_layer_stride, stack0, st);

    float *wpack[PCM_MAX_LAYERS], *bpack[PCM_MAX_LAYERS];
    for (int l = 0; l < lay.L; ++l) {
        wpack[l] = sv + lay.off_wpack[l];
        bpack[l] = sv + lay.off_bpack[l];
    }
    // the tuned shape walks all steps in ONE persistent launch
    const bool persist = !active_rows && pcm_persist_fwd_supported(lay, lengths);
    if (persist) {
        rc = (pcm_persist_fwd_mode() == 2 ? pcm_launch_persist2_fwd
                                          : pcm_launch_persist_fwd)(
            lay, sv, weights, tokens, dropout_p, seed, status,
            (unsigned *)(sv + lay.off_sync), st);
        if (rc) return rc;
    }
    // one chain over the whole batch, or two half-batch chains on two streams
    int split = persist ? 0 : chain_split(lay);
    SideStream *cs = split ? chain_stream() : nullptr;
    if (!cs) split = 0;
    const int nch = split ? 2 : 1;
    if (split) stream_wait(cs, st, cs->stream);   // fork: state is initialised
    // partial stack-control sums handed from a step's top cell to the next
    // step's stack kernel (PCM_CTRL_PARTIALS=0: off)
    static const bool ctrl_partials = [] {
        const char *e = getenv("PCM_CTRL_PARTIALS");
        return !e || atoi(e) != 0;
    }();
    float *cs_part = (ctrl_partials && lay.use_stack) ? sv + lay.off_cspart
                                                      : nullptr;
    int have_part[2] = {0, 0};
    for (int t = 0; t < steps && !persist; ++t) {
        StepBufs sb;
        memset(&sb, 0, sizeof(sb));
        for (int l = 0; l < lay.L; ++l) {
            sb.h_prev[l] = sv + lay.off_hs[l] + (size_t)t * lay.hs_step();
            sb.h_new[l] = sv + lay.off_hs[l] + (size_t)(t + 1) * lay.hs_step();
            sb.gates[l] = training ? sv + lay.off_gates[l] +
                                         (size_t)t * lay.B * 4 * lay.H
                                   : nullptr;
            sb.xdrop[l] = (drop && l >= 1)
                              ? sv + lay.off_xdrop[l] + (size_t)t * lay.hs_step()
                              : nullptr;
        }
        if (lay.use_stack) {
            sb.S_prev = sv + lay.s_index(t);
            sb.S_new = sv + lay.s_index(t + 1);
            sb.av = sv + lay.av_index(t);
            sb.xtop = sv + lay.xtop_index(t);
        }
        sb.tok = tokens + (size_t)t * lay.B;
        const int total = active_rows ? active_rows[t] : lay.B;
        for (int c = 0; c < nch; ++c) {
            int b0, n;
            chain_rows(c, split, total, &b0, &n);
            rc = enqueue_step(lay, weights, sv + lay.off_tokg, wpack, bpack, sb,
                              lengths, t, dropout_p, seed, status,
                              c == 0 ? st : cs->stream, n, b0, cs_part,
                              &have_part[c],
                              lay.L >= 2 && lay.use_stack ? sv + lay.off_hraw
                                                          : nullptr);
            if (rc) return rc;
        }
    }
    if (split) stream_wait(cs, cs->stream, st);   // join
    if (active_rows) {
        // each row's state after ITS last step (slot lengths[b])
        if (h_final)
            for (int l = 0; l < lay.L; ++l)
                pcm_launch_gather_at_length(
                    sv + lay.off_hs[l], (long long)lay.hs_step(), lay.Hp, steps,
                    0, lengths, h_final + (size_t)l * lay.B * lay.H, lay.H,
                    lay.B, st);
        if (stack_final && lay.use_stack)
            pcm_launch_gather_at_length(
                sv + lay.off_S, (long long)lay.B * lay.DW, lay.DW, steps,
                training ? 0 : 1, lengths, stack_final, lay.DW, lay.B, st);
        return pcm_check_launch("stackgru_forward");
    }
    if (h_final)
        for (int l = 0; l < lay.L; ++l)
            pcm_launch_copy_rows(
                sv + lay.off_hs[l] + (size_t)steps * lay.hs_step(), lay.Hp,
                h_final + (size_t)l * lay.B * lay.H, lay.H, lay.B, lay.H, st);
    if (stack_final && lay.use_stack)
        pcm_launch_copy_rows(sv + lay.s_index(steps), lay.DW, stack_final,
                             lay.DW, lay.B, lay.DW, st);
    return pcm_check_launch("stackgru_forward");
}

int pcm_stackgru_backward(const pcm_dims *dims,
                          const pcm_stackgru_params *weights,
                          const int64_t *tokens, int32_t steps,
                          const int32_t *lengths, const int32_t *active_rows,
                          float dropout_p, uint64_t seed, void *saved,
                          size_t saved_bytes,
                          const float *d_out, int64_t d_out_row_stride,
                          const float *d_h_final, const float *d_stack_final,
                          float *d_h0, float *d_stack0,
                          const pcm_stackgru_params *grads, int32_t accumulate,
                          void *scratch, size_t scratch_bytes, void *stream) {
    PCM_API_GUARD;
    SGLayout lay;
    int rc = pcm_sg_layout(dims, steps, 1, 1, &lay);
    if (rc) return rc;
    rc = check_params(lay, weights, "stackgru_backward");
    if (rc) return rc;
    PCM_REQUIRE(tokens && saved && scratch && grads, PCM_EINVAL,
                "stackgru_backward: null pointer");
    PCM_REQUIRE(saved_bytes >= lay.total * sizeof(float), PCM_ENOMEM,
                "stackgru_backward: saved workspace too small");
    SGScratch sc;
    pcm_sg_scratch(lay, &sc);
    PCM_REQUIRE(scratch_bytes >= sc.total * sizeof(float), PCM_ENOMEM,
                "stackgru_backward: scratch %zu < %zu bytes", scratch_bytes,
                sc.total * sizeof(float));
    rc = check_active_rows(active_rows, lengths, steps, lay.B,
                           "stackgru_backward");
    if (rc) return rc;
    if (lay.use_stack) {
        rc = pcm_stack_bwd_check(lay.D, lay.W, lay.AVW);
        if (rc) return rc;
    }
    const bool drop = dropout_p > 0.f && lay.L > 1;
    lay.dropout = drop ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    float *sv = (float *)saved;
    float *scf = (float *)scratch;
    const int B = lay.B, H = lay.H, Hp = lay.Hp, L = lay.L, T = steps;
    const int ld0 = lay.E + lay.W;
    void *gs = scf + sc.off_gemm;
    const size_t gsb = sc.gemm_bytes;

    // running gradients
    float *dh = scf + sc.off_dh;  // [L][B][Hp]
    pcm_launch_fill(dh, 0.f, (long long)L * B * Hp, st);
    if (d_h_final)
        for (int l = 0; l < L; ++l)
            pcm_launch_copy_rows(d_h_final + (size_t)l * B * H, H,
                                 dh + (size_t)l * B * Hp, Hp, B, H, st);
    float *dS[2] = {scf + sc.off_dS, scf + sc.off_dS + (size_t)B * lay.DW};
    int cur = 0;
    if (lay.use_stack) {
        // (with a shrinking active batch a row joins the walk at its own last
        // step, in whichever ping-pong slot is current then: seed both)
        for (int q = 0; q < (active_rows ? 2 : 1); ++q) {
            if (d_stack_final)
                pcm_launch_copy_rows(d_stack_final, lay.DW, dS[q], lay.DW, B,
                                     lay.DW, st);
            else
                pcm_launch_fill(dS[q], 0.f, (long long)B * lay.DW, st);
        }
    }
    float *dtop = scf + sc.off_dtop;
    float *tmp = scf + sc.off_tmp;
    float *psum = scf + sc.off_psum;
    // PCM_STACK_SPLIT=0: the whole stack adjoint in its own launch on the chain
    // rows [0, Bt * top%) ride in the top layer's launch, the rest in layer 0's
    static const int rider_top_percent = [] {
        const char *e = getenv("PCM_STACK_RIDER_TOP");
        return e ? std::max(0, std::min(100, atoi(e))) : 0;
    }();
    static const bool stack_split_on = [] {
        const char *e = getenv("PCM_STACK_SPLIT");
        return !e || atoi(e) != 0;
    }();
    // fused reverse cell needs whole 64-wide chunks inside every gate segment
    const bool fused_bwd = (H % 64) == 0;
    if (fused_bwd) {
        for (int l = 0; l < L; ++l) {
            PackTArgs pt;
            pt.w_ih = weights->w_ih[l]; pt.w_hh = weights->w_hh[l];
            pt.wpackT = scf + sc.off_wpackT[l];
            pt.H = H; pt.Kx = lay.Kx[l]; pt.Kxp = lay.Kxp[l];
            pt.ld_ih = l == 0 ? ld0 : H;
            pt.x_off = l == 0 ? lay.E : 0;
            pcm_launch_packT(pt, st);
        }
    }

    // ---- time-parallel parameter gradients, in chunks of steps ---------------------
    // Rows [t_lo*B, t_hi*B) of the gate-gradient / activation buffers are final
    // once the chain has passed step t_lo, so their products are accumulated
    // on the side stream while the chain keeps walking down.
    SideStream *side = side_stream();
    hipStream_t gst = side ? side->stream : st;
    void *gs2 = side ? (void *)(scf + sc.off_gemm2) : gs;
    const int acc = accumulate ? 1 : 0;
    // Chunk boundaries (in steps already walked): shrinking chunks, so that the
    // work still queued on the gradient stream when the chain ends is small.
    int bounds[5];
    int nbounds = 0;
    {
        static int cut_percent[5] = {100, 100, 100, 100, 100};
        static const bool parsed = [] {
            const char *e = getenv("PCM_DW_CUTS");  // e.g. "45,75,90,97"
            if (!e) return true;
            int n = 0;
            for (const char *q = e; *q && n < 5; ++n) {
                cut_percent[n] = atoi(q);
                while (*q && *q != ',') ++q;
                if (*q == ',') ++q;
            }
            for (; n < 5; ++n) cut_percent[n] = 100;
            return true;
        }();
        (void)parsed;
        for (int i = 0; i < 5; ++i) {
            const int b = (int)((long long)T * cut_percent[i] / 100);
            if (b > 0 && b < T && (nbounds == 0 || b > bounds[nbounds - 1]))
                bounds[nbounds++] = b;
        }
    }
    int next_bound = 0;
    int pending_hi = T;      // steps [?, pending_hi) not yet handed to gst
    bool first_chunk = true;
    if (side) stream_wait(side, st, gst);  // inputs of this call are ready
    auto param_grads = [&](int t_lo, int t_hi, bool first, bool last) -> int {
        const size_t r0 = (size_t)t_lo * B;
        const int R = (t_hi - t_lo) * B;
        const int cacc = first ? acc : 1;
        int r;
        GemmBatch gb;
        gb.count = 0;
        // packed mode: the products below skip the 16-row chunks no sequence
        // reaches (PCM_ROW_SKIP=0: multiply the zeros, as in padded mode)
        static const bool row_skip_enabled = [] {
            const char *e = getenv("PCM_ROW_SKIP");
            return !e || atoi(e) != 0;
        }();
        GemmRowSkip skip;
        const bool use_skip =
            row_skip_enabled && active_rows &&
            pcm_gemm_row_skip(active_rows + t_lo, t_hi - t_lo, B, &skip);
        auto flush = [&]() -> int {
            if (gb.count == 0) return PCM_OK;
            pcm_probe_begin(PCM_K_DW_GEMM);
            int q = pcm_gemm_launch(gb, gs2, gsb, gst,
                                    use_skip ? &skip : nullptr);
            pcm_probe_end(PCM_K_DW_GEMM);
            gb.count = 0;
            return q;
        };
        // large, square-ish products first.  Bias gradients ride along as column
        // sums of the A operand where the K-major kernel takes the batch.
        bool bih_done[PCM_MAX_LAYERS] = {}, bhh_done[PCM_MAX_LAYERS] = {};
        bool cb_done = false;
        const float *G0 = sv + lay.off_gates[0] + r0 * 4 * H;
        const bool pad_ok = (R % 16) == 0 && (H % 4) == 0;
        float *dwx = scf + sc.off_dwx, *dcw = scf + sc.off_dcw;
        float *onehot = scf + sc.off_onehot;
        float *dtokg = scf + sc.off_dtokg;
        const bool tok_path = grads->w_ih[0] || grads->emb;
        // One call over all steps (the default): the stack-top columns of W_ih0
        // and the token table share their left operand (layer 0's gate
        // gradients), so they are ONE wide product against [stack top | one-hot
        // token] (128 columns) in the wide launch -- the gate gradients are read
        // once and no tile is half empty: dwx[3H][128] = [dW_ih0[:, E:] | dtokg^T]
        const bool merged = pad_ok && first && last && lay.use_stack &&
                            grads->w_ih[0] && lay.Wp <= 64 && lay.V <= 64;
        {
            GemmProb big[3 * PCM_MAX_LAYERS + 1];
            int tag[3 * PCM_MAX_LAYERS + 1];  // layer * 2 + (0: b_hh, 1: b_ih)
            int nbig = 0;
            if (merged) {
                pcm_launch_xcat(sv + lay.off_xtop + r0 * lay.Wp, lay.Wp,
                                tokens + r0, (long long)R, lay.V, onehot, gst);
                GemmProb p = pcm_gemm_prob(G0, 1, 4LL * H, onehot, 128, 1, dwx,
                                           128, 3 * H, 128, R);
                if (grads->b_ih[0]) {
                    p.colsum1 = grads->b_ih[0];
                    p.cs_accumulate = cacc;
                }
                // a launch of its own: 12 tiles x 64 splits = one full round of
                // resident blocks (inside the wide batch its 12 tiles left the
                // last round of 156 x 16 blocks a quarter full: +250 us)
                gb.count = 0;
                gb.p[gb.count++] = p;
                if (pcm_gemm_takes_colsum(gb, true)) {
                    if (p.colsum1) bih_done[0] = true;
                } else {
                    gb.p[0].colsum1 = nullptr;
                }
                r = flush();
                if (r) return r;
            }
            for (int l = 0; l < L; ++l) {
                const float *G = sv + lay.off_gates[l] + r0 * 4 * H;
                const float *hprev = sv + lay.off_hs[l] + r0 * Hp;
                if (grads->w_hh[l]) {
                    GemmProb p1 = pcm_gemm_prob(G, 1, 4LL * H, hprev, Hp, 1,
                                                grads->w_hh[l], H, 2 * H, H, R);
                    GemmProb p2 = pcm_gemm_prob(
                        G + 3 * H, 1, 4LL * H, hprev, Hp, 1,
                        grads->w_hh[l] + (size_t)2 * H * H, H, H, H, R);
                    if (grads->b_hh[l]) {
                        p1.colsum1 = grads->b_hh[l];
                        p2.colsum1 = grads->b_hh[l] + 2 * H;
                    }
                    p1.accumulate = p2.accumulate = cacc;
                    tag[nbig] = l * 2; big[nbig++] = p1;
                    tag[nbig] = l * 2; big[nbig++] = p2;
                }
                if (l > 0 && grads->w_ih[l]) {
                    const float *x =
                        (drop ? sv + lay.off_xdrop[l]
                              : sv + lay.off_hs[l - 1] + lay.hs_step()) + r0 * Hp;
                    GemmProb p3 = pcm_gemm_prob(G, 1, 4LL * H, x, Hp, 1,
                                                grads->w_ih[l], H, 3 * H, H, R);
                    if (grads->b_ih[l]) p3.colsum1 = grads->b_ih[l];
                    p3.accumulate = cacc;
                    tag[nbig] = l * 2 + 1; big[nbig++] = p3;
                }
            }
            // w_hh's two products of a layer stay in one batch (even count)
            for (int i0 = 0; i0 < nbig;) {
                int i1 = std::min(nbig, i0 + PCM_GEMM_MAX_PROBS);
                if (i1 < nbig && tag[i1] == tag[i1 - 1] && !(tag[i1] & 1)) --i1;
                gb.count = 0;
                for (int i = i0; i < i1; ++i) gb.p[gb.count++] = big[i];
                const bool takes = pcm_gemm_takes_colsum(gb, true);
                for (int i = i0; i < i1; ++i) {
                    if (!takes) {
                        gb.p[i - i0].colsum1 = nullptr;
                    } else if (big[i].colsum1) {
                        if (tag[i] & 1) bih_done[tag[i] >> 1] = true;
                        else bhh_done[tag[i] >> 1] = true;
                    }
                }
                r = flush();
                if (r) return r;
                i0 = i1;
            }
        }
        // skinny products over the same long reduction.  When the rows allow
        // 16-byte loads they go through padded scratch products (the stack top
        // is stored Wp wide with zero padding, the controls AVW wide) so the
        // K-major kernel takes them; the valid block is then added to the
        // gradient.  Otherwise the strided generic kernel computes them in place.
        if (tok_path && !merged)
            pcm_launch_onehot(tokens + r0, (long long)R, lay.V, onehot, gst);
        if (grads->w_ih[0] && lay.use_stack && !merged) {
            GemmProb p = pad_ok
                ? pcm_gemm_prob(G0, 1, 4LL * H, sv + lay.off_xtop + r0 * lay.Wp,
                                lay.Wp, 1, dwx, lay.Wp, 3 * H, lay.Wp, R)
                : pcm_gemm_prob(G0, 1, 4LL * H, sv + lay.off_xtop + r0 * lay.Wp,
                                lay.Wp, 1, grads->w_ih[0] + lay.E, ld0, 3 * H,
                                lay.W, R);
            p.accumulate = pad_ok ? 0 : cacc;
            if (pad_ok && grads->b_ih[0]) {
                p.colsum1 = grads->b_ih[0];
                p.cs_accumulate = cacc;
            }
            gb.p[gb.count++] = p;
        }
        if (lay.use_stack && (grads->ctrl_w || grads->push_w)) {
            const float *dav = sv + lay.off_av + r0 * lay.AVW;
            const float *htop = sv + lay.off_hs[L - 1] + r0 * Hp;
            if (pad_ok) {
                GemmProb p = pcm_gemm_prob(dav, 1, lay.AVW, htop, Hp, 1, dcw, H,
                                           lay.AVW, H, R);
                p.colsum1 = dcw + (size_t)lay.AVW * H;  // [AVW] control biases
                gb.p[gb.count++] = p;
            } else {
                if (grads->ctrl_w) {
                    GemmProb p = pcm_gemm_prob(dav, 1, lay.AVW, htop, Hp, 1,
                                               grads->ctrl_w, H, 3, H, R);
                    p.accumulate = cacc;
                    gb.p[gb.count++] = p;
                }
                if (grads->push_w) {
                    GemmProb p = pcm_gemm_prob(dav + 3, 1, lay.AVW, htop, Hp, 1,
                                               grads->push_w, H, lay.W, H, R);
                    p.accumulate = cacc;
                    gb.p[gb.count++] = p;
                }
            }
        }
        // token path: dtokg (+)= onehot^T . dGi0; at the end dW_ih0[:, :E], dEmb
        if (tok_path && !merged) {
            GemmProb p = pcm_gemm_prob(onehot, 1, lay.V, G0, 4LL * H, 1, dtokg,
                                       3LL * H, lay.V, 3 * H, R);
            p.accumulate = first ? 0 : 1;
            gb.p[gb.count++] = p;
        }
        if (gb.count > 0 && pcm_gemm_takes_colsum(gb, true)) {
            for (int i = 0; i < gb.count; ++i) {
                if (gb.p[i].colsum1 == grads->b_ih[0] && grads->b_ih[0])
                    bih_done[0] = true;
                if (gb.p[i].colsum1 == dcw + (size_t)lay.AVW * H) cb_done = true;
            }
        } else {
            for (int i = 0; i < gb.count; ++i)
                gb.p[i].colsum1 = gb.p[i].colsum2 = nullptr;
        }
        r = flush();
        if (r) return r;
        if (cb_done) {
            const float *dcb = dcw + (size_t)lay.AVW * H;
            if (grads->ctrl_b)
                pcm_launch_add_block(dcb, lay.AVW, grads->ctrl_b, 3, 1, 3, cacc,
                                     gst);
            if (grads->push_b)
                pcm_launch_add_block(dcb + 3, lay.AVW, grads->push_b, lay.W, 1,
                                     lay.W, cacc, gst);
        }
        if (pad_ok && lay.use_stack) {
            if (grads->w_ih[0])
                pcm_launch_add_block(dwx, merged ? 128 : lay.Wp,
                                     grads->w_ih[0] + lay.E, ld0, 3 * H, lay.W,
                                     cacc, gst);
            if (grads->ctrl_w)
                pcm_launch_add_block(dcw, H, grads->ctrl_w, H, 3, H, cacc, gst);
            if (grads->push_w)
                pcm_launch_add_block(dcw + (size_t)3 * H, H, grads->push_w, H,
                                     lay.W, H, cacc, gst);
        }
        if (tok_path && last) {
            GemmBatch g2;
            g2.count = 0;
            // dtokg[v][c]: its own [V][3H] buffer, or (merged) column 64 + v of
            // row c of dwx
            const float *tg = merged ? dwx + 64 : dtokg;
            const long long tg_v = merged ? 1 : 3LL * H;    // stride over v
            const long long tg_c = merged ? 128 : 1;        // stride over c
            if (grads->w_ih[0]) {
                GemmProb p = pcm_gemm_prob(tg, tg_c, tg_v, weights->emb,
                                           lay.E, 1, grads->w_ih[0], ld0,
                                           3 * H, lay.E, lay.V);
                p.accumulate = acc;
                g2.p[g2.count++] = p;
            }
            if (grads->emb) {
                GemmProb p = pcm_gemm_prob(tg, tg_v, tg_c,
                                           weights->w_ih[0], ld0, 1,
                                           grads->emb, lay.E, lay.V, lay.E,
                                           3 * H);
                p.accumulate = acc;
                g2.p[g2.count++] = p;
            }
            r = pcm_gemm_launch(g2, gs2, gsb, gst);
            if (r) return r;
        }
        // biases: column sums of the gate gradients
        for (int l = 0; l < L; ++l) {
            const float *G = sv + lay.off_gates[l] + r0 * 4 * H;
            if (grads->b_ih[l] && !bih_done[l]) {
                r = pcm_colsum_launch(G, 4LL * H, R, 3 * H, grads->b_ih[l],
                                      nullptr, cacc, gs2, gsb, gst);
                if (r) return r;
            }
            if (grads->b_hh[l] && !bhh_done[l]) {
                r = pcm_colsum_launch(G, 4LL * H, R, 2 * H, grads->b_hh[l],
                                      nullptr, cacc, gs2, gsb, gst);
                if (r) return r;
                r = pcm_colsum_launch(G + 3 * H, 4LL * H, R, H,
                                      grads->b_hh[l] + 2 * H, nullptr, cacc, gs2,
                                      gsb, gst);
                if (r) return r;
            }
        }
        if (lay.use_stack && !cb_done) {
            const float *dav = sv + lay.off_av + r0 * lay.AVW;
            if (grads->ctrl_b) {
                r = pcm_colsum_launch(dav, lay.AVW, R, 3, grads->ctrl_b, nullptr,
                                      cacc, gs2, gsb, gst);
                if (r) return r;
            }
            if (grads->push_b) {
                r = pcm_colsum_launch(dav + 3, lay.AVW, R, lay.W, grads->push_b,
                                      nullptr, cacc, gs2, gsb, gst);
                if (r) return r;
            }
        }
        return PCM_OK;
    };
    // one chain over the whole batch, or two half-batch chains on two streams
    // (see chain_stream); everything below the loop runs on `st` after the join
    int split = fused_bwd ? chain_split(lay) : 0;
    SideStream *cs = split ? chain_stream() : nullptr;
    if (!cs) split = 0;
    const int nch = split ? 2 : 1;
    // hand steps [t_lo, pending_hi) to the gradient stream
    auto flush_param_grads = [&](int t_lo) -> int {
        if (t_lo >= pending_hi) return PCM_OK;
        if (side) stream_wait(side, st, gst);
        // the second chain's rows of these steps are operands too (gst is the
        // caller's stream when there is no side stream)
        if (split) stream_wait(cs, cs->stream, gst);
        int r = param_grads(t_lo, pending_hi, first_chunk, t_lo == 0);
        first_chunk = false;
        pending_hi = t_lo;
        return r;
    };

    if (split) stream_wait(cs, st, cs->stream);   // fork
    auto total_rows = [&](int t) { return active_rows ? active_rows[t] : B; };
    int cur_c[2] = {cur, cur};
    for (int t = T - 1; t >= 0; --t) {
      for (int c = 0; c < nch; ++c) {
        hipStream_t cst = c == 0 ? st : cs->stream;
        // rows [b0, b0 + Bt) still running at step t in this chain
        int b0, Bt, b0n, Btn = 0, b0p, Btp = 0;
        chain_rows(c, split, total_rows(t), &b0, &Bt);
        if (t + 1 < T) chain_rows(c, split, total_rows(t + 1), &b0n, &Btn);
        if (t > 0) chain_rows(c, split, total_rows(t - 1), &b0p, &Btp);
        const size_t oH = (size_t)b0 * Hp, oG = (size_t)b0 * 4 * H;
        const int32_t *len = lengths ? lengths + b0 : nullptr;
        int &cur = cur_c[c];
        int dtop_parts = 1;
        bool bulk_done = false;
        for (int l = L - 1; l >= 0 && Bt > 0; --l) {
            float *gates = sv + lay.off_gates[l] + (size_t)t * B * 4 * H + oG;
            float *dhl = dh + (size_t)l * B * Hp + oH;
            // gate derivatives: the top layer (it also receives d_out and
            // the stack-control path of step t+1) and every layer on the
            // generic path take them in their own launch; on the fused path
            // the lower layers' are taken in the epilogue of the cell above
            // (with a stack, the top layer's are taken by the stack adjoint of
            // step t+1, which produces their last operand; only the first
            // step walked has no such launch above it)
            const bool by_stack =
                l == L - 1 && lay.use_stack && t + 1 < T && Btn > 0;
            if ((l == L - 1 || !fused_bwd) && !by_stack) {
                GateBwdArgs g;
                memset(&g, 0, sizeof(g));
                g.dh = dhl;
                g.dout = (l == L - 1 && d_out)
                             ? d_out + ((size_t)t * B + b0) * d_out_row_stride
                             : nullptr;
                g.ld_dout = d_out_row_stride;
                g.gates = gates;
                g.hprev = sv + lay.off_hs[l] + (size_t)t * lay.hs_step() + oH;
                g.lengths = len; g.B = Bt; g.H = H; g.Hp = Hp; g.t = t;
                pcm_probe_begin(PCM_K_GATE_BWD);
                pcm_launch_gate_bwd(g, cst);
                pcm_probe_end(PCM_K_GATE_BWD);
            }

            if (fused_bwd) {
                CellBwdArgs cb;
                cb.dgates = gates;
                cb.wpackT = scf + sc.off_wpackT[l];
                cb.dh = dhl;
                cb.dx = l > 0 ? dh + (size_t)(l - 1) * B * Hp + oH
                              : dtop + (size_t)b0 * lay.Wp;
                cb.ldx = l > 0 ? Hp : lay.Wp;
                cb.dx_accumulate = l > 0 ? 1 : 0;
                cb.dx_cols = l > 0 ? H : lay.Wp;
                cb.B = Bt; cb.H = H; cb.Hp = Hp;
                cb.NT = H + lay.Kxp[l];
                cb.t = t; cb.layer = l; cb.b_off = b0;
                cb.drop_p = (l > 0 && drop) ? dropout_p : 0.f;
                cb.inv_keep = cb.drop_p > 0.f ? 1.f / (1.f - dropout_p) : 1.f;
                cb.seed = seed;
                cb.gates_low = nullptr; cb.hprev_low = nullptr;
                cb.lengths = len;
                cb.xparts = 1; cb.xpart_stride = (long long)B * lay.Wp;
                if (l > 0) {
                    cb.gates_low = sv + lay.off_gates[l - 1] +
                                   (size_t)t * B * 4 * H + oG;
                    cb.hprev_low = sv + lay.off_hs[l - 1] +
                                   (size_t)t * lay.hs_step() + oH;
                }
                if (l == 0) dtop_parts = pcm_cell_bwd_xparts(cb);
                // the bulk of this step's stack adjoint rides in the top
                // layer's launch (it needs nothing the cells produce)
                cb.rider_rows = 0; cb.ncell = 0;
                memset(&cb.rider, 0, sizeof(cb.rider));
                if ((l == L - 1 || l == 0) && L >= 2 && lay.use_stack &&
                    stack_split_on && lay.W <= 61 && lay.AVW <= 64 &&
                    (bulk_done || pcm_cell_bwd_takes_rider(cb, lay.D, lay.W))) {
                    // (both launches of a step take the narrow fp32 tile or
                    // neither does: the first decides)
                    const int n_top = (int)((long long)Bt * rider_top_percent / 100);
                    const int r0 = l == 0 ? n_top : 0;
                    const int rn = l == 0 ? Bt - n_top : n_top;
                    bulk_done = true;
                    if (rn > 0) {
                        const size_t oS = (size_t)(b0 + r0) * lay.DW;
                        cb.rider_rows = rn;
                        cb.rider.dS_next = dS[cur] + oS;
                        cb.rider.dS_prev = dS[cur ^ 1] + oS;
                        cb.rider.S_prev = sv + lay.s_index(t) + oS;
                        cb.rider.av =
                            sv + lay.av_index(t) + (size_t)(b0 + r0) * lay.AVW;
                        cb.rider.psum = psum + (size_t)(b0 + r0) * 4;
                        cb.rider.lengths = len ? len + r0 : nullptr;
                        cb.rider.B = rn; cb.rider.D = lay.D; cb.rider.W = lay.W;
                        cb.rider.AVW = lay.AVW; cb.rider.t = t;
                    }
                }
                pcm_probe_begin(PCM_K_CELL_BWD, l);
                pcm_launch_cell_bwd(cb, cst);
                pcm_probe_end(PCM_K_CELL_BWD, l);
                continue;
            }
            // generic path (H not a multiple of 64; always one chain):
            // dh[l] += dGh . W_hh  (dGh = cols [0,2H) and [3H,4H) of gates)
            GemmBatch gb;
            gb.count = 0;
            const bool seg_ok = ((2 * H) % 16) == 0;
            GemmProb p1 = pcm_gemm_prob(gates, 4LL * H, 1, weights->w_hh[l], H,
                                        1, dhl, Hp, Bt, H, seg_ok ? 3 * H : 2 * H);
            p1.accumulate = 1;
            if (seg_ok) {
                p1.K1 = 2 * H;
                p1.A2 = gates + 3 * H;
                p1.B2 = weights->w_hh[l] + (size_t)2 * H * H;
            }
            gb.p[gb.count++] = p1;
            // input-side product
            const bool use_tmp = (l > 0 && drop);
            if (l > 0) {
                GemmProb p2 = pcm_gemm_prob(
                    gates, 4LL * H, 1, weights->w_ih[l], H, 1,
                    use_tmp ? tmp : dh + (size_t)(l - 1) * B * Hp, Hp, Bt, H,
                    3 * H);
                p2.accumulate = use_tmp ? 0 : 1;
                gb.p[gb.count++] = p2;
            } else if (lay.use_stack) {
                GemmProb p2 = pcm_gemm_prob(gates, 4LL * H, 1,
                                            weights->w_ih[0] + lay.E, ld0, 1,
                                            dtop, lay.Wp, Bt, lay.W, 3 * H);
                gb.p[gb.count++] = p2;
            }
            rc = pcm_gemm_launch(gb, gs, gsb, cst);
            if (rc) return rc;
            if (!seg_ok) {
                GemmBatch g2;
                g2.count = 1;
                GemmProb p3 = pcm_gemm_prob(gates + 3 * H, 4LL * H, 1,
                                            weights->w_hh[l] + (size_t)2 * H * H,
                                            H, 1, dhl, Hp, Bt, H, H);
                p3.accumulate = 1;
                g2.p[0] = p3;
                rc = pcm_gemm_launch(g2, gs, gsb, cst);
                if (rc) return rc;
            }
            if (use_tmp)
                pcm_launch_dropout_add(dh + (size_t)(l - 1) * B * Hp, Hp, tmp, Hp,
                                       Bt, H, seed, l, t, dropout_p, cst);
        }
        if (lay.use_stack && Bt > 0) {
            const size_t oS = (size_t)b0 * lay.DW;
            StackBwdArgs s;
            s.dS_next = dS[cur] + oS; s.dS_prev = dS[cur ^ 1] + oS;
            s.dtop = dtop + (size_t)b0 * lay.Wp;
            s.dtop_parts = dtop_parts;
            s.dtop_stride = (long long)B * lay.Wp;
            s.S_prev = sv + lay.s_index(t) + oS;
            s.av = sv + lay.av_index(t) + (size_t)b0 * lay.AVW;
            s.lengths = len;
            s.B = Bt; s.H = H; s.Hp = Hp; s.D = lay.D; s.W = lay.W;
            s.Wp = lay.Wp; s.AVW = lay.AVW; s.t = t;
            // ... followed by the top layer's gate gradient of step t-1 (or,
            // at t == 0, the control path into the initial hidden state)
            s.gate_mode = t > 0 ? 1 : 2;
            s.gate_B = t > 0 ? Btp : Bt;
            s.g_dh = dh + (size_t)(L - 1) * B * Hp + oH;
            s.g_dout = (t > 0 && d_out)
                           ? d_out + ((size_t)(t - 1) * B + b0) * d_out_row_stride
                           : nullptr;
            s.g_ld_dout = d_out_row_stride;
            s.g_gates = t > 0 ? sv + lay.off_gates[L - 1] +
                                    (size_t)(t - 1) * B * 4 * H + oG
                              : nullptr;
            s.g_hprev = t > 0 ? sv + lay.off_hs[L - 1] +
                                    (size_t)(t - 1) * lay.hs_step() + oH
                              : nullptr;
            s.ctrl_w = weights->ctrl_w; s.push_w = weights->push_w;
            s.psum = bulk_done ? psum + (size_t)b0 * 4 : nullptr;
            pcm_probe_begin(PCM_K_STACK_BWD);
            rc = bulk_done ? pcm_launch_stack_bwd_finish(s, cst)
                           : pcm_launch_stack_bwd(s, cst);
            pcm_probe_end(PCM_K_STACK_BWD);
            if (rc) return rc;
            cur ^= 1;
        }
      }
        // the control gradients of step t feed the gate kernel of step t-1
        // only as a READ, so rows >= t are final here
        if (t > 0 && next_bound < nbounds && T - t == bounds[next_bound]) {
            ++next_bound;
            rc = flush_param_grads(t);
            if (rc) return rc;
        }
    }
    if (split) stream_wait(cs, cs->stream, st);   // join
    if (d_h0)
        for (int l = 0; l < L; ++l)
            pcm_launch_copy_rows(dh + (size_t)l * B * Hp, Hp,
                                 d_h0 + (size_t)l * B * H, H, B, H, st);
    if (d_stack0 && lay.use_stack)
        for (int c = 0; c < nch; ++c) {
            // (each chain's ping-pong slot flips on the steps IT walked)
            const int b0 = c == 0 ? 0 : split, n = split ? (c == 0 ? split : B - split) : B;
            pcm_launch_copy_rows(dS[cur_c[c]] + (size_t)b0 * lay.DW, lay.DW,
                                 d_stack0 + (size_t)b0 * lay.DW, lay.DW, n,
                                 lay.DW, st);
        }

    // the remaining chunk(s) and the join with the side stream
    rc = flush_param_grads(0);
    if (rc) return rc;
    if (side) stream_wait(side, gst, st);
    return pcm_check_launch("stackgru_backward");
}

// ---- dense layers -------------------------------------------------------------------
int pcm_linear_forward(const float *x, int64_t ldx, const float *w,
                       const float *b, float *y, int64_t ldy, int32_t m,
                       int32_t n, int32_t k, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(x && w && y && m > 0 && n > 0 && k > 0, PCM_EINVAL,
                "linear_forward: bad args");
    GemmBatch gb;
    gb.count = 1;
    gb.p[0] = pcm_gemm_prob(x, ldx, 1, w, 1, k, y, ldy, m, n, k);
    gb.p[0].bias = b;
    return pcm_gemm_launch(gb, nullptr, 0, (hipStream_t)stream);
}

size_t pcm_linear_scratch_bytes(int32_t m, int32_t n, int32_t k) {
    (void)k;
    size_t a = GEMM_SCRATCH_FLOATS * sizeof(float);
    size_t b = pcm_colsum_scratch_bytes(m, n);
    return a > b ? a : b;
}

int pcm_linear_backward(const float *x, int64_t ldx, const float *w,
                        const float *dy, const float *dy_scale, int64_t lddy,
                        float *dx, int64_t lddx,
                        float *dw, float *db, int32_t m, int32_t n, int32_t k,
                        int32_t accumulate, void *scratch,
                        size_t scratch_bytes, void *stream) {
    PCM_API_GUARD;
    PCM_REQUIRE(dy && m > 0 && n > 0 && k > 0, PCM_EINVAL,
                "linear_backward: bad args");
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (dx) {
        PCM_REQUIRE(w, PCM_EINVAL, "linear_backward: dx needs w");
        GemmBatch gb;
        gb.count = 1;
        gb.p[0] = pcm_gemm_prob(dy, lddy, 1, w, k, 1, dx, lddx, m, k, n);
        gb.p[0].alpha_ptr = dy_scale;
        rc = pcm_gemm_launch(gb, scratch, scratch_bytes, st);
        if (rc) return rc;
    }
    if (dw) {
        PCM_REQUIRE(x, PCM_EINVAL, "linear_backward: dw needs x");
        GemmBatch gb;
        gb.count = 1;
        gb.p[0] = pcm_gemm_prob(dy, 1, lddy, x, ldx, 1, dw, k, n, k, m);
        gb.p[0].accumulate = accumulate ? 1 : 0;
        gb.p[0].alpha_ptr = dy_scale;
        rc = pcm_gemm_launch(gb, scratch, scratch_bytes, st);
        if (rc) return rc;
    }
    if (db) {
        rc = pcm_colsum_launch(dy, lddy, m, n, db, dy_scale, accumulate ? 1 : 0,
                               scratch, scratch_bytes, st);
        if (rc) return rc;
    }
    return PCM_OK;
}

// ---- generation ----------------------------------------------------------------------
struct GenLayout {
    SGLayout lay;
    size_t off_h[2][PCM_MAX_LAYERS];
    size_t off_S[2];
    size_t off_av, off_xtop, off_tokg, off_logits, off_tok, off_cspart;
    size_t off_wpack[PCM_MAX_LAYERS], off_bpack[PCM_MAX_LAYERS];
    size_t total;
};

static int gen_layout(const pcm_dims *dims, GenLayout *g) {
    int rc = pcm_sg_layout(dims, 1, 0, 0, &g->lay);
    if (rc) return rc;
    const SGLayout &lay = g->lay;
    size_t off = 0;
    for (int s = 0; s < 2; ++s)
        for (int l = 0; l < lay.L; ++l)
            g->off_h[s][l] = take(off, (size_t)lay.B * lay.Hp);
    for (int s = 0; s < 2; ++s)
        g->off_S[s] = take(off, (size_t)lay.B * (lay.use_stack ? lay.DW : 1));
    g->off_av = take(off, (size_t)lay.B * (lay.use_stack ? lay.AVW : 1));
    g->off_xtop = take(off, (size_t)lay.B * (lay.use_stack ? lay.Wp : 1));
    g->off_tokg = take(off, (size_t)lay.V * 3 * lay.Hn);
    g->off_logits = take(off, (size_t)lay.B * lay.V);
    g->off_tok = take(off, (size_t)lay.B * 2);  // int64[B]
    g->off_cspart =
        take(off, lay.use_stack ? (size_t)lay.B * (lay.Hn / 16) * 64 : 0);
    for (int l = 0; l < lay.L; ++l) {
        g->off_wpack[l] = take(off, (size_t)3 * lay.Hn * lay.Kp[l]);
        g->off_bpack[l] = take(off, (size_t)4 * lay.Hn);
    }
    g->total = off;
    return PCM_OK;
}

size_t pcm_generate_workspace_bytes(const pcm_dims *dims) {
    GenLayout g;
    if (gen_layout(dims, &g)) return 0;
    return g.total * sizeof(float);
}

int pcm_stackgru_generate(const pcm_dims *dims,
                          const pcm_stackgru_params *weights,
                          const float *out_w, const float *out_b,
                          const float *h0, int64_t h0_layer_stride,
                          const int64_t *prime, int32_t prime_len,
                          int32_t generate_len, int32_t search,
                          float temperature, uint64_t seed,
                          int64_t *tokens_out, float *logits_out,
                          void *workspace, size_t workspace_bytes,
                          int32_t *status, void *stream) {
    PCM_API_GUARD;
    GenLayout g;
    int rc = gen_layout(dims, &g);
    if (rc) return rc;
    SGLayout &lay = g.lay;
    rc = check_params(lay, weights, "stackgru_generate");
    if (rc) return rc;
    PCM_REQUIRE(out_w && prime && tokens_out && workspace, PCM_EINVAL,
                "stackgru_generate: null pointer");
    PCM_REQUIRE(prime_len >= 1 && generate_len >= 0, PCM_EINVAL,
                "stackgru_generate: prime_len=%d generate_len=%d", prime_len,
                generate_len);
    PCM_REQUIRE(search == PCM_SEARCH_GREEDY || search == PCM_SEARCH_SAMPLING,
                PCM_EINVAL, "stackgru_generate: unknown search %d", search);
    PCM_REQUIRE(search != PCM_SEARCH_SAMPLING || temperature > 0.f, PCM_EINVAL,
                "stackgru_generate: temperature must be positive");
    PCM_REQUIRE(workspace_bytes >= g.total * sizeof(float), PCM_ENOMEM,
                "stackgru_generate: workspace %zu < %zu bytes", workspace_bytes,
                g.total * sizeof(float));
    if (lay.use_stack) {
        rc = pcm_stack_fwd_check(lay.D, lay.W, lay.H, lay.AVW);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const int B = lay.B, H = lay.H, Hp = lay.Hp, L = lay.L, V = lay.V;
    const long long out_stride = (long long)prime_len + generate_len;

    // weights, token table: reuse the forward preparation on a layout whose
    // offsets point into this workspace
    SGLayout prep = lay;
    prep.off_tokg = g.off_tokg;
    for (int l = 0; l < L; ++l) {
        prep.off_wpack[l] = g.off_wpack[l];
        prep.off_bpack[l] = g.off_bpack[l];
    }
    rc = enqueue_prepare(prep, weights, ws, st);
    if (rc) return rc;
    float *wpack[PCM_MAX_LAYERS], *bpack[PCM_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        wpack[l] = ws + g.off_wpack[l];
        bpack[l] = ws + g.off_bpack[l];
    }
    // initial state
    if (Hp != H)
        for (int s = 0; s < 2; ++s)
            for (int l = 0; l < L; ++l)
                pcm_launch_fill(ws + g.off_h[s][l], 0.f, (long long)B * Hp, st);
    for (int l = 0; l < L; ++l) {
        if (h0)
            pcm_launch_copy_rows(h0 + (long long)l * h0_layer_stride, H,
                                 ws + g.off_h[0][l], Hp, B, H, st);
        else
            pcm_launch_fill(ws + g.off_h[0][l], 0.f, (long long)B * Hp, st);
    }
    if (lay.use_stack)
        pcm_launch_fill(ws + g.off_S[0], 0.f, (long long)B * lay.DW, st);
    int64_t *cur_tok = (int64_t *)(ws + g.off_tok);
    for (int p = 0; p < prime_len; ++p)
        pcm_launch_broadcast_token(prime + p, B, nullptr, tokens_out, out_stride,
                                   p, st);

    const int total_steps = (prime_len - 1) + generate_len;
    int slot = 0;
    // partial stack-control sums from a step's top cell to the next step's
    // stack kernel, as in pcm_stackgru_forward
    static const bool ctrl_partials = [] {
        const char *e = getenv("PCM_CTRL_PARTIALS");
        return !e || atoi(e) != 0;
    }();
    float *cs_part = (ctrl_partials && lay.use_stack) ? ws + g.off_cspart
                                                      : nullptr;
    int have_part = 0;
    for (int s = 0; s < total_steps; ++s) {
        const bool priming = s < prime_len - 1;
        if (s < prime_len)
            pcm_launch_broadcast_token(prime + s, B, cur_tok, nullptr, 0, 0, st);
        StepBufs sb;
        memset(&sb, 0, sizeof(sb));
        for (int l = 0; l < L; ++l) {
            sb.h_prev[l] = ws + g.off_h[slot][l];
            sb.h_new[l] = ws + g.off_h[slot ^ 1][l];
        }
        if (lay.use_stack) {
            sb.S_prev = ws + g.off_S[slot];
            sb.S_new = ws + g.off_S[slot ^ 1];
            sb.av = ws + g.off_av;
            sb.xtop = ws + g.off_xtop;
        }
        sb.tok = cur_tok;
        rc = enqueue_step(lay, weights, ws + g.off_tokg, wpack, bpack, sb,
                          nullptr, s, 0.f, 0, status, st, -1, 0, cs_part,
                          &have_part);
        if (rc) return rc;
        slot ^= 1;
        if (!priming) {
            const int gi = s - (prime_len - 1);
            // output layer + search step, fused when the shape allows
            static const bool fuse = [] {
                const char *e = getenv("PCM_FUSED_CHOOSE");
                return !e || atoi(e) != 0;
            }();
            if (fuse &&
                pcm_launch_logits_choose(
                    ws + g.off_h[slot][L - 1], Hp, out_w, out_b, B, V, H, search,
                    temperature, seed, gi, cur_tok, tokens_out, out_stride,
                    prime_len + gi,
                    logits_out ? logits_out + (size_t)gi * B * V : nullptr, st))
                continue;
            float *logits = logits_out ? logits_out + (size_t)gi * B * V
                                       : ws + g.off_logits;
            rc = pcm_linear_forward(ws + g.off_h[slot][L - 1], Hp, out_w, out_b,
                                    logits, V, B, V, H, st);
            if (rc) return rc;
            pcm_launch_choose(logits, B, V, search, temperature, seed, gi,
                              cur_tok, tokens_out, out_stride, prime_len + gi,
                              st);
        }
    }
    return pcm_check_launch("stackgru_generate");
}

}  // extern "C"

// ---- beam-search generation ---------------------------------------------------------
// The beam branch of generate_from_latent keeps beam_width copies of the
// recurrent state (vae.py:442-453); here they are the rows k * B + b of ONE
// batch of beam_width * B sequences, so a beam step is the ordinary fused step
// over that batch, one output-layer GEMM and one beam_step_kernel launch.
struct BeamLayout {
    GenLayout g;           // state of the beam_width * B rows
    size_t off_scores;     // double[B][width]
    size_t off_hist_tok;   // int32[G][B][width]
    size_t off_hist_par;   // int32[G][B][width]
    size_t total;          // floats
};

static int beam_layout(const pcm_dims *dims, int width, int generate_len,
                       BeamLayout *o) {
    PCM_REQUIRE(dims, PCM_EINVAL, "dims is null");
    PCM_REQUIRE(width >= 1 && width <= 16, PCM_EINVAL,
                "beam_width=%d outside 1..16", width);
    PCM_REQUIRE(generate_len >= 0, PCM_EINVAL, "generate_len=%d", generate_len);
    PCM_REQUIRE((long long)dims->batch * width < (1ll << 24), PCM_EINVAL,
                "batch x beam_width too large");
    pcm_dims big = *dims;
    big.batch = dims->batch * width;
    int rc = gen_layout(&big, &o->g);
    if (rc) return rc;
    size_t off = o->g.total;
    const size_t bw = (size_t)dims->batch * width;
    o->off_scores = take(off, 2 * bw);
    o->off_hist_tok = take(off, bw * (generate_len > 0 ? generate_len : 1));
    o->off_hist_par = take(off, bw * (generate_len > 0 ? generate_len : 1));
    o->total = off;
    return PCM_OK;
}

extern "C" {

size_t pcm_generate_beam_workspace_bytes(const pcm_dims *dims,
                                         int32_t beam_width,
                                         int32_t generate_len) {
    BeamLayout bl;
    if (beam_layout(dims, beam_width, generate_len, &bl)) return 0;
    return bl.total * sizeof(float);
}

int pcm_stackgru_generate_beam(const pcm_dims *dims,
                               const pcm_stackgru_params *weights,
                               const float *out_w, const float *out_b,
                               const float *h0, int64_t h0_layer_stride,
                               const int64_t *prime, int32_t prime_len,
                               int32_t generate_len, int32_t beam_width,
                               int32_t top_tokens, float temperature,
                               int64_t *tokens_out, int64_t *beam_tokens_out,
                               double *beam_scores_out, void *workspace,
                               size_t workspace_bytes, int32_t *status,
                               void *stream) {
    PCM_API_GUARD;
    BeamLayout bl;
    int rc = beam_layout(dims, beam_width, generate_len, &bl);
    if (rc) return rc;
    GenLayout &g = bl.g;
    SGLayout &lay = g.lay;   // lay.B = beam_width * B
    rc = check_params(lay, weights, "stackgru_generate_beam");
    if (rc) return rc;
    PCM_REQUIRE(out_w && prime && tokens_out && workspace, PCM_EINVAL,
                "stackgru_generate_beam: null pointer");
    PCM_REQUIRE(prime_len >= 1 && generate_len >= 1, PCM_EINVAL,
                "stackgru_generate_beam: prime_len=%d generate_len=%d",
                prime_len, generate_len);
    PCM_REQUIRE(top_tokens >= 1 && temperature > 0.f, PCM_EINVAL,
                "stackgru_generate_beam: top_tokens=%d temperature=%f",
                top_tokens, temperature);
    PCM_REQUIRE(workspace_bytes >= bl.total * sizeof(float), PCM_ENOMEM,
                "stackgru_generate_beam: workspace %zu < %zu bytes",
                workspace_bytes, bl.total * sizeof(float));
    PCM_REQUIRE(((uintptr_t)workspace & 15) == 0, PCM_EINVAL,
                "stackgru_generate_beam: workspace must be 16-byte aligned");
    if (lay.use_stack) {
        rc = pcm_stack_fwd_check(lay.D, lay.W, lay.H, lay.AVW);
        if (rc) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    const int B = dims->batch, W = beam_width, BW = lay.B;
    const int H = lay.H, Hp = lay.Hp, L = lay.L, V = lay.V;
    const int top = top_tokens < V ? top_tokens : V;

    SGLayout prep = lay;
    prep.off_tokg = g.off_tokg;
    for (int l = 0; l < L; ++l) {
        prep.off_wpack[l] = g.off_wpack[l];
        prep.off_bpack[l] = g.off_bpack[l];
    }
    rc = enqueue_prepare(prep, weights, ws, st);
    if (rc) return rc;
    float *wpack[PCM_MAX_LAYERS], *bpack[PCM_MAX_LAYERS];
    for (int l = 0; l < L; ++l) {
        wpack[l] = ws + g.off_wpack[l];
        bpack[l] = ws + g.off_bpack[l];
    }
    // every slot starts from the same state (vae.py:446-453); the prime is
    // run on all of them, which is what priming once and copying gives
    if (Hp != H)
        for (int s = 0; s < 2; ++s)
            for (int l = 0; l < L; ++l)
                pcm_launch_fill(ws + g.off_h[s][l], 0.f, (long long)BW * Hp, st);
    for (int l = 0; l < L; ++l)
        for (int k = 0; k < W; ++k) {
            float *dst = ws + g.off_h[0][l] + (size_t)k * B * Hp;
            if (h0)
                pcm_launch_copy_rows(h0 + (long long)l * h0_layer_stride, H, dst,
                                     Hp, B, H, st);
            else
                pcm_launch_fill(dst, 0.f, (long long)B * Hp, st);
        }
    if (lay.use_stack)
        pcm_launch_fill(ws + g.off_S[0], 0.f, (long long)BW * lay.DW, st);
    int64_t *cur_tok = (int64_t *)(ws + g.off_tok);
    double *scores = (double *)(ws + bl.off_scores);
    int32_t *hist_tok = (int32_t *)(ws + bl.off_hist_tok);
    int32_t *hist_par = (int32_t *)(ws + bl.off_hist_par);

    // slots: state copies still fed (the reference's zip() over tokens and
    // states shrinks them when fewer beams than slots exist, vae.py:471-483);
    // beams: entries of the beam list (search.py:186 zips them with the slots)
    int slots = W, beams = 1, slot = 0;
    const int total_steps = (prime_len - 1) + generate_len;
    for (int s = 0; s < total_steps; ++s) {
        const bool priming = s < prime_len - 1;
        if (s < prime_len)
            pcm_launch_broadcast_token(prime + s, BW, cur_tok, nullptr, 0, 0, st);
        StepBufs sb;
        memset(&sb, 0, sizeof(sb));
        for (int l = 0; l < L; ++l) {
            sb.h_prev[l] = ws + g.off_h[slot][l];
            sb.h_new[l] = ws + g.off_h[slot ^ 1][l];
        }
        if (lay.use_stack) {
            sb.S_prev = ws + g.off_S[slot];
            sb.S_new = ws + g.off_S[slot ^ 1];
            sb.av = ws + g.off_av;
            sb.xtop = ws + g.off_xtop;
        }
        sb.tok = cur_tok;
        rc = enqueue_step(lay, weights, ws + g.off_tokg, wpack, bpack, sb,
                          nullptr, s, 0.f, 0, status, st,
                          priming ? BW : slots * B);
        if (rc) return rc;
        slot ^= 1;
        if (priming) continue;
        const int gi = s - (prime_len - 1);
        float *logits = ws + g.off_logits;   // [slots * B, V]
        rc = pcm_linear_forward(ws + g.off_h[slot][L - 1], Hp, out_w, out_b,
                                logits, V, slots * B, V, H, st);
        if (rc) return rc;
        BeamStepArgs a;
        a.logits = logits; a.scores = scores;
        a.hist_tok = hist_tok; a.hist_parent = hist_par; a.next_tok = cur_tok;
        a.B = B; a.V = V; a.width = W;
        a.used = slots < beams ? slots : beams;
        a.top = top;
        a.nb_next = a.used * top < W ? a.used * top : W;
        a.slots_next = slots < a.nb_next ? slots : a.nb_next;
        a.step = gi; a.temperature = temperature;
        rc = pcm_launch_beam_step(a, st);
        if (rc) return rc;
        beams = a.nb_next;
        slots = a.slots_next;
    }
    pcm_launch_beam_backtrack(hist_tok, hist_par, scores, B, W, beams,
                              generate_len, tokens_out, beam_tokens_out,
                              beam_scores_out, st);
    return pcm_check_launch("stackgru_generate_beam");
}

}  // extern "C"
