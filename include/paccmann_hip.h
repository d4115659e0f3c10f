/*
 * paccmann_hip.h -- C ABI of libpaccmann_hip.so: the stack-augmented GRU VAE hot
 * path of PaccMann/paccmann_chemistry as hand-written gfx950 (MI355X) kernels.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own; the entry
 * points below are what a binding for this path replaces, each citing the
 * reference code it stands in for (path:line under the reference checkout).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to a dense row-major array unless a
 *     stride is given; floats are IEEE fp32, token ids / targets int64,
 *     lengths int32;
 *   - `stream` is a hipStream_t (NULL = the default stream); every call only
 *     enqueues work on that stream and never synchronises, allocates or frees
 *     device memory: the caller owns every buffer, including the `saved` and
 *     `scratch` workspaces whose sizes the *_bytes() helpers return;
 *   - return value 0 = success; otherwise a negative PCM_E* code and
 *     pcm_last_error() (thread-local) describes the failure;
 *   - calls that share a workspace must not overlap in time.
 */
#ifndef PACCMANN_HIP_H
#define PACCMANN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCM_MAX_LAYERS 8

#define PCM_OK 0
#define PCM_EINVAL (-1)   /* bad argument (shape, null pointer, size)        */
#define PCM_ENOMEM (-2)   /* caller workspace too small                      */
#define PCM_EDEVICE (-3)  /* HIP runtime error (launch failed, no device)    */

/* bits of the int32 status word a forward call leaves on the device */
#define PCM_STATUS_BAD_TOKEN 1 /* a token id outside [0, vocab) was clamped */
#define PCM_STATUS_TIMEOUT 2   /* a hand-off inside the persistent forward kernel
                                  gave up (its blocks were not all resident):
                                  the call's outputs are not valid */

/* Model geometry of one StackGRU (stack_rnn.py:46-56). */
typedef struct pcm_dims {
    int32_t batch;       /* B  sequences in the batch                       */
    int32_t vocab;       /* V  rows of the embedding table                  */
    int32_t emb;         /* E  embedding width (= V for embedding: one_hot) */
    int32_t hidden;      /* H  rnn_cell_size                                */
    int32_t layers;      /* L  n_layers, 1..PCM_MAX_LAYERS                  */
    int32_t stack_depth; /* D                                               */
    int32_t stack_width; /* W                                               */
    int32_t use_stack;   /* 0: plain GRU (stack_rnn.py:64-66), 1: stack on  */
} pcm_dims;

/*
 * Parameters of one StackGRU in state_dict layout (stack_rnn.py:68-105).  The
 * same struct carries gradients (pcm_stackgru_backward); a NULL gradient
 * pointer means "not wanted" (e.g. the frozen one-hot embedding).
 */
typedef struct pcm_stackgru_params {
    float *ctrl_w;                 /* stack_controls_layer.weight [3,H]     */
    float *ctrl_b;                 /* stack_controls_layer.bias   [3]       */
    float *push_w;                 /* stack_input_layer.weight    [W,H]     */
    float *push_b;                 /* stack_input_layer.bias      [W]       */
    float *emb;                    /* embedding.weight            [V,E]     */
    float *w_ih[PCM_MAX_LAYERS];   /* gru.weight_ih_l{k} [3H, E+W | H]      */
    float *w_hh[PCM_MAX_LAYERS];   /* gru.weight_hh_l{k} [3H, H]            */
    float *b_ih[PCM_MAX_LAYERS];   /* gru.bias_ih_l{k}   [3H]               */
    float *b_hh[PCM_MAX_LAYERS];   /* gru.bias_hh_l{k}   [3H]               */
} pcm_stackgru_params;

/* Library / device ---------------------------------------------------------*/

/* ABI version of this header (bumped on any signature change). */
int pcm_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char *pcm_last_error(void);
/* 0 if a gfx950 device is current and usable, PCM_EDEVICE otherwise.  Also the
 * library's per-device set-up call: it resolves, for the current device, the
 * address of the static scratch array pcm_softmax_ce_static uses (a symbol
 * lookup that must not be issued inside a stream capture).  Call it once per
 * device, with that device current, before capturing a graph. */
int pcm_device_check(void);

/*
 * Mixed-precision switch (the reference's counterpart is running the model under
 * torch autocast, which it never does: models/training.py:132-157 trains in
 * fp32).  mode 0 (default): every product on fp32 MFMAs -- the only mode parity
 * with the reference is claimed on.  mode 1: the GRU cells' products of the
 * forward and reverse recurrence and the recurrent weight-gradient GEMMs round
 * their operands to bf16 (nearest even) and accumulate in fp32 on bf16 MFMAs;
 * hidden state, gates, stack, output layer, losses and optimizer stay fp32.
 * Process-wide; set it between calls, not while a call is being enqueued.
 * Returns PCM_EINVAL for any other mode.  pcm_get_matmul_precision reads it.
 */
int pcm_set_matmul_precision(int32_t mode);
int pcm_get_matmul_precision(void);

/*
 * Measurement aid (no counterpart in the reference): while enabled for one
 * kernel id, each launch of that kernel is made with a (start, stop) event
 * pair bound to the kernel's own dispatch (hipExtLaunchKernelGGL), so the
 * elapsed time is the kernel's begin -> end on the device -- the interval a
 * rocprofv3 kernel trace reports -- and no marker packet joins the stream.
 * pcm_probe_read synchronises on the recorded events and returns the summed
 * duration and launch count since enable.
 * Ids: 1 fused GRU cell, forward (11: layer 0 only, 12: layers >= 1 only);
 * 2 stack step, forward; 3 gate gradient; 4 stack step, backward (with the top
 * layer's fused gate gradient); 5 fused reverse GRU cell (51 / 52 as above);
 * 6 wide K-major weight-gradient GEMM launches (dW_hh, dW_ih); 0 disables.  At most max_launches
 * launches are recorded.  Not thread safe; one device per process.
 */
int pcm_probe_enable(int32_t kernel_id, int32_t max_launches);
/*
 * What the event pair itself adds to a reading: `launches` empty dependent
 * kernels timed twice on `stream` -- by two events around the whole run
 * (chain_us: the chain's wall clock per launch, the cost a launch has inside the
 * recurrence) and by a pair bound to each launch (probe_us).  Subtracting
 * probe_us - chain_us from a probed average gives the launch's share of the
 * chain's wall clock, which is what a rocprofv3 kernel trace shows for
 * back-to-back dependent launches.  Synchronises the stream.
 */
int pcm_probe_calibrate(int32_t launches, double *probe_us, double *chain_us,
                        void *stream);
int pcm_probe_read(double *total_ms, int64_t *launches, int64_t *dropped);

/* StackGRU over a whole sequence ------------------------------------------*/

/*
 * Bytes of the `saved` workspace pcm_stackgru_forward needs for T steps.
 * training != 0 keeps every step's hidden states, gates, stacks and stack
 * controls for pcm_stackgru_backward; training == 0 keeps the hidden states
 * only (the stack ping-pongs between two slots).
 */
size_t pcm_stackgru_saved_bytes(const pcm_dims *dims, int32_t steps,
                                int32_t training);
/*
 * Where the per-step top-layer outputs live inside `saved` after a forward:
 * out[t][b][j] = ((float*)saved)[*offset + (t*B + b) * *row_stride + j],
 * t < steps, j < H  (the `output` of stack_rnn.py:155 for every step).
 */
int pcm_stackgru_out_view(const pcm_dims *dims, int32_t steps,
                          int32_t training, size_t *offset_floats,
                          size_t *row_stride_floats);
/* Bytes of the `scratch` workspace pcm_stackgru_backward needs. */
size_t pcm_stackgru_scratch_bytes(const pcm_dims *dims, int32_t steps);

/*
 * T steps of StackGRU.forward (stack_rnn.py:131-156, _stack_update 188-199,
 * stack_augmentation 201-232, nn.GRU 99-105) -- the loops of
 * StackGRUEncoder._forward_pass_padded/_packed (vae.py:111-113, 165-183) and
 * StackGRUDecoder._forward_pass_padded/_packed (vae.py:337-341, 370-383).
 *
 *   tokens   int64[T,B]   input token per step (ids outside [0,V) are clamped
 *                         and PCM_STATUS_BAD_TOKEN is or-ed into *status)
 *   lengths  int32[B] or NULL.  NULL: padded mode, every row runs T steps.
 *            Otherwise packed semantics (utils/utils.py:226-243): row b is
 *            updated while t < lengths[b] and frozen afterwards; for
 *            t >= lengths[b] its out[t][b][:] repeats the frozen top-layer
 *            hidden state and carries no gradient.
 *   active_rows  HOST int32[T] or NULL (needs lengths): the rows still running
 *            at each step when the batch is sorted by decreasing length, i.e.
 *            PackedSequence.batch_sizes (utils/utils.py:215-243, where the
 *            reference slices hidden/stack to the active prefix every step).
 *            Steps then launch over rows [0, active_rows[t]) only; must be
 *            non-increasing and agree with lengths.  out[t][b][:] is ZERO for
 *            t >= lengths[b] in this mode.
 *   h0       fp32[L,B,H] initial hidden, element (l,b,j) at
 *            h0[l*h0_layer_stride + b*H + j] (stride 0 broadcasts one [B,H]
 *            to every layer, vae.py:564-565); NULL = zeros (stack_rnn.py:122)
 *   stack0   fp32[B,D,W] initial stack; NULL = zeros (stack_rnn.py:126)
 *   dropout_p / seed / training: inter-layer GRU dropout (stack_rnn.py:104),
 *            applied only when training != 0 and dropout_p > 0; the mask is a
 *            counter-based hash of (seed, layer, t, b, j), regenerated by the
 *            backward pass from the same seed.
 *   h_final  fp32[L,B,H] (nullable) hidden after each row's last step
 *   stack_final fp32[B,D,W] (nullable)
 *   status   int32[1] device word (nullable), or-ed with PCM_STATUS_* bits
 */
int pcm_stackgru_forward(const pcm_dims *dims,
                         const pcm_stackgru_params *weights,
                         const int64_t *tokens, int32_t steps,
                         const int32_t *lengths, const int32_t *active_rows,
                         const float *h0, int64_t h0_layer_stride,
                         const float *stack0, float dropout_p, uint64_t seed,
                         int32_t training, void *saved, size_t saved_bytes,
                         float *h_final, float *stack_final, int32_t *status,
                         void *stream);

/*
 * Reverse pass of pcm_stackgru_forward (what autograd replays for
 * training.py:153).  `saved` is the workspace of a training-mode forward with
 * the same dims/steps/tokens/lengths/active_rows/dropout_p/seed; it is consumed (the gate
 * slots are overwritten by gate gradients).
 *
 *   d_out        fp32[T,B,H] gradient of every step's top-layer output,
 *                row stride d_out_row_stride floats (nullable = zeros)
 *   d_h_final    fp32[L,B,H] (nullable), d_stack_final fp32[B,D,W] (nullable)
 *   d_h0         fp32[L,B,H] (nullable) receives dL/dh0
 *   d_stack0     fp32[B,D,W] (nullable) receives dL/dstack0
 *   grads        parameter gradients, overwritten (accumulate == 0) or added
 *                to (accumulate != 0); NULL members are skipped
 */
int pcm_stackgru_backward(const pcm_dims *dims,
                          const pcm_stackgru_params *weights,
                          const int64_t *tokens, int32_t steps,
                          const int32_t *lengths, const int32_t *active_rows,
                          float dropout_p, uint64_t seed, void *saved,
                          size_t saved_bytes,
                          const float *d_out, int64_t d_out_row_stride,
                          const float *d_h_final, const float *d_stack_final,
                          float *d_h0, float *d_stack0,
                          const pcm_stackgru_params *grads,
                          int32_t accumulate, void *scratch,
                          size_t scratch_bytes, void *stream);

/* Dense layers ---------------------------------------------------------------*/

/*
 * y[M,N] = x[M,K] . w[N,K]^T + b[N]   (torch.nn.Linear: hidden_to_mu /
 * hidden_to_logvar vae.py:84-85, latent_to_hidden vae.py:316, output_layer
 * vae.py:342).  ldx / ldy are row strides in floats; b may be NULL.
 */
int pcm_linear_forward(const float *x, int64_t ldx, const float *w,
                       const float *b, float *y, int64_t ldy, int32_t m,
                       int32_t n, int32_t k, void *stream);
size_t pcm_linear_scratch_bytes(int32_t m, int32_t n, int32_t k);
/*
 * Gradients of pcm_linear_forward: dx = s*dy.w, dw = s*dy^T.x, db = s*column
 * sums of dy, with s = dy_scale[0] (device scalar; NULL = 1).  Any of
 * dx/dw/db may be NULL.  accumulate != 0 adds into dw/db.
 */
int pcm_linear_backward(const float *x, int64_t ldx, const float *w,
                        const float *dy, const float *dy_scale, int64_t lddy,
                        float *dx, int64_t lddx,
                        float *dw, float *db, int32_t m, int32_t n, int32_t k,
                        int32_t accumulate, void *scratch,
                        size_t scratch_bytes, void *stream);

/* Decoder head: cross entropy over every step -----------------------------*/

/*
 * Sum over steps of the batch-mean cross entropy (criterion of vae.py:288
 * applied at vae.py:343 / 386) plus the monitoring argmax (vae.py:347-348).
 *
 *   logits   fp32[T,B,V]
 *   targets  int64[T,B]
 *   lengths  int32[B] or NULL: step t averages over the rows with
 *            lengths[b] > t (all B rows when NULL; no ignore_index)
 *   step_weight fp32[T] or NULL: when given, step t's row losses are scaled by
 *            step_weight[t] INSTEAD of 1 / (rows active at step t).  Data
 *            parallel training passes 1 / (rows active at step t over ALL
 *            ranks), so that the sum of the ranks' losses is the single-process
 *            loss of the global batch (vae.py:386 averages over the rows that
 *            are active in the whole batch).
 *   loss     fp32[1]
 *   argmax   int64[B,T] (nullable) first maximum on ties; 0 where inactive
 *   dlogits  fp32[T,B,V] (nullable) d loss / d logits
 *   scratch  fp32[T*B + T] row losses and per-step weights (required)
 */
int pcm_softmax_ce(const float *logits, const int64_t *targets,
                   const int32_t *lengths, const float *step_weight,
                   int32_t steps, int32_t batch,
                   int32_t vocab, float *loss, int64_t *argmax, float *dlogits,
                   float *scratch, int32_t *status, void *stream);

/*
 * Output layer fused with the cross entropy: logits = x . w^T + bias
 * (output_layer, vae.py:342) and, in the same kernel, the per-step criterion
 * and the monitoring argmax (vae.py:343-348, 386-388) -- the logits do not
 * travel through memory between the two.
 *
 *   x        fp32 rows [T*B] of `hidden` floats, row stride ldx
 *   w, bias  output_layer.weight [V,H], .bias [V]
 *   logits   fp32[T,B,V] (nullable: not written; required for shapes outside
 *            the fused kernel -- V > 128 or H not a multiple of 8 --, which run
 *            as pcm_linear_forward + pcm_softmax_ce)
 *   everything else as pcm_softmax_ce
 */
int pcm_output_ce(const float *x, int64_t ldx, const float *w,
                  const float *bias, const int64_t *targets,
                  const int32_t *lengths, const float *step_weight,
                  int32_t steps, int32_t batch, int32_t vocab, int32_t hidden,
                  float *logits, float *loss, int64_t *argmax, float *dlogits,
                  float *scratch, int32_t *status, void *stream);

/*
 * pcm_softmax_ce for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving / evaluation loops, harnesses that forbid
 * hipMalloc).  NO counterpart in the reference: the same computation with the
 * T*B + T floats of scratch taken from a static per-GPU array inside the
 * library instead of a caller workspace; it calls no hipMalloc / hipFree or
 * any allocating routine.  Limits: steps*batch + steps <= PCM_CE_STATIC_MAX_FLOATS
 * (PCM_EINVAL beyond it).  Every call on one GPU shares that GPU's array:
 * calls that overlap in time on one GPU (different streams / threads) race;
 * calls ordered on one stream are safe, nothing needs clearing in between.
 * The array's device address is looked up once per device, by pcm_device_check()
 * or else by the first call on that device: a caller that records this call
 * into a HIP graph runs pcm_device_check() (or one eager call) on the device
 * first, so that no symbol lookup happens under capture.
 */
#define PCM_CE_STATIC_MAX_FLOATS (1 << 20) /* 4 MiB per GPU */
int pcm_softmax_ce_static(const float *logits, const int64_t *targets,
                          const int32_t *lengths, const float *step_weight,
                          int32_t steps, int32_t batch, int32_t vocab,
                          float *loss, int64_t *argmax, float *dlogits,
                          int32_t *status, void *stream);

/* Latent ---------------------------------------------------------------------*/

/* z = eps * exp(0.5*logvar) + mu  (vae.py:538-540), n elements. */
int pcm_reparam_forward(const float *mu, const float *logvar, const float *eps,
                        float *z, int64_t n, void *stream);
/* dmu = dz ; dlogvar = dz * eps * 0.5 * exp(0.5*logvar). */
int pcm_reparam_backward(const float *dz, const float *logvar,
                         const float *eps, float *dmu, float *dlogvar,
                         int64_t n, void *stream);
/*
 * kl[0] = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) (loss_functions.py:39);
 * scratch: fp32[pcm_kl_scratch_floats(n)].
 */
size_t pcm_kl_scratch_floats(int64_t n);
int pcm_kl_forward(const float *mu, const float *logvar, float *kl,
                   float *scratch, int64_t n, void *stream);
/* dmu = dkl[0]*mu ; dlogvar = dkl[0] * 0.5 * (exp(logvar) - 1). */
int pcm_kl_backward(const float *dkl, const float *mu, const float *logvar,
                    float *dmu, float *dlogvar, int64_t n, void *stream);

/* Generation -------------------------------------------------------------------*/

#define PCM_SEARCH_GREEDY 0   /* GreedySearch.step   utils/search.py:76-87   */
#define PCM_SEARCH_SAMPLING 1 /* SamplingSearch.step utils/search.py:124-141 */

size_t pcm_generate_workspace_bytes(const pcm_dims *dims);
/*
 * StackGRUDecoder.generate_from_latent, non-beam branch (vae.py:422-468):
 * prime the state with prime[0..P-2], then `generate_len` times: one StackGRU
 * step, output_layer, search.step, feed the chosen token back.
 *
 *   h0        fp32[L,B,H] (layer stride as in pcm_stackgru_forward): the
 *             latent_to_hidden projection of the latent (vae.py:425-427)
 *   out_w/out_b  output_layer weight [V,H] / bias [V]
 *   prime     int64[P] priming tokens, P >= 1 (same for every row)
 *   tokens_out int64[B, P+generate_len], the prime included (vae.py:430,464)
 *   logits_out fp32[generate_len,B,V] (nullable) per-step logits
 *   search    PCM_SEARCH_*; temperature and seed only used for sampling
 *             (token = first index whose cumulative softmax(logits/T) exceeds
 *             a counter-based uniform of (seed, step, row))
 */
int pcm_stackgru_generate(const pcm_dims *dims,
                          const pcm_stackgru_params *weights,
                          const float *out_w, const float *out_b,
                          const float *h0, int64_t h0_layer_stride,
                          const int64_t *prime, int32_t prime_len,
                          int32_t generate_len, int32_t search,
                          float temperature, uint64_t seed,
                          int64_t *tokens_out, float *logits_out,
                          void *workspace, size_t workspace_bytes,
                          int32_t *status, void *stream);

/*
 * GreedySearch.step / SamplingSearch.step and their forward() forms
 * (utils/search.py:66-87, 105-141) on logits that already live on the device:
 * one token per row of logits[rows, vocab] -- the first maximum (greedy), or
 * the first index whose cumulative softmax(logits / temperature) exceeds a
 * counter-based uniform of (seed, step, row) (sampling: the batched form of
 * the reference's per-row torch.multinomial loop).
 */
int pcm_search_step(const float *logits, int64_t rows, int32_t vocab,
                    int32_t search, float temperature, uint64_t seed,
                    int32_t step, int64_t *tokens_out, void *stream);

size_t pcm_generate_beam_workspace_bytes(const pcm_dims *dims,
                                         int32_t beam_width,
                                         int32_t generate_len);
/*
 * StackGRUDecoder.generate_from_latent, beam branch (vae.py:439-488) with
 * BeamSearch.step (utils/search.py:169-201, 247-275) on the device: the
 * beam_width state copies are rows k*B + b of one batch of beam_width * B
 * sequences; per generated token one fused StackGRU step over that batch,
 * output_layer, and one beam step (softmax(logits / temperature), the
 * top_tokens most probable tokens per beam, candidates ranked by
 * score + log(p + eps) with ties in candidate order, the best beam_width
 * kept).  As in the reference, a state copy stays with its slot; it does not
 * follow the parent of the beam that lands there.
 *
 *   dims->batch  B (the true batch); beam_width 1..16
 *   tokens_out       int64[B, generate_len]: the best-scoring sequence of
 *                    each row, WITHOUT the prime (vae.py:485-488)
 *   beam_tokens_out  int64[B, beam_width, generate_len] (nullable): every
 *                    beam, best first; -1 where fewer beams exist
 *   beam_scores_out  fp64[B, beam_width] (nullable): their scores (sums of
 *                    log-probabilities, double like the reference's floats)
 */
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
                               void *stream);

/* Optimiser ------------------------------------------------------------------*/

/*
 * One torch.optim.Adam step (OPTIMIZER_FACTORY['adam'], hyperparams.py:14,
 * as constructed at training.py:132: betas (0.9, 0.999), eps 1e-8, no weight
 * decay, no amsgrad) over a flat fp32 buffer of n elements.  `step` is the
 * 1-based step count used for bias correction; grad_scale multiplies the
 * gradient first (1/world_size after a sum all-reduce).
 */
int pcm_adam_step(float *param, const float *grad, float *exp_avg,
                  float *exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t step, float grad_scale,
                  void *stream);

/* Batch preparation -----------------------------------------------------------*/

/*
 * The three training sequences of a batch, built on the device from the
 * concatenated samples (one host->device copy per batch): what
 * sequential_data_preparation (utils/utils.py:29-81, after training.py:339-348
 * pad_sequence) and packed_sequential_data_preparation (utils/utils.py:84-146)
 * build sample by sample on the host.
 *
 *   flat      int64[N]    the samples' tokens back to back
 *   offsets   int64[B+1]  sample b is flat[offsets[b] .. offsets[b+1])
 *   steps     T = the longest sample
 *   packed == 0: enc / dec / tgt are int64[T,B], zero padded.  tgt is enc
 *            shifted by one step with a zero row appended; word dropout
 *            replaces WHOLE time rows of dec by dropout_index, padding
 *            included (utils/utils.py:65-75).
 *   packed != 0: enc / dec / tgt are the `data` of three PackedSequences,
 *            element (t, b) at step_base[t] + b, step_base int64[T] = exclusive
 *            prefix sum of batch_sizes (samples sorted by decreasing length);
 *            word dropout is per token (utils/utils.py:118-137), tgt is each
 *            sample shifted by one with a zero appended.
 *   A token (packed) or time row (padded) is KEPT when keep_mask[i] != 0, i =
 *   flat position / time step; with keep_mask == NULL when a counter-based
 *   uniform of (seed, i) is below input_keep (Bernoulli(input_keep), always
 *   kept at input_keep >= 1).
 */
int pcm_prepare_batch(const int64_t *flat, const int64_t *offsets,
                      int32_t batch, int32_t steps, int32_t packed,
                      const int64_t *step_base, float input_keep,
                      uint64_t seed, const uint8_t *keep_mask,
                      int64_t dropout_index, int64_t *enc, int64_t *dec,
                      int64_t *tgt, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PACCMANN_HIP_H */
