"""Autograd-visible operators over the C ABI (include/paccmann_hip.h).

torch is used for device memory, streams and the autograd graph only; every
arithmetic operation of the path runs in libpaccmann_hip.so.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, load, ptr, require_device, stream_ptr


_PRECISIONS = {'fp32': 0, 'bf16': 1}


def set_matmul_precision(precision):
    """'fp32' (default; the mode parity with the reference is claimed on) or
    'bf16': the recurrent cells' products and the recurrent weight gradients
    take bf16-rounded operands with fp32 accumulation (pcm_set_matmul_precision).
    The reference trains in fp32 only (models/training.py:132-157)."""
    if precision not in _PRECISIONS:
        raise ValueError(
            f'precision {precision!r}: expected one of {sorted(_PRECISIONS)}'
        )
    check(
        load().pcm_set_matmul_precision(_PRECISIONS[precision]),
        'set_matmul_precision'
    )


def get_matmul_precision():
    mode = load().pcm_get_matmul_precision()
    return next(k for k, v in _PRECISIONS.items() if v == mode)


def _f32(t, what):
    require_device(t, what)
    if t.dtype != torch.float32:
        raise TypeError(f'{what}: expected float32, got {t.dtype}')
    return t


def _rows2d(x, what):
    """[.., K] tensor -> (tensor kept alive, rows, row stride) with unit
    column stride and a uniform row stride; copies only when it must."""
    k = x.shape[-1]
    x2 = x.reshape(-1, k)  # a view whenever the strides allow it
    if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < k):
        x2 = x2.contiguous()
    ld = x2.stride(0) if x2.shape[0] > 1 else max(k, 1)
    return x2, x2.shape[0], ld


def check_status(status, what='token ids'):
    """Raise the RuntimeError the reference's embedding lookup would raise for
    ids outside the vocabulary (generate_latent_codes.py:126 relies on it).
    One device->host read, so callers do it once per forward."""
    if status is None:
        return
    word = int(status.item())
    if word:
        status.zero_()
    if word & _lib.STATUS_TIMEOUT:
        raise _lib.PaccmannHipError(
            'the persistent forward kernel gave up waiting for its peer '
            'blocks (GPU shared with other work?); results are invalid -- '
            'set PCM_PERSIST=0 to use the launch chain'
        )
    if word & _lib.STATUS_BAD_TOKEN:
        raise RuntimeError(f'index out of range in self ({what})')


# Packed batches launch over the still-running prefix of the batch (True) or
# over every row, freezing the finished ones (False; also what happens when
# the lengths are not sorted).  Both give the same mu / logvar / loss /
# gradients; tests flip it to keep the frozen-row path covered.
SHRINK_ACTIVE_BATCH = True


def _active_rows(lengths, steps):
    """Host int32[T] of rows still running at each step, when the lengths
    (a CPU tensor, as pad_packed_sequence returns them) are sorted by
    decreasing length -- PackedSequence.batch_sizes, which the reference
    slices its hidden / stack by (utils/utils.py:226-243).  None otherwise:
    the kernels then run every row and freeze the finished ones."""
    if not SHRINK_ACTIVE_BATCH or lengths.device.type != 'cpu':
        return None
    lens = lengths.to(torch.int64)
    if lens.numel() == 0 or bool((lens[1:] > lens[:-1]).any()):
        return None
    steps_idx = torch.arange(steps).view(-1, 1)
    return (lens.view(1, -1) > steps_idx).sum(dim=1).to(torch.int32).contiguous()


def _direct_sinks(params, needs):
    """Gradient sinks (dp.GradSink) of ``params`` when EVERY wanted gradient
    can be written straight into its parameter's bucket view, all in the same
    state: (sinks, accumulate) with accumulate = 0 for the first write since
    ``zero_grad`` and 1 for a later one; None when any of them must take the
    ordinary route (a returned tensor that AccumulateGrad adds)."""
    sinks, states = [], set()
    for param, need in zip(params, needs):
        if param is None or not need:
            sinks.append(None)
            continue
        sink = getattr(param, '_pcm_grad_sink', None)
        if sink is None or not sink.usable_by(param):
            return None
        sinks.append(sink)
        states.add(sink.written)
    if len(states) != 1:
        return None
    return sinks, 1 if states.pop() else 0


def _host_ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# --------------------------------------------------------------------------
# StackGRU over a sequence
# --------------------------------------------------------------------------
class StackGRUSequenceFn(torch.autograd.Function):
    """out[T,B,H], h_final[L,B,H], stack_final[B,D,W] =
    f(tokens[T,B], lengths, h0, stack0, *weights)."""

    @staticmethod
    def forward(ctx, cfg, tokens, lengths, h0, stack0, status, *weights):
        lib = load()
        require_device(tokens, 'StackGRU tokens')
        dims, n_layers = cfg['dims'], cfg['dims'].layers
        steps, batch = tokens.shape
        if batch != dims.batch:
            raise ValueError(f'batch {batch} != dims.batch {dims.batch}')
        dev = tokens.device
        tokens = tokens.contiguous()
        if tokens.dtype != torch.int64:
            tokens = tokens.long()
        training = 1 if cfg['training'] else 0
        hid = dims.hidden
        ctx.params = weights
        weights = [None if w is None else _f32(w, 'weight').contiguous()
                   for w in weights]
        wstruct = _lib.make_params(weights, n_layers)
        nbytes = lib.pcm_stackgru_saved_bytes(
            ctypes.byref(dims), steps, training
        )
        saved = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        h_final = torch.empty(n_layers, batch, hid, device=dev)
        stack_final = None
        if dims.use_stack:
            stack_final = torch.empty(
                batch, dims.stack_depth, dims.stack_width, device=dev
            )
        h0_stride = 0
        if h0 is not None:
            _f32(h0, 'h0')
            if h0.dim() == 2:  # one [B,H] broadcast to every layer
                h0 = h0.contiguous()
                h0_stride = 0
            else:
                h0 = h0.contiguous()
                h0_stride = batch * hid
        if stack0 is not None:
            stack0 = _f32(stack0, 'stack0').contiguous()
        active = None
        if lengths is not None:
            active = _active_rows(lengths, steps)
            lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
        check(
            lib.pcm_stackgru_forward(
                ctypes.byref(dims), ctypes.byref(wstruct), ptr(tokens), steps,
                ptr(lengths), _host_ptr(active), ptr(h0), h0_stride,
                ptr(stack0),
                float(cfg['dropout']), int(cfg['seed']), training, ptr(saved),
                nbytes, ptr(h_final), ptr(stack_final), ptr(status),
                stream_ptr(dev)
            ), 'pcm_stackgru_forward'
        )
        off, ld = ctypes.c_size_t(), ctypes.c_size_t()
        check(
            lib.pcm_stackgru_out_view(
                ctypes.byref(dims), steps, training, ctypes.byref(off),
                ctypes.byref(ld)
            ), 'pcm_stackgru_out_view'
        )
        out = saved.as_strided(
            (steps, batch, hid), (batch * ld.value, ld.value, 1), off.value
        )
        ctx.cfg = cfg
        ctx.steps = steps
        ctx.h0_broadcast = h0 is not None and h0.dim() == 2
        ctx.has_h0 = h0 is not None
        ctx.has_stack0 = stack0 is not None
        ctx.saved_buf = saved
        ctx.tokens = tokens
        ctx.lengths = lengths
        ctx.active = active
        ctx.weights = weights
        ctx.set_materialize_grads(False)
        if stack_final is None:
            stack_final = torch.empty(0, device=dev)
            ctx.mark_non_differentiable(stack_final)
        return out, h_final, stack_final

    @staticmethod
    def backward(ctx, d_out, d_h_final, d_stack_final):
        lib = load()
        cfg = ctx.cfg
        if not cfg['training']:
            raise RuntimeError(
                'StackGRU backward needs a training-mode forward '
                '(module.train()): the eval-mode forward keeps no activations'
            )
        if ctx.saved_buf is None:
            raise RuntimeError(
                'StackGRU backward called twice: the saved activations are '
                'consumed by the first call (retain_graph is not supported)'
            )
        dims, n_layers = cfg['dims'], cfg['dims'].layers
        steps, batch, hid = ctx.steps, dims.batch, dims.hidden
        dev = ctx.tokens.device
        weights = ctx.weights
        wstruct = _lib.make_params(weights, n_layers)
        needs = ctx.needs_input_grad[6:]
        direct = _direct_sinks(ctx.params, needs)
        if direct is not None:
            sinks, accumulate = direct
            grads = [None if s is None else s.view for s in sinks]
        else:
            accumulate = 0
            grads = [
                torch.empty_like(w) if (w is not None and need) else None
                for w, need in zip(weights, needs)
            ]
        gstruct = _lib.make_params(grads, n_layers)
        d_out_ld = 0
        if d_out is not None:
            d_out = _f32(d_out, 'd_out')
            if d_out.stride(2) != 1 or d_out.stride(0) != batch * d_out.stride(1):
                d_out = d_out.contiguous()
            d_out_ld = d_out.stride(1)
        if d_h_final is not None:
            d_h_final = _f32(d_h_final, 'd_h_final').contiguous()
        if d_stack_final is not None and d_stack_final.numel() == 0:
            d_stack_final = None
        if d_stack_final is not None:
            d_stack_final = _f32(d_stack_final, 'd_stack_final').contiguous()
        d_h0 = None
        if ctx.has_h0 and ctx.needs_input_grad[3]:
            d_h0 = torch.empty(n_layers, batch, hid, device=dev)
        d_stack0 = None
        if ctx.has_stack0 and ctx.needs_input_grad[4] and dims.use_stack:
            d_stack0 = torch.empty(
                batch, dims.stack_depth, dims.stack_width, device=dev
            )
        sbytes = lib.pcm_stackgru_scratch_bytes(ctypes.byref(dims), steps)
        scratch = torch.empty(sbytes // 4, dtype=torch.float32, device=dev)
        saved = ctx.saved_buf
        check(
            lib.pcm_stackgru_backward(
                ctypes.byref(dims), ctypes.byref(wstruct), ptr(ctx.tokens),
                steps, ptr(ctx.lengths), _host_ptr(ctx.active),
                float(cfg['dropout']),
                int(cfg['seed']), ptr(saved), saved.numel() * 4, ptr(d_out),
                d_out_ld, ptr(d_h_final), ptr(d_stack_final), ptr(d_h0),
                ptr(d_stack0), ctypes.byref(gstruct), accumulate, ptr(scratch),
                sbytes, stream_ptr(dev)
            ), 'pcm_stackgru_backward'
        )
        ctx.saved_buf = None
        pad = cfg.get('pad_index')
        if grads[4] is not None and pad is not None:
            grads[4][pad].zero_()  # nn.Embedding(padding_idx=...) semantics
        if d_h0 is not None and ctx.h0_broadcast:
            d_h0 = d_h0.sum(dim=0)
        if direct is not None:
            # written in place: nothing for AccumulateGrad to add
            for sink in sinks:
                if sink is not None:
                    sink.mark_written()
            grads = [None] * len(grads)
        ctx.params = None
        return (None, None, None, d_h0, d_stack0, None) + tuple(grads)


# --------------------------------------------------------------------------
# dense layer
# --------------------------------------------------------------------------
_scratch_cache = {}


def _linear_scratch(dev, m, n, k):
    """One reusable scratch buffer per (device, stream): calls on a stream are
    ordered and the buffer is only used inside a single library call; two
    streams (two host threads, or the two directions of a bidirectional
    encoder) must not share one."""
    lib = load()
    need = lib.pcm_linear_scratch_bytes(m, n, k)
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() * 4 < need:
        if len(_scratch_cache) >= 16:   # streams come and go: do not hoard
            _scratch_cache.clear()
        buf = torch.empty(need // 4 + 1, dtype=torch.float32, device=dev)
        _scratch_cache[key] = buf
    return buf


class LinearFn(torch.autograd.Function):
    """y = x . w^T + b on the fp32 MFMA GEMM (torch.nn.Linear semantics)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = load()
        _f32(x, 'linear input')
        weight_param, bias_param = weight, bias
        weight = _f32(weight, 'linear weight').contiguous()
        n, k = weight.shape
        x2, m, ldx = _rows2d(x, 'linear input')
        y = torch.empty(x.shape[:-1] + (n, ), device=x.device)
        if bias is not None:
            bias = _f32(bias, 'linear bias').contiguous()
        check(
            lib.pcm_linear_forward(
                ptr(x2), ldx, ptr(weight), ptr(bias), ptr(y), n, m, n, k,
                stream_ptr(x.device)
            ), 'pcm_linear_forward'
        )
        ctx.save_for_backward(x2, weight)
        ctx.ldx = ldx
        ctx.in_shape = x.shape
        ctx.has_bias = bias is not None
        ctx.params = (weight_param, bias_param)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = load()
        x2, weight = ctx.saved_tensors
        n, k = weight.shape
        dy2, m, lddy = _rows2d(_f32(dy, 'linear grad'), 'linear grad')
        dev = dy.device
        need_x, need_w, need_b = ctx.needs_input_grad
        dx = torch.empty(m, k, device=dev) if need_x else None
        need_b = need_b and ctx.has_bias
        direct = _direct_sinks(ctx.params, (need_w, need_b))
        accumulate = 0
        if direct is not None:
            (sink_w, sink_b), accumulate = direct
            dw = sink_w.view if sink_w is not None else None
            db = sink_b.view if sink_b is not None else None
        else:
            dw = torch.empty(n, k, device=dev) if need_w else None
            db = torch.empty(n, device=dev) if need_b else None
        scratch = _linear_scratch(dev, m, n, k)
        check(
            lib.pcm_linear_backward(
                ptr(x2), ctx.ldx, ptr(weight), ptr(dy2), None, lddy, ptr(dx),
                k, ptr(dw), ptr(db), m, n, k, accumulate, ptr(scratch),
                scratch.numel() * 4, stream_ptr(dev)
            ), 'pcm_linear_backward'
        )
        if dx is not None:
            dx = dx.view(ctx.in_shape)
        if direct is not None:
            for sink in (sink_w, sink_b):
                if sink is not None:
                    sink.mark_written()
            dw = db = None
        return dx, dw, db


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


# --------------------------------------------------------------------------
# decoder head: logits + cross entropy summed over steps (vae.py:342-343)
# --------------------------------------------------------------------------
class DecoderHeadFn(torch.autograd.Function):
    """loss, argmax[B,T], logits[T,B,V] = head(out[T,B,H], W, b, targets)."""

    @staticmethod
    def forward(ctx, out, weight, bias, targets, lengths, status,
                step_weight=None):
        lib = load()
        _f32(out, 'decoder outputs')
        steps, batch, hid = out.shape
        ctx.params = (weight, bias)
        weight = _f32(weight, 'output_layer.weight').contiguous()
        bias = _f32(bias, 'output_layer.bias').contiguous()
        vocab = weight.shape[0]
        dev = out.device
        x2, m, ldx = _rows2d(out, 'decoder outputs')
        logits = torch.empty(steps, batch, vocab, device=dev)
        targets = targets.to(device=dev, dtype=torch.int64).contiguous()
        if lengths is not None:
            lengths = lengths.to(device=dev, dtype=torch.int32).contiguous()
        loss = torch.empty((), device=dev)
        argmax = torch.empty(batch, steps, dtype=torch.int64, device=dev)
        want_grad = any(ctx.needs_input_grad[:3])
        dlogits = torch.empty_like(logits) if want_grad else None
        scratch = torch.empty(m + steps, device=dev)
        if step_weight is not None:
            # data parallel: 1 / (rows active at step t over all ranks)
            step_weight = step_weight.to(
                device=dev, dtype=torch.float32
            ).contiguous()
            if step_weight.numel() != steps:
                raise ValueError('step_weight must have one entry per step')
        # output layer, softmax, cross entropy and argmax in one kernel
        check(
            lib.pcm_output_ce(
                ptr(x2), ldx, ptr(weight), ptr(bias), ptr(targets),
                ptr(lengths), ptr(step_weight), steps, batch, vocab, hid,
                ptr(logits), ptr(loss), ptr(argmax), ptr(dlogits),
                ptr(scratch), ptr(status), stream_ptr(dev)
            ), 'pcm_output_ce'
        )
        ctx.save_for_backward(x2, weight)
        ctx.dlogits = dlogits
        ctx.ldx = ldx
        ctx.shape = (steps, batch, hid, vocab)
        ctx.mark_non_differentiable(argmax, logits)
        return loss, argmax, logits

    @staticmethod
    def backward(ctx, dloss, _dargmax, _dlogits):
        lib = load()
        x2, weight = ctx.saved_tensors
        steps, batch, hid, vocab = ctx.shape
        m = steps * batch
        dev = x2.device
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        dx = torch.empty(steps, batch, hid, device=dev) if need_x else None
        direct = _direct_sinks(ctx.params, (need_w, need_b))
        accumulate = 0
        if direct is not None:
            (sink_w, sink_b), accumulate = direct
            dw = sink_w.view if sink_w is not None else None
            db = sink_b.view if sink_b is not None else None
        else:
            dw = torch.empty(vocab, hid, device=dev) if need_w else None
            db = torch.empty(vocab, device=dev) if need_b else None
        dloss = _f32(dloss, 'dloss').reshape(1).contiguous()
        scratch = _linear_scratch(dev, m, vocab, hid)
        check(
            lib.pcm_linear_backward(
                ptr(x2), ctx.ldx, ptr(weight), ptr(ctx.dlogits), ptr(dloss),
                vocab, ptr(dx), hid, ptr(dw), ptr(db), m, vocab, hid,
                accumulate, ptr(scratch), scratch.numel() * 4, stream_ptr(dev)
            ), 'pcm_linear_backward(output_layer)'
        )
        ctx.dlogits = None
        if direct is not None:
            for sink in (sink_w, sink_b):
                if sink is not None:
                    sink.mark_written()
            dw = db = None
        return dx, dw, db, None, None, None, None


# --------------------------------------------------------------------------
# latent
# --------------------------------------------------------------------------
class ReparamFn(torch.autograd.Function):
    """z = eps * exp(0.5*logvar) + mu (vae.py:538-540)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        lib = load()
        mu = _f32(mu, 'mu').contiguous()
        logvar = _f32(logvar, 'logvar').contiguous()
        eps = _f32(eps, 'eps').contiguous()
        z = torch.empty_like(mu)
        check(
            lib.pcm_reparam_forward(
                ptr(mu), ptr(logvar), ptr(eps), ptr(z), mu.numel(),
                stream_ptr(mu.device)
            ), 'pcm_reparam_forward'
        )
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = load()
        logvar, eps = ctx.saved_tensors
        dz = _f32(dz, 'dz').contiguous()
        dmu = torch.empty_like(dz) if ctx.needs_input_grad[0] else None
        dlv = torch.empty_like(dz) if ctx.needs_input_grad[1] else None
        check(
            lib.pcm_reparam_backward(
                ptr(dz), ptr(logvar), ptr(eps), ptr(dmu), ptr(dlv),
                dz.numel(), stream_ptr(dz.device)
            ), 'pcm_reparam_backward'
        )
        return dmu, dlv, None


class KLDivFn(torch.autograd.Function):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar)) (loss_functions.py:39)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        lib = load()
        mu = _f32(mu, 'mu').contiguous()
        logvar = _f32(logvar, 'logvar').contiguous()
        n = mu.numel()
        kl = torch.empty((), device=mu.device)
        scratch = torch.empty(
            max(1, lib.pcm_kl_scratch_floats(n)), device=mu.device
        )
        check(
            lib.pcm_kl_forward(
                ptr(mu), ptr(logvar), ptr(kl), ptr(scratch), n,
                stream_ptr(mu.device)
            ), 'pcm_kl_forward'
        )
        ctx.save_for_backward(mu, logvar)
        return kl

    @staticmethod
    def backward(ctx, dkl):
        lib = load()
        mu, logvar = ctx.saved_tensors
        dkl = _f32(dkl, 'dkl').reshape(1).contiguous()
        dmu = torch.empty_like(mu) if ctx.needs_input_grad[0] else None
        dlv = torch.empty_like(mu) if ctx.needs_input_grad[1] else None
        check(
            lib.pcm_kl_backward(
                ptr(dkl), ptr(mu), ptr(logvar), ptr(dmu), ptr(dlv),
                mu.numel(), stream_ptr(mu.device)
            ), 'pcm_kl_backward'
        )
        return dmu, dlv


# --------------------------------------------------------------------------
# generation (no autograd)
# --------------------------------------------------------------------------
def stackgru_generate(
    dims, weights, out_w, out_b, h0, prime, generate_len, search, temperature,
    seed, want_logits=False, status=None
):
    """pcm_stackgru_generate; h0 is [B,H] (broadcast) or [L,B,H].  Returns
    (tokens i64[B, P+G], logits [G,B,V] or None)."""
    lib = load()
    require_device(h0, 'generate h0')
    dev = h0.device
    n_layers, batch, hid = dims.layers, dims.batch, dims.hidden
    weights = [None if w is None else _f32(w, 'weight').contiguous()
               for w in weights]
    wstruct = _lib.make_params(weights, n_layers)
    h0 = _f32(h0, 'h0').contiguous()
    h0_stride = 0 if h0.dim() == 2 else batch * hid
    prime = prime.to(device=dev, dtype=torch.int64).contiguous().view(-1)
    plen = prime.numel()
    tokens = torch.empty(batch, plen + generate_len, dtype=torch.int64,
                         device=dev)
    logits = None
    if want_logits and generate_len > 0:
        logits = torch.empty(generate_len, batch, dims.vocab, device=dev)
    nbytes = lib.pcm_generate_workspace_bytes(ctypes.byref(dims))
    work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out_w = _f32(out_w, 'output_layer.weight').contiguous()
    out_b = _f32(out_b, 'output_layer.bias').contiguous()
    check(
        lib.pcm_stackgru_generate(
            ctypes.byref(dims), ctypes.byref(wstruct), ptr(out_w), ptr(out_b),
            ptr(h0), h0_stride, ptr(prime), plen, int(generate_len),
            int(search), float(temperature), int(seed), ptr(tokens),
            ptr(logits), ptr(work), nbytes, ptr(status), stream_ptr(dev)
        ), 'pcm_stackgru_generate'
    )
    return tokens, logits


def stackgru_generate_beam(
    dims, weights, out_w, out_b, h0, prime, generate_len, beam_width,
    top_tokens, temperature, want_beams=False, status=None
):
    """pcm_stackgru_generate_beam; h0 is [B,H] (broadcast) or [L,B,H].
    Returns tokens i64[B, G] (the best beam of every row, without the prime),
    and with ``want_beams`` also (beam tokens i64[B, width, G], scores
    f64[B, width])."""
    lib = load()
    require_device(h0, 'generate h0')
    dev = h0.device
    n_layers, batch, hid = dims.layers, dims.batch, dims.hidden
    weights = [None if w is None else _f32(w, 'weight').contiguous()
               for w in weights]
    wstruct = _lib.make_params(weights, n_layers)
    h0 = _f32(h0, 'h0').contiguous()
    h0_stride = 0 if h0.dim() == 2 else batch * hid
    prime = prime.to(device=dev, dtype=torch.int64).contiguous().view(-1)
    tokens = torch.empty(batch, generate_len, dtype=torch.int64, device=dev)
    beams = scores = None
    if want_beams:
        beams = torch.empty(batch, beam_width, generate_len,
                            dtype=torch.int64, device=dev)
        scores = torch.empty(batch, beam_width, dtype=torch.float64,
                             device=dev)
    nbytes = lib.pcm_generate_beam_workspace_bytes(
        ctypes.byref(dims), int(beam_width), int(generate_len)
    )
    if nbytes == 0:
        check(-1, 'pcm_generate_beam_workspace_bytes')
    work = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    out_w = _f32(out_w, 'output_layer.weight').contiguous()
    out_b = _f32(out_b, 'output_layer.bias').contiguous()
    check(
        lib.pcm_stackgru_generate_beam(
            ctypes.byref(dims), ctypes.byref(wstruct), ptr(out_w), ptr(out_b),
            ptr(h0), h0_stride, ptr(prime), prime.numel(), int(generate_len),
            int(beam_width), int(top_tokens), float(temperature), ptr(tokens),
            ptr(beams), ptr(scores), ptr(work), nbytes, ptr(status),
            stream_ptr(dev)
        ), 'pcm_stackgru_generate_beam'
    )
    if want_beams:
        return tokens, beams, scores
    return tokens


def search_step(logits, search, temperature=1.0, seed=0, step=0):
    """pcm_search_step: one token per row of ``logits[..., V]`` (a GPU
    tensor); returns i64 of shape ``logits.shape[:-1]``."""
    lib = load()
    require_device(logits, 'search logits')
    x2, rows, ld = _rows2d(_f32(logits, 'search logits'), 'search logits')
    if ld != x2.shape[1]:
        x2 = x2.contiguous()
    out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    check(
        lib.pcm_search_step(
            ptr(x2), rows, x2.shape[1], int(search), float(temperature),
            int(seed), int(step), ptr(out), stream_ptr(logits.device)
        ), 'pcm_search_step'
    )
    return out.view(logits.shape[:-1])


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step,
              grad_scale=1.0):
    """In-place torch.optim.Adam update of a flat fp32 buffer."""
    lib = load()
    require_device(param, 'adam param')
    check(
        lib.pcm_adam_step(
            ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq),
            param.numel(), float(lr), float(beta1), float(beta2), float(eps),
            int(step), float(grad_scale), stream_ptr(param.device)
        ), 'pcm_adam_step'
    )
