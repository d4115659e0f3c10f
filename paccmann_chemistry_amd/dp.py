"""Data parallelism for the training step: one process per GPU, the batch
sharded across ranks, parameters replicated, gradients summed with RCCL
(``torch.distributed`` backend ``nccl``) -- or ``gloo`` in the CPU tests.

The reference has no distributed code; this is the one exchange step the
sharded path needs.  Loss scaling: the decoder loss is a MEAN over the local
batch (vae.py:343) while the KL term is a SUM over the batch
(loss_functions.py:39).  To reproduce the single-process gradient of the
global batch, each rank back-propagates ``decoder_loss / world + kl_w * KL``
and the gradients are SUM-reduced.  In packed mode (and for unequal shards)
the divisor of step t is the number of rows active over ALL ranks:
``global_step_weights`` all-reduces the T counts and the decoder head scales
its row losses by them (``StackGRUDecoder.global_batch_loss``).

Gradients travel as two flat fp32 buckets, decoder first: the decoder's
gradients are final before the encoder's backward starts, so its all-reduce
is issued asynchronously and overlaps the encoder's reverse recurrence.
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def scale_losses(decoder_loss, kl_div, kl_w, global_share=False):
    """Per-rank scalar to back-propagate (see module docstring).

    global_share: the decoder loss already is this rank's share of the global
    batch's loss (``StackGRUDecoder.global_batch_loss = True``: every step is
    divided by the rows active over ALL ranks), so it is not divided again.
    That form is exact for packed batches and unequal shards too; the plain
    ``/ world`` form only for padded batches with equal shards."""
    if global_share:
        return decoder_loss + kl_w * kl_div
    size, _ = world()
    return decoder_loss / size + kl_w * kl_div


def global_step_weights(lengths, steps, batch, device=None):
    """fp32[T]: 1 / (rows active at step t over ALL ranks) -- the divisor of the
    reference's per-step cross-entropy mean (vae.py:343 padded, vae.py:386
    packed: the mean runs over the rows of the whole batch that are still
    active), when the batch is sharded over ranks.  One all-reduce of T counts
    per step; a single process gets 1 / local counts.

    lengths: per-row lengths of THIS rank's shard (None = padded mode, every
    row runs ``steps`` steps).  Padded mode: every rank must run the same
    number of steps (pad the shards of one global batch to one length, as the
    single-process ``pad_sequence`` does); the exchange is then ONE all-reduce
    on the device with no host synchronisation.  Packed mode: shards may differ
    in their longest sequence, so the longest T is exchanged first."""
    size, _ = world()
    on = 'cpu'
    if size > 1 and dist.get_backend() == 'nccl':
        on = device if device is not None else 'cuda'
    if lengths is None:
        counts = torch.full((steps, ), int(batch), dtype=torch.int64, device=on)
        if size > 1:
            dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    else:
        lens = torch.as_tensor(lengths).to('cpu', torch.int64).view(1, -1)
        counts = (lens > torch.arange(steps).view(-1, 1)).sum(dim=1)
        if size > 1:
            # count over the longest T of any rank (reduced with MAX first)
            longest = torch.tensor([steps], dtype=torch.int64, device=on)
            dist.all_reduce(longest, op=dist.ReduceOp.MAX)
            padded = torch.zeros(int(longest.item()), dtype=torch.int64,
                                 device=on)
            padded[:steps] = counts.to(on)
            dist.all_reduce(padded, op=dist.ReduceOp.SUM)
            counts = padded[:steps]
    weights = torch.where(
        counts > 0, 1.0 / counts.clamp(min=1).to(torch.float32),
        torch.zeros(steps, dtype=torch.float32, device=counts.device)
    )
    return weights if device is None else weights.to(device)


class GradSink:
    """Where the library writes one parameter's gradient: a view of its
    bucket's flat buffer.  The autograd Functions of ops.py hand ``view`` to
    the C ABI as the gradient output (``=`` on the first write after
    ``zero_grad``, ``+=`` on later ones) and return None for the parameter, so
    no AccumulateGrad add -- one small launch per parameter and step -- runs."""

    __slots__ = ('view', 'written', 'counted', 'bucket', 'owner')

    def __init__(self, view, bucket, owner):
        self.view, self.bucket, self.owner = view, bucket, owner
        self.written = self.counted = False

    def usable_by(self, param):
        """True while ``param.grad`` is this sink's view (the caller follows
        the GradientBuckets.zero_grad protocol)."""
        grad = param.grad
        return grad is not None and grad.data_ptr() == self.view.data_ptr()

    def mark_written(self):
        """The library has enqueued this gradient's write into the view."""
        self.written = True
        if not self.counted:
            self.counted = True
            self.owner._on_grad(self.bucket)


ALIGN = 64   # floats: every view starts on a 256-byte boundary


class GradientBuckets:
    """Flat gradient buckets of a TeacherVAE, reduced with async all-reduce.

    ``param.grad`` of every parameter is a view of its bucket's flat gradient
    buffer, so a bucket's gradients are contiguous and no flatten / unflatten
    copies are needed (FlatAdam below does the same for the parameters).  With
    ``direct`` (the default) every parameter also carries a ``GradSink``: the
    reverse pass writes its gradients straight into the views."""

    def __init__(self, model, overlap=True, force_collective=False,
                 direct=None):
        self.model = model
        self.overlap = overlap
        if direct is None:   # the library only ever writes device memory
            direct = all(p.is_cuda for p in model.parameters())
        self.direct = direct
        # issue the all-reduce even in a one-rank group (rehearses the RCCL
        # path on a single GPU; the sum over one rank is the identity)
        self.force_collective = force_collective
        dec = [p for n, p in model.named_parameters()
               if n.startswith('decoder.') and p.requires_grad]
        enc = [p for n, p in model.named_parameters()
               if not n.startswith('decoder.') and p.requires_grad]
        self._pending = []
        self._hooks = []
        self.buckets = [self._make_bucket(dec), self._make_bucket(enc)]
        if overlap:
            self._install_hooks()

    def _make_bucket(self, params):
        if not params:
            return None
        offsets, off = [], 0
        for p in params:
            offsets.append(off)
            off += -(-p.numel() // ALIGN) * ALIGN
        dev, dtype = params[0].device, params[0].dtype
        flat = torch.zeros(off, device=dev, dtype=dtype)
        views = [flat[o:o + p.numel()].view_as(p)
                 for p, o in zip(params, offsets)]
        bucket = {'params': params, 'flat': flat, 'views': views,
                  'offsets': offsets, 'ready': 0, 'sinks': []}
        for p, view in zip(params, views):
            sink = GradSink(view, bucket, self)
            bucket['sinks'].append(sink)
            if self.direct:
                p._pcm_grad_sink = sink
        return bucket

    def _install_hooks(self):
        """Launch a bucket's all-reduce as soon as its last gradient has been
        accumulated (post-accumulate-grad hooks; gradients written through a
        GradSink report by hand, see GradSink.mark_written)."""
        for bucket in self.buckets:
            if bucket is None:
                continue
            for p, sink in zip(bucket['params'], bucket['sinks']):
                self._hooks.append(
                    p.register_post_accumulate_grad_hook(
                        lambda param, b=bucket, s=sink: self._on_hook(b, s)
                    )
                )

    def _on_hook(self, bucket, sink):
        # AccumulateGrad ran for this parameter (a producer that did not use
        # the sink); count it once per step
        sink.written = True
        if not sink.counted:
            sink.counted = True
            self._on_grad(bucket)

    def _on_grad(self, bucket):
        if not self.overlap:
            return
        bucket['ready'] += 1
        if bucket['ready'] == len(bucket['params']):
            self._launch(bucket)

    def _gather_flat(self, bucket):
        for p, view, sink in zip(bucket['params'], bucket['views'],
                                 bucket['sinks']):
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            sink.written = True
        return bucket['flat']

    def _launch(self, bucket):
        size, _ = world()
        flat = self._gather_flat(bucket)
        if size > 1 or (self.force_collective and dist.is_initialized()):
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            self._pending.append(work)
        bucket['ready'] = -(1 << 30)  # launched

    def finish(self):
        """Call after backward: reduce whatever was not launched by the hooks
        and wait for every bucket."""
        for bucket in self.buckets:
            if bucket is not None and bucket['ready'] >= 0:
                self._launch(bucket)
        for work in self._pending:
            work.wait()
        self._pending = []
        for bucket in self.buckets:
            if bucket is not None:
                bucket['ready'] = 0

    def zero_grad(self):
        """Start a step: the flat buffers are cleared (one fill per bucket) and
        every ``param.grad`` is its bucket view again.  The library's first
        write of a step into a view overwrites it (GradSink), later ones add;
        a gradient that arrives through AccumulateGrad, or that the caller
        copies into ``param.grad`` by hand, lands on the zeros as usual."""
        for bucket in self.buckets:
            if bucket is None:
                continue
            bucket['ready'] = 0
            bucket['flat'].zero_()
            for p, view, sink in zip(bucket['params'], bucket['views'],
                                     bucket['sinks']):
                sink.written = sink.counted = False
                p.grad = view


class FlatAdam:
    """``torch.optim.Adam`` (betas, eps as given, no weight decay / amsgrad --
    the reference's ``OPTIMIZER_FACTORY['adam'](params, lr=lr)``,
    training.py:132) over the flat buffers of a GradientBuckets: the
    parameters are re-pointed into one flat buffer per bucket and a step is one
    ``pcm_adam_step`` launch per bucket instead of a multi-tensor sweep."""

    def __init__(self, buckets, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.buckets = buckets
        self.lr, self.betas, self.eps = lr, betas, eps
        self.step_count = 0
        self.state = []
        for bucket in buckets.buckets:
            if bucket is None:
                self.state.append(None)
                continue
            flat = torch.zeros_like(bucket['flat'])
            with torch.no_grad():
                for p, off in zip(bucket['params'], bucket['offsets']):
                    view = flat[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
            self.state.append({
                'param': flat,
                'exp_avg': torch.zeros_like(flat),
                'exp_avg_sq': torch.zeros_like(flat),
            })

    def zero_grad(self):
        self.buckets.zero_grad()

    def state_dict(self):
        """Step count and moments per bucket (checkpoint / resume)."""
        return {
            'step_count': self.step_count, 'lr': self.lr,
            'betas': tuple(self.betas), 'eps': self.eps,
            'buckets': [
                None if st is None else {
                    'exp_avg': st['exp_avg'].clone(),
                    'exp_avg_sq': st['exp_avg_sq'].clone(),
                } for st in self.state
            ],
        }

    def load_state_dict(self, state):
        if len(state['buckets']) != len(self.state):
            raise ValueError('FlatAdam: bucket count mismatch')
        for st, saved in zip(self.state, state['buckets']):
            if (st is None) != (saved is None):
                raise ValueError('FlatAdam: bucket layout mismatch')
            if st is None:
                continue
            for key in ('exp_avg', 'exp_avg_sq'):
                if saved[key].numel() != st[key].numel():
                    raise ValueError(f'FlatAdam: {key} size mismatch')
                st[key].copy_(saved[key])
        self.step_count = int(state['step_count'])
        self.lr, self.eps = state['lr'], state['eps']
        self.betas = tuple(state['betas'])

    def step(self):
        from .ops import adam_step
        self.step_count += 1
        for bucket, st in zip(self.buckets.buckets, self.state):
            if bucket is None:
                continue
            if bucket['ready'] >= 0:
                # not reduced through GradientBuckets.finish() (a plain
                # single-process loop such as train_vae): make sure every
                # gradient sits in the flat buffer
                self.buckets._gather_flat(bucket)
            adam_step(st['param'], bucket['flat'], st['exp_avg'],
                      st['exp_avg_sq'], self.lr, self.betas[0], self.betas[1],
                      self.eps, self.step_count)
