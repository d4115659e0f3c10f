"""Data parallelism for the training step: one process per GPU, the batch
sharded across ranks, parameters replicated, gradients summed with RCCL
(``torch.distributed`` backend ``nccl``) -- or ``gloo`` in the CPU tests.

The reference has no distributed code; this is the one exchange step the
sharded path needs.  Loss scaling: the decoder loss is a MEAN over the local
batch (vae.py:343) while the KL term is a SUM over the batch
(loss_functions.py:39).  To reproduce the single-process gradient of the
global batch, each rank back-propagates ``decoder_loss / world + kl_w * KL``
and the gradients are SUM-reduced.

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


def scale_losses(decoder_loss, kl_div, kl_w):
    """Per-rank scalar to back-propagate (see module docstring)."""
    size, _ = world()
    return decoder_loss / size + kl_w * kl_div


class GradientBuckets:
    """Flat gradient buckets of a TeacherVAE, reduced with async all-reduce.

    ``param.grad`` of every parameter is a view of its bucket's flat gradient
    buffer, so a bucket's gradients are contiguous and no flatten / unflatten
    copies are needed (FlatAdam below does the same for the parameters)."""

    def __init__(self, model, overlap=True, force_collective=False):
        self.model = model
        self.overlap = overlap
        # issue the all-reduce even in a one-rank group (rehearses the RCCL
        # path on a single GPU; the sum over one rank is the identity)
        self.force_collective = force_collective
        dec = [p for n, p in model.named_parameters()
               if n.startswith('decoder.') and p.requires_grad]
        enc = [p for n, p in model.named_parameters()
               if not n.startswith('decoder.') and p.requires_grad]
        self.buckets = [self._make_bucket(dec), self._make_bucket(enc)]
        self._pending = []
        self._hooks = []
        if overlap:
            self._install_hooks()

    @staticmethod
    def _make_bucket(params):
        if not params:
            return None
        total = sum(p.numel() for p in params)
        dev, dtype = params[0].device, params[0].dtype
        flat = torch.zeros(total, device=dev, dtype=dtype)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return {'params': params, 'flat': flat, 'views': views, 'ready': 0}

    def _install_hooks(self):
        """Launch a bucket's all-reduce as soon as its last gradient has been
        accumulated (post-accumulate-grad hooks)."""
        for bucket in self.buckets:
            if bucket is None:
                continue
            for p in bucket['params']:
                self._hooks.append(
                    p.register_post_accumulate_grad_hook(
                        lambda param, b=bucket: self._on_grad(b)
                    )
                )

    def _on_grad(self, bucket):
        bucket['ready'] += 1
        if bucket['ready'] == len(bucket['params']):
            self._launch(bucket)

    def _gather_flat(self, bucket):
        for p, view in zip(bucket['params'], bucket['views']):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
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
        for bucket in self.buckets:
            if bucket is None:
                continue
            bucket['flat'].zero_()
            bucket['ready'] = 0
            for p, view in zip(bucket['params'], bucket['views']):
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
            flat = torch.empty_like(bucket['flat'])
            off = 0
            with torch.no_grad():
                for p in bucket['params']:
                    view = flat[off:off + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view
                    off += p.numel()
            self.state.append({
                'param': flat,
                'exp_avg': torch.zeros_like(flat),
                'exp_avg_sq': torch.zeros_like(flat),
            })

    def zero_grad(self):
        self.buckets.zero_grad()

    def step(self):
        from .ops import adam_step
        self.step_count += 1
        for bucket, st in zip(self.buckets.buckets, self.state):
            if bucket is None:
                continue
            adam_step(st['param'], bucket['flat'], st['exp_avg'],
                      st['exp_avg_sq'], self.lr, self.betas[0], self.betas[1],
                      self.eps, self.step_count)
