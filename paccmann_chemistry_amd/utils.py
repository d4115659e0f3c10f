"""Host-side helpers either side of the hot path: device selection, batch
preparation (padded / packed) and the KL annealing weight.  Mirrors the names
exported by the reference's ``paccmann_chemistry/utils/__init__.py``; the
rdkit / pytoda dependent cosmetics are not part of this path.
"""
import math

import torch
from torch.nn.utils.rnn import (
    PackedSequence, pack_sequence, pad_packed_sequence
)


def cuda():
    return torch.cuda.is_available()


def get_device():
    """utils/utils.py:18-19."""
    return torch.device('cuda' if cuda() else 'cpu')


def kl_weight(step, growth_rate=0.004):
    """Sigmoid KL annealing weight (utils/utils.py:246-264)."""
    return 1 / (1 + math.exp((15 - growth_rate * step)))


def _dropout_rows(shape_rows, input_keep):
    """Bernoulli(keep) per position, 1 = keep (utils/utils.py:71,127)."""
    return torch.bernoulli(torch.full((shape_rows, ), float(input_keep)))


def sequential_data_preparation(
    input_batch, input_keep=1, start_index=2, end_index=3, dropout_index=1,
    device=None
):
    """Padded mode (utils/utils.py:29-81): returns (encoder_seq, decoder_seq,
    target_seq), each i64[T,B].  Word dropout replaces WHOLE time rows of the
    decoder input by ``dropout_index`` (<UNK>); the target is the input shifted
    by one step with a zero row appended."""
    device = get_device() if device is None else device
    input_batch = input_batch.long().to(device)
    decoder_batch = input_batch.clone()
    if input_keep != 1:
        keep = _dropout_rows(input_batch.shape[0], input_keep).to(device)
        decoder_batch[keep == 0] = dropout_index
    end_padding = torch.zeros(
        1, input_batch.shape[1], dtype=torch.long, device=device
    )
    target_seq = torch.cat((input_batch[1:], end_padding), dim=0)
    return input_batch, decoder_batch, target_seq


def packed_sequential_data_preparation(
    input_batch, input_keep=1, start_index=2, end_index=3, dropout_index=1,
    device=None
):
    """Packed mode (utils/utils.py:84-146): ``input_batch`` is a list of 1-D
    token tensors sorted by length (longest first); word dropout is applied
    per token; returns three ``PackedSequence`` objects."""
    device = get_device() if device is None else device
    enc, dec, tgt = [], [], []
    for sample in input_batch:
        if sample.dim() != 1:
            raise ValueError
        inp = sample.long().to(device)
        decoder = inp.clone()
        if input_keep != 1:
            keep = _dropout_rows(inp.shape[0], input_keep).to(device)
            decoder[keep == 0] = dropout_index
        target = torch.cat(
            [inp[1:], torch.zeros(1, dtype=torch.long, device=device)]
        )
        enc.append(inp)
        dec.append(decoder)
        tgt.append(target)
    return [pack_sequence(entry) for entry in (enc, dec, tgt)]


class _PinnedStaging:
    """Two reusable page-locked host buffers per device, used alternately: a
    batch's tokens, offsets and step bases are gathered into one of them and
    copied to the device with ONE truly asynchronous host->device copy (from
    pageable memory ``non_blocking=True`` is a synchronous staged copy).  A
    buffer is only refilled after the copy that last read it has finished (an
    event recorded behind the copy), so batch k+1 can be staged while batch
    k's copy and kernels are still in flight."""

    def __init__(self):
        self.slots = [[None, None], [None, None]]   # [buffer, copy event]
        self.turn = 0

    def stage(self, parts, device):
        need = sum(int(p.numel()) for p in parts)
        slot = self.slots[self.turn]
        self.turn ^= 1
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < need:
            slot[0] = torch.empty(max(need, 1 << 16), dtype=torch.int64,
                                  pin_memory=True)
        host = slot[0][:need]
        torch.cat(parts, out=host)
        staged = host.to(device, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(device))
        return staged


_staging = {}


def _stage_to_device(parts, device):
    key = torch.device(device).index or 0
    if key not in _staging:
        _staging[key] = _PinnedStaging()
    return _staging[key].stage(parts, device)


def prepare_batch_on_device(
    batch, packed, input_keep=1, dropout_index=1, device=None, keep_mask=None,
    seed=None
):
    """Both preparations above as ONE host->device copy and ONE launch
    (pcm_prepare_batch): ``batch`` is the DataLoader's list of 1-D token
    tensors (for ``packed`` sorted longest first, as ``collate_fn`` leaves it).

    Returns what ``sequential_data_preparation(pad_sequence(batch))`` /
    ``packed_sequential_data_preparation(batch)`` return: three i64[T,B]
    tensors, or three PackedSequences.  The Bernoulli word-dropout draws come
    from a counter hash seeded from torch's global generator (``seed``), or
    from ``keep_mask`` (uint8: per time step when padded, per token of the
    concatenated batch when packed) when the caller supplies the draws."""
    import ctypes

    from . import _lib
    device = get_device() if device is None else torch.device(device)
    if device.type != 'cuda':
        raise _lib.PaccmannHipError(
            'prepare_batch_on_device needs a HIP device (the host-side '
            'restatements are sequential_data_preparation / '
            'packed_sequential_data_preparation)'
        )
    samples = []
    for sample in batch:
        if sample.dim() != 1:
            raise ValueError
        samples.append(sample.long().cpu())
    lens = torch.tensor([s.shape[0] for s in samples], dtype=torch.int64)
    if len(samples) == 0 or int(lens.min()) < 1:
        raise ValueError('empty batch or empty sample')
    n_batch, steps = len(samples), int(lens.max())
    offsets = torch.zeros(n_batch + 1, dtype=torch.int64)
    offsets[1:] = torch.cumsum(lens, 0)
    parts = [torch.cat(samples), offsets]
    batch_sizes = None
    if packed:
        if bool((lens[1:] > lens[:-1]).any()):
            raise RuntimeError(
                '`lengths` array must be sorted in decreasing order when '
                '`enforce_sorted` is True.'
            )
        batch_sizes = (lens.view(1, -1) > torch.arange(steps).view(-1, 1)).sum(1)
        parts.append(torch.cumsum(batch_sizes, 0) - batch_sizes)
    staged = _stage_to_device(parts, device)
    total = int(offsets[-1])
    flat, offs = staged[:total], staged[total:total + n_batch + 1]
    base = staged[total + n_batch + 1:] if packed else None
    shape = (total, ) if packed else (steps, n_batch)
    out = torch.empty((3, ) + shape, dtype=torch.int64, device=device)
    if keep_mask is not None:
        need = total if packed else steps
        keep_mask = keep_mask.to(device=device, dtype=torch.uint8).contiguous()
        if keep_mask.numel() != need:
            raise ValueError(f'keep_mask needs {need} entries')
    if seed is None:
        seed = int(torch.randint(0, 2**62, (1, )).item())
    lib = _lib.load()
    _lib.check(
        lib.pcm_prepare_batch(
            _lib.ptr(flat), _lib.ptr(offs), n_batch, steps, 1 if packed else 0,
            _lib.ptr(base), float(input_keep), ctypes.c_uint64(seed),
            _lib.ptr(keep_mask), int(dropout_index), _lib.ptr(out[0]),
            _lib.ptr(out[1]), _lib.ptr(out[2]), _lib.stream_ptr(device)
        ), 'pcm_prepare_batch'
    )
    if not packed:
        return out[0], out[1], out[2]
    return [PackedSequence(out[i], batch_sizes) for i in range(3)]


def collate_fn(batch):
    """Sort a batch from longest to shortest (utils/utils.py:149-168)."""
    order = sorted(
        range(len(batch)), key=lambda i: batch[i].shape[0], reverse=True
    )
    return [batch[i] for i in order]


def unpack_sequence(seq):
    tensor_seqs, seq_lens = pad_packed_sequence(seq)
    return [s[:n] for s, n in zip(tensor_seqs.unbind(dim=1), seq_lens)]


def repack_sequence(seq):
    return pack_sequence(seq)


def perpare_packed_input(input):  # (sic) reference spelling
    """Split PackedSequence.data by batch_sizes (utils/utils.py:215-223)."""
    data, prev = [], 0
    for batch in input.batch_sizes:
        size = prev + int(batch)
        data.append(input.data[prev:size])
        prev = size
    return data, input.batch_sizes


def manage_step_packed_vars(final_var, var, batch_size, prev_batch, batch_dim):
    """utils/utils.py:226-243 (kept for API compatibility; the fused kernels
    honour packed semantics through per-row lengths instead)."""
    if batch_size < prev_batch:
        finished = int(prev_batch - batch_size)
        brk = var.shape[batch_dim] - finished
        if batch_dim == 0:
            if final_var is not None:
                final_var[brk:] = var[brk:]
            var = var[:brk]
        elif batch_dim == 1:
            if final_var is not None:
                final_var[:, brk:] = var[:, brk:]
            var = var[:, :brk]
        else:
            raise ValueError('Allowed batch dim are 1 and 2')
    return final_var, var


def packed_to_lengths(packed):
    """PackedSequence -> (padded i64[T,B] on the data's device, lengths
    i64[B] on CPU).  Requires the length-sorted layout ``pack_sequence``
    produces (enforce_sorted), which is what the reference's loops assume."""
    if not isinstance(packed, PackedSequence):
        raise TypeError('Input is not PackedSequence')
    if packed.sorted_indices is not None:
        idx = packed.sorted_indices.cpu()
        if not torch.equal(idx, torch.arange(len(idx))):
            raise ValueError(
                'PackedSequence must be built from length-sorted sequences'
            )
    return _unpack_sorted(packed)


_UNPACK_CACHE = {}


def _unpack_sorted(packed):
    """``pad_packed_sequence`` for a length-sorted PackedSequence without its
    loop over the time steps (one slice copy per step: 120 tiny copies per
    sequence tensor at T=120): ``data`` is row-major over (t, b < batch_sizes[t]),
    so the padded tensor is ONE gather through an index built from
    ``batch_sizes`` on the host.  The index is kept for the next call (encoder
    input, decoder input and target of a batch share their ``batch_sizes``)."""
    sizes = packed.batch_sizes
    steps, batch = int(sizes.numel()), int(sizes[0])
    device = packed.data.device
    # (per stream: the cached index is filled by a copy in stream order, and a
    # model may serve independent batches on several streams)
    stream = torch.cuda.current_stream(device).cuda_stream \
        if device.type == 'cuda' else 0
    key = (device.type, device.index, stream)
    cached = _UNPACK_CACHE.get(key)
    if cached is None or not torch.equal(cached[0], sizes):
        offsets = torch.cumsum(sizes, 0) - sizes
        cols = torch.arange(batch).view(1, -1)
        valid = cols < sizes.view(-1, 1)
        index = (offsets.view(-1, 1) + cols).masked_fill(~valid, 0)
        lengths = valid.sum(dim=0)
        if device.type == 'cuda':
            n = index.numel()
            staged = _stage_to_device(
                [index.reshape(-1), valid.reshape(-1).to(torch.int64)], device
            )
            index_d, valid_d = staged[:n], staged[n:].bool()
        else:
            index_d, valid_d = index.reshape(-1), valid.reshape(-1)
        cached = (sizes.clone(), index_d, valid_d, lengths)
        _UNPACK_CACHE[key] = cached
    _, index, valid, lengths = cached
    data = packed.data
    padded = data[index].view(steps, batch, *data.shape[1:])
    mask = valid.view(steps, batch, *([1] * (data.dim() - 1)))
    padded = torch.where(mask, padded, torch.zeros((), dtype=data.dtype,
                                                   device=device))
    return padded, lengths.clone()


def to_np(x):
    return x.data.cpu().numpy()


def crop_start_stop(smiles, smiles_language):
    """utils/utils.py:271-284."""
    smiles = smiles.tolist()
    try:
        start_idx = smiles.index(smiles_language.start_index)
        stop_idx = smiles.index(smiles_language.stop_index)
        return smiles[start_idx + 1:stop_idx]
    except Exception:
        return smiles


def crop_start(smiles, smiles_language):
    """utils/utils.py:287-299."""
    smiles = smiles.tolist()
    try:
        start_idx = smiles.index(smiles_language.start_index)
        return smiles[start_idx + 1:]
    except Exception:
        return smiles


def packed_to_padded(seq, target_packed=None):
    """utils/utils.py:171-203: per-step outputs of a packed run (``seq[t]``
    holds one value per sequence still active at step t) -> f32[B, T] padded
    with 0.  Like the reference, the rows that have ended are derived from the
    STOP tokens (3) of the packed target: a row whose target at step t-1 was
    STOP is absent from step t+1 on.  Without ``target_packed`` the active rows
    of step t are the first ``len(seq[t])`` ones (the length-sorted layout)."""
    steps = len(seq)
    batch_size = len(seq[0])
    padded = torch.zeros(batch_size, steps)
    ended = torch.zeros(batch_size, dtype=torch.bool)
    if target_packed is not None and len(target_packed) == 0:
        target_packed = None
    for t in range(steps):
        row = torch.as_tensor(seq[t]).reshape(-1).to(padded.dtype)
        if target_packed is None:
            padded[:row.numel(), t] = row
            continue
        alive = (~ended).nonzero().view(-1)
        if alive.numel() != row.numel():
            raise ValueError(
                f'step {t}: {row.numel()} outputs for {alive.numel()} '
                'sequences without a STOP target so far'
            )
        padded[alive, t] = row
        if t > 0:
            prev = torch.as_tensor(target_packed[t - 1]).reshape(-1)
            # positions in the step's own (shrunken) list -> batch rows
            # (rows alive at t-1, two steps behind the STOP as in the reference)
            ended[alive_prev[(prev == 3).nonzero().view(-1)]] = True
        alive_prev = alive
    return padded


def print_example_reconstruction(
    reconstruction, inp, language, selfies=False, crop_stop=True
):
    """utils/utils.py:302-343: (target text, reconstructed text) of one random
    sample.  Any language object with ``token_indexes_to_smiles`` (or
    ``token_indexes_to_sequence``) is accepted -- pytoda is not a dependency."""
    if hasattr(language, 'token_indexes_to_smiles'):
        to_text = language.token_indexes_to_smiles
        if selfies and hasattr(language, 'selfies_to_smiles'):
            inner = to_text

            def to_text(tokens):
                return language.selfies_to_smiles(inner(tokens))
    elif hasattr(language, 'token_indexes_to_sequence'):
        to_text = language.token_indexes_to_sequence
    else:
        raise TypeError(f'Unknown language class: {type(language)}')
    import numpy as np
    sample_idx = np.random.randint(len(reconstruction))
    crop = crop_start_stop if crop_stop else crop_start
    reconstructed = crop(torch.as_tensor(reconstruction[sample_idx]).long(),
                         language)
    if isinstance(inp, torch.Tensor):
        inp = inp.permute(1, 0)
    elif not isinstance(inp, list):
        raise TypeError(f'Unknown input type {type(inp)}')
    sample = torch.as_tensor(inp[sample_idx]).tolist()
    return to_text(sample), to_text(reconstructed)


def disable_rdkit_logging():
    """utils/utils.py:346-352; a no-op when rdkit is not installed."""
    try:
        import rdkit.RDLogger as rkl
        import rdkit.rdBase as rkrb
    except ImportError:
        return
    rkl.logger().setLevel(rkl.ERROR)
    rkrb.DisableLog('rdApp.error')
