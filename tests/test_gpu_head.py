"""pcm_output_ce -- the output layer fused with softmax / cross entropy / argmax
(vae.py:342-348, 386-388) -- against the oracle's per-step criterion on CPU and
against the two-call path (pcm_linear_forward + pcm_softmax_ce) it replaces."""
import ctypes

import numpy as np
import pytest
import torch

from test_gpu_parity import _dev

pytestmark = pytest.mark.gpu


def _reference(x, w, b, targets, lengths, step_weight):
    """float64 restatement: sum over steps of the mean cross entropy over the
    active rows (or step_weight[t] * sum), logits, first-max argmax, dlogits."""
    T, B, _ = x.shape
    logits = x.double() @ w.double().t() + b.double()
    active = torch.ones(T, B, dtype=torch.bool)
    if lengths is not None:
        active = torch.arange(T)[:, None] < lengths[None, :]
    logp = torch.log_softmax(logits, dim=-1)
    nll = -logp.gather(-1, targets[..., None]).squeeze(-1)
    if step_weight is None:
        cnt = active.sum(1).clamp(min=1).double()
        wt = 1.0 / cnt
    else:
        wt = step_weight.double()
    loss = (nll * active * wt[:, None]).sum()
    onehot = torch.nn.functional.one_hot(targets, logits.shape[-1]).double()
    dlogits = (logp.exp() - onehot) * (active * wt[:, None])[..., None]
    argmax = torch.where(active, logits.argmax(-1), 0).t()
    return logits, loss, argmax, dlogits


@pytest.mark.parametrize('T,B,V,H,ragged,weighted', [
    (5, 7, 60, 512, False, False),      # the config head, ragged row count
    (120, 64, 60, 512, True, False),    # packed mode
    (9, 33, 128, 64, True, True),       # four column tiles, global step weights
    (3, 130, 33, 72, False, True),      # two tiles, odd vocabulary
    (4, 10, 7, 8, False, False),        # one tile, one k group
    (6, 20, 200, 64, True, False),      # V > 128: the two-call path inside
])
def test_output_ce_matches_reference_and_two_call_path(T, B, V, H, ragged,
                                                       weighted):
    from paccmann_chemistry_amd import _lib
    from paccmann_chemistry_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    dev = _dev()
    g = torch.Generator().manual_seed(T * 1000 + B * 10 + V)
    x = torch.randn(T, B, H, generator=g)
    w = torch.randn(V, H, generator=g) / np.sqrt(H)
    b = torch.randn(V, generator=g) * 0.1
    targets = torch.randint(0, V, (T, B), generator=g)
    lengths = None
    if ragged:
        lengths = torch.randint(1, T + 1, (B, ), generator=g)
        lengths[0] = T
        x = x * (torch.arange(T)[:, None] < lengths[None, :])[..., None]
    step_weight = torch.rand(T, generator=g) + 0.1 if weighted else None
    r_logits, r_loss, r_argmax, r_dlogits = _reference(
        x, w, b, targets, lengths, step_weight)

    # operands at odd 4-byte offsets of larger buffers, as parameters that live
    # inside a flat optimizer buffer are
    def odd(t, shift):
        flat = torch.empty(t.numel() + shift, device=dev)
        view = flat[shift:].view(t.shape)
        view.copy_(t)
        return view
    xd, wd, bd = odd(x, 3), odd(w, 1), odd(b, 2)
    td = targets.to(dev)
    ld = lengths.to(dev, torch.int32) if ragged else None
    swd = step_weight.to(dev) if weighted else None
    status = torch.zeros(1, dtype=torch.int32, device=dev)

    def run(fused, want_logits=True):
        logits = torch.full((T, B, V), float('nan'), device=dev)
        dlogits = torch.full((T, B, V), float('nan'), device=dev)
        loss = torch.empty((), device=dev)
        argmax = torch.full((B, T), -1, dtype=torch.int64, device=dev)
        scratch = torch.empty(T * B + T, device=dev)
        if fused:
            _lib.check(lib.pcm_output_ce(
                ptr(xd), H, ptr(wd), ptr(bd), ptr(td), ptr(ld), ptr(swd), T, B,
                V, H, ptr(logits) if want_logits else None, ptr(loss),
                ptr(argmax), ptr(dlogits), ptr(scratch), ptr(status),
                stream_ptr(dev)), 'pcm_output_ce')
        else:
            _lib.check(lib.pcm_linear_forward(
                ptr(xd), H, ptr(wd), ptr(bd), ptr(logits), V, T * B, V, H,
                stream_ptr(dev)), 'linear')
            _lib.check(lib.pcm_softmax_ce(
                ptr(logits), ptr(td), ptr(ld), ptr(swd), T, B, V, ptr(loss),
                ptr(argmax), ptr(dlogits), ptr(scratch), ptr(status),
                stream_ptr(dev)), 'ce')
        torch.cuda.synchronize()
        return logits.cpu(), loss.item(), argmax.cpu(), dlogits.cpu()

    logits, loss, argmax, dlogits = run(True)
    scale = r_logits.abs().max().item()
    assert (logits.double() - r_logits).abs().max().item() < 1e-5 * scale
    assert abs(loss - r_loss.item()) < 1e-5 * abs(r_loss.item())
    assert torch.equal(argmax, r_argmax)
    dscale = r_dlogits.abs().max().item()
    assert (dlogits.double() - r_dlogits).abs().max().item() < 1e-5 * dscale
    assert status.item() == 0

    l2, loss2, argmax2, dl2 = run(False)
    assert (logits - l2).abs().max().item() < 1e-5 * scale
    assert abs(loss - loss2) < 1e-6 * abs(loss2)
    assert torch.equal(argmax, argmax2)
    assert (dlogits - dl2).abs().max().item() < 1e-5 * dscale

    if V <= 128:
        # logits are optional in the fused kernel: nothing else changes
        nolog, loss3, argmax3, dl3 = run(True, want_logits=False)
        assert torch.isnan(nolog).all()
        assert loss3 == loss and torch.equal(argmax3, argmax)
        assert torch.equal(dl3, dlogits)
    else:
        logits_missing = lib.pcm_output_ce(
            ptr(xd), H, ptr(wd), ptr(bd), ptr(td), ptr(ld), ptr(swd), T, B, V,
            H, None, ptr(torch.empty((), device=dev)), None, None,
            ptr(torch.empty(T * B + T, device=dev)), ptr(status),
            stream_ptr(dev))
        assert logits_missing != 0


def test_output_ce_flags_bad_targets():
    from paccmann_chemistry_amd import _lib
    from paccmann_chemistry_amd._lib import ptr, stream_ptr
    lib = _lib.load()
    dev = _dev()
    T, B, V, H = 2, 3, 11, 16
    x = torch.randn(T, B, H, device=dev)
    w = torch.randn(V, H, device=dev)
    b = torch.zeros(V, device=dev)
    targets = torch.zeros(T, B, dtype=torch.int64, device=dev)
    targets[1, 2] = V   # out of range
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    loss = torch.empty((), device=dev)
    scratch = torch.empty(T * B + T, device=dev)
    _lib.check(lib.pcm_output_ce(
        ptr(x), H, ptr(w), ptr(b), ptr(targets), None, None, T, B, V, H, None,
        ptr(loss), None, None, ptr(scratch), ptr(status), stream_ptr(dev)),
        'pcm_output_ce')
    torch.cuda.synchronize()
    assert status.item() & _lib.STATUS_BAD_TOKEN
    assert np.isfinite(loss.item())
