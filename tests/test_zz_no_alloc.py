"""pcm_softmax_ce_static -- the entry point that allocates nothing -- against
pcm_softmax_ce (caller workspace) through the C ABI: same loss, argmax and
dlogits bit for bit, on two calls in a row (the second on a smaller input,
sharing the GPU's static scratch array), and a clear error past the maximum.

Collected last on purpose (the round-end run uses `pytest -x`)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(lib, steps, batch, vocab, ragged, seed, dev, weights=False):
    from paccmann_chemistry_amd._lib import check, ptr, stream_ptr
    rng = np.random.default_rng(seed)
    logits = torch.from_numpy(
        rng.standard_normal((steps, batch, vocab)).astype(np.float32) * 3
    ).to(dev)
    targets = torch.from_numpy(
        rng.integers(0, vocab, size=(steps, batch), dtype=np.int64)).to(dev)
    lengths = None
    if ragged:
        lengths = torch.from_numpy(np.sort(rng.integers(
            1, steps + 1, size=batch))[::-1].astype(np.int32).copy()).to(dev)
    step_weight = None
    if weights:
        step_weight = torch.from_numpy(
            rng.random(steps).astype(np.float32)).to(dev)
    outs = []
    for static in (False, True):
        loss = torch.full((1, ), -7.0, device=dev)
        argmax = torch.full((batch, steps), -1, dtype=torch.int64, device=dev)
        dlogits = torch.full_like(logits, 9.0)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        if static:
            check(lib.pcm_softmax_ce_static(
                ptr(logits), ptr(targets), ptr(lengths), ptr(step_weight),
                steps, batch, vocab, ptr(loss), ptr(argmax), ptr(dlogits),
                ptr(status), stream_ptr(dev)), 'pcm_softmax_ce_static')
        else:
            scratch = torch.empty(steps * batch + steps, device=dev)
            check(lib.pcm_softmax_ce(
                ptr(logits), ptr(targets), ptr(lengths), ptr(step_weight),
                steps, batch, vocab, ptr(loss), ptr(argmax), ptr(dlogits),
                ptr(scratch), ptr(status), stream_ptr(dev)), 'pcm_softmax_ce')
        torch.cuda.synchronize()
        outs.append((loss.cpu(), argmax.cpu(), dlogits.cpu()))
    (l0, a0, d0), (l1, a1, d1) = outs
    assert torch.isfinite(l0).all() and l0.item() > 0
    assert torch.equal(l0, l1)
    assert torch.equal(a0, a1)
    assert torch.equal(d0, d1)
    # and the value itself: sum over steps of the mean CE over active rows
    lg = logits.cpu().double()
    nll = -torch.log_softmax(lg, -1).gather(
        2, targets.cpu().unsqueeze(-1)).squeeze(-1)
    if ragged:
        active = (lengths.cpu().view(1, -1) >
                  torch.arange(steps).view(-1, 1)).double()
    else:
        active = torch.ones(steps, batch, dtype=torch.float64)
    if weights:
        per_step = step_weight.cpu().double()
    else:
        per_step = 1.0 / active.sum(1).clamp(min=1)
    want = ((nll * active).sum(1) * per_step).sum().item()
    assert abs(l1.item() - want) <= 1e-5 * abs(want)


def test_softmax_ce_static_matches_workspace_variant():
    from paccmann_chemistry_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    torch.zeros(1, device=dev)
    # config-2 head shape first, then a smaller ragged one on the same array
    _case(lib, 120, 512, 60, False, 1, dev)
    _case(lib, 17, 37, 23, True, 2, dev)
    _case(lib, 9, 5, 300, True, 3, dev, weights=True)


def test_softmax_ce_static_rejects_oversize():
    from paccmann_chemistry_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    one = torch.zeros(4, device=dev)
    tgt = torch.zeros(4, dtype=torch.int64, device=dev)
    # T*B + T = 1024 * 1024 + 1024 > 2^20: refused before any launch
    rc = lib.pcm_softmax_ce_static(
        one.data_ptr(), tgt.data_ptr(), None, None, 1024, 1024, 2,
        one.data_ptr(), None, None, None, None)
    assert rc == -1
    msg = lib.pcm_last_error().decode()
    assert 'static maximum' in msg and str(1 << 20) in msg


def test_softmax_ce_static_allocates_nothing_and_takes_the_maximum():
    """The property the entry point exists for: device memory is the same
    before and after a call (torch's allocator statistics AND the driver's free
    memory), at the largest accepted shape, T*B + T == 2^20 floats of scratch.
    pcm_device_check() resolves the array's address beforehand, as the header
    asks of callers that will capture."""
    from paccmann_chemistry_amd import _lib
    from paccmann_chemistry_amd._lib import check, ptr, stream_ptr
    lib = _lib.load()
    dev = torch.device('cuda:0')
    steps, batch, vocab = 1024, 1023, 2
    assert steps * batch + steps == 1 << 20
    rng = np.random.default_rng(4)
    logits = torch.from_numpy(
        rng.standard_normal((steps, batch, vocab)).astype(np.float32)).to(dev)
    targets = torch.from_numpy(
        rng.integers(0, vocab, size=(steps, batch), dtype=np.int64)).to(dev)
    loss = torch.zeros(1, device=dev)
    ref_loss = torch.zeros(1, device=dev)
    scratch = torch.empty(steps * batch + steps, device=dev)
    check(lib.pcm_device_check(), 'pcm_device_check')
    check(lib.pcm_softmax_ce(
        ptr(logits), ptr(targets), None, None, steps, batch, vocab,
        ptr(ref_loss), None, None, ptr(scratch), None, stream_ptr(dev)),
        'pcm_softmax_ce')
    torch.cuda.synchronize()
    before = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev),
              torch.cuda.mem_get_info(dev)[0])
    for _ in range(3):
        check(lib.pcm_softmax_ce_static(
            ptr(logits), ptr(targets), None, None, steps, batch, vocab,
            ptr(loss), None, None, None, stream_ptr(dev)),
            'pcm_softmax_ce_static')
    torch.cuda.synchronize()
    after = (torch.cuda.memory_allocated(dev), torch.cuda.memory_reserved(dev),
             torch.cuda.mem_get_info(dev)[0])
    assert after == before
    assert torch.equal(loss.cpu(), ref_loss.cpu())
    # one float more is refused
    rc = lib.pcm_softmax_ce_static(
        ptr(logits), ptr(targets), None, None, steps, batch + 1, vocab,
        ptr(loss), None, None, None, stream_ptr(dev))
    assert rc == -1


def test_softmax_ce_static_under_stream_capture():
    """A caller the entry point is for: the call is recorded into a HIP graph
    (no allocation is legal there) and replayed on fresh inputs."""
    from paccmann_chemistry_amd import _lib
    from paccmann_chemistry_amd._lib import check, ptr, stream_ptr
    lib = _lib.load()
    dev = torch.device('cuda:0')
    steps, batch, vocab = 12, 40, 33
    rng = np.random.default_rng(5)
    logits = torch.zeros(steps, batch, vocab, device=dev)
    targets = torch.zeros(steps, batch, dtype=torch.int64, device=dev)
    loss = torch.zeros(1, device=dev)
    dlogits = torch.zeros_like(logits)
    argmax = torch.zeros(batch, steps, dtype=torch.int64, device=dev)
    check(lib.pcm_device_check(), 'pcm_device_check')   # address resolved here
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        check(lib.pcm_softmax_ce_static(
            ptr(logits), ptr(targets), None, None, steps, batch, vocab,
            ptr(loss), ptr(argmax), ptr(dlogits), None, stream_ptr(dev)),
            'pcm_softmax_ce_static (captured)')
    for seed in (6, 7):
        rng = np.random.default_rng(seed)
        logits.copy_(torch.from_numpy(rng.standard_normal(
            (steps, batch, vocab)).astype(np.float32) * 2))
        targets.copy_(torch.from_numpy(
            rng.integers(0, vocab, size=(steps, batch), dtype=np.int64)))
        graph.replay()
        torch.cuda.synchronize()
        want_loss = torch.zeros(1, device=dev)
        want_d = torch.zeros_like(logits)
        want_a = torch.zeros_like(argmax)
        scratch = torch.empty(steps * batch + steps, device=dev)
        check(lib.pcm_softmax_ce(
            ptr(logits), ptr(targets), None, None, steps, batch, vocab,
            ptr(want_loss), ptr(want_a), ptr(want_d), ptr(scratch), None,
            stream_ptr(dev)), 'pcm_softmax_ce')
        torch.cuda.synchronize()
        assert torch.equal(loss, want_loss)
        assert torch.equal(argmax, want_a)
        assert torch.equal(dlogits, want_d)
