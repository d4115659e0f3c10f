"""Pins the CPU oracle (oracle/stackgru_oracle.py) against outputs of the
reference itself (tests/golden, produced by oracle/make_golden.py).

Tolerances: the oracle and the reference run the same torch CPU kernels in
fp32 but group the work differently (sequence-level masks instead of
shrinking slices, one big logits matmul), so agreement is to a few ulps:
rtol 1e-5 / atol 1e-6 on activations, 2e-5 relative-to-max on gradients.
"""
import numpy as np
import pytest
import torch

from conftest import Golden, golden_names
from oracle import stackgru_oracle as O


def _close(a, b, rtol=1e-5, atol=1e-6):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'max err {err.max().item():.3e}'


def _grad_close(a, b, rel=2e-5):
    a, b = a.double(), b.double()
    scale = b.abs().max().clamp(min=1e-6)
    err = (a - b).abs().max() / scale
    assert err.item() <= rel, f'rel-to-max err {err.item():.3e}'


@pytest.mark.parametrize('name', golden_names('step_'))
def test_single_step(name):
    g = Golden(name)
    p = g.params
    out, hid, stack = O.stackgru_step(
        g.weights, g.tensor('token')[0], g.tensor('hidden'),
        g.tensor('stack'), use_stack=p['use_stack']
    )
    _close(out, g.data['output'][0])
    _close(hid, g.data['hidden_out'])
    _close(stack, g.data['stack_out'])


def _run_vae(g, dtype=torch.float32):
    p = g.params
    # gradients only where the reference has them (frozen one-hot
    # embeddings and the unused stack layers of use_stack=False have none)
    wanted = set(g.grads) | set(g.grad_norms)
    sd = {
        k: v.to(dtype).clone().requires_grad_(k in wanted)
        for k, v in g.weights.items()
    }
    loss, mu, logvar, logits, argmax = O.vae_forward(
        sd, g.tensor('tokens'), g.tensor('dec_tokens'), g.tensor('targets'),
        g.tensor('eps').to(dtype), p['stack_depth'], g.lengths,
        p['use_stack'], p['bidirectional']
    )
    total, kl = O.vae_loss(
        loss, mu, logvar, kl_growth=float(g.data['kl_growth']),
        step=int(g.data['step'])
    )
    return sd, loss, mu, logvar, logits, argmax, total, kl


@pytest.mark.parametrize(
    'name', [n for n in golden_names('vae_') if 'config' not in n]
)
def test_vae_forward_backward_small(name):
    g = Golden(name)
    sd, loss, mu, logvar, logits, argmax, total, kl = _run_vae(g)
    _close(mu, g.data['mu'])
    _close(logvar, g.data['logvar'])
    ref_logits = g.tensor('logits')
    if g.lengths is not None:
        steps = torch.arange(ref_logits.shape[0]).view(-1, 1)
        active = (g.lengths.view(1, -1) > steps).unsqueeze(-1)
        logits = logits * active
    _close(logits, ref_logits, rtol=1e-5, atol=2e-6)
    _close(loss, g.data['decoder_loss'], rtol=2e-6, atol=0)
    _close(kl, g.data['kl'], rtol=1e-5)
    _close(total, g.data['loss'], rtol=2e-6, atol=0)
    assert np.array_equal(
        argmax.numpy(), g.data['outputs'].astype(np.int64)
    )
    total.backward()
    assert g.grads
    for key, ref in g.grads.items():
        _grad_close(sd[key].grad, ref)


@pytest.mark.parametrize('name', golden_names('vae_config'))
def test_vae_config_shape(name):
    """H=512, V=60, D=W=50, L=2, Z=256, T=120 (weights from the numpy seed).
    Per-step logits to the north-star tolerance (1e-4 relative)."""
    g = Golden(name)
    sd, loss, mu, logvar, logits, argmax, total, kl = _run_vae(g)
    ref_logits = g.tensor('logits')
    if g.lengths is not None:
        steps = torch.arange(ref_logits.shape[0]).view(-1, 1)
        active = (g.lengths.view(1, -1) > steps).unsqueeze(-1)
        logits = logits * active
    scale = ref_logits.abs().max()
    assert ((logits - ref_logits).abs().max() / scale).item() < 1e-4
    _close(loss, g.data['decoder_loss'], rtol=5e-6, atol=0)
    _close(kl, g.data['kl'], rtol=1e-4)
    total.backward()
    for key, norm in g.grad_norms.items():
        got = sd[key].grad.double().norm().item()
        assert abs(got - norm) <= 1e-4 * max(norm, 1e-8), (key, got, norm)
        sample = sd[key].grad.reshape(-1)[::997]
        _grad_close(sample, g.grad_samples[key], rel=1e-4)


@pytest.mark.parametrize('name', golden_names('greedy_'))
def test_greedy_decode(name):
    g = Golden(name)
    p = g.params
    dec = O.sub_params(g.weights, 'decoder')
    tokens, logits = O.generate_from_latent(
        dec, g.tensor('z')[0], g.tensor('prime'),
        int(g.data['generate_len']), p['stack_depth'], p['use_stack']
    )
    ref_logits = g.tensor('logits')
    scale = ref_logits.abs().max()
    assert ((logits - ref_logits).abs().max() / scale).item() < 1e-4
    assert np.array_equal(tokens.numpy(), g.data['tokens'])


def test_kl_weight():
    g = Golden('kl_weight')
    for growth in (0.0015, 0.003, 0.004):
        got = [O.kl_weight(int(s), growth) for s in g.data['steps']]
        assert np.array_equal(np.array(got), g.data[f'g{growth}'])


def test_search_golden_tokens():
    """The reference's own golden vectors for GreedySearch
    (utils/tests/test_search.py:7-30)."""
    logits = torch.tensor(
        [[[0.1, 0.2, 0.3, 0.4, 0.5], [0.5, 0.4, 0.3, 0.2, 0.1]] * 5] * 2
    )
    truth = [4, 0] * logits.shape[2]
    for step_logits, token in zip(logits.permute(1, 0, 2), truth):
        got = O.greedy_step(step_logits)
        assert got.view(-1).tolist() == [token] * logits.shape[0]
    probs = O.sampling_probabilities(logits[:, 0], 0.8)
    assert torch.allclose(probs.sum(1), torch.ones(2))


def test_batch_independence_invariant():
    """Reference invariant (models/tests/test_vae.py:96-142): the same sample
    encodes to the same mu whatever the batch size."""
    from oracle import weights as W
    p = {
        'latent_dim': 128, 'rnn_cell_size': 32, 'embedding_size': 30,
        'vocab_size': 55, 'stack_width': 50, 'stack_depth': 51,
        'n_layers': 3, 'batch_size': 1, 'embedding': 'learned',
        'use_stack': True
    }
    sd = O.sub_params(W.make_state_dict(p, 5), 'encoder')
    sample = torch.tensor(np.concatenate([[2], np.arange(3, 53)]))
    results = []
    for batch in (1, 2, 12, 55):
        toks = sample.view(-1, 1).repeat(1, batch)
        mu, _ = O.encoder_forward(sd, toks, p['stack_depth'])
        results.append(mu[0])
        # no cross-talk inside the batch (test_vae.py:68-94)
        for row in mu:
            _close(row, mu[0], rtol=1e-5, atol=1e-7)
    for other in results[1:]:
        _close(other, results[0], rtol=1e-5, atol=1e-7)


def test_aten_cell_switch_matches_restatement():
    """ATEN_CELL (what bench.py's cpu_baseline times) changes the kernel of
    the layer-step, not the numbers."""
    from oracle import stackgru_oracle as O
    gen = torch.Generator().manual_seed(3)
    x, h = torch.randn(7, 13, generator=gen), torch.randn(7, 24, generator=gen)
    w_ih, w_hh = (torch.randn(72, 13, generator=gen),
                  torch.randn(72, 24, generator=gen))
    b_ih, b_hh = torch.randn(72, generator=gen), torch.randn(72, generator=gen)
    plain = O.gru_cell(x, h, w_ih, w_hh, b_ih, b_hh)
    O.ATEN_CELL = True
    try:
        aten = O.gru_cell(x, h, w_ih, w_hh, b_ih, b_hh)
    finally:
        O.ATEN_CELL = False
    assert torch.allclose(plain, aten, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name', golden_names('beam_'))
def test_beam_generation(name):
    """The oracle's beam branch (vae.py:439-488 + search.py:169-275) against
    the reference's own beams: best sequence, every beam and its score."""
    g = Golden(name)
    p = g.params
    dec = O.sub_params(g.weights, 'decoder')
    with torch.no_grad():
        best, beams = O.generate_from_latent_beam(
            dec, g.tensor('z')[0], g.tensor('prime'),
            int(g.data['generate_len']), p['stack_depth'],
            int(g.data['beam_width']), int(g.data['top_tokens']),
            float(g.data['temperature']), p.get('use_stack', True))
    assert np.array_equal(best.numpy(), g.data['tokens'])
    got_tokens = np.array([[beam[0] for beam in row] for row in beams])
    got_scores = np.array([[beam[1] for beam in row] for row in beams])
    assert np.array_equal(got_tokens, g.data['beam_tokens'])
    assert np.allclose(got_scores, g.data['beam_scores'], rtol=1e-6, atol=0)
