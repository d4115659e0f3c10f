"""The mixed-precision mode (pcm_set_matmul_precision(1)) on the GPU.

No parity with the reference is claimed for it (the reference trains in fp32
only, models/training.py:132-157): bf16-rounded MFMA operands give about three
significant digits per product.  What is checked: the switch takes effect (the
results differ from the fp32 run's), the results stay within bf16's rounding of
the fp32 run (loss, logits, every parameter gradient by relative L2 error and
direction), the mode is off again afterwards, and bad modes are rejected.
"""
import pytest
import torch

from oracle import weights as W
from test_gpu_parity import _build_vae, _dev

pytestmark = pytest.mark.gpu

CONFIG = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.2, 'batch_size': 512, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}
# tolerances of the bf16 mode against the fp32 run of the same product path
LOSS_RTOL = 2e-3
LOGIT_TOL = 5e-2     # max |difference| / max |logit|
GRAD_L2_TOL = 8e-2   # ||g_bf16 - g_fp32|| / ||g_fp32||
GRAD_COS_MIN = 0.995


def _run(p, sd, tokens, targets, eps, seq_len):
    from paccmann_chemistry_amd import vae_loss_function
    vae = _build_vae(p, sd)
    vae.encoder._next_seed = lambda: 0x1234567
    vae.decoder._next_seed = lambda: 0x7654321
    dev = _dev()
    dec_loss, mu, logvar = vae(tokens.to(dev), tokens.to(dev), targets.to(dev),
                               eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()
    grads = {n: q.grad.detach().clone() for n, q in vae.named_parameters()
             if q.grad is not None}
    return total.item(), vae.decoder.logits.detach().clone(), grads


@pytest.mark.parametrize('batch,seq_len', [(512, 120), (48, 17)])
def test_bf16_mode_tracks_fp32(batch, seq_len):
    """Both tile families: batch 512 (wide forward tiles, XPART / narrow reverse
    tiles) and a small ragged-edged batch (narrow 16-unit forward tiles)."""
    import paccmann_chemistry_amd as P
    p = dict(CONFIG, batch_size=batch)
    sd = W.make_state_dict(p, 5100)
    tokens = W.synthetic_tokens(seq_len, batch, p['vocab_size'], 5101)
    targets = W.shifted_targets(tokens)
    eps = torch.randn(batch, p['latent_dim'],
                      generator=torch.Generator().manual_seed(5102))
    assert P.get_matmul_precision() == 'fp32'
    loss32, logits32, grads32 = _run(p, sd, tokens, targets, eps, seq_len)
    try:
        P.set_matmul_precision('bf16')
        assert P.get_matmul_precision() == 'bf16'
        loss16, logits16, grads16 = _run(p, sd, tokens, targets, eps, seq_len)
    finally:
        P.set_matmul_precision('fp32')
    assert P.get_matmul_precision() == 'fp32'
    # the switch did something ...
    assert not torch.equal(logits16, logits32)
    # ... and stayed within bf16 rounding of the fp32 run
    assert abs(loss16 - loss32) < LOSS_RTOL * abs(loss32)
    scale = logits32.abs().max().item()
    assert (logits16 - logits32).abs().max().item() < LOGIT_TOL * scale
    bad = {}
    for name, g32 in grads32.items():
        g16 = grads16[name]
        n32 = g32.norm().item()
        if n32 == 0.0:
            assert g16.norm().item() == 0.0
            continue
        rel = (g16 - g32).norm().item() / n32
        cos = torch.dot(g16.flatten(), g32.flatten()).item() / (
            n32 * g16.norm().item())
        if not (rel < GRAD_L2_TOL and cos > GRAD_COS_MIN):
            bad[name] = (rel, cos)
    assert not bad, bad
    # fp32 again: bit-identical to the first fp32 run (deterministic kernels)
    loss_again, logits_again, _ = _run(p, sd, tokens, targets, eps, seq_len)
    assert loss_again == loss32 and torch.equal(logits_again, logits32)


def test_bad_precision_is_rejected():
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import _lib
    with pytest.raises(ValueError):
        P.set_matmul_precision('fp8')
    assert _lib.load().pcm_set_matmul_precision(7) != 0
    assert P.get_matmul_precision() == 'fp32'
