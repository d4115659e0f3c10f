"""GPU parity at the sizes that are benchmarked (BASELINE.json configs 2, 4, 5).

The launchers pick kernel variants by batch size (two blocks per CU for the
layer-1 reverse cell, the 256-block layer-0 variant, one-row stack blocks, the
MFMA stack step from 4096 rows on), so the variants the bench times are
checked here at their real shapes:

* B=512 x T=120 training step (forward + backward), every parameter gradient
  against the oracle's autograd, without and with inter-layer dropout
  (models/vae.py:571-599, training.py:147-153);
* SamplingSearch.step (utils/search.py:124-141) bit-exact against the oracle's
  distribution with the counter-hash uniforms regenerated in numpy;
* B=4096 encoder-only and greedy decode rows against the same rows at B=4,
  which the oracle checks.
"""
import numpy as np
import pytest
import torch

from oracle import stackgru_oracle as O
from oracle import weights as W
from test_gpu_parity import (
    GRAD_TOL, LOGIT_TOL, _build_vae, _cpu_dropout_masks, _dev, _rel_to_max
)

pytestmark = pytest.mark.gpu

CONFIG = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.0, 'batch_size': 512, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}
SEQ_LEN = 120


def _mix(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _uniform(seed, counters):
    """numpy restatement of pcm_uniform (csrc/common.h)."""
    with np.errstate(over='ignore'):
        h = _mix(np.uint64(seed) ^ _mix(counters.astype(np.uint64)))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(1 / 16777216)


@pytest.mark.parametrize('dropout', [0.0, 0.2])
def test_full_size_training_step_matches_oracle(dropout):
    """The timed configuration itself: B=512, T=120, train mode.  Loss, mu,
    per-step logits and EVERY parameter gradient against the oracle; with
    dropout 0.2 the device masks (a pure function of seed, layer, t, b, j) are
    regenerated in numpy and fed to the oracle."""
    from paccmann_chemistry_amd import vae_loss_function
    p = dict(CONFIG, dropout=dropout)
    batch = p['batch_size']
    sd = W.make_state_dict(p, 4100)
    tokens = W.synthetic_tokens(SEQ_LEN, batch, p['vocab_size'], 4101)
    targets = W.shifted_targets(tokens)
    keep = np.random.default_rng(4102).random(SEQ_LEN) < 0.85
    dec_tokens = tokens.clone()
    dec_tokens[torch.from_numpy(~keep)] = 1   # whole-row word dropout
    eps = torch.from_numpy(np.random.default_rng(4103).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    vae = _build_vae(p, sd)
    seeds = {'encoder': 0x1234567, 'decoder': 0x7654321}
    vae.encoder._next_seed = lambda: seeds['encoder']
    vae.decoder._next_seed = lambda: seeds['decoder']
    dev = _dev()
    dec_loss, mu, logvar = vae(tokens.to(dev), dec_tokens.to(dev),
                               targets.to(dev), eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()

    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    enc_masks = dec_masks = None
    if dropout > 0:
        enc_masks = _cpu_dropout_masks(seeds['encoder'], SEQ_LEN, 2, batch,
                                       512, dropout)
        dec_masks = _cpu_dropout_masks(seeds['decoder'], SEQ_LEN, 2, batch,
                                       512, dropout)
    r_mu, r_logvar = O.encoder_forward(
        O.sub_params(ref, 'encoder'), tokens, p['stack_depth'], None, True,
        enc_masks)
    z = O.reparameterize(r_mu, r_logvar, eps)
    r_loss, r_logits, r_argmax = O.decoder_forward(
        O.sub_params(ref, 'decoder'), z, dec_tokens, targets,
        p['stack_depth'], None, True, dec_masks)
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()

    assert _rel_to_max(mu, r_mu) < 1e-5
    assert _rel_to_max(vae.decoder.logits, r_logits) < LOGIT_TOL
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    bad = {}
    for name, param in vae.named_parameters():
        if ref[name].grad is None:
            continue
        err = _rel_to_max(param.grad, ref[name].grad)
        if not err < GRAD_TOL:
            bad[name] = err
    assert not bad, bad


@pytest.mark.parametrize('layers,depth,width,hidden,batch', [
    (1, 3, 5, 64, 40),      # one layer: no rider host, the fused adjoint launch
    (3, 3, 5, 64, 48),      # a middle layer between the rider's two hosts
    (2, 1, 4, 64, 40),      # stack of depth 1: no row 1 for the pop term
    (2, 6, 61, 128, 33),    # the widest stack the split adjoint takes, odd batch
    (2, 4, 62, 64, 32),     # one wider: the fused adjoint launch
    (2, 5, 7, 512, 512),    # hidden 512 / batch 512: forward stack step fused
                            # into the cells, small stack
], ids=['l1', 'l3', 'depth1', 'w61', 'w62', 'b512_small_stack'])
def test_fast_path_edge_shapes_match_oracle(layers, depth, width, hidden, batch):
    """The kernels of the hidden % 64 == 0 path at the edges of what round 3's
    two restructurings accept (the stack adjoint as rider blocks + finish launch:
    two or more layers, stack width <= 61; the forward stack step fused into
    the cells: one block per CU): a short training step, every gradient
    against the oracle's autograd."""
    from paccmann_chemistry_amd import vae_loss_function
    steps = 9
    p = dict(CONFIG, n_layers=layers, stack_depth=depth, stack_width=width,
             rnn_cell_size=hidden, batch_size=batch, latent_dim=16,
             vocab_size=20, embedding_size=20)
    sd = W.make_state_dict(p, 4300 + layers + depth)
    tokens = W.synthetic_tokens(steps, batch, p['vocab_size'], 4301)
    targets = W.shifted_targets(tokens)
    eps = torch.from_numpy(np.random.default_rng(4302).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    vae = _build_vae(p, sd)
    dev = _dev()
    dec_loss, mu, logvar = vae(tokens.to(dev), tokens.to(dev), targets.to(dev),
                               eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()
    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    r_mu, r_logvar = O.encoder_forward(
        O.sub_params(ref, 'encoder'), tokens, p['stack_depth'])
    z = O.reparameterize(r_mu, r_logvar, eps)
    r_loss, r_logits, _ = O.decoder_forward(
        O.sub_params(ref, 'decoder'), z, tokens, targets, p['stack_depth'])
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()
    assert _rel_to_max(mu, r_mu) < 1e-5
    assert _rel_to_max(vae.decoder.logits, r_logits) < LOGIT_TOL
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    bad = {}
    for name, param in vae.named_parameters():
        if ref[name].grad is None:
            continue
        err = _rel_to_max(param.grad, ref[name].grad)
        if not err < GRAD_TOL:
            bad[name] = err
    assert not bad, bad


@pytest.mark.parametrize('shrink', [True, False],
                         ids=['active_prefix', 'frozen_rows'])
def test_packed_training_step_matches_oracle(shrink, monkeypatch):
    """Packed mode at a batch the K-major weight-gradient kernel takes (B=256,
    T=48, lengths 2..48, batch_sizes shrinking to 3 rows): the launches walk
    the still-running prefix of the batch and the weight-gradient products
    skip the 16-row chunks no sequence reaches (PCM_ROW_SKIP=0 multiplies the
    zeros instead; test_opt_in_recurrence_variants runs this test that way).
    Loss, mu, logits of the real steps and every gradient against the oracle's
    packed semantics (vae.py:138-236, 352-389).  frozen_rows: the same batch
    through the all-rows path (finished rows frozen in place, what unsorted
    lengths get), i.e. the length checks of the hidden-512 kernels -- the
    reverse cells, the stack adjoint's rider blocks and its finish launch."""
    from paccmann_chemistry_amd import ops, vae_loss_function
    if not shrink:
        monkeypatch.setattr(ops, 'SHRINK_ACTIVE_BATCH', False)
    batch, steps = 256, 48
    p = dict(CONFIG, batch_size=batch, batch_mode='packed')
    sd = W.make_state_dict(p, 4150)
    tokens = W.synthetic_tokens(steps, batch, p['vocab_size'], 4151)
    targets = W.shifted_targets(tokens)
    rng = np.random.default_rng(4152)
    lens = np.sort(rng.integers(2, steps + 1, size=batch))[::-1].copy()
    lens[:3] = steps
    lengths = torch.from_numpy(lens.astype(np.int64))
    eps = torch.from_numpy(np.random.default_rng(4153).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    vae = _build_vae(p, sd)
    dev = _dev()
    pack = torch.nn.utils.rnn.pack_sequence

    def packed(x):
        return pack([x[:n, b] for b, n in enumerate(lens.tolist())]).to(dev)

    dec_loss, mu, logvar = vae(packed(tokens), packed(tokens), packed(targets),
                               eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()

    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    r_mu, r_logvar = O.encoder_forward(
        O.sub_params(ref, 'encoder'), tokens, p['stack_depth'], lengths)
    z = O.reparameterize(r_mu, r_logvar, eps)
    r_loss, r_logits, _ = O.decoder_forward(
        O.sub_params(ref, 'decoder'), z, tokens, targets, p['stack_depth'],
        lengths)
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()

    assert _rel_to_max(mu, r_mu) < 1e-5
    active = (lengths.view(1, -1) > torch.arange(steps).view(-1, 1))
    assert _rel_to_max(vae.decoder.logits.detach().cpu() * active.unsqueeze(-1),
                       r_logits.detach() * active.unsqueeze(-1)) < LOGIT_TOL
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    bad = {}
    for name, param in vae.named_parameters():
        if ref[name].grad is None:
            continue
        err = _rel_to_max(param.grad, ref[name].grad)
        if not err < GRAD_TOL:
            bad[name] = err
    assert not bad, bad


def test_example_params_shape_matches_oracle():
    """The reference's examples/example_params.json model (vocabulary 380
    one-hot, H=512, L=2, stack 50x50, latent 256, packed mode) at batch 64 x
    T=40: the 380-wide vocabulary takes the paths the 60-token benchmark model
    does not -- output layer and cross entropy as two launches (V > 128), the
    token-table gradient as a product of its own (V > 64), the generic output
    layer + search step in decoding.  Loss, mu, every gradient and 30 greedy
    tokens against the oracle."""
    from paccmann_chemistry_amd import GreedySearch, vae_loss_function
    batch, steps = 64, 40
    p = dict(CONFIG, batch_size=batch, batch_mode='packed', vocab_size=380,
             embedding_size=380)
    sd = W.make_state_dict(p, 4500)
    tokens = W.synthetic_tokens(steps, batch, 380, 4501)
    targets = W.shifted_targets(tokens)
    rng = np.random.default_rng(4502)
    lens = np.sort(rng.integers(5, steps + 1, size=batch))[::-1].copy()
    lens[0] = steps
    lengths = torch.from_numpy(lens.astype(np.int64))
    eps = torch.from_numpy(np.random.default_rng(4503).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    vae = _build_vae(p, sd)
    dev = _dev()
    pack = torch.nn.utils.rnn.pack_sequence

    def packed(x):
        return pack([x[:n, b] for b, n in enumerate(lens.tolist())]).to(dev)

    dec_loss, mu, logvar = vae(packed(tokens), packed(tokens), packed(targets),
                               eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()

    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    r_mu, r_logvar = O.encoder_forward(
        O.sub_params(ref, 'encoder'), tokens, p['stack_depth'], lengths)
    z = O.reparameterize(r_mu, r_logvar, eps)
    r_loss, r_logits, _ = O.decoder_forward(
        O.sub_params(ref, 'decoder'), z, tokens, targets, p['stack_depth'],
        lengths)
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()
    assert _rel_to_max(mu, r_mu) < 1e-5
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    bad = {}
    for name, param in vae.named_parameters():
        if ref[name].grad is None:
            continue
        err = _rel_to_max(param.grad, ref[name].grad)
        if not err < GRAD_TOL:
            bad[name] = err
    assert not bad, bad

    vae.eval()
    zg = torch.from_numpy(np.random.default_rng(4504).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    prime, gen_len = torch.tensor([2]), 30
    with torch.no_grad():
        toks, logits = vae.decoder.generate_from_latent(
            zg.unsqueeze(0).to(dev), prime.to(dev), torch.tensor([3]),
            search=GreedySearch(), generate_len=gen_len, return_logits=True)
        r_toks, r_logits = O.generate_from_latent(
            O.sub_params(sd, 'decoder'), zg, prime, gen_len, p['stack_depth'])
    assert _rel_to_max(logits, r_logits) < LOGIT_TOL
    assert torch.equal(toks.cpu(), r_toks)


def test_sampling_tokens_bit_exact_against_oracle_distribution():
    """SamplingSearch.step at the config model (V=60, H=512), B=70: the device
    draw is a counter hash of (seed, step, row); with the same uniforms the
    token must be the first index whose cumulative
    oracle.sampling_probabilities(logits, T) exceeds the draw, at every step
    of a 12-token generation (the oracle replays the generation with that
    chooser, so the per-step logits are compared too)."""
    import paccmann_chemistry_amd as P
    p = dict(CONFIG, batch_size=70)
    temperature, seed, gen_len = 0.8, 20241111, 12
    sd = W.make_state_dict(p, 4200)
    vae = _build_vae(p, sd, train=False)
    z = torch.from_numpy(np.random.default_rng(4201).standard_normal(
        (70, p['latent_dim'])).astype(np.float32))
    prime = torch.tensor([2])
    tokens, logits = vae.decoder.generate_from_latent(
        z.unsqueeze(0).to(_dev()), prime.to(_dev()), torch.tensor([3]),
        search=P.SamplingSearch(temperature=temperature),
        generate_len=gen_len, seed=seed, return_logits=True
    )
    step_box = {'i': 0}
    margins = []

    def chooser(step_logits):
        i = step_box['i']
        step_box['i'] += 1
        rows = np.arange(70, dtype=np.uint64)
        u = _uniform(seed, (np.uint64(i) << np.uint64(32)) | rows)
        probs = O.sampling_probabilities(step_logits, temperature)
        cum = torch.cumsum(probs.double(), dim=1).numpy()
        pick = (cum > u[:, None].astype(np.float64)).argmax(axis=1)
        none = ~(cum > u[:, None]).any(axis=1)
        pick[none] = cum.shape[1] - 1
        margins.append(np.abs(cum - u[:, None]).min(axis=1))
        return torch.from_numpy(pick).view(-1, 1)

    with torch.no_grad():
        r_tokens, r_logits = O.generate_from_latent(
            O.sub_params(sd, 'decoder'), z, prime, gen_len, p['stack_depth'],
            True, chooser)
    margin = float(np.min(margins))
    assert torch.equal(tokens.cpu(), r_tokens), (
        f'smallest |cumulative - draw| over the run: {margin:.3e}')
    assert _rel_to_max(logits, r_logits) < LOGIT_TOL
    # the draws do move the choice away from the argmax
    greedy = r_logits.argmax(dim=2).t()
    assert (r_tokens[:, 1:] != greedy).float().mean().item() > 0.2


def test_batch_4096_rows_match_small_batch():
    """Config 4 / 5 shape (B=4096): encoder-only mu / logvar and 100 greedy
    tokens.  Rows 0, 1023, 2048, 4095 of the big batch against the same
    sequences / latents at B=4, and those against the oracle.  Covers the
    MFMA stack step (>= 4096 rows) and the many-blocks-per-CU forward cell."""
    from paccmann_chemistry_amd import GreedySearch
    rows = [0, 1023, 2048, 4095]
    p = dict(CONFIG, batch_size=4096)
    sd = W.make_state_dict(p, 4300)
    tokens = W.synthetic_tokens(SEQ_LEN, 4096, p['vocab_size'], 4301)
    z = torch.from_numpy(np.random.default_rng(4302).standard_normal(
        (4096, p['latent_dim'])).astype(np.float32))
    prime, gen_len = torch.tensor([2]), 100
    big = _build_vae(p, sd, train=False)
    dev = _dev()
    with torch.no_grad():
        mu, logvar = big.encode(tokens.to(dev))
        toks, logits = big.decoder.generate_from_latent(
            z.unsqueeze(0).to(dev), prime.to(dev), torch.tensor([3]),
            search=GreedySearch(), generate_len=gen_len, return_logits=True)
    big.check_tokens()
    mu, logvar = mu.cpu()[rows], logvar.cpu()[rows]
    toks, logits = toks.cpu()[rows], logits.cpu()[:, rows]
    del big
    small = _build_vae(dict(p, batch_size=4), sd, train=False)
    with torch.no_grad():
        mu_s, logvar_s = small.encode(tokens[:, rows].to(dev))
        toks_s, logits_s = small.decoder.generate_from_latent(
            z[rows].unsqueeze(0).to(dev), prime.to(dev), torch.tensor([3]),
            search=GreedySearch(), generate_len=gen_len, return_logits=True)
    assert _rel_to_max(mu, mu_s) < 1e-5
    assert _rel_to_max(logvar, logvar_s) < 1e-5
    assert torch.equal(toks, toks_s.cpu())
    assert _rel_to_max(logits, logits_s) < 1e-5
    # the small batch against the oracle
    with torch.no_grad():
        r_mu, r_logvar = O.encoder_forward(
            O.sub_params(sd, 'encoder'), tokens[:, rows], p['stack_depth'])
        r_toks, r_logits = O.generate_from_latent(
            O.sub_params(sd, 'decoder'), z[rows], prime, gen_len,
            p['stack_depth'])
    assert _rel_to_max(mu_s, r_mu) < 1e-4
    assert _rel_to_max(logits_s, r_logits) < LOGIT_TOL
    assert torch.equal(toks_s.cpu(), r_toks)


def test_sampling_at_batch_4096_rows_match_oracle():
    """SamplingSearch.step at the benchmarked decode batch (config 5: B=4096,
    temperature 0.8): logits_choose_kernel runs a 256-block grid there, 5 blocks
    at the B=70 test above.  Rows 0, 1023, 2048, 4095 of a 20-token generation
    against the oracle replaying those four rows with the draws the device
    hashes from (seed, step, GLOBAL row) regenerated in numpy: tokens bit for
    bit, per-step logits within tolerance."""
    import paccmann_chemistry_amd as P
    rows = [0, 1023, 2048, 4095]
    p = dict(CONFIG, batch_size=4096)
    temperature, seed, gen_len = 0.8, 20261111, 20
    sd = W.make_state_dict(p, 4310)
    z = torch.from_numpy(np.random.default_rng(4311).standard_normal(
        (4096, p['latent_dim'])).astype(np.float32))
    prime = torch.tensor([2])
    big = _build_vae(p, sd, train=False)
    with torch.no_grad():
        tokens, logits = big.decoder.generate_from_latent(
            z.unsqueeze(0).to(_dev()), prime.to(_dev()), torch.tensor([3]),
            search=P.SamplingSearch(temperature=temperature),
            generate_len=gen_len, seed=seed, return_logits=True)
    big.check_tokens()
    tokens, logits = tokens.cpu()[rows], logits.cpu()[:, rows]
    step_box = {'i': 0}

    def chooser(step_logits):
        i = step_box['i']
        step_box['i'] += 1
        u = _uniform(seed, (np.uint64(i) << np.uint64(32)) |
                     np.asarray(rows, dtype=np.uint64))
        probs = O.sampling_probabilities(step_logits, temperature)
        cum = torch.cumsum(probs.double(), dim=1).numpy()
        above = cum > u[:, None].astype(np.float64)
        pick = above.argmax(axis=1)
        pick[~above.any(axis=1)] = cum.shape[1] - 1
        return torch.from_numpy(pick).view(-1, 1)

    with torch.no_grad():
        r_tokens, r_logits = O.generate_from_latent(
            O.sub_params(sd, 'decoder'), z[rows], prime, gen_len,
            p['stack_depth'], True, chooser)
    assert torch.equal(tokens, r_tokens)
    assert _rel_to_max(logits, r_logits) < LOGIT_TOL
    assert (r_tokens[:, 1:] != r_logits.argmax(dim=2).t()).any()


def test_large_stack_gradients_and_limit():
    """Stack sizes beyond the tuned 50x50: a 64x64 stack (one row per block in
    both stack kernels) trains with oracle-matching gradients; a stack whose
    row no longer fits the reverse pass's LDS row cache is refused by the
    training-mode forward, before any launch, while the eval-mode forward
    (smaller footprint) still runs and matches."""
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd._lib import PaccmannHipError
    base = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 64, 'stack_depth': 64, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 6, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    steps = 5
    sd = W.make_state_dict(base, 4400)
    enc_sd = O.sub_params(sd, 'encoder')
    tokens = W.synthetic_tokens(steps, 6, 20, 4401)
    enc = P.StackGRUEncoder(base)
    enc.load_state_dict(enc_sd)
    enc = enc.to(_dev()).train()
    mu, logvar = enc.encoder_train_step(tokens.to(_dev()))
    (mu.sum() + 0.5 * logvar.sum()).backward()
    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in enc_sd.items()}
    r_mu, r_logvar = O.encoder_forward(ref, tokens, 64)
    (r_mu.sum() + 0.5 * r_logvar.sum()).backward()
    assert _rel_to_max(mu, r_mu) < 1e-5
    for name, param in enc.named_parameters():
        if ref[name].grad is not None:
            assert _rel_to_max(param.grad, ref[name].grad) < GRAD_TOL, name

    # 130 x 64: 8320 floats per row; the reverse pass would need 2 x that
    huge = dict(base, stack_depth=130)
    sd = W.make_state_dict(huge, 4402)
    enc_sd = O.sub_params(sd, 'encoder')
    enc = P.StackGRUEncoder(huge)
    enc.load_state_dict(enc_sd)
    enc = enc.to(_dev()).train()
    with pytest.raises(PaccmannHipError, match='LDS row cache'):
        enc.encoder_train_step(tokens.to(_dev()))
    enc.eval()
    with torch.no_grad():
        mu, _ = enc.encoder_train_step(tokens.to(_dev()))
        r_mu, _ = O.encoder_forward(enc_sd, tokens, 130)
    assert _rel_to_max(mu, r_mu) < 1e-5


@pytest.mark.parametrize('env', [
    {'PCM_PERSIST': '1'},
    {'PCM_PERSIST': '2'},
    {'PCM_CHAINS': '2'},
    {'PCM_CHAINS': '2', 'PCM_DW_CUTS': '45,75,90'},
    {'PCM_FUSE_STACK': '0'},
    {'PCM_ROW_SKIP': '0'},
    {'PCM_STACK_SPLIT': '0'},
    # (switches that only touch one shape run the tests of that shape)
    {'PCM_STACK_SPLIT64': '8', '-k': 'batch_4096'},
    {'PCM_FWD_HOIST': '1', '-k': 'full_size_training or vae_config_golden'},
    # the measured-slower placements of the split stack adjoint: half the rider
    # rows in the top layer's launch, two rows per rider block, four rows per
    # finish block
    {'PCM_STACK_RIDER_TOP': '50', 'PCM_STACK_RIDER_PER': '2',
     'PCM_STACK_FINISH_ROWS': '4',
     '-k': 'full_size_training or packed_training_step'},
])
def test_opt_in_recurrence_variants(env):
    """The forms of the recurrence that are built but not the default (measured
    slower than, or level with, the launch chain at batch 512): the
    persistent weight-stationary forward kernel with in-launch group hand-offs
    (PCM_PERSIST=1), the batch walked as two half-batch chains on two
    streams (PCM_CHAINS=2) and the forward stack step as a launch of its own
    (PCM_FUSE_STACK=0: three launches per forward step; since round 3 the
    default at batch 512 fuses it into the cells); PCM_ROW_SKIP=0 is the
    packed-mode weight-gradient reduction over all rows, zeros included;
    PCM_STACK_SPLIT=0 the stack adjoint as one launch on the chain (the
    default since round 3 splits it: bulk as rider blocks of the layer-0
    reverse cell, stack-top terms + gate gradient on the chain);
    PCM_STACK_SPLIT64=8 the large-batch forward stack step as a top-only
    launch + update riders in the 64 x 64 top cell (measured slower);
    PCM_FWD_HOIST=1 layer 1's hidden-side product as extra blocks of layer
    0's fused launch (measured level to slower).  The switches are read once per process, hence a
    child process: the full-size training step (every gradient against the
    oracle, with and without dropout), the B=4096 rows and the packed / padded
    data-parallel shard test."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(env)
    select = env.pop('-k', 'full_size_training or batch_4096 or shards_sum or '
                           'determinism or vae_config_golden or '
                           'packed_training_step')
    cmd = [sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider',
           os.path.join(ROOT, 'tests', 'test_gpu_fullsize.py'),
           os.path.join(ROOT, 'tests', 'test_gpu_parity.py'), '-k', select]
    res = subprocess.run(cmd, env=dict(os.environ, **env), cwd=ROOT,
                         capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert ' passed' in res.stdout
