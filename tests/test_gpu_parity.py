"""GPU parity: the HIP path (through the C ABI) against the reference's golden
vectors and against the CPU oracle on the same seeded inputs.

Tolerances (north star): per-step fp32 logits within 1e-4 relative (to the
largest logit), loss within 1e-5 relative, gradients within 1e-4 of the
largest entry of each parameter gradient; greedy tokens and argmax bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import Golden, golden_names
from oracle import stackgru_oracle as O
from oracle import weights as W

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-4
GRAD_TOL = 1e-4


def _dev():
    return torch.device('cuda:0')


def _rel_to_max(a, b):
    a = torch.as_tensor(a).detach().double().cpu()
    b = torch.as_tensor(b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / b.abs().max().clamp(min=1e-12)).item()


def _build_vae(params, weights, train=True):
    import paccmann_chemistry_amd as P
    p = dict(params)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p))
    missing = vae.load_state_dict(
        {k: v.clone() for k, v in weights.items()}, strict=True
    )
    assert not missing.missing_keys and not missing.unexpected_keys
    vae = vae.to(_dev())
    vae.train(train)
    return vae


def _inputs(g, vae):
    dev = _dev()
    tokens, dec_tokens = g.tensor('tokens'), g.tensor('dec_tokens')
    targets = g.tensor('targets')
    if g.lengths is None:
        return tokens.to(dev), dec_tokens.to(dev), targets.to(dev)
    pack = torch.nn.utils.rnn.pack_sequence
    lens = g.lengths.tolist()
    enc = pack([tokens[:n, b].to(dev) for b, n in enumerate(lens)])
    dec = pack([dec_tokens[:n, b].to(dev) for b, n in enumerate(lens)])
    tgt = pack([targets[:n, b].to(dev) for b, n in enumerate(lens)])
    return enc, dec, tgt


def _run_gpu_vae(g):
    from paccmann_chemistry_amd import vae_loss_function
    p = dict(g.params)
    p['dropout'] = 0.0
    vae = _build_vae(p, g.weights)
    enc, dec, tgt = _inputs(g, vae)
    eps = g.tensor('eps').to(_dev())
    dec_loss, mu, logvar = vae(enc, dec, tgt, eps=eps)
    total, kl = vae_loss_function(
        dec_loss, mu, logvar, kl_growth=float(g.data['kl_growth']),
        step=int(g.data['step'])
    )
    total.backward()
    vae.check_tokens()
    return vae, dec_loss, mu, logvar, total, kl


def _masked_logits(g, logits):
    ref = g.tensor('logits')
    logits = logits.detach().cpu()
    if g.lengths is not None:
        steps = torch.arange(ref.shape[0]).view(-1, 1)
        active = (g.lengths.view(1, -1) > steps).unsqueeze(-1)
        logits = logits * active
    return logits, ref


# --------------------------------------------------------------------------
def test_library_reports_gfx950():
    from paccmann_chemistry_amd import _lib
    lib = _lib.load()
    torch.zeros(1, device=_dev())
    assert lib.pcm_device_check() == 0, lib.pcm_last_error()


@pytest.mark.parametrize(
    'm,n,k', [(1, 1, 1), (3, 5, 7), (64, 64, 16), (70, 33, 129),
              (512, 60, 512), (257, 1536, 110), (1000, 50, 50)]
)
def test_linear_forward_backward(m, n, k):
    """The generic MFMA GEMM behind every dense product (odd sizes, all three
    operand layouts, split-K) against an fp64 torch product."""
    from paccmann_chemistry_amd.ops import linear
    gen = torch.Generator().manual_seed(m * 131 + n * 17 + k)
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(n, k, generator=gen)
    b = torch.randn(n, generator=gen)
    dy = torch.randn(m, n, generator=gen)
    xd = x.to(_dev()).requires_grad_(True)
    wd = w.to(_dev()).requires_grad_(True)
    bd = b.to(_dev()).requires_grad_(True)
    y = linear(xd, wd, bd)
    y.backward(dy.to(_dev()))
    x64, w64, b64 = x.double(), w.double(), b.double()
    assert _rel_to_max(y, x64 @ w64.t() + b64) < 2e-6
    assert _rel_to_max(xd.grad, dy.double() @ w64) < 2e-6
    assert _rel_to_max(wd.grad, dy.double().t() @ x64) < 2e-6
    assert _rel_to_max(bd.grad, dy.double().sum(0)) < 2e-6


def test_linear_large_k_splitk():
    """Weight-gradient shape of the path: K = T*B rows reduced by split-K."""
    from paccmann_chemistry_amd.ops import linear
    gen = torch.Generator().manual_seed(5)
    m, n, k = 6000, 96, 40
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(n, k, generator=gen)
    dy = torch.randn(m, n, generator=gen)
    xd = x.to(_dev())
    wd = w.to(_dev()).requires_grad_(True)
    y = linear(xd, wd, None)
    y.backward(dy.to(_dev()))
    assert _rel_to_max(wd.grad, dy.double().t() @ x.double()) < 3e-6
    first = wd.grad.clone()
    wd.grad = None
    linear(xd, wd, None).backward(dy.to(_dev()))
    assert torch.equal(first, wd.grad)  # deterministic reduction order


@pytest.mark.parametrize('name', golden_names('step_'))
def test_single_step_golden(name):
    """StackGRU.forward for one token with non-zero incoming state."""
    from paccmann_chemistry_amd import StackGRU
    g = Golden(name)
    p = dict(g.params)
    p['dropout'] = 0.0
    cell = StackGRU(p)
    cell.load_state_dict({k: v.clone() for k, v in g.weights.items()})
    cell = cell.to(_dev()).eval()
    with torch.no_grad():
        out, hid, stack = cell(
            g.tensor('token').to(_dev()), g.tensor('hidden').to(_dev()),
            g.tensor('stack').to(_dev())
        )
    assert _rel_to_max(out, g.data['output']) < 2e-6
    assert _rel_to_max(hid, g.data['hidden_out']) < 2e-6
    assert _rel_to_max(stack, g.data['stack_out']) < 2e-6


@pytest.mark.parametrize(
    'name', [n for n in golden_names('vae_') if 'config' not in n]
)
def test_vae_small_golden(name):
    g = Golden(name)
    vae, dec_loss, mu, logvar, total, kl = _run_gpu_vae(g)
    assert _rel_to_max(mu, g.data['mu']) < 1e-5
    assert _rel_to_max(logvar, g.data['logvar']) < 1e-5
    logits, ref = _masked_logits(g, vae.decoder.logits)
    assert _rel_to_max(logits, ref) < LOGIT_TOL
    assert abs(dec_loss.item() - float(g.data['decoder_loss'])) <= \
        1e-5 * abs(float(g.data['decoder_loss']))
    assert abs(kl.item() - float(g.data['kl'])) <= 1e-5 * abs(
        float(g.data['kl']))
    assert abs(total.item() - float(g.data['loss'])) <= 1e-5 * abs(
        float(g.data['loss']))
    outputs = vae.decoder.outputs.cpu().numpy().astype(np.int64)
    assert np.array_equal(outputs, g.data['outputs'].astype(np.int64))
    named = dict(vae.named_parameters())
    for key, ref_grad in g.grads.items():
        assert named[key].grad is not None, key
        err = _rel_to_max(named[key].grad, ref_grad)
        assert err < GRAD_TOL, (key, err)
    for key, param in named.items():
        if key not in g.grads:
            assert param.grad is None or param.grad.abs().max().item() == 0, key


@pytest.mark.parametrize(
    'name', [n for n in golden_names('vae_') if 'packed' in n]
)
def test_packed_golden_with_frozen_rows(name, monkeypatch):
    """Packed batches normally launch over the still-running prefix of the
    batch (PackedSequence.batch_sizes); the same fixtures through the
    all-rows path that freezes finished rows instead."""
    from paccmann_chemistry_amd import ops
    monkeypatch.setattr(ops, 'SHRINK_ACTIVE_BATCH', False)
    g = Golden(name)
    vae, dec_loss, mu, logvar, total, kl = _run_gpu_vae(g)
    assert _rel_to_max(mu, g.data['mu']) < 1e-5
    logits, ref = _masked_logits(g, vae.decoder.logits)
    assert _rel_to_max(logits, ref) < LOGIT_TOL
    assert abs(total.item() - float(g.data['loss'])) <= 1e-5 * abs(
        float(g.data['loss']))
    named = dict(vae.named_parameters())
    if hasattr(g, 'grads') and g.grads:
        for key, ref_grad in g.grads.items():
            assert _rel_to_max(named[key].grad, ref_grad) < GRAD_TOL, key


@pytest.mark.parametrize('name', golden_names('vae_config'))
def test_vae_config_golden(name):
    """H=512, V=60, D=W=50, L=2, Z=256, T=120 against the reference."""
    g = Golden(name)
    vae, dec_loss, mu, logvar, total, kl = _run_gpu_vae(g)
    logits, ref = _masked_logits(g, vae.decoder.logits)
    assert _rel_to_max(logits, ref) < LOGIT_TOL
    assert abs(dec_loss.item() - float(g.data['decoder_loss'])) <= \
        1e-5 * abs(float(g.data['decoder_loss']))
    assert abs(kl.item() - float(g.data['kl'])) <= 1e-4 * abs(
        float(g.data['kl']))
    outputs = vae.decoder.outputs.cpu().numpy().astype(np.int64)
    assert np.array_equal(outputs, g.data['outputs'].astype(np.int64))
    named = dict(vae.named_parameters())
    for key, norm in g.grad_norms.items():
        got = named[key].grad.double().norm().item()
        assert abs(got - norm) <= 2e-4 * max(norm, 1e-8), (key, got, norm)
        sample = named[key].grad.reshape(-1)[::997]
        assert _rel_to_max(sample, g.grad_samples[key]) < 2e-4, key


@pytest.mark.parametrize('name', golden_names('greedy_'))
def test_greedy_decode_golden(name):
    """Greedy tokens bit-exact, per-step logits within tolerance."""
    from paccmann_chemistry_amd import GreedySearch
    g = Golden(name)
    p = dict(g.params)
    vae = _build_vae(p, g.weights, train=False)
    tokens, logits = vae.decoder.generate_from_latent(
        g.tensor('z').to(_dev()), g.tensor('prime').to(_dev()),
        torch.tensor([3]), search=GreedySearch(),
        generate_len=int(g.data['generate_len']), return_logits=True
    )
    assert _rel_to_max(logits, g.data['logits']) < LOGIT_TOL
    assert np.array_equal(tokens.cpu().numpy(), g.data['tokens'])


def test_one_model_serves_independent_batches_on_several_streams():
    """bench_inference.py --streams N: consecutive, independent batches are
    enqueued on N HIP streams in turn with ONE model (workspaces are allocated
    per call in stream order, cached tables are per stream).  Encoder mu /
    logvar (padded and packed) and greedy tokens of three different batches
    over three streams equal the same calls one after the other, bit for
    bit."""
    import paccmann_chemistry_amd as P
    from oracle import weights as W
    p = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 48, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    sd = W.make_state_dict(p, 5100)
    vae = _build_vae(p, sd, train=False)
    vae_packed = _build_vae(dict(p, batch_mode='packed'), sd, train=False)
    dev = _dev()
    steps, batch = 17, 48
    pack = torch.nn.utils.rnn.pack_sequence
    jobs = []
    for k in range(3):
        tokens = W.synthetic_tokens(steps, batch, 20, 5101 + k)
        lens = sorted(np.random.default_rng(5110 + k).integers(
            2, steps + 1, size=batch).tolist(), reverse=True)
        packed = pack([tokens[:n, b] for b, n in enumerate(lens)]).to(dev)
        z = torch.from_numpy(np.random.default_rng(5120 + k).standard_normal(
            (1, batch, 16)).astype(np.float32)).to(dev)
        jobs.append((tokens.to(dev), packed, z))
    prime = torch.tensor([2], device=dev)
    end = torch.tensor([3], device=dev)

    def work(job):
        tokens, packed, z = job
        mu, logvar = vae.encode(tokens)
        mu_p, logvar_p = vae_packed.encode(packed)
        toks = vae.decoder.generate_from_latent(
            z, prime, end, search=P.GreedySearch(), generate_len=12)
        return mu, logvar, mu_p, logvar_p, toks

    with torch.no_grad():
        serial = [work(job) for job in jobs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(3)]
        for _ in range(2):   # second round: cached tables in use
            outs = []
            for k, job in enumerate(jobs):
                with torch.cuda.stream(streams[k]):
                    outs.append(work(job))
            torch.cuda.synchronize()
            for got, want in zip(outs, serial):
                for a, b in zip(got, want):
                    assert torch.equal(a, b)
    vae.check_tokens()

    # the product's pipelined loops (inference.py) over the same batches
    with torch.no_grad():
        enc = list(P.encode_batches(vae, [j[0] for j in jobs], streams=3))
        enc_p = list(P.encode_batches(vae_packed, [j[1] for j in jobs],
                                      streams=2))
        gen = list(P.generate_batches(vae, [j[2] for j in jobs], prime, end,
                                      generate_len=12,
                                      search=P.GreedySearch(), streams=3))
    torch.cuda.synchronize()
    for k, want in enumerate(serial):
        assert torch.equal(enc[k][0], want[0])
        assert torch.equal(enc[k][1], want[1])
        assert torch.equal(enc_p[k][0], want[2])
        assert torch.equal(enc_p[k][1], want[3])
        assert torch.equal(gen[k], want[4])


def test_pipelined_loops_keep_generated_inputs_alive():
    """the batches may come from a lazy generator that allocates
    each one on the fly and keeps no reference.  The stream ring must hold the
    input until the caller's stream is ordered behind the side stream's reads;
    otherwise the caching allocator hands the block to the next batch (here:
    to the scribble tensor allocated right after every yield) while the encoder
    on the side stream still reads its tokens at every step.  Also: the loops
    check the token status once after draining (the reference's embedding
    lookup raises for ids outside the vocabulary)."""
    import paccmann_chemistry_amd as P
    from oracle import weights as W
    p = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 256, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    vae = _build_vae(p, W.make_state_dict(p, 5200), train=False)
    dev = _dev()
    steps, batch, n = 64, 256, 6
    host = [W.synthetic_tokens(steps, batch, 20, 5201 + k) for k in range(n)]

    with torch.no_grad():
        want = [vae.encode(t.to(dev)) for t in host]
    torch.cuda.synchronize()

    def lazy():
        for t in host:
            yield t.to(dev)       # the only reference is the generator's
            # same size and dtype as a token batch: takes the freed block if
            # the ring let go of it; its fill runs on the caller's stream
            scribble = torch.full((steps, batch), 7, dtype=torch.int64,
                                  device=dev)
            del scribble

    with torch.no_grad():
        got = list(P.encode_batches(vae, lazy(), streams=3))
    torch.cuda.synchronize()
    for (mu, logvar), (mu_w, logvar_w) in zip(got, want):
        assert torch.equal(mu, mu_w) and torch.equal(logvar, logvar_w)

    bad = host[1].clone()
    bad[3, 5] = 20                # one id outside the vocabulary
    with torch.no_grad(), pytest.raises(RuntimeError, match='out of range'):
        list(P.encode_batches(vae, [host[0].to(dev), bad.to(dev)], streams=2))
    with torch.no_grad():         # the flag was consumed; a clean loop passes
        list(P.encode_batches(vae, [host[0].to(dev)], streams=2))


def test_host_threads_share_the_library():
    """the library's process-wide state (launch probe,
    precision switch, side streams, lazily set kernel attributes) is guarded by
    one host-side lock, so several host threads may enter at once -- each with
    its own model and its own HIP stream here, as a serving process would, and
    forward + backward both (autograd runs backward on its device thread
    anyway).  Results equal the serial ones bit for bit."""
    import threading
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import vae_loss_function
    from oracle import weights as W
    p = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 24, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    dev = _dev()
    n_threads, rounds, steps = 4, 6, 11
    sd = W.make_state_dict(p, 5300)
    jobs = []
    for k in range(n_threads):
        tokens = W.synthetic_tokens(steps, 24, 20, 5301 + k).to(dev)
        eps = torch.from_numpy(np.random.default_rng(5310 + k).standard_normal(
            (24, 16)).astype(np.float32)).to(dev)
        jobs.append((tokens, W.shifted_targets(tokens.cpu()).to(dev), eps))

    def one(vae, job):
        tokens, targets, eps = job
        vae.zero_grad()
        dec_loss, mu, logvar = vae(tokens, tokens, targets, eps=eps)
        total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                     step=4000)
        total.backward()
        grads = torch.cat([q.grad.reshape(-1) for q in vae.parameters()
                           if q.grad is not None])
        return total.detach().clone(), grads.clone()

    models = [_build_vae(p, sd) for _ in range(n_threads)]
    serial = [one(models[k], jobs[k]) for k in range(n_threads)]
    torch.cuda.synchronize()
    results = [[] for _ in range(n_threads)]
    errors = []
    start = threading.Barrier(n_threads)

    def worker(k):
        try:
            stream = torch.cuda.Stream(dev)
            start.wait()
            with torch.cuda.stream(stream):
                for _ in range(rounds):
                    results[k].append(one(models[k], jobs[k]))
            stream.synchronize()
        except Exception as exc:   # noqa: BLE001 (reported below)
            errors.append((k, repr(exc)))

    threads = [threading.Thread(target=worker, args=(k, ))
               for k in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    torch.cuda.synchronize()
    assert not errors, errors
    for k in range(n_threads):
        assert len(results[k]) == rounds
        for loss, grads in results[k]:
            assert torch.equal(loss, serial[k][0])
            assert torch.equal(grads, serial[k][1])
        models[k].check_tokens()


def test_determinism_bitwise():
    """Same inputs twice => bitwise identical loss and gradients."""
    g = Golden('vae_odd_padded')
    runs = []
    for _ in range(2):
        vae, dec_loss, mu, logvar, total, kl = _run_gpu_vae(g)
        grads = {k: p.grad.clone() for k, p in vae.named_parameters()
                 if p.grad is not None}
        runs.append((total.clone(), grads))
    assert torch.equal(runs[0][0], runs[1][0])
    for key in runs[0][1]:
        assert torch.equal(runs[0][1][key], runs[1][1][key]), key


def _cpu_dropout_masks(seed, steps, n_layers, batch, hid, p):
    """numpy restatement of pcm_dropout_scale (csrc/common.h)."""
    def mix(x):
        x = (x + np.uint64(0x9E3779B97F4A7C15))
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))
    masks = []
    b = np.arange(batch, dtype=np.uint64).reshape(-1, 1)
    j = np.arange(hid, dtype=np.uint64).reshape(1, -1)
    with np.errstate(over='ignore'):
        for t in range(steps):
            per_layer = []
            for layer in range(1, n_layers):
                ctr = (np.uint64(layer) << np.uint64(56)) ^ \
                    (np.uint64(t) << np.uint64(40)) ^ (b << np.uint64(16)) ^ j
                h = mix(np.uint64(seed) ^ mix(ctr))
                u = (h >> np.uint64(40)).astype(np.float32) * np.float32(
                    1.0 / 16777216.0)
                keep = (u >= np.float32(p)).astype(np.float32)
                per_layer.append(
                    torch.from_numpy(keep * np.float32(1.0 / (1.0 - p)))
                )
            masks.append(per_layer)
    return masks


def test_dropout_forward_backward_matches_oracle():
    """Inter-layer GRU dropout (stack_rnn.py:104): the device mask is a pure
    function of (seed, layer, t, b, j); regenerate it on the CPU, feed it to
    the oracle and compare outputs and gradients (forward and backward must
    use the same mask)."""
    import paccmann_chemistry_amd as P
    p = {
        'latent_dim': 16, 'rnn_cell_size': 48, 'embedding_size': 12,
        'vocab_size': 20, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 3,
        'dropout': 0.3, 'batch_size': 6, 'batch_mode': 'padded',
        'embedding': 'learned', 'use_stack': True, 'bidirectional': False
    }
    steps, seed = 7, 123456789
    sd = W.make_state_dict(p, 77)
    enc_sd = O.sub_params(sd, 'encoder')
    enc = P.StackGRUEncoder(p)
    enc.load_state_dict(enc_sd)
    enc = enc.to(_dev()).train()
    tokens = W.synthetic_tokens(steps, 6, 20, 78)
    out, h_final, _ = enc._run_sequence(
        tokens.to(_dev()), None, None, None, seed=seed
    )
    weight = torch.linspace(-1, 1, out.numel()).view(out.shape)
    (out * weight.to(_dev())).sum().backward()

    ref = {k: v.clone().requires_grad_(k != 'embedding.weight' or True)
           for k, v in enc_sd.items()}
    masks = _cpu_dropout_masks(seed, steps, 3, 6, 48, 0.3)
    hidden = torch.zeros(3, 6, 48)
    stack = torch.zeros(6, 5, 6)
    ref_out, ref_h, _ = O.stackgru_sequence(
        ref, tokens, hidden, stack, None, True, masks
    )
    (ref_out * weight).sum().backward()
    assert _rel_to_max(out, ref_out) < 1e-5
    assert _rel_to_max(h_final, ref_h) < 1e-5
    frac = np.mean([m.eq(0).float().mean().item() for row in masks for m in row])
    assert 0.2 < frac < 0.4
    named = dict(enc.named_parameters())
    for key, val in ref.items():
        if val.grad is None or key.startswith('hidden_to'):
            continue
        err = _rel_to_max(named[key].grad, val.grad)
        assert err < GRAD_TOL, (key, err)


def test_batch_independence_on_device():
    """Reference invariants (models/tests/test_vae.py:68-142): identical rows
    encode identically, and independently of the batch size."""
    import paccmann_chemistry_amd as P
    p = {
        'latent_dim': 128, 'rnn_cell_size': 32, 'embedding_size': 30,
        'vocab_size': 55, 'stack_width': 50, 'stack_depth': 51,
        'n_layers': 3, 'dropout': .7, 'batch_size': 128, 'bidirectional': True,
        'batch_mode': 'padded', 'embedding': 'learned', 'use_stack': True
    }
    sd = W.make_state_dict(p, 5)
    enc_sd = O.sub_params(sd, 'encoder')
    sample = torch.tensor(np.concatenate([[2], np.arange(3, 53)]))
    results = []
    for batch in (1, 2, 4, 12, 55, 128):
        q = dict(p, batch_size=batch)
        enc = P.StackGRUEncoder(q)
        enc.load_state_dict(enc_sd)
        enc = enc.to(_dev()).eval()
        toks = sample.view(-1, 1).repeat(1, batch).to(_dev())
        with torch.no_grad():
            mu, logvar = enc.encoder_train_step(toks)
        mu = mu.cpu()
        for row in mu:
            assert np.allclose(row, mu[0], rtol=1e-5, atol=1e-7)
        results.append(mu[0])
    for other in results[1:]:
        assert np.allclose(other, results[0], rtol=1e-5, atol=1e-6)
    # and against the oracle
    bw = O.sub_params(sd, 'encoder.backward_stackgru')
    ref_mu, _ = O.encoder_forward(
        enc_sd, sample.view(-1, 1), p['stack_depth'], None, True, None, bw
    )
    assert _rel_to_max(results[0], ref_mu[0]) < 1e-5


def test_full_batch_rows_match_small_batch():
    """BASELINE config 2 size (B=512, T=120): every row of the big batch must
    equal the same sequence run in a small batch that the oracle checks."""
    import paccmann_chemistry_amd as P
    p = {
        'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
        'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 512, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    sd = W.make_state_dict(p, 300)
    tokens = W.synthetic_tokens(120, 512, 60, 9)
    targets = W.shifted_targets(tokens)
    vae = _build_vae(p, sd, train=False)
    with torch.no_grad():
        mu, _ = vae.encode(tokens.to(_dev()))
        loss = vae.decode(mu.unsqueeze(0), tokens.to(_dev()),
                          targets.to(_dev()))
    logits_big = vae.decoder.logits.cpu()
    rows = [0, 101, 317, 511]
    small = _build_vae(dict(p, batch_size=len(rows)), sd, train=False)
    with torch.no_grad():
        mu_s, _ = small.encode(tokens[:, rows].to(_dev()))
        small.decode(mu_s.unsqueeze(0), tokens[:, rows].to(_dev()),
                     targets[:, rows].to(_dev()))
    assert _rel_to_max(mu.cpu()[rows], mu_s) < 1e-5
    assert _rel_to_max(logits_big[:, rows], small.decoder.logits) < 1e-5
    # oracle on the small batch
    enc_sd, dec_sd = O.sub_params(sd, 'encoder'), O.sub_params(sd, 'decoder')
    ref_mu, _ = O.encoder_forward(enc_sd, tokens[:, rows], 50)
    _, ref_logits, _ = O.decoder_forward(
        dec_sd, ref_mu, tokens[:, rows], targets[:, rows], 50
    )
    assert _rel_to_max(mu_s, ref_mu) < 1e-4
    assert _rel_to_max(small.decoder.logits, ref_logits) < LOGIT_TOL
    assert torch.isfinite(loss).item()


def test_long_reduction_gradients_match_oracle():
    """Config-shaped model at batch 32 x 120 steps: the weight gradients sum
    over 3840 rows, which takes the K-major GEMM through its split-K slabs,
    the folded bias column sums and the narrow (skinny) wave mappings.  Every
    parameter gradient against the oracle's autograd."""
    from paccmann_chemistry_amd import vae_loss_function
    p = {
        'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
        'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 32, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    seq_len = 120
    sd = W.make_state_dict(p, 311)
    tokens = W.synthetic_tokens(seq_len, 32, 60, 12)
    targets = W.shifted_targets(tokens)
    eps = torch.from_numpy(np.random.default_rng(5).standard_normal(
        (32, 256)).astype(np.float32))
    vae = _build_vae(p, sd)
    dev = _dev()
    dec_loss, mu, logvar = vae(tokens.to(dev), tokens.to(dev), targets.to(dev),
                               eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    ref = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    r_loss, r_mu, r_logvar, _, _ = O.vae_forward(
        ref, tokens, tokens, targets, eps, p['stack_depth'])
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    worst = {}
    for name, param in vae.named_parameters():
        if ref[name].grad is None:
            continue
        worst[name] = _rel_to_max(param.grad, ref[name].grad)
    bad = {k: v for k, v in worst.items() if not v < GRAD_TOL}
    assert not bad, bad


SWEEP = {
    # name: (params overrides, T, ragged lengths?)
    'h192_l3_learned_packed': (dict(
        rnn_cell_size=192, n_layers=3, stack_width=33, stack_depth=9,
        vocab_size=37, embedding_size=20, embedding='learned', latent_dim=40,
        batch_size=37, batch_mode='packed'), 11, True),
    'h1024_l1_nostack': (dict(
        rnn_cell_size=1024, n_layers=1, vocab_size=24, embedding_size=24,
        latent_dim=48, batch_size=64, use_stack=False), 6, False),
    'h1024_l2_stack': (dict(
        rnn_cell_size=1024, n_layers=2, stack_width=20, stack_depth=7,
        vocab_size=24, embedding_size=24, latent_dim=32, batch_size=40), 5,
        False),
    'h64_bidir_b200': (dict(
        rnn_cell_size=64, n_layers=2, stack_width=12, stack_depth=6,
        vocab_size=30, embedding_size=30, latent_dim=24, batch_size=200,
        bidirectional=True), 7, False),
    'h256_w64_b33_packed': (dict(
        rnn_cell_size=256, n_layers=2, stack_width=64, stack_depth=4,
        vocab_size=61, embedding_size=61, latent_dim=16, batch_size=33,
        batch_mode='packed'), 9, True),
    'h320_wide_stack': (dict(
        rnn_cell_size=320, n_layers=2, stack_width=70, stack_depth=3,
        vocab_size=28, embedding_size=28, latent_dim=16, batch_size=48), 4,
        False),
}


@pytest.mark.parametrize('name', sorted(SWEEP))
def test_shape_sweep_matches_oracle(name):
    """Shapes off the tuned path (hidden sizes above 512 / not tile multiples,
    deep, wide stacks, ragged packed batches, bidirectional, large batch):
    loss, per-step logits and every parameter gradient against the oracle."""
    from paccmann_chemistry_amd import vae_loss_function
    over, seq_len, ragged = SWEEP[name]
    p = {
        'latent_dim': 16, 'rnn_cell_size': 32, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 5, 'stack_depth': 7, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 4, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    p.update(over)
    batch = p['batch_size']
    sd = W.make_state_dict(p, 900)
    lengths = None
    if ragged:
        lengths = sorted(W.ragged_lengths(batch, seq_len, 901,
                                          mean=seq_len * 0.6, std=3, lo=2),
                         reverse=True)
        lengths[0] = seq_len
    tokens = W.synthetic_tokens(seq_len, batch, p['vocab_size'], 902, lengths)
    targets = W.shifted_targets(tokens)
    if lengths is not None:
        for b, n in enumerate(lengths):
            targets[n - 1:, b] = 0
    eps = torch.from_numpy(np.random.default_rng(903).standard_normal(
        (batch, p['latent_dim'])).astype(np.float32))
    vae = _build_vae(p, sd)
    dev = _dev()
    if lengths is None:
        enc = dec = tokens.to(dev)
        tgt = targets.to(dev)
    else:
        pack = torch.nn.utils.rnn.pack_sequence
        enc = dec = pack([tokens[:n, b].to(dev) for b, n in enumerate(lengths)])
        tgt = pack([targets[:n, b].to(dev) for b, n in enumerate(lengths)])
    dec_loss, mu, logvar = vae(enc, dec, tgt, eps=eps.to(dev))
    total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                 step=5000)
    total.backward()
    vae.check_tokens()

    trainable = {k for k, q in vae.named_parameters() if q.requires_grad}
    ref = {k: v.clone().requires_grad_(k in trainable) for k, v in sd.items()}
    len_t = None if lengths is None else torch.tensor(lengths)
    r_loss, r_mu, r_logvar, r_logits, _ = O.vae_forward(
        ref, tokens, tokens, targets, eps, p['stack_depth'], len_t,
        p['use_stack'], p['bidirectional'])
    r_total, _ = O.vae_loss(r_loss, r_mu, r_logvar, 0.003, step=5000)
    r_total.backward()
    assert _rel_to_max(mu, r_mu) < 1e-5
    logits = vae.decoder.logits
    if lengths is not None:
        active = (len_t.view(1, -1) > torch.arange(seq_len).view(-1, 1))
        mask = active.unsqueeze(-1).to(logits.dtype)
        assert _rel_to_max(logits.cpu() * mask, r_logits * mask) < LOGIT_TOL
    else:
        assert _rel_to_max(logits, r_logits) < LOGIT_TOL
    assert abs(total.item() - r_total.item()) < 1e-5 * abs(r_total.item())
    bad = {}
    for key, param in vae.named_parameters():
        if ref[key].grad is None:
            continue
        err = _rel_to_max(param.grad, ref[key].grad)
        if not err < GRAD_TOL:
            bad[key] = err
    assert not bad, bad


def test_flat_adam_matches_torch_adam():
    """dp.FlatAdam (one pcm_adam_step per flat bucket) against torch.optim.Adam
    on the same parameters and gradient sequence, 6 steps."""
    from paccmann_chemistry_amd import dp
    g = Golden('vae_tiny_padded_onehot')
    p = dict(g.params)
    p['dropout'] = 0.0
    vae_a = _build_vae(p, g.weights)
    vae_b = _build_vae(p, g.weights)
    buckets = dp.GradientBuckets(vae_a, overlap=False)
    opt_a = dp.FlatAdam(buckets, lr=1e-2)
    params_b = [q for q in vae_b.parameters() if q.requires_grad]
    opt_b = torch.optim.Adam(params_b, lr=1e-2)
    gen = torch.Generator(device='cpu').manual_seed(3)
    for _ in range(6):
        opt_a.zero_grad()
        opt_b.zero_grad()
        for (na, qa), (nb, qb) in zip(vae_a.named_parameters(),
                                      vae_b.named_parameters()):
            if not qa.requires_grad:
                continue
            grad = torch.randn(qa.shape, generator=gen).to(_dev())
            qa.grad.copy_(grad)
            qb.grad = grad.clone()
        buckets.finish()
        opt_a.step()
        opt_b.step()
    for (na, qa), (nb, qb) in zip(vae_a.named_parameters(),
                                  vae_b.named_parameters()):
        assert na == nb
        assert _rel_to_max(qa, qb) < 1e-6, na
    # the model still runs on the re-pointed parameters
    enc, dec, tgt = _inputs(g, vae_a)
    loss, _, _ = vae_a(enc, dec, tgt, eps=g.tensor('eps').to(_dev()))
    assert torch.isfinite(loss).item()


DECODE_SWEEP = {
    # vocab tiles of the fused output-layer + search kernel: one (V <= 32),
    # two with a K split (the tuned V = 60 shape), four / five (no K split),
    # and shapes it does not take (H % 32 != 0, V > 256: GEMM + scan fallback)
    'v20': dict(vocab_size=20, embedding_size=20, rnn_cell_size=64),
    'v60_h512': dict(vocab_size=60, embedding_size=60, rnn_cell_size=512,
                     stack_width=50, stack_depth=50, latent_dim=256),
    'v100': dict(vocab_size=100, embedding_size=100, rnn_cell_size=96),
    'v150_l1': dict(vocab_size=150, embedding_size=150, rnn_cell_size=128,
                    n_layers=1),
    'v30_h48': dict(vocab_size=30, embedding_size=30, rnn_cell_size=48),
    'v300': dict(vocab_size=300, embedding_size=300, rnn_cell_size=64),
}


@pytest.mark.parametrize('name', sorted(DECODE_SWEEP))
def test_greedy_decode_matches_oracle(name):
    """generate_from_latent, greedy (vae.py:422-468, search.py:76-87): tokens
    bit-exact and per-step logits within tolerance against the oracle, at a
    batch that is not a multiple of the 32-row tile."""
    from paccmann_chemistry_amd import GreedySearch
    p = {
        'latent_dim': 16, 'rnn_cell_size': 32, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 5, 'stack_depth': 7, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 70, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    p.update(DECODE_SWEEP[name])
    sd = W.make_state_dict(p, 77)
    vae = _build_vae(p, sd, train=False)
    z = torch.from_numpy(np.random.default_rng(78).standard_normal(
        (p['batch_size'], p['latent_dim'])).astype(np.float32))
    prime = torch.tensor([2, 5])
    tokens, logits = vae.decoder.generate_from_latent(
        z.unsqueeze(0).to(_dev()), prime.to(_dev()), torch.tensor([3]),
        search=GreedySearch(), generate_len=12, return_logits=True
    )
    dec = O.sub_params(sd, 'decoder')
    with torch.no_grad():
        r_tokens, r_logits = O.generate_from_latent(
            dec, z, prime, 12, p['stack_depth'], p['use_stack'])
    assert _rel_to_max(logits, r_logits.numpy()) < LOGIT_TOL
    assert torch.equal(tokens.cpu(), r_tokens)


def test_packed_to_lengths_matches_pad_packed_sequence_on_device():
    """The single-gather unpacking of a length-sorted PackedSequence against
    torch's pad_packed_sequence, token and feature data, repeated and changing
    batch_sizes (the gather index is cached per batch_sizes)."""
    from torch.nn.utils.rnn import pack_sequence, pad_packed_sequence
    from paccmann_chemistry_amd.utils import packed_to_lengths
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    for lens in ((9, 9, 4, 1), (9, 9, 4, 1), (120, 77, 77, 30, 2), (3, )):
        tokens = [torch.randint(0, 50, (n, ), generator=g) for n in lens]
        feats = [torch.randn(n, 3, generator=g) for n in lens]
        for seqs in (tokens, feats):
            packed = pack_sequence([x.to(dev) for x in seqs])
            got, got_len = packed_to_lengths(packed)
            ref, ref_len = pad_packed_sequence(packed)
            assert got.is_cuda and torch.equal(got, ref)
            assert torch.equal(got_len, ref_len)


def test_generate_with_changing_batch_sizes():
    """generate() resizes the decoder per call (stack_rnn.py:111-129); the zero
    initial states are rebuilt only when the size changes.  One decoder, batch
    5 -> 3 -> 5 -> 70: every call against the oracle, and the attributes the
    reference exposes keep the shapes of the last call."""
    from paccmann_chemistry_amd import GreedySearch
    p = {
        'latent_dim': 16, 'rnn_cell_size': 32, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 5, 'stack_depth': 7, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 4, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    sd = W.make_state_dict(p, 91)
    vae = _build_vae(p, sd, train=False)
    dec = O.sub_params(sd, 'decoder')
    prime = torch.tensor([2])
    for call, batch in enumerate((5, 3, 5, 70)):
        z = torch.from_numpy(np.random.default_rng(92 + call).standard_normal(
            (batch, p['latent_dim'])).astype(np.float32))
        tokens = vae.decoder.generate_from_latent(
            z.unsqueeze(0).to(_dev()), prime.to(_dev()), torch.tensor([3]),
            search=GreedySearch(), generate_len=9
        )
        with torch.no_grad():
            r_tokens, _ = O.generate_from_latent(
                dec, z, prime, 9, p['stack_depth'], p['use_stack'])
        assert torch.equal(tokens.cpu(), r_tokens), batch
        assert vae.decoder.batch_size == batch
        assert tuple(vae.decoder.init_hidden.shape) == (2, batch, 32)
        assert tuple(vae.decoder.init_stack.shape) == (batch, 7, 5)
        assert vae.decoder.init_hidden.is_cuda
        assert float(vae.decoder.init_stack.abs().sum()) == 0.0


@pytest.mark.parametrize('name', golden_names('beam_'))
def test_beam_decode_golden(name):
    """generate_from_latent(search=BeamSearch) on the device
    (pcm_stackgru_generate_beam) against the reference's own beams
    (vae.py:439-488, search.py:169-275): the returned best sequence, every
    beam and every score (sums of double log-probabilities)."""
    from paccmann_chemistry_amd import BeamSearch
    g = Golden(name)
    vae = _build_vae(dict(g.params), g.weights, train=False)
    search = BeamSearch(beam_width=int(g.data['beam_width']),
                        temperature=float(g.data['temperature']),
                        top_tokens=int(g.data['top_tokens']))
    z, prime = g.tensor('z').to(_dev()), g.tensor('prime').to(_dev())
    best = vae.decoder.generate_from_latent(
        z, prime, torch.tensor([3]), search=search,
        generate_len=int(g.data['generate_len']))
    assert np.array_equal(best.cpu().numpy(), g.data['tokens'])
    vae.decoder._update_batch_size(z.shape[1])
    tokens, beams, scores = vae.decoder._generate_beam(
        z, prime, search, int(g.data['generate_len']), return_beams=True)
    assert torch.equal(tokens, best)
    assert np.array_equal(beams.cpu().numpy(), g.data['beam_tokens'])
    assert np.allclose(scores.cpu().numpy(), g.data['beam_scores'],
                       rtol=2e-6, atol=0)


@pytest.mark.parametrize('width,top', [(2, 5), (8, 3), (5, 30), (1, 1)])
def test_beam_decode_matches_oracle(width, top):
    """Beam shapes beyond the fixtures, against the oracle's beam branch:
    more state copies than first-step candidates (8 x 3: the reference's zip()
    drops copies), top_tokens above the vocabulary tile, a single beam, a
    batch that is no tile multiple."""
    from paccmann_chemistry_amd import BeamSearch
    p = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 37,
        'vocab_size': 37, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 21, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    sd = W.make_state_dict(p, 88, scale=0.35)
    vae = _build_vae(p, sd, train=False)
    z = torch.from_numpy(np.random.default_rng(89).standard_normal(
        (21, 16)).astype(np.float32))
    prime, gen_len = torch.tensor([2, 7]), 9
    search = BeamSearch(beam_width=width, temperature=0.9, top_tokens=top)
    vae.decoder._update_batch_size(21)
    tokens, beams, scores = vae.decoder._generate_beam(
        z.unsqueeze(0).to(_dev()), prime.to(_dev()), search, gen_len,
        return_beams=True)
    with torch.no_grad():
        r_best, r_beams = O.generate_from_latent_beam(
            O.sub_params(sd, 'decoder'), z, prime, gen_len, p['stack_depth'],
            width, top, 0.9)
    assert torch.equal(tokens.cpu(), r_best)
    alive = len(r_beams[0])
    want = np.array([[beam[0] for beam in row] for row in r_beams])
    assert np.array_equal(beams.cpu().numpy()[:, :alive], want)
    assert (beams.cpu().numpy()[:, alive:] == -1).all()
    want_scores = np.array([[beam[1] for beam in row] for row in r_beams])
    assert np.allclose(scores.cpu().numpy()[:, :alive], want_scores,
                       rtol=2e-6, atol=0)


def test_search_steps_on_device_logits():
    """GreedySearch / SamplingSearch forward() and step() on GPU logits
    (pcm_search_step: the batched form of the reference's per-row multinomial
    loop, search.py:117-122,136-141): greedy equals argmax with the first
    maximum on ties, sampling picks the inverse-CDF token of the regenerated
    uniforms, both keep the reference's output shapes."""
    import paccmann_chemistry_amd as P
    from test_gpu_fullsize import _uniform
    rng = np.random.default_rng(31)
    logits = torch.from_numpy(
        rng.standard_normal((6, 11, 23)).astype(np.float32) * 2)
    logits[0, 0, 3] = logits[0, 0, 9] = 7.0          # a tie: index 3 wins
    dev_logits = logits.to(_dev())
    greedy = P.GreedySearch()
    assert torch.equal(greedy(dev_logits).cpu(), torch.argmax(logits, 2))
    assert greedy(dev_logits)[0, 0].item() == 3
    step = greedy.step(dev_logits[:, 0])
    assert step.shape == (6, 1)
    assert torch.equal(step.cpu(), torch.argmax(logits[:, 0], 1, keepdim=True))
    sampler = P.SamplingSearch(temperature=0.6)
    draws = sampler.forward(dev_logits, seed=1234)
    assert draws.shape == (6, 11)
    u = _uniform(1234, np.arange(66, dtype=np.uint64)).astype(np.float64)
    cum = torch.cumsum(torch.softmax(logits.double().view(66, 23) / 0.6, 1), 1)
    want = (cum.numpy() > u[:, None]).argmax(axis=1)
    assert np.array_equal(draws.cpu().numpy().reshape(-1), want)
    one = sampler.step(dev_logits[:, 2], seed=77)
    assert one.shape == (6, 1) and one.dtype == torch.int64
    # CPU logits keep the reference's torch path
    assert greedy.step(logits[:, 0]).shape == (6, 1)
    assert sampler.step(logits[:, 0]).shape == (6, 1)


def test_sampling_search_distribution():
    """SamplingSearch.step (search.py:124-141): device draws follow
    softmax(logits / T); checked through generation statistics."""
    import paccmann_chemistry_amd as P
    p = {
        'latent_dim': 8, 'rnn_cell_size': 32, 'embedding_size': 12,
        'vocab_size': 12, 'stack_width': 5, 'stack_depth': 4, 'n_layers': 1,
        'dropout': 0.0, 'batch_size': 4096, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True
    }
    sd = W.make_state_dict(p, 12, scale=0.5)
    vae = _build_vae(p, sd, train=False)
    z = torch.zeros(1, 4096, 8, device=_dev())  # identical rows
    tokens, logits = vae.decoder.generate_from_latent(
        z, torch.tensor([2]), torch.tensor([3]),
        search=P.SamplingSearch(temperature=0.7), generate_len=1, seed=99,
        return_logits=True
    )
    probs = torch.softmax(logits[0, 0].cpu().double() / 0.7, 0)
    counts = torch.bincount(tokens[:, 1].cpu(), minlength=12).double()
    freq = counts / counts.sum()
    assert (freq - probs).abs().max().item() < 0.03
    again, _ = vae.decoder.generate_from_latent(
        z, torch.tensor([2]), torch.tensor([3]),
        search=P.SamplingSearch(temperature=0.7), generate_len=1, seed=99,
        return_logits=True
    )
    assert torch.equal(tokens, again)


def test_bad_token_raises():
    """Out-of-vocabulary ids surface as RuntimeError
    (generate_latent_codes.py:126 depends on it)."""
    g = Golden('vae_tiny_padded_onehot')
    vae = _build_vae(dict(g.params), g.weights, train=False)
    toks = g.tensor('tokens').clone()
    toks[2, 1] = 999
    with torch.no_grad():
        vae.encode(toks.to(_dev()))
    with pytest.raises(RuntimeError):
        vae.check_tokens()


def test_cpu_tensor_fails_loudly():
    from paccmann_chemistry_amd._lib import PaccmannHipError
    g = Golden('vae_tiny_padded_onehot')
    import paccmann_chemistry_amd as P
    enc = P.StackGRUEncoder(dict(g.params))
    with pytest.raises(PaccmannHipError):
        enc.cpu().encoder_train_step(g.tensor('tokens'))


@pytest.mark.parametrize('ragged', [False, True])
def test_bench_distributed_path_on_one_rank(ragged):
    """bench.py's N > 1 code path (RCCL process group, gradient buckets reduced
    asynchronously from autograd hooks, the all-reduced per-step row counts of
    the decoder loss, barriers, MAX over ranks) rehearsed in a one-rank group
    on this GPU: same final loss as the plain run, padded and ragged packed."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '2',
           '--warmup', '1', '--batch', '64', '--no-cpu-baseline', '--probe', '0']
    if ragged:
        cmd.append('--ragged')
    losses = []
    for force in ('0', '1'):
        env = dict(os.environ, PCM_BENCH_FORCE_DIST=force,
                   MASTER_ADDR='127.0.0.1', MASTER_PORT='29541')
        for key in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
            env.pop(key, None)
        res = subprocess.run(cmd, env=env, capture_output=True, text=True,
                             timeout=280)
        assert res.returncode == 0, res.stderr[-2000:]
        line = json.loads(res.stdout.strip().splitlines()[-1])
        assert line['n_gpus'] == 1 and line['value'] > 0
        losses.append(line['final_loss'])
    assert losses[0] == losses[1]


@pytest.mark.parametrize('mode', ['padded', 'packed'])
def test_data_parallel_shards_sum_to_global_batch(mode, monkeypatch):
    """Data-parallel loss scaling on the product path, without a second GPU:
    the global batch once as a single process, then as two unequal shards whose
    decoder heads get the GLOBAL per-step row counts (what
    dp.global_step_weights all-reduces).  The shards' losses and gradients must
    sum to the single-process ones (vae.py:343 / 386: the cross-entropy mean
    runs over the rows of the whole batch active at each step)."""
    from paccmann_chemistry_amd import dp, vae_loss_function
    p = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 20,
        'vocab_size': 20, 'stack_width': 6, 'stack_depth': 5, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 10, 'batch_mode': mode,
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    steps, batch = 9, 10
    lengths = [9, 9, 8, 6, 6, 5, 3, 3, 2, 2] if mode == 'packed' else None
    shards = [list(range(0, 7)), list(range(7, 10))]
    sd = W.make_state_dict(p, 610)
    tokens = W.synthetic_tokens(steps, batch, 20, 611, lengths)
    targets = W.shifted_targets(tokens)
    if lengths is not None:
        for b, n in enumerate(lengths):
            targets[n - 1:, b] = 0
    eps = torch.from_numpy(np.random.default_rng(612).standard_normal(
        (batch, 16)).astype(np.float32))
    dev = _dev()
    pack = torch.nn.utils.rnn.pack_sequence

    def run(rows, weights):
        q = dict(p, batch_size=len(rows))
        vae = _build_vae(q, sd)
        vae.decoder.global_batch_loss = weights is not None
        if weights is not None:
            monkeypatch.setattr(
                dp, 'global_step_weights',
                lambda lens, t, b, device=None: weights[:t].to(device))
        if lengths is None:
            enc = tokens[:, rows].to(dev)
            tgt = targets[:, rows].to(dev)
        else:
            enc = pack([tokens[:lengths[b], b].to(dev) for b in rows])
            tgt = pack([targets[:lengths[b], b].to(dev) for b in rows])
        dec_loss, mu, logvar = vae(enc, enc, tgt, eps=eps[rows].to(dev))
        total, _ = vae_loss_function(dec_loss, mu, logvar, kl_growth=0.003,
                                     step=5000)
        total.backward()
        grads = {k: q_.grad.detach().double().cpu()
                 for k, q_ in vae.named_parameters() if q_.grad is not None}
        return total.item(), grads

    want_loss, want = run(list(range(batch)), None)
    lens = torch.tensor(lengths if lengths else [steps] * batch)
    counts = (lens.view(1, -1) > torch.arange(steps).view(-1, 1)).sum(1)
    weights = 1.0 / counts.float()
    parts = [run(rows, weights) for rows in shards]
    got_loss = sum(part[0] for part in parts)
    assert abs(got_loss - want_loss) < 1e-5 * abs(want_loss)
    for key, ref in want.items():
        got = sum(part[1][key] for part in parts)
        assert _rel_to_max(got, ref) < 2e-5, key


@pytest.mark.parametrize('packed', [False, True])
def test_prepare_batch_on_device_matches_host(packed):
    """pcm_prepare_batch against the host-side restatements of
    sequential_data_preparation / packed_sequential_data_preparation
    (utils/utils.py:29-146) with the Bernoulli draws supplied."""
    from paccmann_chemistry_amd import utils as U
    rng = np.random.default_rng(5)
    lens = sorted(rng.integers(1, 40, size=37).tolist(), reverse=True)
    batch = [torch.from_numpy(rng.integers(2, 60, size=n)) for n in lens]
    steps, total = max(lens), sum(lens)
    keep = torch.from_numpy(
        (rng.random(total if packed else steps) < 0.7).astype(np.uint8))
    got = U.prepare_batch_on_device(batch, packed, input_keep=0.7,
                                    device=_dev(), keep_mask=keep)
    if packed:
        pad = torch.nn.utils.rnn.pad_packed_sequence
        # host restatement with the same draws, sample by sample
        want, off = ([], [], []), 0
        for s in batch:
            k = keep[off:off + len(s)].bool()
            off += len(s)
            want[0].append(s.clone())
            want[1].append(torch.where(k, s, torch.ones_like(s)))
            want[2].append(torch.cat([s[1:], torch.zeros(1, dtype=s.dtype)]))
        for g, w in zip(got, want):
            assert isinstance(g, torch.nn.utils.rnn.PackedSequence)
            ref = torch.nn.utils.rnn.pack_sequence(w)
            assert torch.equal(g.data.cpu(), ref.data)
            assert torch.equal(g.batch_sizes, ref.batch_sizes)
            assert torch.equal(pad(g)[1], pad(ref)[1])
    else:
        padded = torch.nn.utils.rnn.pad_sequence(batch)
        dec = padded.clone()
        dec[keep == 0] = 1
        tgt = torch.cat([padded[1:], torch.zeros(1, len(lens),
                                                 dtype=padded.dtype)])
        for g, w in zip(got, (padded, dec, tgt)):
            assert torch.equal(g.cpu(), w)


@pytest.mark.parametrize('name', ['prepare_padded', 'prepare_packed'])
def test_prepare_batch_matches_reference_fixture(name):
    """pcm_prepare_batch bit-exact against the REFERENCE's own output for the
    same batch and the same Bernoulli draws (tests/golden/prepare_*.npz, made
    by oracle/make_golden.py from utils/utils.py:29-146), staged through the
    pinned host buffers; twice, so both staging buffers are used."""
    import os
    from conftest import GOLDEN
    from paccmann_chemistry_amd import utils as U
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    flat, offs = torch.from_numpy(g['flat']), g['offsets']
    batch = [flat[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    packed = bool(int(g['packed']))
    for _ in range(3):
        got = U.prepare_batch_on_device(
            batch, packed, input_keep=float(g['input_keep']), device=_dev(),
            keep_mask=torch.from_numpy(g['keep']))
        for out, key in zip(got, ('enc', 'dec', 'tgt')):
            if packed:
                assert torch.equal(out.batch_sizes,
                                   torch.from_numpy(g['batch_sizes']))
                out = out.data
            assert torch.equal(out.cpu(), torch.from_numpy(g[key])), key


def test_prepare_batch_on_device_bernoulli():
    """Without supplied draws the word dropout is Bernoulli(input_keep) from a
    counter hash: rate, determinism under a seed, keep == 1 leaves dec == enc,
    and get_data_preparation routes list batches through it."""
    from paccmann_chemistry_amd import utils as U
    from paccmann_chemistry_amd.training import get_data_preparation
    batch = [torch.full((200, ), 7) for _ in range(64)]
    enc, dec, tgt = U.prepare_batch_on_device(batch, True, input_keep=0.8,
                                              device=_dev(), seed=11)
    kept = (dec.data == 7).float().mean().item()
    assert abs(kept - 0.8) < 0.02, kept
    assert set(dec.data.unique().tolist()) == {1, 7}
    again = U.prepare_batch_on_device(batch, True, input_keep=0.8,
                                      device=_dev(), seed=11)
    assert torch.equal(dec.data, again[1].data)
    enc, dec, tgt = get_data_preparation('padded')(batch, 1, 2, 3, _dev())
    assert torch.equal(enc, dec) and enc.shape == (200, 64)
    assert enc.is_cuda and int(tgt[-1].abs().sum()) == 0
    enc, dec, tgt = get_data_preparation('padded')(batch, 0.5, 2, 3, _dev())
    rows = (dec == 1).all(dim=1) | (dec == 7).all(dim=1)
    assert bool(rows.all())      # whole time rows are dropped, or none
    with pytest.raises(RuntimeError):
        U.prepare_batch_on_device([torch.ones(3), torch.ones(5)], True,
                                  device=_dev())


@pytest.mark.parametrize('env', [
    {'PCM_BWD_TILE': '64', 'PCM_FWD_TILE': '32', 'PCM_STACK_LEAN': '0'},
    {'PCM_FWD_TILE': '16', 'PCM_BWD_XPART': '0', 'PCM_STACK_LEAN': '1'},
    {'PCM_FUSED_CHOOSE': '0', 'PCM_STACK_MFMA': '1'},
    {'PCM_FWD_TILE': '64'},
    {'PCM_FUSE_STACK': '1'},
    {'PCM_CTRL_PARTIALS': '0'},
])
def test_alternate_kernel_variants(env):
    """The launchers pick tile widths / kernel variants by shape; the variants
    a given shape would not pick (wide reverse tiles, forced narrow / wide
    forward tiles, lean / prefetching / MFMA stack step, plain layer-0 reverse cell,
    two-launch output layer + search) are forced through their environment
    switches -- read once per process, hence a child process -- over the
    config-shaped, packed and greedy-decode fixtures (PCM_FWD_TILE=64: the
    large-batch 64 x 64 forward tile on small ragged batches; PCM_FUSE_STACK=1:
    the stack step fused into the cells; PCM_CTRL_PARTIALS=0: the stack step
    computing its control products itself at every step)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    cmd = [sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests',
           'test_gpu_parity.py'), '-x', '-q', '-p', 'no:cacheprovider', '-k',
           'vae_config_golden or vae_small_golden or greedy_decode']
    res = subprocess.run(cmd, env=dict(os.environ, **env), cwd=ROOT,
                         capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert ' passed' in res.stdout
