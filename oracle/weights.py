"""Deterministic synthetic weights and token batches (test infrastructure).

Used on both sides of a parity check -- by ``oracle/make_golden.py`` when it
loads the reference modules, and by the tests/bench when they load the HIP
modules -- so that large (config-shaped) fixtures need not store 22 MB of
weights: both sides regenerate them from a numpy seed.

State-dict layout follows the reference (``stack_rnn.py:68-105``,
``vae.py:54-61,283-286``; key list in SURVEY.md section 8b).
"""
import numpy as np
import torch


def stackgru_shapes(p, prefix):
    hid, width = p['rnn_cell_size'], p['stack_width']
    vocab = p['vocab_size']
    one_hot = p.get('embedding', 'learned') == 'one_hot'
    emb = vocab if one_hot else p['embedding_size']
    gru_in = emb + (width if p.get('use_stack', True) else 0)
    shapes = {
        f'{prefix}.stack_controls_layer.weight': (3, hid),
        f'{prefix}.stack_controls_layer.bias': (3, ),
        f'{prefix}.stack_input_layer.weight': (width, hid),
        f'{prefix}.stack_input_layer.bias': (width, ),
        f'{prefix}.embedding.weight': (vocab, emb),
    }
    for layer in range(p['n_layers']):
        k_in = gru_in if layer == 0 else hid
        shapes[f'{prefix}.gru.weight_ih_l{layer}'] = (3 * hid, k_in)
        shapes[f'{prefix}.gru.weight_hh_l{layer}'] = (3 * hid, hid)
        shapes[f'{prefix}.gru.bias_ih_l{layer}'] = (3 * hid, )
        shapes[f'{prefix}.gru.bias_hh_l{layer}'] = (3 * hid, )
    return shapes


def vae_shapes(p):
    hid, lat, vocab = p['rnn_cell_size'], p['latent_dim'], p['vocab_size']
    dirs = 2 if p.get('bidirectional', False) else 1
    shapes = stackgru_shapes(p, 'encoder')
    if dirs == 2:
        shapes.update(stackgru_shapes(p, 'encoder.backward_stackgru'))
    shapes['encoder.hidden_to_mu.weight'] = (lat, hid * dirs)
    shapes['encoder.hidden_to_mu.bias'] = (lat, )
    shapes['encoder.hidden_to_logvar.weight'] = (lat, hid * dirs)
    shapes['encoder.hidden_to_logvar.bias'] = (lat, )
    shapes.update(stackgru_shapes(p, 'decoder'))
    shapes['decoder.latent_to_hidden.weight'] = (hid, lat)
    shapes['decoder.latent_to_hidden.bias'] = (hid, )
    shapes['decoder.output_layer.weight'] = (vocab, hid)
    shapes['decoder.output_layer.bias'] = (vocab, )
    return shapes


def make_state_dict(p, seed, scale=None, dtype=torch.float32):
    """Uniform(-s, s) weights, s = scale or 1/sqrt(H) (torch's default GRU /
    Linear range); one-hot embeddings are the identity (stack_rnn.py:88-97),
    learned ones get a zero ``pad_index`` row (nn.Embedding padding_idx)."""
    rng = np.random.default_rng(seed)
    hid = p['rnn_cell_size']
    s = scale if scale is not None else 1.0 / np.sqrt(hid)
    one_hot = p.get('embedding', 'learned') == 'one_hot'
    out = {}
    for name, shape in vae_shapes(p).items():
        if name.endswith('embedding.weight'):
            if one_hot:
                w = np.eye(shape[0], dtype=np.float64)
            else:
                w = rng.standard_normal(shape)
                w[p.get('pad_index', 0)] = 0.0
        else:
            w = rng.uniform(-s, s, size=shape)
        out[name] = torch.tensor(w, dtype=dtype)
    return out


def synthetic_tokens(seq_len, batch, vocab, seed, lengths=None):
    """SMILES-shaped batch: START(2) first, tokens
    uniform in [4, V), STOP(3) at len-1, PAD(0) after.  Returns i64[T,B]."""
    rng = np.random.default_rng(seed)
    toks = rng.integers(4, vocab, size=(seq_len, batch), dtype=np.int64)
    toks[0, :] = 2
    if lengths is None:
        toks[seq_len - 1, :] = 3
    else:
        for b, n in enumerate(lengths):
            toks[n - 1, b] = 3
            toks[n:, b] = 0
    return torch.from_numpy(toks)


def shifted_targets(tokens):
    """target = input shifted by one with a zero row appended
    (utils/utils.py:77-79)."""
    return torch.cat((tokens[1:], torch.zeros_like(tokens[:1])), dim=0)


def ragged_lengths(batch, seq_len, seed, mean=60, std=20, lo=10):
    """len ~ clip(N(mean,std), lo, T) sorted descending (packed mode)."""
    rng = np.random.default_rng(seed)
    n = np.clip(np.rint(rng.normal(mean, std, size=batch)), lo,
                seq_len).astype(np.int64)
    n[::-1].sort()
    n[0] = seq_len
    return n


# ---- loss-curve protocol (fixture F5; shared by oracle/make_loss_curve.py and
# the tests so both sides train on identical batches, eps and schedule) --------
LOSS_CURVES = {
    # >= 1k optimizer steps at a reduced model; 4 epochs (Adam is re-created at
    # the top of every epoch, training.py:132, so its moments reset there)
    'loss_curve_small': {
        'params': {
            'latent_dim': 64, 'rnn_cell_size': 128, 'embedding_size': 40,
            'vocab_size': 40, 'stack_width': 20, 'stack_depth': 20,
            'n_layers': 2, 'dropout': 0.0, 'batch_size': 32,
            'batch_mode': 'padded', 'embedding': 'one_hot',
            'use_stack': True, 'bidirectional': False
        },
        'seq_len': 40, 'steps': 1000, 'steps_per_epoch': 250, 'lr': 1e-3,
        'kl_growth': 0.0015, 'seed': 7000,
    },
    # the BASELINE config's model (H=512, stack 50x50, T=120) at batch 32
    'loss_curve_config': {
        'params': {
            'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
            'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50,
            'n_layers': 2, 'dropout': 0.0, 'batch_size': 32,
            'batch_mode': 'padded', 'embedding': 'one_hot',
            'use_stack': True, 'bidirectional': False
        },
        'seq_len': 120, 'steps': 60, 'steps_per_epoch': 30, 'lr': 5e-4,
        'kl_growth': 0.0015, 'seed': 7100,
    },
}


def loss_curve_batch(cfg, index):
    """Batch `index` of the epoch: (encoder tokens, decoder tokens with
    precomputed <UNK>=1 word dropout, shifted targets), all [T, B] int64.
    The sequences are learnable (START, an arithmetic progression over the
    non-special tokens with a per-sequence start and stride, STOP) so the
    curve actually descends and a wrong gradient shows up in it."""
    p = cfg['params']
    seq_len, batch, vocab = cfg['seq_len'], p['batch_size'], p['vocab_size']
    seed = cfg['seed'] + 10 * index
    rng = np.random.default_rng(seed)
    start = rng.integers(0, vocab - 4, size=batch)
    stride = rng.integers(1, 4, size=batch)
    t = np.arange(seq_len)[:, None]
    toks = 4 + (start[None, :] + stride[None, :] * t) % (vocab - 4)
    toks[0, :] = 2
    toks[seq_len - 1, :] = 3
    tokens = torch.from_numpy(toks.astype(np.int64))
    drop = torch.from_numpy(rng.random((seq_len, batch)) < 0.1)
    dec_tokens = tokens.clone()
    dec_tokens[drop & (tokens != 0)] = 1
    return tokens, dec_tokens, shifted_targets(tokens)


def loss_curve_eps(cfg, global_step):
    p = cfg['params']
    rng = np.random.default_rng(cfg['seed'] + 100000 + global_step)
    return torch.from_numpy(
        rng.standard_normal((p['batch_size'], p['latent_dim'])
                            ).astype(np.float32))


def run_loss_curve(cfg, make_model, train_step, steps=None):
    """Drive the protocol: `make_model(state_dict)` builds a model from the
    seeded weights, `train_step(model, optimizer, enc, dec, tgt, eps,
    global_step)` does zero_grad/forward/loss/backward/step and returns
    (loss, decoder_loss, kl) as floats.  Returns float64 [steps, 3]."""
    steps = cfg['steps'] if steps is None else steps
    sd = make_state_dict(cfg['params'], cfg['seed'])
    model, make_optimizer = make_model(sd)
    per_epoch = cfg['steps_per_epoch']
    curve = np.zeros((steps, 3), np.float64)
    optimizer = None
    for g in range(steps):
        if g % per_epoch == 0:
            optimizer = make_optimizer(cfg['lr'])
        enc, dec, tgt = loss_curve_batch(cfg, g % per_epoch)
        curve[g] = train_step(model, optimizer, enc, dec, tgt,
                              loss_curve_eps(cfg, g), g)
    return curve
