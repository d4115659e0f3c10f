"""Deterministic synthetic weights and token batches (test infrastructure).

Used on both sides of a parity check -- by ``oracle/make_golden.py`` when it
loads the reference modules, and by the tests when they load the HIP modules.
The generators themselves live in ``paccmann_chemistry_amd/synthetic.py`` (the
benchmarks use them too and must not import ``oracle/``); this module re-exports
them and adds the loss-curve protocol.
"""
import numpy as np
import torch

from paccmann_chemistry_amd.synthetic import (  # noqa: F401
    make_state_dict, ragged_lengths, shifted_targets, stackgru_shapes,
    synthetic_tokens, vae_shapes
)


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
        'seq_len': 120, 'steps': 240, 'steps_per_epoch': 30, 'lr': 5e-4,
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
