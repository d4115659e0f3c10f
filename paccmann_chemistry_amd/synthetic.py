"""Deterministic synthetic weights and SMILES-shaped token batches.

Input synthesis for benchmarks, examples and tests (no arithmetic of the hot
path lives here): the workloads of SURVEY.md section 8d are regenerated from a
numpy seed wherever they are needed, so that config-shaped fixtures need not
store 22 MB of weights.

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
