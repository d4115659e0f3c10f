"""Generate the golden fixtures under tests/golden from the REFERENCE itself.

Runs only in the build container (it needs /root/reference; the reference
never travels to the GPU box).  pytoda / rdkit, which the reference imports at
module top but only touches in off-path code (vae.py:664, utils.py:317-352,
training.py:191-199), are replaced by empty stub modules.

    python oracle/make_golden.py            # all fixtures
    python oracle/make_golden.py tiny       # only names containing 'tiny'

Fixtures are .npz files: inputs, (small) weights or the numpy seed they are
regenerated from (oracle/weights.py), and the reference's outputs -- mu,
logvar, per-step logits (hook on decoder.output_layer), decoder loss,
decoder.outputs, KL, total loss and all parameter gradients.
"""
import os
import sys
import types

import numpy as np
import torch

REFERENCE = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)

from oracle import weights as W  # noqa: E402


def import_reference():
    if not os.path.isdir(REFERENCE):
        raise SystemExit('reference checkout not present; nothing to do')
    sys.dont_write_bytecode = True
    names = [
        'pytoda', 'pytoda.transforms', 'pytoda.smiles',
        'pytoda.smiles.smiles_language', 'pytoda.proteins', 'rdkit',
        'rdkit.rdBase', 'rdkit.RDLogger', 'rdkit.Chem', 'rdkit.Chem.Draw'
    ]
    for name in names:
        sys.modules[name] = types.ModuleType(name)

    class _Lang:
        pass

    sys.modules['pytoda.transforms'].Compose = object
    sys.modules['pytoda.smiles'].SMILESLanguage = _Lang
    sys.modules['pytoda.smiles.smiles_language'].SMILESLanguage = _Lang
    sys.modules['pytoda.proteins'].ProteinLanguage = _Lang
    pytoda = sys.modules['pytoda']
    pytoda.smiles = sys.modules['pytoda.smiles']
    pytoda.transforms = sys.modules['pytoda.transforms']
    pytoda.proteins = sys.modules['pytoda.proteins']
    pytoda.smiles.smiles_language = sys.modules[
        'pytoda.smiles.smiles_language']
    rdkit = sys.modules['rdkit']
    rdkit.Chem = sys.modules['rdkit.Chem']
    rdkit.Chem.Draw = sys.modules['rdkit.Chem.Draw']
    sys.path.insert(0, REFERENCE)
    from paccmann_chemistry.models import vae as ref_vae
    from paccmann_chemistry.utils import loss_functions as ref_loss
    from paccmann_chemistry.utils import search as ref_search
    return ref_vae, ref_loss, ref_search


def build_reference_vae(ref_vae, p, state_dict):
    enc = ref_vae.StackGRUEncoder(p)
    dec = ref_vae.StackGRUDecoder(p)
    vae = ref_vae.TeacherVAE(enc, dec)
    vae.load_state_dict(state_dict)
    return vae


def np_state(state_dict):
    return {'w::' + k: v.numpy() for k, v in state_dict.items()}


def run_vae_case(ref_vae, ref_loss, name, p, seq_len, seed, lengths=None,
                 store_weights=True, step=5000, kl_growth=0.003):
    """One TeacherVAE forward+backward in eval-mode GRU dropout (p['dropout']
    is forced to 0 so the run is reproducible) with eps injected."""
    p = dict(p)
    p['dropout'] = 0.0
    batch = p['batch_size']
    sd = W.make_state_dict(p, seed)
    vae = build_reference_vae(ref_vae, p, sd)
    vae.train()
    tokens = W.synthetic_tokens(seq_len, batch, p['vocab_size'], seed + 1,
                                lengths)
    # decoder input with a few <UNK>=1 word-dropouts (utils/utils.py:65-75)
    rng = np.random.default_rng(seed + 2)
    dec_tokens = tokens.clone()
    drop = torch.from_numpy(rng.random(tokens.shape) < 0.15)
    dec_tokens[drop & (tokens != 0)] = 1
    targets = W.shifted_targets(tokens)
    if lengths is not None:
        for b, n in enumerate(lengths):
            targets[n - 1:, b] = 0
    eps = torch.from_numpy(
        rng.standard_normal((batch, p['latent_dim'])).astype(np.float32)
    )

    step_logits = []
    hook = vae.decoder.output_layer.register_forward_hook(
        lambda m, i, o: step_logits.append(o.detach().squeeze(0).clone())
    )
    real_randn_like = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
    try:
        if lengths is None:
            enc_in, dec_in, tgt_in = tokens, dec_tokens, targets
        else:
            pack = torch.nn.utils.rnn.pack_sequence
            enc_in = pack([tokens[:n, b] for b, n in enumerate(lengths)])
            dec_in = pack([dec_tokens[:n, b] for b, n in enumerate(lengths)])
            tgt_in = pack([targets[:n, b] for b, n in enumerate(lengths)])
        dec_loss, mu, logvar = vae(enc_in, dec_in, tgt_in)
    finally:
        torch.randn_like = real_randn_like
        hook.remove()
    loss, kl = ref_loss.vae_loss_function(
        dec_loss, mu, logvar, kl_growth=kl_growth, step=step
    )
    loss.backward()
    vocab = p['vocab_size']
    logits = torch.zeros(seq_len, batch, vocab)
    for t, lg in enumerate(step_logits):
        logits[t, :lg.shape[0]] = lg
    out = {
        'tokens': tokens.numpy(),
        'dec_tokens': dec_tokens.numpy(),
        'targets': targets.numpy(),
        'eps': eps.numpy(),
        'seed': np.int64(seed),
        'step': np.int64(step),
        'kl_growth': np.float64(kl_growth),
        'mu': mu.detach().numpy(),
        'logvar': logvar.detach().numpy(),
        'logits': logits.numpy(),
        'decoder_loss': dec_loss.detach().numpy(),
        'kl': kl.detach().numpy(),
        'loss': loss.detach().numpy(),
        'outputs': vae.decoder.outputs.detach().numpy(),
    }
    if lengths is not None:
        out['lengths'] = np.asarray(lengths, dtype=np.int64)
    for k, v in p.items():
        out['p::' + k] = np.asarray(v)
    if store_weights:
        out.update(np_state(sd))
        for k, param in vae.named_parameters():
            if param.grad is not None:
                out['g::' + k] = param.grad.numpy()
    else:
        # config-shaped: keep gradient norms and a strided sample only
        for k, param in vae.named_parameters():
            if param.grad is not None:
                g = param.grad
                out['gn::' + k] = np.float64(g.double().norm().item())
                out['gs::' + k] = g.reshape(-1)[::997].numpy().copy()
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: loss {loss.item():.6f} dec {dec_loss.item():.6f} '
          f'kl {kl.item():.6f}')


def run_step_case(ref_vae, name, p, seed):
    """F1: a single StackGRU.forward with random incoming hidden/stack."""
    from paccmann_chemistry.models.stack_rnn import StackGRU
    p = dict(p)
    p['dropout'] = 0.0
    sd = W.make_state_dict(p, seed)
    enc_sd = {
        k[len('encoder.'):]: v
        for k, v in sd.items()
        if k.startswith('encoder.') and 'hidden_to' not in k
    }
    cell = StackGRU(p)
    cell.load_state_dict(enc_sd)
    cell.eval()
    rng = np.random.default_rng(seed + 7)
    B, L, H = p['batch_size'], p['n_layers'], p['rnn_cell_size']
    D, Wd = p['stack_depth'], p['stack_width']
    token = torch.from_numpy(
        rng.integers(0, p['vocab_size'], size=(1, B), dtype=np.int64)
    )
    hidden = torch.from_numpy(
        rng.uniform(-1, 1, size=(L, B, H)).astype(np.float32)
    )
    stack = torch.from_numpy(
        rng.uniform(-1, 1, size=(B, D, Wd)).astype(np.float32)
    )
    with torch.no_grad():
        out, hid2, stack2 = cell(token, hidden, stack)
    res = {
        'token': token.numpy(), 'hidden': hidden.numpy(),
        'stack': stack.numpy(), 'output': out.numpy(),
        'hidden_out': hid2.numpy(), 'stack_out': stack2.numpy(),
        'seed': np.int64(seed)
    }
    for k, v in p.items():
        res['p::' + k] = np.asarray(v)
    res.update({'w::' + k: v.numpy() for k, v in enc_sd.items()})
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **res)
    print(f'{name}: |out| {out.norm().item():.6f}')


def run_greedy_case(ref_vae, ref_search, name, p, seed, generate_len,
                    store_weights=True):
    """F4: greedy generate_from_latent (vae.py:391-468)."""
    p = dict(p)
    p['dropout'] = 0.0
    sd = W.make_state_dict(p, seed)
    vae = build_reference_vae(ref_vae, p, sd)
    vae.eval()
    rng = np.random.default_rng(seed + 11)
    B = p['batch_size']
    z = torch.from_numpy(
        rng.standard_normal((1, B, p['latent_dim'])).astype(np.float32)
    )
    prime = torch.tensor([2, 5, 7]) if 'prime3' in name else torch.tensor([2])
    step_logits = []
    hook = vae.decoder.output_layer.register_forward_hook(
        lambda m, i, o: step_logits.append(o.detach().squeeze(0).clone())
    )
    with torch.no_grad():
        tokens = vae.decoder.generate_from_latent(
            z, prime, torch.tensor([3]), search=ref_search.GreedySearch(),
            generate_len=generate_len
        )
    hook.remove()
    res = {
        'z': z.numpy(), 'prime': prime.numpy(), 'tokens': tokens.numpy(),
        'logits': torch.stack(step_logits).numpy(),
        'generate_len': np.int64(generate_len), 'seed': np.int64(seed)
    }
    for k, v in p.items():
        res['p::' + k] = np.asarray(v)
    if store_weights:
        res.update(np_state(sd))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **res)
    print(f'{name}: tokens[0,:8] {tokens[0, :8].tolist()}')


def run_beam_case(ref_vae, ref_search, name, p, seed, generate_len, width,
                  top_tokens, temperature, store_weights=True):
    """Beam branch of generate_from_latent (vae.py:439-488 with
    search.py:144-275): the best-scoring sequence per row, plus every beam and
    its score after the last step."""
    p = dict(p)
    p['dropout'] = 0.0
    sd = W.make_state_dict(p, seed)
    vae = build_reference_vae(ref_vae, p, sd)
    vae.eval()
    rng = np.random.default_rng(seed + 11)
    B = p['batch_size']
    z = torch.from_numpy(
        rng.standard_normal((1, B, p['latent_dim'])).astype(np.float32)
    )
    prime = torch.tensor([2, 5]) if 'prime2' in name else torch.tensor([2])
    search = ref_search.BeamSearch(
        beam_width=width, temperature=temperature, top_tokens=top_tokens
    )
    kept = {}
    real_step = search.step

    def recording_step(logits, beams):
        tokens, updated = real_step(logits, beams)
        kept['beams'] = updated
        return tokens, updated

    search.step = recording_step
    with torch.no_grad():
        tokens = vae.decoder.generate_from_latent(
            z, prime, torch.tensor([3]), search=search,
            generate_len=generate_len
        )
    beams = kept['beams']
    all_tokens = np.array(
        [[[int(t) for t in beam[0]] for beam in row] for row in beams],
        np.int64
    )   # [B, width, G]
    scores = np.array([[float(beam[1]) for beam in row] for row in beams],
                      np.float64)
    res = {
        'z': z.numpy(), 'prime': prime.numpy(), 'tokens': tokens.numpy(),
        'beam_tokens': all_tokens, 'beam_scores': scores,
        'generate_len': np.int64(generate_len), 'seed': np.int64(seed),
        'beam_width': np.int64(width), 'top_tokens': np.int64(top_tokens),
        'temperature': np.float64(temperature),
    }
    for k, v in p.items():
        res['p::' + k] = np.asarray(v)
    if store_weights:
        res.update(np_state(sd))
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **res)
    print(f'{name}: best[0,:8] {tokens[0, :8].tolist()} '
          f'score {scores[0, 0]:.6f}')


def run_prepare_case(name, seed, packed, input_keep=0.7):
    """Batch preparation (utils/utils.py:29-146 behind training.py:328-348):
    the reference's own encoder / decoder / target sequences for a ragged,
    length-sorted batch, with the Bernoulli word-dropout draws it took
    recorded, so that the device kernel can be fed the same draws."""
    from paccmann_chemistry.utils import utils as ref_utils
    real = ref_utils.Bernoulli
    draws = []

    class Recording:
        def __init__(self, probs):
            self.dist = real(probs)

        def sample(self, shape):
            mask = self.dist.sample(shape)
            draws.append(mask.clone())
            return mask

    rng = np.random.default_rng(seed)
    lens = sorted(rng.integers(2, 40, size=23).tolist(), reverse=True)
    batch = []
    for n in lens:
        body = rng.integers(4, 60, size=n - 2)
        batch.append(torch.from_numpy(
            np.concatenate([[2], body, [3]]).astype(np.int64)))
    cpu = torch.device('cpu')
    ref_utils.Bernoulli = Recording
    try:
        if packed:
            enc, dec, tgt = ref_utils.packed_sequential_data_preparation(
                batch, input_keep=input_keep, start_index=2, end_index=3,
                device=cpu)
            out = {
                'enc': enc.data.numpy(), 'dec': dec.data.numpy(),
                'tgt': tgt.data.numpy(),
                'batch_sizes': enc.batch_sizes.numpy(),
            }
        else:
            padded = torch.nn.utils.rnn.pad_sequence(batch)
            enc, dec, tgt = ref_utils.sequential_data_preparation(
                padded, input_keep=input_keep, start_index=2, end_index=3,
                device=cpu)
            out = {'enc': enc.numpy(), 'dec': dec.numpy(), 'tgt': tgt.numpy()}
    finally:
        ref_utils.Bernoulli = real
    keep = torch.cat([d.reshape(-1) for d in draws]).numpy().astype(np.uint8)
    out.update({
        'flat': torch.cat(batch).numpy(),
        'offsets': np.concatenate([[0], np.cumsum(lens)]).astype(np.int64),
        'keep': keep, 'input_keep': np.float64(input_keep),
        'packed': np.int64(packed), 'seed': np.int64(seed),
    })
    np.savez_compressed(os.path.join(GOLDEN, name + '.npz'), **out)
    print(f'{name}: {len(lens)} samples, {int(keep.sum())}/{keep.size} kept')


def run_kl_weight_case(name):
    from paccmann_chemistry.utils.utils import kl_weight
    steps = np.array([0, 1, 10, 100, 1000, 5000, 9000, 20000], np.int64)
    vals = {}
    for growth in (0.0015, 0.003, 0.004):
        vals[f'g{growth}'] = np.array(
            [kl_weight(int(s), growth) for s in steps], np.float64
        )
    np.savez_compressed(
        os.path.join(GOLDEN, name + '.npz'), steps=steps, **vals
    )
    print(name)


TINY = {
    'latent_dim': 16, 'rnn_cell_size': 32, 'embedding_size': 12,
    'vocab_size': 20, 'stack_width': 5, 'stack_depth': 7, 'n_layers': 2,
    'dropout': 0.0, 'batch_size': 4, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}

CONFIG = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.0, 'batch_size': 4, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    ref_vae, ref_loss, ref_search = import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    def want(name):
        return only in name

    def tiny(**kw):
        p = dict(TINY)
        p.update(kw)
        # the reference sizes the GRU input from params['embedding_size']
        # BEFORE forcing it to vocab_size for one_hot (stack_rnn.py:55,82),
        # so a one_hot model is only runnable when the two already agree
        if p['embedding'] == 'one_hot':
            p['embedding_size'] = p['vocab_size']
        return p

    if want('step_tiny'):
        run_step_case(ref_vae, 'step_tiny_onehot', tiny(), 100)
        run_step_case(
            ref_vae, 'step_tiny_learned_l3',
            tiny(embedding='learned', n_layers=3), 101
        )
        run_step_case(
            ref_vae, 'step_tiny_nostack', tiny(use_stack=False), 102
        )
    cases = [
        ('vae_tiny_padded_onehot', tiny(), 8, 200, None),
        ('vae_tiny_padded_learned', tiny(embedding='learned'), 8, 201, None),
        ('vae_tiny_padded_nostack', tiny(use_stack=False), 8, 202, None),
        ('vae_tiny_padded_l3', tiny(n_layers=3, batch_size=3), 9, 203, None),
        ('vae_tiny_packed_onehot', tiny(batch_mode='packed', batch_size=5), 8,
         204, [8, 7, 5, 5, 3]),
        ('vae_tiny_packed_learned',
         tiny(batch_mode='packed', batch_size=5, embedding='learned'), 8, 205,
         [8, 8, 6, 4, 2]),
        ('vae_tiny_padded_bidir', tiny(bidirectional=True), 8, 206, None),
        ('vae_tiny_packed_bidir',
         tiny(bidirectional=True, batch_mode='packed', batch_size=5), 8, 207,
         [8, 6, 6, 3, 2]),
        ('vae_odd_padded',
         tiny(rnn_cell_size=48, vocab_size=37, stack_width=11, stack_depth=6,
              latent_dim=24, batch_size=7), 13, 208, None),
    ]
    for name, p, seq_len, seed, lengths in cases:
        if want(name):
            run_vae_case(ref_vae, ref_loss, name, p, seq_len, seed, lengths)
    if want('vae_config_padded'):
        run_vae_case(
            ref_vae, ref_loss, 'vae_config_padded', dict(CONFIG), 120, 300,
            None, store_weights=False
        )
    if want('vae_config_packed'):
        p = dict(CONFIG, batch_mode='packed', batch_size=6)
        run_vae_case(
            ref_vae, ref_loss, 'vae_config_packed', p, 120, 301,
            [120, 97, 80, 61, 33, 12], store_weights=False
        )
    if want('greedy_tiny'):
        run_greedy_case(ref_vae, ref_search, 'greedy_tiny', tiny(), 400, 20)
        run_greedy_case(
            ref_vae, ref_search, 'greedy_tiny_prime3',
            tiny(embedding='learned'), 401, 12
        )
    if want('greedy_config'):
        run_greedy_case(
            ref_vae, ref_search, 'greedy_config', dict(CONFIG), 402, 100,
            store_weights=False
        )
    if want('kl_weight'):
        run_kl_weight_case('kl_weight')
    if want('prepare'):
        run_prepare_case('prepare_padded', 500, False)
        run_prepare_case('prepare_packed', 501, True)
    if want('beam_tiny'):
        run_beam_case(ref_vae, ref_search, 'beam_tiny', tiny(), 410, 14, 3, 5,
                      1.0)
        run_beam_case(ref_vae, ref_search, 'beam_tiny_prime2_t07',
                      tiny(embedding='learned', batch_size=5), 411, 10, 4, 7,
                      0.7)
    if want('beam_config'):
        run_beam_case(ref_vae, ref_search, 'beam_config', dict(CONFIG), 412,
                      40, 3, 5, 1.0, store_weights=False)


if __name__ == '__main__':
    main()
