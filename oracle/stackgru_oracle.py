"""CPU oracle for the stack-augmented GRU VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``paccmann_chemistry_amd`` imports
this module; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by
the ``cpu_baseline`` leg of ``bench.py`` as the *checker*, never as the thing
that is shipped or measured as the product.

It restates, as explicit sequence-level math on CPU tensors, the algorithm of
the reference (citations are ``path:line`` under the reference checkout):

* StackGRU step ............ ``paccmann_chemistry/models/stack_rnn.py:131-156``
* stack control/update ..... ``stack_rnn.py:188-232``
* GRU cell (torch.nn.GRU) .. ``stack_rnn.py:99-105,155`` (PyTorch gate order
  r, z, n; third-party arithmetic: torch, pinned loosely by ``setup.py:37``)
* encoder (padded/packed) .. ``models/vae.py:63-236``
* decoder (padded/packed) .. ``models/vae.py:294-389``
* reparameterise ........... ``models/vae.py:524-540``
* VAE loss, KL weight ...... ``utils/loss_functions.py:39-45``,
  ``utils/utils.py:246-264``
* greedy / sampling step ... ``utils/search.py:76-87,124-141``
* generate_from_latent ..... ``models/vae.py:391-468``
* beam search .............. ``utils/search.py:169-201,247-275`` with the beam
  branch of ``models/vae.py:439-488``

Parity pinning: the functions below are checked against outputs of the
reference itself, generated in the build container by ``oracle/make_golden.py``
(which imports the reference) and committed under ``tests/golden``.

Arithmetic is the reference's: floating point through ``torch`` CPU kernels,
in the dtype of the parameters that are passed in (fp32 to mirror the
reference, fp64 to arbitrate between two fp32 implementations).  Gradients
come from autograd over this restatement.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# parameter handling
# --------------------------------------------------------------------------
def stackgru_param_names(n_layers):
    """state_dict keys of one StackGRU (stack_rnn.py:68-105)."""
    names = [
        'stack_controls_layer.weight', 'stack_controls_layer.bias',
        'stack_input_layer.weight', 'stack_input_layer.bias',
        'embedding.weight'
    ]
    for layer in range(n_layers):
        names += [
            f'gru.weight_ih_l{layer}', f'gru.weight_hh_l{layer}',
            f'gru.bias_ih_l{layer}', f'gru.bias_hh_l{layer}'
        ]
    return names


def n_layers_of(params):
    layer = 0
    while f'gru.weight_hh_l{layer}' in params:
        layer += 1
    return layer


def sub_params(state_dict, prefix):
    """Select ``prefix.*`` entries of a TeacherVAE state_dict."""
    plen = len(prefix) + 1
    return {
        k[plen:]: v
        for k, v in state_dict.items() if k.startswith(prefix + '.')
    }


# --------------------------------------------------------------------------
# one StackGRU step
# --------------------------------------------------------------------------
def stack_augmentation(push_val, prev_stack, controls):
    """stack_rnn.py:201-232. push_val [B,W], prev_stack [B,D,W], controls [B,3]
    (0 = push, 1 = pop, 2 = no-op); same add order as the reference."""
    a_push = controls[:, 0].reshape(-1, 1, 1)
    a_pop = controls[:, 1].reshape(-1, 1, 1)
    a_noop = controls[:, 2].reshape(-1, 1, 1)
    zeros = torch.zeros_like(prev_stack[:, :1])
    stack_down = torch.cat((prev_stack[:, 1:], zeros), dim=1)
    stack_up = torch.cat((push_val.unsqueeze(1), prev_stack[:, :-1]), dim=1)
    new_stack = a_noop * prev_stack
    new_stack = new_stack + a_push * stack_up
    new_stack = new_stack + a_pop * stack_down
    return new_stack


# True: one layer-step runs ATen's own GRU cell (torch.gru_cell, the kernel
# behind the reference's nn.GRU call on CPU) instead of the gate-by-gate
# restatement below.  Same numbers to rounding (tests/test_oracle_golden.py);
# bench.py's cpu_baseline switches it on so that the timed oracle costs what the
# reference costs (oracle/time_vs_reference.py), the parity tests leave it off.
ATEN_CELL = False


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """One layer, one time step of torch.nn.GRU (gate order r, z, n)."""
    if ATEN_CELL:
        return torch.gru_cell(x, h, w_ih, w_hh, b_ih, b_hh)
    hid = h.shape[1]
    gi = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
    z = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
    n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
    return (1.0 - z) * n + z * h


def stackgru_step(
    params, token, hidden, stack, use_stack=True, dropout_masks=None
):
    """stack_rnn.py:131-156.

    token i64[B]; hidden [L,B,H]; stack [B,D,W].
    dropout_masks: optional list (len L-1) of [B,H] multipliers applied to the
    input of layers 1..L-1 (already scaled by 1/(1-p)); None = eval mode.
    Returns (output [B,H], hidden' [L,B,H], stack').
    """
    n_layers = hidden.shape[0]
    emb = params['embedding.weight'][token]  # [B,E]
    if use_stack:
        h_top = hidden[-1]
        controls = F.softmax(
            h_top @ params['stack_controls_layer.weight'].t() +
            params['stack_controls_layer.bias'],
            dim=-1
        )
        push_val = torch.tanh(
            h_top @ params['stack_input_layer.weight'].t() +
            params['stack_input_layer.bias']
        )
        stack = stack_augmentation(push_val, stack, controls)
        x = torch.cat((emb, stack[:, 0, :]), dim=1)
    else:
        x = emb
    new_hidden = []
    for layer in range(n_layers):
        if layer > 0 and dropout_masks is not None:
            x = x * dropout_masks[layer - 1]
        h_new = gru_cell(
            x, hidden[layer], params[f'gru.weight_ih_l{layer}'],
            params[f'gru.weight_hh_l{layer}'],
            params[f'gru.bias_ih_l{layer}'], params[f'gru.bias_hh_l{layer}']
        )
        new_hidden.append(h_new)
        x = h_new
    return x, torch.stack(new_hidden), stack


# --------------------------------------------------------------------------
# sequence level
# --------------------------------------------------------------------------
def stackgru_sequence(
    params,
    tokens,
    hidden,
    stack,
    lengths=None,
    use_stack=True,
    dropout_masks=None
):
    """Run T steps.  tokens i64[T,B].  lengths (i64[B], optional) gives packed
    semantics: row b is updated only while t < lengths[b]; afterwards its
    hidden/stack are frozen -- equivalent to the shrinking active prefix of
    ``vae.py:165-183`` / ``utils/utils.py:226-243``.
    dropout_masks: optional [T][L-1] nested list of [B,H] masks.
    Returns (outputs [T,B,H], hidden_T [L,B,H], stack_T [B,D,W])."""
    outs = []
    for t in range(tokens.shape[0]):
        masks = None if dropout_masks is None else dropout_masks[t]
        out, new_hidden, new_stack = stackgru_step(
            params, tokens[t], hidden, stack, use_stack, masks
        )
        if lengths is not None:
            active = (lengths > t)
            out_keep = active.view(-1, 1)
            new_hidden = torch.where(
                active.view(1, -1, 1), new_hidden, hidden
            )
            new_stack = torch.where(active.view(-1, 1, 1), new_stack, stack)
            out = torch.where(out_keep, out, torch.zeros_like(out))
        hidden, stack = new_hidden, new_stack
        outs.append(out)
    return torch.stack(outs), hidden, stack


def _zeros_state(params, batch, stack_depth):
    n_layers = n_layers_of(params)
    w_hh = params['gru.weight_hh_l0']
    hid = w_hh.shape[1]
    width = params['stack_input_layer.weight'].shape[0]
    hidden = torch.zeros(n_layers, batch, hid, dtype=w_hh.dtype)
    stack = torch.zeros(batch, stack_depth, width, dtype=w_hh.dtype)
    return hidden, stack


def encoder_forward(
    params,
    tokens,
    stack_depth,
    lengths=None,
    use_stack=True,
    dropout_masks=None,
    bw_params=None
):
    """vae.py:63-136 (padded) / 138-236 (packed, via ``lengths``).

    bw_params: parameters of ``backward_stackgru`` for the bidirectional
    encoder (vae.py:118-135; packed: 186-235 -- each sequence reversed within
    its own length).  Returns (mu [B,Z], logvar [B,Z])."""
    batch = tokens.shape[1]
    hidden, stack = _zeros_state(params, batch, stack_depth)
    _, hidden, _ = stackgru_sequence(
        params, tokens, hidden, stack, lengths, use_stack, dropout_masks
    )
    top = hidden[-1]
    if bw_params is not None:
        seq_len = tokens.shape[0]
        if lengths is None:
            rev = torch.flip(tokens, dims=[0])
        else:
            idx = lengths.view(1, -1) - 1 - torch.arange(seq_len).view(-1, 1)
            rev = torch.gather(tokens, 0, idx.clamp(min=0))
        hidden_b, stack_b = _zeros_state(bw_params, batch, stack_depth)
        _, hidden_b, _ = stackgru_sequence(
            bw_params, rev, hidden_b, stack_b, lengths, use_stack, None
        )
        top = torch.cat([top, hidden_b[-1]], dim=1)
    mu = top @ params['hidden_to_mu.weight'].t() + params['hidden_to_mu.bias']
    logvar = (
        top @ params['hidden_to_logvar.weight'].t() +
        params['hidden_to_logvar.bias']
    )
    return mu, logvar


def decoder_forward(
    params,
    latent_z,
    tokens,
    targets,
    stack_depth,
    lengths=None,
    use_stack=True,
    dropout_masks=None
):
    """vae.py:294-350 (padded) / 352-389 (packed, via ``lengths``).

    latent_z [B,Z]: every layer starts from the same Linear(Z->H)(z) row
    (vae.py:316 after the repeat of vae.py:564-565).
    Returns (loss scalar, logits [T,B,V], argmax i64[B,T]).  Loss = sum over
    steps of the CE mean over the rows active at that step (vae.py:343,386);
    padded mode has every row active at every step (no ignore_index)."""
    n_layers = n_layers_of(params)
    batch = tokens.shape[1]
    h0 = (
        latent_z @ params['latent_to_hidden.weight'].t() +
        params['latent_to_hidden.bias']
    )
    hidden = h0.unsqueeze(0).repeat(n_layers, 1, 1)
    _, stack = _zeros_state(params, batch, stack_depth)
    outs, _, _ = stackgru_sequence(
        params, tokens, hidden, stack, lengths, use_stack, dropout_masks
    )
    logits = outs @ params['output_layer.weight'].t() + params[
        'output_layer.bias']
    seq_len = tokens.shape[0]
    logp = F.log_softmax(logits, dim=-1)
    nll = -logp.gather(2, targets.unsqueeze(-1)).squeeze(-1)  # [T,B]
    if lengths is None:
        loss = nll.mean(dim=1).sum()
        argmax = torch.argmax(logits, dim=2).t().contiguous()
    else:
        steps = torch.arange(seq_len).view(-1, 1)
        active = (lengths.view(1, -1) > steps)
        count = active.sum(dim=1).clamp(min=1).to(nll.dtype)
        loss = ((nll * active.to(nll.dtype)).sum(dim=1) / count).sum()
        argmax = torch.argmax(logits, dim=2) * active.to(torch.int64)
        argmax = argmax.t().contiguous()
    return loss, logits, argmax


def reparameterize(mu, logvar, eps):
    """vae.py:538-540 with the normal sample ``eps`` supplied."""
    return eps * torch.exp(0.5 * logvar) + mu


def kl_divergence(mu, logvar):
    """utils/loss_functions.py:39 (a SUM over batch and latent dims)."""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


def kl_weight(step, growth_rate=0.004):
    """utils/utils.py:246-264."""
    return 1 / (1 + math.exp((15 - growth_rate * step)))


def vae_loss(decoder_loss, mu, logvar, kl_growth=0.0015, step=None,
             eval_mode=False):
    """utils/loss_functions.py:39-45."""
    kl_div = kl_divergence(mu, logvar)
    kl_w = 1. if eval_mode else kl_weight(step, growth_rate=kl_growth)
    return kl_w * kl_div + decoder_loss, kl_div


def vae_forward(
    state_dict,
    enc_tokens,
    dec_tokens,
    targets,
    eps,
    stack_depth,
    lengths=None,
    use_stack=True,
    bidirectional=False
):
    """TeacherVAE.forward, vae.py:596-599, with ``eps`` injected.
    Returns (decoder_loss, mu, logvar, logits, argmax)."""
    enc = sub_params(state_dict, 'encoder')
    dec = sub_params(state_dict, 'decoder')
    bw = None
    if bidirectional:
        bw = sub_params(state_dict, 'encoder.backward_stackgru')
    mu, logvar = encoder_forward(
        enc, enc_tokens, stack_depth, lengths, use_stack, None, bw
    )
    z = reparameterize(mu, logvar, eps)
    loss, logits, argmax = decoder_forward(
        dec, z, dec_tokens, targets, stack_depth, lengths, use_stack, None
    )
    return loss, mu, logvar, logits, argmax


# --------------------------------------------------------------------------
# decoding
# --------------------------------------------------------------------------
def greedy_step(logits):
    """utils/search.py:76-87 (first maximum wins on ties)."""
    return torch.argmax(logits, 1, keepdim=True)


def sampling_probabilities(logits, temperature):
    """utils/search.py:135 -- the distribution ``multinomial`` draws from."""
    return torch.softmax(logits / temperature, 1)


def generate_from_latent(
    params, latent_z, prime, generate_len, stack_depth, use_stack=True,
    chooser=greedy_step
):
    """vae.py:422-468,489 (non-beam branch), eval mode.

    latent_z [B,Z]; prime i64[P].  Returns (tokens i64[B,P+G], logits
    [G,B,V])."""
    n_layers = n_layers_of(params)
    batch = latent_z.shape[0]
    h0 = (
        latent_z @ params['latent_to_hidden.weight'].t() +
        params['latent_to_hidden.bias']
    )
    hidden = h0.unsqueeze(0).repeat(n_layers, 1, 1)
    _, stack = _zeros_state(params, batch, stack_depth)
    generated = prime.view(1, -1).repeat(batch, 1)
    for p in range(prime.numel() - 1):
        _, hidden, stack = stackgru_step(
            params, generated[:, p], hidden, stack, use_stack
        )
    token = generated[:, -1]
    all_logits = []
    for _ in range(generate_len):
        out, hidden, stack = stackgru_step(
            params, token, hidden, stack, use_stack
        )
        logits = out @ params['output_layer.weight'].t() + params[
            'output_layer.bias']
        top = chooser(logits)
        token = top.view(-1)
        generated = torch.cat((generated, top.view(-1, 1)), dim=1)
        all_logits.append(logits)
    return generated, torch.stack(all_logits)


def beam_step(probabilities, beams, beam_width, top_tokens):
    """utils/search.py:169-201 for one batch row: ``probabilities`` [slots, V]
    (fp32 tensor), ``beams`` a list of [token list, score].  Each (slot, beam)
    pair -- zip() stops at the shorter of the two -- contributes its
    ``top_tokens`` most probable tokens; candidates are ranked by
    score + log(p + eps), ties in candidate order (Python's sort is stable),
    and the best ``beam_width`` survive."""
    eps = 2.220446049250313e-16   # sys.float_info.epsilon
    candidates = []
    for probability, (sequence, score) in zip(probabilities, beams):
        order = torch.argsort(-probability, stable=True)[:top_tokens]
        for token in order.tolist():
            # p + eps stays fp32 (tensor + Python scalar), the log is a double
            bumped = float(probability[token] + eps)
            candidates.append([sequence + [token], score + math.log(bumped)])
    ranked = sorted(candidates, key=lambda pair: pair[1], reverse=True)
    return ranked[:beam_width]


def generate_from_latent_beam(
    params, latent_z, prime, generate_len, stack_depth, beam_width,
    top_tokens, temperature=1.0, use_stack=True
):
    """Beam branch of vae.py:439-488, eval mode.  latent_z [B,Z]; prime
    i64[P].  Returns (best tokens i64[B,G] -- no prime, vae.py:485-488 --,
    beams: per row a list of [token list, score], best first).

    The reference keeps ``beam_width`` copies of (token, hidden, stack); copy k
    is fed the last token of beam k after every step and is NOT re-pointed at
    that beam's parent state; when fewer beams than copies exist, zip() drops
    the surplus copies for good (vae.py:471-483)."""
    n_layers = n_layers_of(params)
    batch = latent_z.shape[0]
    h0 = (
        latent_z @ params['latent_to_hidden.weight'].t() +
        params['latent_to_hidden.bias']
    )
    hidden = h0.unsqueeze(0).repeat(n_layers, 1, 1)
    _, stack = _zeros_state(params, batch, stack_depth)
    for p in range(prime.numel() - 1):
        _, hidden, stack = stackgru_step(
            params, prime[p].repeat(batch), hidden, stack, use_stack
        )
    tokens = [prime[-1].repeat(batch) for _ in range(beam_width)]
    hiddens = [hidden.clone() for _ in range(beam_width)]
    stacks = [stack.clone() for _ in range(beam_width)]
    beams = [[[list(), 0.0]] for _ in range(batch)]
    for _ in range(generate_len):
        logits = []
        slots = min(len(tokens), len(hiddens))
        for k in range(slots):
            out, hiddens[k], stacks[k] = stackgru_step(
                params, tokens[k], hiddens[k], stacks[k], use_stack
            )
            logits.append(out @ params['output_layer.weight'].t() +
                          params['output_layer.bias'])
        hiddens, stacks = hiddens[:slots], stacks[:slots]
        probabilities = torch.softmax(
            torch.stack(logits).div(temperature), 2
        )   # [slots, B, V]
        beams = [
            beam_step(probabilities[:, b], beams[b], beam_width, top_tokens)
            for b in range(batch)
        ]
        alive = len(beams[0])
        tokens = [
            torch.tensor([beams[b][k][0][-1] for b in range(batch)])
            for k in range(alive)
        ]
    best = torch.tensor([beams[b][0][0] for b in range(batch)])
    return best, beams
