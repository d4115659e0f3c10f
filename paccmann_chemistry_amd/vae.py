"""StackGRU encoder / decoder / TeacherVAE: host-side mirror of the reference's
``paccmann_chemistry/models/vae.py`` (same classes, method names, argument
meaning, ``state_dict`` keys and exceptions).  The per-step Python loops of the
reference are replaced by sequence-level calls into libpaccmann_hip.so.
"""
from itertools import takewhile

import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence

from . import _lib
from .hyperparams import OPTIMIZER_FACTORY
from .ops import (
    DecoderHeadFn, ReparamFn, check_status, linear, stackgru_generate
)
from .search import BeamSearch, GreedySearch, SamplingSearch
from .stack_rnn import StackGRU
from .utils import packed_to_lengths


def _reverse_within_length(tokens, lengths):
    """Reverse each column of tokens[T,B] inside its own length (the packed
    bidirectional pass, vae.py:191-198)."""
    steps = tokens.shape[0]
    lens = lengths.to(tokens.device).view(1, -1)
    idx = lens - 1 - torch.arange(steps, device=tokens.device).view(-1, 1)
    return torch.gather(tokens, 0, idx.clamp(min=0))


class StackGRUEncoder(StackGRU):
    """Stacked GRU Encoder (vae.py:15-249)."""

    def __init__(self, params):
        super(StackGRUEncoder, self).__init__(params)
        self.bidirectional = params.get('bidirectional', False)
        self.n_directions = 2 if self.bidirectional else 1
        if self.bidirectional:
            self.backward_stackgru = StackGRU(params)
        self.latent_dim = params['latent_dim']
        self.hidden_to_mu = nn.Linear(
            in_features=self.rnn_cell_size * self.n_directions,
            out_features=self.latent_dim
        )
        self.hidden_to_logvar = nn.Linear(
            in_features=self.rnn_cell_size * self.n_directions,
            out_features=self.latent_dim
        )

    def encoder_train_step(self, input_seq):
        """vae.py:63-87: returns (mu [B,Z], logvar [B,Z])."""
        hidden = self._forward_fn(input_seq, self.init_hidden, self.init_stack)
        mu = linear(hidden, self.hidden_to_mu.weight, self.hidden_to_mu.bias)
        logvar = linear(
            hidden, self.hidden_to_logvar.weight, self.hidden_to_logvar.bias
        )
        return mu, logvar

    def _initial(self, hidden, stack, owner):
        """The zero initial state is implicit in the kernels."""
        if hidden is owner.init_hidden:
            hidden = None
        if stack is owner.init_stack:
            stack = None
        return hidden, stack

    def _forward_pass_padded(self, input_seq, hidden, stack):
        """vae.py:89-136."""
        if isinstance(input_seq, PackedSequence) or \
                not isinstance(input_seq, torch.Tensor):
            raise TypeError('Input is PackedSequence or is not a Tensor')
        return self._encode(input_seq, None, hidden, stack)

    def _forward_pass_packed(self, input_seq, hidden, stack):
        """vae.py:138-236 (shrinking active batch == per-row lengths)."""
        if not isinstance(input_seq, PackedSequence):
            raise TypeError('Input is not PackedSequence')
        tokens, lengths = packed_to_lengths(input_seq)
        return self._encode(tokens, lengths, hidden, stack)

    def _encode(self, tokens, lengths, hidden, stack):
        h0, s0 = self._initial(hidden, stack, self)
        _, h_final, _ = self._run_sequence(tokens, lengths, h0, s0)
        hidden = self._post_gru_reshape(h_final)
        if self.bidirectional:
            assert len(tokens.shape) == 2, 'Input Seq must be 2D Tensor.'
            if lengths is None:
                rev = torch.flip(tokens, dims=[0])
            else:
                rev = _reverse_within_length(tokens, lengths)
            _, h_back, _ = self.backward_stackgru._run_sequence(
                rev, lengths, None, None
            )
            hidden = torch.cat(
                [hidden, self._post_gru_reshape(h_back)], axis=1
            )
        return hidden

    def _post_gru_reshape(self, hidden: torch.Tensor) -> torch.Tensor:
        """vae.py:238-249."""
        if not torch.equal(torch.tensor(hidden.shape), self.expected_shape):
            raise ValueError(
                f'GRU hidden layer has incorrect shape: {hidden.shape}. '
                f'Expected shape: {self.expected_shape}'
            )
        return hidden[-1, :, :]

    def check_tokens(self):
        super().check_tokens()
        if self.bidirectional:
            self.backward_stackgru.check_tokens()


class StackGRUDecoder(StackGRU):
    """Stack GRU Decoder (vae.py:252-489)."""

    def __init__(self, params, *args, **kwargs):
        super(StackGRUDecoder, self).__init__(params, *args, **kwargs)
        self.params = params
        self.latent_dim = params['latent_dim']
        self.latent_to_hidden = nn.Linear(
            in_features=self.latent_dim, out_features=self.rnn_cell_size
        )
        self.output_layer = nn.Linear(self.rnn_cell_size, self.vocab_size)
        self.criterion = nn.CrossEntropyLoss()
        self.optimizer = OPTIMIZER_FACTORY[
            params.get('optimizer', 'adadelta')
        ](self.parameters(), lr=params.get('lr', 0.01))  # yapf: disable
        self.outputs = None
        self.logits = None

    def _latent_to_hidden(self, latent_z):
        """vae.py:316 / 425-427.  A latent repeated over the layer axis
        (stride 0, or [B,Z]) is projected once and broadcast by the kernel."""
        if latent_z.dim() == 3 and (
            latent_z.shape[0] == 1 or latent_z.stride(0) == 0
        ):
            latent_z = latent_z[0]
        return linear(
            latent_z, self.latent_to_hidden.weight, self.latent_to_hidden.bias
        )

    def decoder_train_step(self, latent_z, input_seq, target_seq):
        """vae.py:294-320: latent_z [L,B,Z] (or [B,Z]) -> scalar loss."""
        hidden = self._latent_to_hidden(latent_z)
        return self._forward_fn(input_seq, target_seq, hidden, self.init_stack)

    def _forward_pass_padded(self, input_seq, target_seq, hidden, stack):
        """vae.py:322-350."""
        if isinstance(input_seq, PackedSequence) or \
                not isinstance(input_seq, torch.Tensor):
            raise TypeError('Input is PackedSequence or is not a Tensor')
        return self._decode(input_seq, target_seq, None, hidden, stack)

    def _forward_pass_packed(self, input_seq, target_seq, hidden, stack):
        """vae.py:352-389."""
        if not isinstance(input_seq, PackedSequence):
            raise TypeError('Input is not PackedSequence')
        tokens, lengths = packed_to_lengths(input_seq)
        targets, _ = packed_to_lengths(target_seq)
        return self._decode(tokens, targets, lengths, hidden, stack)

    def _decode(self, tokens, targets, lengths, hidden, stack):
        if stack is self.init_stack:
            stack = None
        out, _, _ = self._run_sequence(tokens, lengths, hidden, stack)
        loss, argmax, logits = DecoderHeadFn.apply(
            out, self.output_layer.weight, self.output_layer.bias, targets,
            lengths, self._status_word(out.device)
        )
        # For monitoring purposes (vae.py:346-348; packed: float tensor with
        # zeros where a sequence has ended, utils/utils.py:171-203)
        self.outputs = argmax if lengths is None else argmax.float()
        self.logits = logits
        return loss

    def generate_from_latent(
        self, latent_z, prime_input, end_token, search=SamplingSearch,
        generate_len=100, seed=None, return_logits=False
    ):
        """vae.py:391-489.  latent_z [1,B,Z]; prime_input i64[P]; returns
        i64[B, P+generate_len] (the prime included).  Greedy and sampling run
        in one fused generation call; beam search steps the cell from Python.
        """
        if isinstance(search, type):
            search = search()
        batch_size = latent_z.shape[1]
        self._update_batch_size(batch_size)
        _lib.require_device(latent_z, 'generate latent_z')
        if isinstance(search, BeamSearch):
            return self._generate_beam(
                latent_z, prime_input, search, generate_len
            )
        if isinstance(search, GreedySearch):
            mode, temperature = _lib.SEARCH_GREEDY, 1.0
        elif isinstance(search, SamplingSearch):
            mode, temperature = _lib.SEARCH_SAMPLING, search.temperature
        else:
            raise TypeError(f'Unknown search strategy: {type(search)}')
        with torch.no_grad():
            hidden = self._latent_to_hidden(latent_z)
            if seed is None:
                seed = int(torch.randint(0, 2**62, (1, )).item())
            status = self._status_word(hidden.device)
            tokens, logits = stackgru_generate(
                self._dims(batch_size), self._weights(),
                self.output_layer.weight, self.output_layer.bias, hidden,
                prime_input.view(-1), generate_len, mode, temperature, seed,
                want_logits=return_logits, status=status
            )
            if batch_size == 1:
                # early stopping for single sequences (vae.py:466-468)
                row = tokens[0, prime_input.numel():]
                hits = (row == end_token.to(row.device).view(-1)[0]).nonzero()
                if hits.numel() > 0:
                    stop = prime_input.numel() + int(hits[0]) + 1
                    tokens = tokens[:, :stop]
        if return_logits:
            return tokens, logits
        return tokens

    def _generate_beam(self, latent_z, prime_input, search, generate_len):
        """Beam branch of vae.py:439-488, stepping the fused cell per token."""
        batch_size = latent_z.shape[1]
        device = latent_z.device
        with torch.no_grad():
            hidden = self._latent_to_hidden(latent_z)
            hidden = hidden.unsqueeze(0).repeat(self.n_layers, 1, 1)
            stack = self.init_stack.to(device)
            prime = prime_input.view(-1).to(device)
            for token in prime[:-1]:
                _, hidden, stack = self(
                    token.repeat(1, batch_size), hidden, stack
                )
            input_token = prime[-1].repeat(1, batch_size)
            beams = [[[list(), 0.0]]] * batch_size
            width = search.beam_width
            input_token = torch.stack([input_token.clone() for _ in range(width)])
            hidden = torch.stack([hidden.clone() for _ in range(width)])
            stack = torch.stack([stack.clone() for _ in range(width)])
            for _ in range(generate_len):
                outs, hids, stacks = zip(*[
                    self(tok, hid, stk)
                    for tok, hid, stk in zip(input_token, hidden, stack)
                ])
                logits = torch.stack([
                    linear(o, self.output_layer.weight,
                           self.output_layer.bias).squeeze(0) for o in outs
                ])
                hidden = torch.stack(hids)
                stack = torch.stack(stacks)
                input_token, beams = search.step(logits.detach().cpu(), beams)
                input_token = input_token.to(device).unsqueeze(1)
        return torch.stack([
            torch.tensor([int(t) for t in beam[0][0]]) for beam in beams
        ])


class TeacherVAE(nn.Module):
    """vae.py:492-681."""

    def __init__(self, encoder, decoder):
        super(TeacherVAE, self).__init__()
        self.encoder = encoder
        self.decoder = decoder

    def encode(self, input_seq):
        """vae.py:506-522."""
        mu, logvar = self.encoder.encoder_train_step(input_seq)
        return mu, logvar

    def reparameterize(self, mu, logvar, eps=None):
        """vae.py:524-540: always samples (also in eval mode).  ``eps`` may be
        injected for reproducible parity runs."""
        if eps is None:
            eps = torch.randn_like(mu)
        return ReparamFn.apply(mu, logvar, eps)

    def decode(self, latent_z, input_seq, target_seq):
        """vae.py:542-569.  The reference repeats the latent over the layer
        axis; the kernel broadcasts one projected [B,H] instead."""
        if latent_z.dim() == 2:
            latent_z = latent_z.unsqueeze(0)
        latent_z = latent_z.expand(self.decoder.n_layers, -1, -1) \
            if latent_z.shape[0] == 1 else latent_z
        return self.decoder.decoder_train_step(latent_z, input_seq, target_seq)

    def forward(self, input_seq, decoder_seq, target_seq, eps=None):
        """vae.py:571-599: returns (decoder_loss, mu, logvar)."""
        mu, logvar = self.encode(input_seq)
        latent_z = self.reparameterize(mu, logvar, eps).unsqueeze(0)
        decoder_loss = self.decode(latent_z, decoder_seq, target_seq)
        return decoder_loss, mu, logvar

    def check_tokens(self):
        """Raise RuntimeError if any forward since the last check saw a token
        id outside the vocabulary (one device->host read per module)."""
        self.encoder.check_tokens()
        self.decoder.check_tokens()

    def generate(
        self, latent_z, prime_input, end_token, generate_len=100,
        search=SamplingSearch()
    ):
        """vae.py:601-651: iterator over per-molecule 1-D tensors with the
        start token and everything from the first end token on removed."""
        generated_batch = self.decoder.generate_from_latent(
            latent_z, prime_input, end_token, search=search,
            generate_len=generate_len
        )
        generated_batch = generated_batch.cpu()
        end = end_token.cpu()
        molecule_gen = (
            takewhile(lambda x: x != end, molecule[1:])
            for molecule in generated_batch
        )
        molecule_map = map(list, molecule_gen)
        return iter(map(torch.tensor, molecule_map))

    def _associate_language(self, language):
        """vae.py:653-672 (duck-typed: pytoda is not a dependency here)."""
        needed = ('token_indexes_to_smiles', 'start_index', 'stop_index')
        if all(hasattr(language, name) for name in needed):
            self.smiles_language = language
        else:
            raise TypeError(
                'Please insert a smiles language (object of type '
                'pytoda.smiles.smiles_language.SMILESLanguage . Given was '
                f'{type(language)}'
            )

    def save(self, path, *args, **kwargs):
        """Save model to path."""
        torch.save(self.state_dict(), path, *args, **kwargs)

    def load(self, path, *args, **kwargs):
        """Load model from path."""
        weights = torch.load(path, *args, **kwargs)
        self.load_state_dict(weights)
