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
    DecoderHeadFn, ReparamFn, check_status, linear, stackgru_generate,
    stackgru_generate_beam
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

    # The reversed-time recurrence (vae.py:118-135, 186-235) is independent of
    # the forward-time one until the concat: on a GPU it is enqueued on a second
    # stream, so the two launch chains interleave on the chip (autograd replays
    # each on the stream it ran on).  False: one after the other.
    concurrent_directions = True

    def _encode(self, tokens, lengths, hidden, stack):
        h0, s0 = self._initial(hidden, stack, self)
        side = None
        if self.bidirectional:
            assert len(tokens.shape) == 2, 'Input Seq must be 2D Tensor.'
            if lengths is None:
                rev = torch.flip(tokens, dims=[0])
            else:
                rev = _reverse_within_length(tokens, lengths)
            if tokens.is_cuda and self.concurrent_directions:
                main = torch.cuda.current_stream(tokens.device)
                side = self._direction_stream(tokens.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    _, h_back, _ = self.backward_stackgru._run_sequence(
                        rev, lengths, None, None
                    )
        _, h_final, _ = self._run_sequence(tokens, lengths, h0, s0)
        hidden = self._post_gru_reshape(h_final)
        if self.bidirectional:
            if side is None:
                _, h_back, _ = self.backward_stackgru._run_sequence(
                    rev, lengths, None, None
                )
            else:
                main.wait_stream(side)
                # allocated on the side stream, read on this one from here on
                h_back.record_stream(main)
                rev.record_stream(side)
            hidden = torch.cat(
                [hidden, self._post_gru_reshape(h_back)], axis=1
            )
        return hidden

    def _direction_stream(self, device):
        streams = self.__dict__.setdefault('_direction_streams', {})
        key = (device.type, device.index)
        if key not in streams:
            streams[key] = torch.cuda.Stream(device=device)
        return streams[key]

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
        # data parallel: divide every step's cross entropy by the rows active
        # over ALL ranks, so that the ranks' losses (and gradients) sum to the
        # single-process loss of the global batch (dp.global_step_weights)
        self.global_batch_loss = False

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
        step_weight = None
        if self.global_batch_loss:
            from . import dp
            step_weight = dp.global_step_weights(
                lengths, tokens.shape[0], tokens.shape[1], out.device
            )
        loss, argmax, logits = DecoderHeadFn.apply(
            out, self.output_layer.weight, self.output_layer.bias, targets,
            lengths, self._status_word(out.device), step_weight
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
        i64[B, P+generate_len] (the prime included).  Greedy, sampling and
        beam search (which returns i64[B, generate_len], the best beam without
        the prime, vae.py:485-488) each run in one fused generation call.
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

    def _generate_beam(self, latent_z, prime_input, search, generate_len,
                       return_beams=False):
        """Beam branch of vae.py:439-488 in one fused call: the beam_width
        state copies run as one batch, BeamSearch.step (search.py:247-275) is
        a device kernel.  Returns i64[B, generate_len] -- the best-scoring
        sequence per row, without the prime, as the reference does."""
        batch_size = latent_z.shape[1]
        with torch.no_grad():
            hidden = self._latent_to_hidden(latent_z)
            return stackgru_generate_beam(
                self._dims(batch_size), self._weights(),
                self.output_layer.weight, self.output_layer.bias, hidden,
                prime_input.view(-1), generate_len, search.beam_width,
                search.top_tokens, search.temperature,
                want_beams=return_beams,
                status=self._status_word(hidden.device)
            )


class _LossTicket:
    """Pending device->host read of (loss, status flags); see
    ``TeacherVAE.loss_ticket``."""

    def __init__(self, model, loss):
        self._model = model
        self._words = [
            m._status for m in model._stackgrus() if m._status is not None
        ]
        self._event = None
        if not self._words or not loss.is_cuda:
            model.check_tokens()
            self._pair = (loss.detach().item(), 0.0)
            return
        flags = self._words[0]
        for other in self._words[1:]:
            if other is not flags:
                flags = flags | other
        packed = torch.cat(
            [loss.detach().reshape(1).double(), flags.double()]
        )
        self._host = torch.empty(2, dtype=torch.float64, pin_memory=True)
        self._host.copy_(packed, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._pair = None

    def result(self):
        if self._pair is None:
            self._event.synchronize()
            self._pair = tuple(self._host.tolist())
        value, status = self._pair
        if int(status):
            for word in self._words:
                word.zero_()
            self._pair = (value, 0.0)
        if int(status) & _lib.STATUS_TIMEOUT:
            raise _lib.PaccmannHipError(
                'the persistent forward kernel gave up waiting for its peer '
                'blocks; results are invalid (PCM_PERSIST=0 disables it)'
            )
        if int(status) & _lib.STATUS_BAD_TOKEN:
            raise RuntimeError('index out of range in self (token ids)')
        return value


class TeacherVAE(nn.Module):
    """vae.py:492-681."""

    def __init__(self, encoder, decoder):
        super(TeacherVAE, self).__init__()
        self.encoder = encoder
        self.decoder = decoder

    def _stackgrus(self):
        mods = [self.encoder, self.decoder]
        if getattr(self.encoder, 'bidirectional', False):
            mods.append(self.encoder.backward_stackgru)
        return mods

    def _share_status(self, device):
        """One int32 status word for every StackGRU of this model, so a step
        needs a single device->host read to learn about bad token ids."""
        word = self.encoder._status_word(device)
        for mod in self._stackgrus():
            mod._status = word
        return word

    def encode(self, input_seq):
        """vae.py:506-522."""
        data = getattr(input_seq, 'data', input_seq)
        if isinstance(data, torch.Tensor) and data.is_cuda:
            self._share_status(data.device)
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
        if latent_z.is_cuda:
            self._share_status(latent_z.device)
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
        id outside the vocabulary (one device->host read: the modules share
        their status word)."""
        seen = set()
        for mod in self._stackgrus():
            if mod._status is not None and id(mod._status) not in seen:
                seen.add(id(mod._status))
                check_status(mod._status)

    def loss_item(self, loss):
        """``loss.item()`` and ``check_tokens()`` in ONE device->host read (the
        per-step sync point of training.py:154): returns the loss as a float,
        raises RuntimeError for out-of-vocabulary token ids."""
        return self.loss_ticket(loss).result()

    def loss_ticket(self, loss):
        """The same read, asynchronous: the loss and the status words are
        copied to pinned host memory in stream order now; ``result()`` waits
        for that copy only.  ``train_vae`` collects a step's ticket after the
        NEXT step has been enqueued, so the GPU never idles behind the host's
        per-step read (a bad token then raises one step late)."""
        return _LossTicket(self, loss)

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
