"""Stack-augmented GRU: host-side mirror of the reference's
``paccmann_chemistry/models/stack_rnn.py`` (same constructor, ``params`` keys,
attribute names, ``state_dict`` keys and error behaviour) whose arithmetic
runs in libpaccmann_hip.so on gfx950.

There is no CPU path: a forward on CPU tensors raises ``PaccmannHipError``.
"""
import logging
import math

import torch
import torch.nn as nn

from . import _lib
from .ops import StackGRUSequenceFn, check_status
from .utils import get_device

logger = logging.getLogger('stack_gru')


class GRUWeights(nn.Module):
    """Parameter container with ``torch.nn.GRU``'s names, shapes and default
    initialisation (uniform(-1/sqrt(H), 1/sqrt(H)) in ``nn.GRU`` parameter
    order), so ``state_dict`` keys match the reference's ``gru.*`` entries
    (stack_rnn.py:99-105).  The recurrence itself never runs through torch."""

    def __init__(self, input_size, hidden_size, num_layers, dropout=0.):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_layers = num_layers
        self.dropout = float(dropout)
        for layer in range(num_layers):
            k_in = input_size if layer == 0 else hidden_size
            self.register_parameter(
                f'weight_ih_l{layer}',
                nn.Parameter(torch.empty(3 * hidden_size, k_in))
            )
            self.register_parameter(
                f'weight_hh_l{layer}',
                nn.Parameter(torch.empty(3 * hidden_size, hidden_size))
            )
            self.register_parameter(
                f'bias_ih_l{layer}', nn.Parameter(torch.empty(3 * hidden_size))
            )
            self.register_parameter(
                f'bias_hh_l{layer}', nn.Parameter(torch.empty(3 * hidden_size))
            )
        stdv = 1.0 / math.sqrt(hidden_size) if hidden_size > 0 else 0
        for weight in self.parameters():
            nn.init.uniform_(weight, -stdv, stdv)

    def layer_weights(self, layer):
        return [
            getattr(self, f'weight_ih_l{layer}'),
            getattr(self, f'weight_hh_l{layer}'),
            getattr(self, f'bias_ih_l{layer}'),
            getattr(self, f'bias_hh_l{layer}')
        ]


class StackGRU(nn.Module):
    """Stack Augmented Gated Recurrent Unit (stack_rnn.py:15-241)."""

    _instances = 0

    def __init__(self, params):
        super(StackGRU, self).__init__()
        self.device = get_device()
        self.embedding_size = params['embedding_size']
        self.rnn_cell_size = params['rnn_cell_size']
        self.vocab_size = params['vocab_size']
        self.stack_width = params['stack_width']
        self.stack_depth = params['stack_depth']
        self.n_layers = params['n_layers']
        self._update_batch_size(params['batch_size'])
        self.gru_input = self.embedding_size
        self.use_stack = params.get('use_stack', True)
        if self.use_stack:
            self.gru_input += self.stack_width
        else:
            logger.warning('Attention: No stack will be used')
        if not 1 <= self.n_layers <= _lib.PCM_MAX_LAYERS:
            raise ValueError(
                f'n_layers must be in 1..{_lib.PCM_MAX_LAYERS}, '
                f'got {self.n_layers}'
            )
        # Network (same creation order as the reference => same default init
        # under the same torch seed)
        self.stack_controls_layer = nn.Linear(
            in_features=self.rnn_cell_size, out_features=3
        )
        self.stack_input_layer = nn.Linear(
            in_features=self.rnn_cell_size, out_features=self.stack_width
        )
        embedding = params.get('embedding', 'learned')
        if embedding == 'learned':
            self.embedding = nn.Embedding(
                self.vocab_size, self.embedding_size,
                padding_idx=params.get('pad_index', 0)
            )
        elif embedding == 'one_hot':
            # NOTE: like the reference (stack_rnn.py:55,82) the GRU input width
            # was taken from params['embedding_size'] above; one_hot is only
            # runnable when that already equals vocab_size.
            self.embedding_size = self.vocab_size
            self.embedding = nn.Embedding(
                self.vocab_size, self.embedding_size,
                padding_idx=params.get('pad_index', 0)
            )
            self.embedding.load_state_dict({
                'weight': torch.nn.functional.one_hot(
                    torch.arange(self.vocab_size)
                )
            })
            self.embedding.weight.requires_grad = False
        self.pad_index = params.get('pad_index', 0)
        self.gru = GRUWeights(
            self.gru_input, self.rnn_cell_size, self.n_layers,
            dropout=params['dropout']
        )
        self.set_batch_mode(params.get('batch_mode'))
        self._check_params()
        self._status = None
        self._seed_counter = 0
        StackGRU._instances += 1
        self._instance_salt = StackGRU._instances

    # -- reference API ---------------------------------------------------------
    def _update_batch_size(self, batch_size: int) -> None:
        """stack_rnn.py:111-129.  ``init_hidden`` / ``init_stack`` stay
        available as attributes; the kernels treat them as implicit zeros."""
        hidden_shape = (self.n_layers, batch_size, self.rnn_cell_size)
        stack_shape = (batch_size, self.stack_depth, self.stack_width)
        device = torch.device(self.device)
        current = getattr(self, 'init_hidden', None)
        if (
            batch_size == getattr(self, 'batch_size', None)
            and current is not None
            and tuple(current.shape) == hidden_shape
            and current.device.type == device.type
            and tuple(self.init_stack.shape) == stack_shape
        ):
            # same size as last time (generate() calls this once per batch):
            # the zero states are never written to, nothing to rebuild
            return
        self.batch_size = batch_size
        self.expected_shape = torch.tensor(list(hidden_shape))
        # (the reference builds both on the host and copies them over,
        # 58 MB at batch 4096: 3.7 ms of idle GPU per generate() call)
        self.init_hidden = torch.zeros(hidden_shape, device=device)
        self.init_stack = torch.zeros(stack_shape, device=device)

    def forward(self, input_token, hidden, stack):
        """One step (stack_rnn.py:131-156): input_token i64[1,B], hidden
        [L,B,H], stack [B,D,W] -> (output [1,B,H], hidden, stack)."""
        out, h_final, stack_final = self._run_sequence(
            input_token.view(1, -1), None, hidden, stack
        )
        if not self.use_stack:
            stack_final = stack
        return out, h_final, stack_final

    def _forward_fn(self, input_seq, hidden, stack):
        raise NotImplementedError

    def _forward_pass_padded(self, *args):
        raise NotImplementedError

    def _forward_pass_packed(self, *args):
        raise NotImplementedError

    def set_batch_mode(self, mode: str) -> None:
        """stack_rnn.py:167-186."""
        if not isinstance(mode, str):
            raise TypeError('Argument `mode` should be a string.')
        mode = mode.capitalize()
        MODES = {
            'Padded': self._forward_pass_padded,
            'Packed': self._forward_pass_packed
        }
        if mode not in MODES:
            raise NotImplementedError(
                f'Unknown mode: {mode}. Available modes: {MODES.keys()}'
            )
        self._forward_fn = MODES[mode]

    def stack_augmentation(self, input_val, prev_stack, controls):
        """stack_rnn.py:201-232, kept for API compatibility (not on the fused
        path, which updates the stack inside the step kernel)."""
        controls = controls.view(-1, 3, 1, 1)
        zeros = torch.zeros_like(prev_stack[:, :1])
        a_push, a_pop, a_no_op = controls[:, 0], controls[:, 1], controls[:, 2]
        stack_down = torch.cat((prev_stack[:, 1:], zeros), dim=1)
        stack_up = torch.cat((input_val, prev_stack[:, :-1]), dim=1)
        new_stack = a_no_op * prev_stack
        new_stack = new_stack + a_push * stack_up
        new_stack = new_stack + a_pop * stack_down
        return new_stack

    def _check_params(self):
        if self.rnn_cell_size < self.embedding_size:
            logger.warning('Refrain from squashing embeddings in RNN cells')

    # -- fused path --------------------------------------------------------------
    def _weights(self):
        if self.use_stack:
            ws = [
                self.stack_controls_layer.weight,
                self.stack_controls_layer.bias, self.stack_input_layer.weight,
                self.stack_input_layer.bias
            ]
        else:  # the stack layers exist but take no part (stack_rnn.py:64-66)
            ws = [None, None, None, None]
        ws.append(self.embedding.weight)
        for layer in range(self.n_layers):
            ws += self.gru.layer_weights(layer)
        return ws

    def _dims(self, batch):
        return _lib.make_dims(
            batch, self.vocab_size, self.embedding.weight.shape[1],
            self.rnn_cell_size, self.n_layers, self.stack_depth,
            self.stack_width, self.use_stack
        )

    def _status_word(self, device):
        if self._status is None or self._status.device != device:
            self._status = torch.zeros(1, dtype=torch.int32, device=device)
        return self._status

    def check_tokens(self):
        """Raise RuntimeError if a forward since the last check saw a token id
        outside the vocabulary (one device->host read)."""
        check_status(self._status)

    def _next_seed(self):
        # fresh dropout stream per forward and per module instance,
        # reproducible under torch.manual_seed
        self._seed_counter += 1
        mixed = (
            torch.initial_seed() * 0x9E3779B97F4A7C15 +
            self._instance_salt * 0xBF58476D1CE4E5B9 +
            self._seed_counter * 0x94D049BB133111EB
        )
        return mixed & 0xFFFFFFFFFFFFFFFF

    def _run_sequence(self, tokens, lengths, hidden, stack, seed=None):
        """tokens i64[T,B] -> (out [T,B,H], h_final [L,B,H], stack_final)."""
        _lib.require_device(tokens, 'StackGRU input tokens')
        batch = tokens.shape[1]
        # inter-layer dropout follows module.train()/eval() like nn.GRU; the
        # activations are kept whenever a backward pass may follow
        dropout = self.gru.dropout if (self.training and
                                       self.n_layers > 1) else 0.0
        cfg = {
            'dims': self._dims(batch),
            'training': torch.is_grad_enabled() or dropout > 0,
            'dropout': dropout,
            'seed': self._next_seed() if seed is None else seed,
            'pad_index': self.embedding.padding_idx,
        }
        if hidden is not None and hidden.dim() == 3 and \
                tuple(hidden.shape) != (self.n_layers, batch,
                                        self.rnn_cell_size):
            raise ValueError(
                f'hidden has shape {tuple(hidden.shape)}, expected '
                f'{(self.n_layers, batch, self.rnn_cell_size)}'
            )
        if not self.use_stack:
            stack = None
        weights = self._weights()
        if self.embedding.weight.shape[1] + \
                (self.stack_width if self.use_stack else 0) != \
                self.gru.weight_ih_l0.shape[1]:
            raise RuntimeError(
                'GRU input width does not match embedding + stack width: '
                f'{self.gru.weight_ih_l0.shape[1]} vs '
                f'{self.embedding.weight.shape[1]} + {self.stack_width} '
                '(one_hot embeddings need embedding_size == vocab_size)'
            )
        return StackGRUSequenceFn.apply(
            cfg, tokens, lengths, hidden, stack,
            self._status_word(tokens.device), *weights
        )
