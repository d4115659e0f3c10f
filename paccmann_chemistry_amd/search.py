"""Decoding strategies: mirror of the reference's ``utils/search.py`` classes.

``StackGRUDecoder.generate_from_latent`` recognises ``GreedySearch``,
``SamplingSearch`` and ``BeamSearch`` instances and runs their ``step`` inside
the fused HIP generation calls (``pcm_stackgru_generate``,
``pcm_stackgru_generate_beam``).  The methods below keep the reference's
stand-alone API (shape / type checks, ``forward`` / ``step``): greedy and
sampling run ``pcm_search_step`` for logits that live on the GPU (one launch
for the whole batch instead of the reference's per-row ``multinomial`` loop,
search.py:117-122,136-141) and plain torch for CPU logits; ``BeamSearch``'s
stand-alone methods are CPU list processing as in the reference
(search.py:144-275).
"""
from math import log
from sys import float_info

import numpy as np
import torch
from torch import nn

from .utils import get_device


class Search(nn.Module):
    """Base search class (search.py:10-53)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.device = get_device()

    def forward(self, logits: torch.Tensor) -> object:
        if not len(logits.shape) == 3:
            raise ValueError(
                f'Logits need to be 3D Tensor, was: {logits.shape}'
            )
        if not type(logits) == torch.Tensor:
            raise TypeError(
                f'Logits need to be torch.Tensor, was: {type(logits)}'
            )

    def step(self, logits: torch.Tensor) -> object:
        if len(logits.shape) > 3:
            raise ValueError(
                f'Logits need to be 2D or 3D Tensor, was: {logits.shape}'
            )
        if not type(logits) == torch.Tensor:
            raise TypeError(
                f'Logits need to be torch.Tensor, was: {type(logits)}'
            )


def _device_search(logits, mode, temperature=1.0, seed=None):
    """pcm_search_step over every row of GPU logits[..., V]."""
    from .ops import search_step
    if mode == 1 and seed is None:
        seed = int(torch.randint(0, 2**62, (1, )).item())
    return search_step(logits.detach(), mode, temperature, seed or 0)


class GreedySearch(Search):
    """Greedy search (search.py:56-87): first maximum wins on ties."""

    def forward(self, logits: torch.Tensor) -> torch.Tensor:
        super().forward(logits)
        if logits.is_cuda:
            return _device_search(logits, 0)
        return torch.argmax(logits, 2)

    def step(self, logits: torch.Tensor) -> torch.Tensor:
        super().step(logits)
        if logits.is_cuda and logits.dim() == 2:
            return _device_search(logits, 0).unsqueeze(1)
        return torch.argmax(logits, 1, keepdim=True)


class SamplingSearch(Search):
    """Temperature sampling (search.py:90-141)."""

    def __init__(self, temperature: float = 1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.temperature = temperature

    def forward(self, logits: torch.Tensor, seed=None) -> torch.Tensor:
        super().forward(logits)
        if logits.is_cuda:
            # every (row, position) in one launch: the batched multinomial
            return _device_search(logits, 1, self.temperature, seed).squeeze()
        probabilities = torch.softmax(logits.div(self.temperature), 2)
        batch, length, vocab = probabilities.shape
        draws = torch.multinomial(probabilities.reshape(-1, vocab), 1)
        return draws.view(batch, length, 1).squeeze()

    def step(self, logits: torch.Tensor, seed=None) -> torch.Tensor:
        super().step(logits)
        if logits.is_cuda and logits.dim() == 2:
            return _device_search(
                logits, 1, self.temperature, seed
            ).unsqueeze(1)
        probabilities = torch.softmax(logits.div(self.temperature), 1)
        return torch.multinomial(probabilities, 1)


class BeamSearch(Search):
    """Beam search (search.py:144-275).  ``generate_from_latent`` runs it on
    the device (pcm_stackgru_generate_beam); the stand-alone ``forward`` /
    ``step`` below are the reference's CPU list processing."""

    def __init__(self, beam_width: int = 3, temperature: float = 1.0,
                 top_tokens: int = 5):
        super().__init__()
        self.beam_width = beam_width
        self.temperature = temperature
        self.top_tokens = top_tokens

    def _beam_step_per_sequence(self, probabilities, beams):
        candidates = []
        for probability, (sequence, score) in zip(probabilities, beams):
            probability = np.asarray(probability)
            for token in np.argsort(-probability)[:self.top_tokens]:
                candidates.append([
                    sequence + [token],
                    score + log(probability[token] + float_info.epsilon)
                ])
        ordered = sorted(candidates, key=lambda pair: pair[1], reverse=True)
        return ordered[:self.beam_width]

    def _beam_per_sequence(self, logits):
        beams = [[list(), 0.0]]
        probabilities = torch.softmax(logits.div(self.temperature), 1)
        for probability in probabilities:
            stacked = torch.stack([probability] * (self.beam_width + 1))
            beams = self._beam_step_per_sequence(stacked, beams)
        sequences, scores = zip(*beams)
        return (torch.tensor(list(sequences)).T, torch.tensor(list(scores)))

    def forward(self, logits: torch.Tensor) -> tuple:
        super().forward(logits)
        tokens, scores = zip(
            *[self._beam_per_sequence(seq) for seq in logits.detach().cpu()]
        )
        return (torch.stack(tokens), torch.stack(scores))

    def step(self, logits: torch.Tensor, beams: list) -> tuple:
        super().step(logits)
        probabilities = torch.softmax(
            logits.detach().cpu().div(self.temperature), 2
        )
        updated = [
            self._beam_step_per_sequence(sample_probability, sample_beams)
            for sample_probability, sample_beams in
            zip(probabilities.permute(1, 0, 2), beams)
        ]
        token_beams = torch.stack([
            torch.tensor([int(beam[0][-1]) for beam in sample_beams])
            for sample_beams in updated
        ]).permute(1, 0).to(self.device)
        return (token_beams, updated)
