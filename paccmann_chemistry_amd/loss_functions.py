"""VAE loss: mirror of the reference's ``utils/loss_functions.py``."""
import torch

from .ops import KLDivFn
from .utils import kl_weight


def vae_loss_function(
    decoder_loss: torch.Tensor,
    mu: torch.Tensor,
    logvar: torch.Tensor,
    kl_growth: float = 0.0015,
    step: int = None,
    eval_mode: bool = False
) -> (torch.Tensor, torch.Tensor):
    """loss_functions.py:7-45: returns (kl_w * KL + decoder_loss, KL) with
    KL = -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) reduced by the fused HIP
    kernel (a SUM over batch and latent dimensions) and kl_w the sigmoid
    annealing weight (1 in eval mode)."""
    kl_div = KLDivFn.apply(mu, logvar)
    kl_w = 1. if eval_mode else kl_weight(step, growth_rate=kl_growth)
    return kl_w * kl_div + decoder_loss, kl_div
