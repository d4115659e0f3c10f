"""Generate the loss-curve fixtures (F5) from the reference itself.

TEST INFRASTRUCTURE ONLY.  Runs in the build container, where /root/reference
exists; writes tests/golden/loss_curve_*.npz (the curves only: every input is
regenerated from seeds by oracle/weights.py on both sides).  Two curves per
protocol: the reference in fp32 (`curve`) and in fp64 (`curve_f64`).

The protocol (oracle/weights.py:run_loss_curve) mirrors training.py:132-156 of
the reference: Adam re-created at the top of each epoch, zero_grad, forward,
vae_loss_function(kl_growth, step=global_step), backward, step.  What cannot
match across implementations is fixed: dropout 0, eps injected by replacing
torch.randn_like for the call, word dropout and batch order precomputed.

    python oracle/make_loss_curve.py [name-substring]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import weights as W  # noqa: E402
from oracle.make_golden import (  # noqa: E402
    GOLDEN, build_reference_vae, import_reference
)


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ''
    ref_vae, ref_loss, _ = import_reference()
    torch.set_num_threads(8)
    for name, cfg in W.LOSS_CURVES.items():
        if only not in name:
            continue
        p = cfg['params']

        dtype = [torch.float32]

        def make_model(sd):
            vae = build_reference_vae(ref_vae, p, sd).to(dtype[0])
            vae.train()
            return vae, lambda lr: torch.optim.Adam(vae.parameters(), lr=lr)

        def train_step(vae, optimizer, enc, dec, tgt, eps, g):
            real = torch.randn_like
            torch.randn_like = lambda t, *a, **k: eps.to(t.dtype)
            try:
                optimizer.zero_grad()
                dec_loss, mu, logvar = vae(enc, dec, tgt)
            finally:
                torch.randn_like = real
            loss, kl = ref_loss.vae_loss_function(
                dec_loss, mu, logvar, kl_growth=cfg['kl_growth'], step=g
            )
            loss.backward()
            optimizer.step()
            if g % 50 == 0:
                print(f'{name} step {g}: loss {loss.item():.5f}', flush=True)
            return loss.item(), dec_loss.item(), kl.item()

        t0 = time.time()
        curve = W.run_loss_curve(cfg, make_model, train_step)
        # the same protocol with the reference model in float64: the
        # trajectory of (nearly) exact arithmetic, against which the fp32
        # reference's own rounding sensitivity is read off
        # (the reference builds its initial hidden / stack with the default
        # dtype, so that is switched for the run as well)
        dtype[0] = torch.float64
        torch.set_default_dtype(torch.float64)
        try:
            curve_f64 = W.run_loss_curve(cfg, make_model, train_step)
        finally:
            torch.set_default_dtype(torch.float32)
        np.savez_compressed(
            os.path.join(GOLDEN, name + '.npz'), curve=curve,
            curve_f64=curve_f64,
            steps=np.int64(cfg['steps']), seed=np.int64(cfg['seed'])
        )
        print(f'{name}: {cfg["steps"]} steps in {time.time() - t0:.0f}s, '
              f'loss {curve[0, 0]:.4f} -> {curve[-1, 0]:.4f}')


if __name__ == '__main__':
    main()
