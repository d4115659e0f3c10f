"""Build-container check that the CPU baseline bench.py reports (the oracle,
kind "port") costs what the reference itself costs: one training step
(forward + backward + Adam) of the configs[1] model at a bounded batch, timed
for the imported reference (/root/reference, pytoda / rdkit stubbed as in
make_golden.py) and for the oracle on the same weights, tokens and threads.
Prints one line; the result is quoted in DESIGN.md.

    python oracle/time_vs_reference.py [batch]
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import make_golden as G  # noqa: E402
from oracle import stackgru_oracle as O  # noqa: E402
from oracle import weights as W  # noqa: E402

P = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.0, 'batch_size': 128, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}
T = 120


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    p = dict(P, batch_size=batch)
    ref_vae, ref_loss, _ = G.import_reference()
    sd = W.make_state_dict(p, 1234)
    tokens = W.synthetic_tokens(T, batch, 60, 99)
    targets = W.shifted_targets(tokens)

    vae = G.build_reference_vae(ref_vae, p, sd)
    vae.train()
    opt = torch.optim.Adam(vae.parameters(), lr=5e-4)

    def ref_step(i):
        opt.zero_grad()
        dec_loss, mu, logvar = vae(tokens, tokens, targets)
        loss, _ = ref_loss.vae_loss_function(dec_loss, mu, logvar,
                                             kl_growth=0.003, step=i)
        loss.backward()
        opt.step()

    osd = {k: v.clone().requires_grad_('embedding' not in k)
           for k, v in sd.items()}
    oopt = torch.optim.Adam([v for v in osd.values() if v.requires_grad],
                            lr=5e-4)
    gen = torch.Generator().manual_seed(7)

    def oracle_step(i):
        eps = torch.randn(batch, 256, generator=gen)
        oopt.zero_grad()
        dec_loss, mu, logvar, _, _ = O.vae_forward(osd, tokens, tokens,
                                                   targets, eps, 50)
        loss, _ = O.vae_loss(dec_loss, mu, logvar, 0.003, step=i)
        loss.backward()
        oopt.step()

    out = {'reference': [], 'oracle': []}
    O.ATEN_CELL = True   # as bench.py's cpu_baseline times it
    # alternate the two (the container's vCPUs are shared and noisy) and keep
    # each one's best step
    for rnd in range(2):
        for name, fn in (('oracle', oracle_step), ('reference', ref_step)):
            for i in range(3):
                t0 = time.perf_counter()
                fn(i)
                out[name].append(time.perf_counter() - t0)
            print(name, [round(x, 2) for x in out[name][-3:]], flush=True)
    ref, orc = min(out['reference']), min(out['oracle'])
    print(f'batch {batch} x {T}, {torch.get_num_threads()} threads, best step: '
          f'reference {ref:.2f} s ({batch * T / ref:.0f} tokens/s), oracle '
          f'{orc:.2f} s ({batch * T / orc:.0f} tokens/s)')


if __name__ == '__main__':
    main()
