"""Bidirectional encoder (B=512, T=120, config model): the two directions on two
streams against one after the other.  Forward + backward, ms per call."""
import sys
import time

import torch

sys.path.insert(0, '.')
import paccmann_chemistry_amd as P
from paccmann_chemistry_amd import synthetic as W

p = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.2, 'batch_size': 512, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': True
}
dev = torch.device('cuda:0')
torch.manual_seed(0)
enc = P.StackGRUEncoder(p).to(dev).train()
tokens = W.synthetic_tokens(120, 512, 60, 1).to(dev)
for mode in (True, False, True, False):
    enc.concurrent_directions = mode
    for it in range(8):
        if it == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        enc.zero_grad()
        mu, logvar = enc.encoder_train_step(tokens)
        (mu.sum() + logvar.sum()).backward()
    torch.cuda.synchronize()
    print('concurrent' if mode else 'sequential',
          round((time.perf_counter() - t0) / 5 * 1e3, 3), 'ms per fwd+bwd',
          flush=True)
