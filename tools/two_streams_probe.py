"""Probe (not part of the product): do two independent half-batch training
steps, each on its own HIP stream and host thread, overlap on the chip?"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import paccmann_chemistry_amd as P  # noqa: E402
from oracle import weights as W  # noqa: E402
from paccmann_chemistry_amd.ops import KLDivFn  # noqa: E402

dev = torch.device('cuda:0')
half = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def make(seed):
    p = dict(bench.PARAMS, batch_size=half)
    torch.manual_seed(seed)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.train()
    tok = W.synthetic_tokens(bench.SEQ_LEN, half, 60, seed).to(dev)
    return vae, tok, W.shifted_targets(tok)


def step(m, stream, n):
    vae, tok, tgt = m
    with torch.cuda.stream(stream):
        for _ in range(n):
            for q in vae.parameters():
                q.grad = None
            loss, mu, logvar = vae(tok, tok, tgt)
            (loss + 0.1 * KLDivFn.apply(mu, logvar)).backward()
    stream.synchronize()


a, b = make(1), make(2)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
step(a, sa, 2); step(b, sb, 2)
torch.cuda.synchronize()
N = 5
t0 = time.perf_counter(); step(a, sa, N); one = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
ta = threading.Thread(target=step, args=(a, sa, N))
tb = threading.Thread(target=step, args=(b, sb, N))
ta.start(); tb.start(); ta.join(); tb.join()
two = (time.perf_counter() - t0) / N
print(f'half batch {half}: one chain {one*1e3:.2f} ms/step; two chains on two '
      f'streams {two*1e3:.2f} ms per pair of steps')
