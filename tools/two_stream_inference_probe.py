"""Encoder-only and greedy-decode throughput at batch 4096 with independent
batches alternating over 1, 2 or 3 HIP streams (independent launch chains:
one chain's prologue / epilogue phases and kernel boundaries can run under the
other's MFMA loops).

    python tools/two_stream_inference_probe.py [batch [stream counts]]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import synthetic as W
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    p = dict(bench.PARAMS, batch_size=batch)
    torch.manual_seed(1234)
    counts = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 \
        else [1, 2, 3]
    for nstreams in counts:
        # one model replica per stream (per-module status words / caches)
        vaes = []
        for _ in range(nstreams):
            torch.manual_seed(1234)
            vaes.append(P.TeacherVAE(P.StackGRUEncoder(p),
                                     P.StackGRUDecoder(p)).to(dev).eval())
        tokens = W.synthetic_tokens(bench.SEQ_LEN, batch, p['vocab_size'],
                                    1234).to(dev)
        z = torch.randn(1, batch, p['latent_dim'], device=dev)
        prime = torch.tensor([2], device=dev)
        end = torch.tensor([3], device=dev)
        streams = [torch.cuda.Stream() for _ in range(nstreams)]
        torch.cuda.synchronize()

        def run(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                k = i % nstreams
                with torch.cuda.stream(streams[k]):
                    fn(vaes[k])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n

        with torch.no_grad():
            enc = lambda m: m.encode(tokens)
            dec = lambda m: m.decoder.generate_from_latent(
                z, prime, end, search=P.GreedySearch(), generate_len=100)
            run(enc, 2 * nstreams)
            sec = run(enc, 12)
            print(f'batch {batch} streams {nstreams}: encoder-only {batch / sec:,.0f} '
                  f'molecules/s', flush=True)
            run(dec, 2 * nstreams)
            sec = run(dec, 12)
            print(f'batch {batch} streams {nstreams}: greedy decode {batch / sec:,.0f} '
                  f'molecules/s', flush=True)
        del vaes


if __name__ == '__main__':
    main()
