"""Secondary measurements of SURVEY.md section 8(d): encoder-only molecules/s
(BASELINE.json configs[3], generate_latent_codes-style mu/logvar throughput)
and decode molecules/s (configs[4], greedy and temperature-0.8 sampling from
prior latents), on a bounded sample of the named workloads (the full 10 M /
1 M sets are the same batches repeated).  One JSON line per workload.

    python bench_inference.py [--batch 4096] [--batches 10]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402  (model configuration of the headline benchmark)


STREAMS = 1   # --streams N > 1: N independent batches in flight (inference.py)


def timed(fn, batches, warmup=2, pipe=None):
    """Seconds per batch: fn() once per batch, or -- with --streams N > 1 --
    pipe(n), the product's pipelined loop over n batches."""
    if STREAMS > 1 and pipe is not None:
        for _ in pipe(max(warmup, 2 * STREAMS)):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in pipe(batches):
            pass
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / batches
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batches):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / batches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--batches', type=int, default=10)
    ap.add_argument('--generate-len', type=int, default=100)
    ap.add_argument('--streams', type=int, default=1,
                    help='> 1: that many independent batches in flight on '
                         'separate HIP streams (paccmann_chemistry_amd.'
                         'inference.encode_batches / generate_batches), one '
                         "model; a chain's kernel boundaries and prologue / "
                         "epilogue phases then run under another batch's MFMA "
                         'loops')
    args = ap.parse_args()
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import synthetic as W
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    p = dict(bench.PARAMS, batch_size=args.batch)
    torch.manual_seed(1234)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.eval()
    seq_len, batch = bench.SEQ_LEN, args.batch
    common = {'n_gpus': 1, 'dtype': 'f32', 'data': 'synthetic',
              'higher_is_better': True, 'batches': args.batches,
              'streams': args.streams}
    if args.streams > 1:
        global STREAMS
        STREAMS = args.streams
        args.batches = max(args.batches, 4 * args.streams)
        common['batches'] = args.batches

    # ---- encoder only, fixed length (padded) and ragged (packed) ----------
    tokens = W.synthetic_tokens(seq_len, batch, p['vocab_size'], 1234).to(dev)
    with torch.no_grad():
        sec = timed(lambda: vae.encode(tokens), args.batches,
                    pipe=lambda n: P.encode_batches(vae, [tokens] * n,
                                                    streams=args.streams))
    print(json.dumps(dict(
        common, metric='encoder-only molecules/sec (mu/logvar)',
        value=batch / sec, unit='molecules/s', ms_per_batch=sec * 1e3,
        tokens_per_s=batch * seq_len / sec,
        config={'workload': f'StackGRUEncoder mu/logvar, no_grad, batch '
                            f'{batch}, seq len {seq_len}, padded'})), flush=True)

    lengths = sorted(W.ragged_lengths(batch, seq_len, 4321), reverse=True)
    ragged = W.synthetic_tokens(seq_len, batch, p['vocab_size'], 1235, lengths)
    # (packed on the host, ONE copy to the device: 4096 per-row copies would
    # show up in a profile as if the library issued them)
    packed = torch.nn.utils.rnn.pack_sequence(
        [ragged[:n, b] for b, n in enumerate(lengths)]).to(dev)
    torch.manual_seed(1234)
    q = dict(p, batch_mode='packed')
    vae_packed = P.TeacherVAE(P.StackGRUEncoder(q), P.StackGRUDecoder(q)).to(dev)
    vae_packed.eval()
    with torch.no_grad():
        sec = timed(lambda: vae_packed.encode(packed), args.batches,
                    pipe=lambda n: P.encode_batches(vae_packed, [packed] * n,
                                                    streams=args.streams))
    print(json.dumps(dict(
        common, metric='encoder-only molecules/sec (mu/logvar)',
        value=batch / sec, unit='molecules/s', ms_per_batch=sec * 1e3,
        tokens_per_s=float(sum(lengths)) / sec,
        config={'workload': f'StackGRUEncoder mu/logvar, no_grad, batch '
                            f'{batch}, ragged lengths ~N(60,20) clipped to '
                            f'[10,{seq_len}], packed'})), flush=True)

    # ---- decode from prior latents -----------------------------------------
    gen = torch.Generator(device='cpu').manual_seed(4321)
    z = torch.randn(1, batch, p['latent_dim'], generator=gen).to(dev)
    prime = torch.tensor([2], device=dev)
    end = torch.tensor([3], device=dev)
    for name, search in (('greedy', P.GreedySearch()),
                         ('sampling T=0.8', P.SamplingSearch(temperature=0.8))):
        def run():
            return vae.decoder.generate_from_latent(
                z, prime, end, search=search,
                generate_len=args.generate_len, seed=99)
        sec = timed(run, args.batches, warmup=3,
                    pipe=lambda n: P.generate_batches(
                        vae, [z] * n, prime, end,
                        generate_len=args.generate_len, search=search,
                        streams=args.streams, seed=99))
        print(json.dumps(dict(
            common, metric=f'decode molecules/sec ({name})',
            value=batch / sec, unit='molecules/s', ms_per_batch=sec * 1e3,
            tokens_per_s=batch * args.generate_len / sec,
            config={'workload': f'StackGRUDecoder.generate_from_latent, '
                                f'{name}, batch {batch}, prime [2], '
                                f'generate_len {args.generate_len}'})),
            flush=True)


if __name__ == '__main__':
    main()
