"""Secondary measurements of SURVEY.md section 8(d): encoder-only molecules/s
(BASELINE.json configs[3], generate_latent_codes-style mu/logvar throughput)
and decode molecules/s (configs[4], greedy and temperature-0.8 sampling from
prior latents), on a bounded sample of the named workloads (the full 10 M /
1 M sets are the same batches repeated).  One JSON line per workload; the same
measurements ride on bench.py's line as `inference` (bench.inference_lines).

    python bench_inference.py [--batch 4096] [--batches 10] [--streams N]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402  (model configuration and the measurement itself)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=4096)
    ap.add_argument('--batches', type=int, default=10)
    ap.add_argument('--generate-len', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--streams', type=int, default=1,
                    help='> 1: that many independent batches in flight on '
                         'separate HIP streams (paccmann_chemistry_amd.'
                         'inference.encode_batches / generate_batches), one '
                         "model; a chain's kernel boundaries and prologue / "
                         "epilogue phases then run under another batch's MFMA "
                         'loops')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    for line in bench.inference_lines(
            dev, batch=args.batch, batches=args.batches,
            generate_len=args.generate_len, streams=args.streams,
            cpu=not args.no_cpu_baseline):
        print(json.dumps(line), flush=True)


if __name__ == '__main__':
    main()
