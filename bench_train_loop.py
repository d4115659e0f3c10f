"""Secondary measurement: tokens/s of the reference-style epoch loop
(paccmann_chemistry_amd.training.train_vae = models/training.py:70-325:
DataLoader -> batch preparation -> forward -> loss -> backward -> optimizer,
with its per-step host reads) on the configs[1] model, next to bench.py's bare
training step.  One JSON line per batch mode.

    python bench_train_loop.py [--batch 512] [--batches 24]
"""
import argparse
import json
import logging
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


class _Language:
    start_index, stop_index, padding_index, unknown_index = 2, 3, 0, 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--batches', type=int, default=24)
    args = ap.parse_args()
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import synthetic as W
    from paccmann_chemistry_amd.training import train_vae
    from paccmann_chemistry_amd.utils import collate_fn
    logger = logging.getLogger('bench_train_loop')
    for mode in ('padded', 'packed'):
        p = dict(bench.PARAMS, batch_size=args.batch, batch_mode=mode)
        torch.manual_seed(1234)
        vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p))
        n = args.batch * args.batches
        if mode == 'padded':
            lengths = [bench.SEQ_LEN] * n
        else:
            lengths = list(W.ragged_lengths(n, bench.SEQ_LEN, 4321))
        tokens = W.synthetic_tokens(bench.SEQ_LEN, n, p['vocab_size'], 7,
                                    lengths)
        data = [tokens[:ln, i].clone() for i, ln in enumerate(lengths)]
        loader = torch.utils.data.DataLoader(
            data, batch_size=args.batch, shuffle=True, drop_last=True,
            collate_fn=collate_fn)
        with tempfile.TemporaryDirectory() as tmp:
            os.makedirs(os.path.join(tmp, 'weights'))
            kw = dict(search=P.GreedySearch(), optimizer='adam', lr=5e-4,
                      kl_growth=0.003, input_keep=0.85, generate_len=100,
                      log_interval=10 ** 9, eval_interval=10 ** 9,
                      save_interval=10 ** 9, logger=logger, batch_mode=mode)
            train_vae(0, vae, loader, loader, _Language(), tmp, **kw)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            train_vae(1, vae, loader, loader, _Language(), tmp, **kw)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
        real = float(sum(lengths))
        print(json.dumps({
            'metric': 'SMILES tokens/sec (train_vae epoch loop)',
            'value': real / sec, 'unit': 'tokens/s', 'n_gpus': 1,
            'ms_per_step': sec / args.batches * 1e3, 'dtype': 'f32',
            'data': 'synthetic', 'higher_is_better': True,
            'config': {'workload': f'train_vae epoch, batch {args.batch}, '
                                   f'{mode} mode, {args.batches} batches, '
                                   'torch.optim.Adam, DataLoader + device '
                                   'batch preparation'}}), flush=True)


if __name__ == '__main__':
    main()
