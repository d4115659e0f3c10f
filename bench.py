"""Headline benchmark: SMILES tokens/s of one training step (forward +
backward + Adam) of the stack-augmented GRU VAE on synthetic SMILES-shaped
batches -- BASELINE.json configs[1]: batch 512 per GPU, seq len 120, vocab 60
(one-hot), H=512, L=2, stack 50x50, latent 256, fp32.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the field notes in DESIGN.md section
"Measurement").  `roofline` is for the dominant kernel (by summed time: the
fused GRU cell of the reverse recurrence, `--probe 5`; `--probe 1` is the
forward cell), its duration measured live with HIP events on the launch stream
during the timed steps; `roofline.traffic` is the HBM-side byte count per launch
from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), not a live
measurement; `cpu_baseline` times the CPU oracle (oracle/stackgru_oracle.py,
kind "port") on a bounded sample on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PARAMS = {
    'latent_dim': 256, 'rnn_cell_size': 512, 'embedding_size': 60,
    'vocab_size': 60, 'stack_width': 50, 'stack_depth': 50, 'n_layers': 2,
    'dropout': 0.2, 'batch_size': 512, 'batch_mode': 'padded',
    'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
}
SEQ_LEN = 120
LR = 5e-4
KL_GROWTH = 0.003
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md, dense fp32 matrix peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=PARAMS['batch_size'],
                    help='sequences per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-batch', type=int, default=512)
    ap.add_argument('--ragged', action='store_true',
                    help='secondary workload: packed mode on ragged lengths '
                         '~N(60,20) clipped to [10,120] (tokens/s counts the '
                         'real tokens); the default is the fixed-length '
                         'padded batch BASELINE.json names')
    ap.add_argument('--probe', type=int, default=5,
                    help='kernel id for the roofline probe (5 = reverse fused '
                         'cell, 1 = forward fused cell, 0 = off)')
    args = ap.parse_args()
    if args.ragged:
        args.probe = 0   # per-launch FLOPs vary with the active row count
    return args


def cell_flops_per_launch(p, batch):
    """Algorithmic FLOPs of one fused-cell launch, averaged over the layers
    (2*3H*(W+H) layer 0, 2*3H*2H further layers, per
    sequence-step; the one-hot token part is a gather, 0 FLOP)."""
    hid, width, layers = p['rnn_cell_size'], p['stack_width'], p['n_layers']
    per_seq = [2 * 3 * hid * (width + hid)] + \
        [2 * 3 * hid * 2 * hid] * (layers - 1)
    return batch * sum(per_seq) / layers


PROBE_KERNELS = {1: 'gru_cell_fwd_kernel', 5: 'gru_cell_bwd_kernel'}


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus
    WRITE_SIZE); None when no pass has been recorded."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f)['kernels'][kernel]['bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(p, cpu_batch):
    """The oracle (CPU restatement of the reference, torch CPU kernels +
    autograd + torch.optim.Adam as training.py:132-156) on a bounded sample."""
    from oracle import stackgru_oracle as O
    from oracle import weights as W
    O.ATEN_CELL = True   # the layer-step on ATen's GRU cell, as nn.GRU runs it
    threads = torch.get_num_threads()
    q = dict(p, batch_size=cpu_batch, dropout=0.0)
    sd = W.make_state_dict(q, 1234)
    sd = {k: v.clone().requires_grad_('embedding' not in k)
          for k, v in sd.items()}
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=LR)
    tokens = W.synthetic_tokens(SEQ_LEN, cpu_batch, q['vocab_size'], 99)
    targets = W.shifted_targets(tokens)
    gen = torch.Generator().manual_seed(7)
    times = []
    for it in range(3):
        eps = torch.randn(cpu_batch, q['latent_dim'], generator=gen)
        t0 = time.perf_counter()
        opt.zero_grad()
        dec_loss, mu, logvar, _, _ = O.vae_forward(
            sd, tokens, tokens, targets, eps, q['stack_depth']
        )
        loss, _ = O.vae_loss(dec_loss, mu, logvar, KL_GROWTH, step=it)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    return {
        'value': cpu_batch * SEQ_LEN / best,
        'unit': 'tokens/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'oracle train step (fwd+bwd+Adam), batch {cpu_batch} x '
                  f'seq {SEQ_LEN}, 1 warm-up + best of 2 timed steps, '
                  f'{threads} torch threads',
    }


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    import torch.distributed as dist
    # PCM_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, hook-driven
    # async all-reduce, barriers, MAX over ranks) in a one-rank RCCL group: the
    # rehearsal tests/test_gpu_parity.py does on a single GPU
    use_dist = world > 1 or os.environ.get('PCM_BENCH_FORCE_DIST') == '1'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        dist.init_process_group('nccl', device_id=torch.device(
            'cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import _lib, dp
    from paccmann_chemistry_amd.ops import KLDivFn
    from paccmann_chemistry_amd.utils import kl_weight
    from oracle import weights as W  # synthetic inputs only

    p = dict(PARAMS, batch_size=args.batch)
    if args.ragged:
        p['batch_mode'] = 'packed'
    torch.manual_seed(1234)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.train()
    # identical random-init weights on every rank (seeded above)
    tokens = W.synthetic_tokens(SEQ_LEN, args.batch, p['vocab_size'],
                                1234 + rank).to(dev)
    targets = W.shifted_targets(tokens)
    # decoder input with whole-row word dropout (utils/utils.py:65-75)
    g = torch.Generator().manual_seed(4321 + rank)
    keep = torch.bernoulli(torch.full((SEQ_LEN, ), 0.85), generator=g)
    dec_tokens = tokens.clone()
    dec_tokens[(keep == 0).to(dev)] = 1
    tokens_per_batch = args.batch * SEQ_LEN
    if args.ragged:
        lengths = sorted(W.ragged_lengths(args.batch, SEQ_LEN, 4321 + rank),
                         reverse=True)
        cpu_tokens = W.synthetic_tokens(SEQ_LEN, args.batch, p['vocab_size'],
                                        1234 + rank, lengths)
        cpu_targets = W.shifted_targets(cpu_tokens)
        pack = torch.nn.utils.rnn.pack_sequence
        tokens = dec_tokens = pack(
            [cpu_tokens[:n, b].to(dev) for b, n in enumerate(lengths)])
        targets = pack(
            [cpu_targets[:n, b].to(dev) for b, n in enumerate(lengths)])
        tokens_per_batch = int(sum(lengths))

    buckets = dp.GradientBuckets(vae, overlap=use_dist,
                                 force_collective=use_dist)
    # torch.optim.Adam's update (training.py:132) as one fused launch per
    # flat bucket (tests/test_gpu_parity.py: test_flat_adam_matches_torch_adam)
    optimizer = dp.FlatAdam(buckets, lr=LR)
    lib = _lib.load()

    def step(global_step):
        buckets.zero_grad()
        dec_loss, mu, logvar = vae(tokens, dec_tokens, targets)
        kl = KLDivFn.apply(mu, logvar)
        kl_w = kl_weight(global_step, KL_GROWTH)
        loss = dp.scale_losses(dec_loss, kl, kl_w)
        loss.backward()
        buckets.finish()
        optimizer.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    cells_per_step = 2 * SEQ_LEN * p['n_layers']
    if args.probe:
        # the event brackets cost ~8 % when every launch carries them, so
        # only the launches of the FIRST timed step are bracketed (the probe
        # stops recording once its event budget is used)
        _lib.check(lib.pcm_probe_enable(args.probe, cells_per_step), 'probe')
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    vae.check_tokens()
    probe_ms, probe_n, dropped = (ctypes.c_double(), ctypes.c_int64(),
                                  ctypes.c_int64())
    if args.probe:
        lib.pcm_probe_read(ctypes.byref(probe_ms), ctypes.byref(probe_n),
                           ctypes.byref(dropped))
        lib.pcm_probe_enable(0, 0)
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    # what an empty HIP-event bracket reads on this stream (marker cost that
    # every probed duration carries on top of the kernel itself)
    pairs = [(torch.cuda.Event(enable_timing=True),
              torch.cuda.Event(enable_timing=True)) for _ in range(200)]
    for e0, e1 in pairs:
        e0.record()
        e1.record()
    torch.cuda.synchronize()
    empty_us = sum(e0.elapsed_time(e1) for e0, e1 in pairs) / len(pairs) * 1e3
    assert final_loss == final_loss, 'loss is NaN'

    if rank == 0:
        # (ragged: rank 0's token count stands for every rank's, the
        # length distributions are the same)
        tokens_per_step = world * tokens_per_batch
        value = tokens_per_step * args.steps / elapsed
        roofline = None
        if args.probe in PROBE_KERNELS and probe_n.value > 0:
            # a bracket reads the kernel plus the cost of its two event
            # markers; the empty bracket measured above is taken off so the
            # figure is the launch duration rocprofv3 reports
            raw_s = probe_ms.value / probe_n.value * 1e-3
            avg_s = max(raw_s - empty_us * 1e-6, 1e-9)
            achieved = cell_flops_per_launch(p, args.batch) / avg_s / 1e12
            roofline = {
                'bound': 'mfma', 'kernel': PROBE_KERNELS[args.probe],
                'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS,
                'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                'traffic': pmc_traffic(PROBE_KERNELS[args.probe]),
                'avg_launch_us': avg_s * 1e6, 'launches': probe_n.value,
                'bracket_us': raw_s * 1e6, 'empty_bracket_us': empty_us,
                'sample': 'HIP-event brackets on every launch of the first '
                          'timed step, minus the empty-bracket reading',
            }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(p, args.cpu_batch)
        line = {
            'metric': 'SMILES tokens/sec (train fwd+bwd)',
            'value': value, 'unit': 'tokens/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {
                'workload': 'Stack-GRU VAE training step (fwd+bwd+Adam), '
                            f'batch {args.batch}/GPU, seq len {SEQ_LEN}, '
                            'vocab 60 one-hot, H=512, L=2, stack 50x50, '
                            'latent 256, dropout 0.2, ' +
                            ('packed mode, ragged lengths ~N(60,20) in '
                             '[10,120]' if args.ragged else 'padded mode'),
                'global_batch': world * args.batch, 'seq_len': SEQ_LEN,
                'parallelism': f'dp{world}',
            },
            'roofline': roofline, 'cpu_baseline': cpu,
            'final_loss': final_loss,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
