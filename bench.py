"""Headline benchmark: SMILES tokens/s of one training step (forward +
backward + Adam) of the stack-augmented GRU VAE on synthetic SMILES-shaped
batches -- BASELINE.json configs[1]: batch 512 per GPU, seq len 120, vocab 60
(one-hot), H=512, L=2, stack 50x50, latent 256, fp32.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (field notes in DESIGN.md section "Measurement").
`value` comes from the timed steps, which carry no instrumentation.  After
them, one extra untimed step per kernel family is run with that family's
launches made through hipExtLaunchKernelGGL with a (start, stop) event pair
bound to each kernel's own dispatch (`pcm_probe_enable`): the per-launch
durations are the kernels' begin -> end on the device, the interval a rocprofv3
kernel trace reports.  `roofline` is the family with the largest summed time,
`roofline_kernels` lists every family, `step_roofline` prices the whole step
(tokens/s x 29.8 MFLOP against the fp32 MFMA peak).
`roofline.traffic` is the memory-side byte count per launch from the committed
rocprofv3 PMC passes (profiles/pmc_traffic.json), not a live measurement;
`cpu_baseline` times the CPU oracle (oracle/stackgru_oracle.py, kind "port") on
a bounded sample on rank 0.

`python bench.py --gpus N` with N > 1 and no rendezvous in the environment starts
the N ranks itself (launch_ranks); `--scaling strong` splits BASELINE.json config
3's global batch of 4096 over them (weak, the default: --batch sequences on every
GPU).  At N = 1 the line also carries `inference` (configs 4 and 5: encoder-only
and greedy / temperature-0.8 decode molecules/s at batch 4096, each with its MFMA
roofline and a CPU baseline) and `ragged_training` (packed mode on ragged
lengths), measured after the timed training steps (--inference 0: off).
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
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, dense bf16 matrix peak


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
    ap.add_argument('--precision', choices=['fp32', 'bf16'], default='fp32',
                    help='fp32: the headline line (the only mode parity is '
                         'claimed on); bf16: a SEPARATE line with dtype '
                         '"bf16": bf16-rounded MFMA operands in the cells and '
                         'the recurrent weight gradients, fp32 accumulation, '
                         'state and optimizer')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='N > 1: weak = --batch sequences on EVERY GPU (global '
                         'batch N x batch); strong = BASELINE.json config 3: '
                         'the global batch --global-batch (4096) split over '
                         'the N GPUs')
    ap.add_argument('--global-batch', type=int, default=4096,
                    help='global batch of --scaling strong')
    ap.add_argument('--inference', type=int, default=1,
                    help='1: after the training measurement also time the '
                         'encoder-only and decode workloads (configs 4, 5) and '
                         'the ragged packed training step; 0: off')
    ap.add_argument('--probe', type=int, default=1,
                    help='1: per-kernel launch durations from extra untimed '
                         'steps after the timed region (roofline); 0: off')
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error('--gpus must be >= 1')
    if args.scaling == 'strong':
        if args.global_batch % args.gpus:
            ap.error('--global-batch must be a multiple of --gpus')
        args.batch = args.global_batch // args.gpus
    if args.ragged:
        args.probe = 0   # per-launch FLOPs vary with the active row count
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous in the
    environment: start the N ranks ourselves, exactly as the driver would
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...`), from
    a parent that has touched no GPU; relay the children's output (rank 0's
    JSON line) and exit with their status."""
    import socket
    import subprocess
    port = os.environ.get('MASTER_PORT')
    if port is None:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = str(s.getsockname()[1])
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """PCM_BENCH_DRYRUN=1 (tests/test_host_cpu.py): the launcher, rendezvous,
    barrier and MAX-over-ranks legs of this file on the CPU (gloo), no GPU and
    no library call; rank 0 prints a line that says so."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group('gloo')
    seen = torch.ones(1, dtype=torch.int64)
    elapsed = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.barrier()
        dist.all_reduce(seen)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({
            'dry_run': True, 'n_gpus': world, 'ranks_seen': int(seen.item()),
            'scaling': args.scaling if world > 1 else 'weak',
            'batch_per_gpu': args.batch,
            'global_batch': world * args.batch,
            'max_elapsed': float(elapsed.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md, HBM3E


def probe_families(p, batch, seq_len):
    """(probe id, kernel, bound, algorithmic work per launch) of the kernels of
    the recurrence and the weight-gradient GEMM.  FLOPs / bytes per
    sequence-step as SURVEY.md section 8d counts them (the one-hot token part is
    a gather, 0 FLOP; the stack moves 2*D*W*4 bytes forward and 3*D*W*4 in the
    reverse pass, whose launch also carries the top layer's gate gradient:
    9 reads / writes of H floats)."""
    hid, width, depth = p['rnn_cell_size'], p['stack_width'], p['stack_depth']
    layers = p['n_layers']
    l0 = 2 * 3 * hid * (width + hid) * batch
    l1 = 2 * 3 * hid * 2 * hid * batch
    stack_fwd = 2 * depth * width * 4 * batch
    stack_bwd = (3 * depth * width * 4 + 9 * hid * 4) * batch
    # weight gradients of one StackGRU, the square-ish products that go
    # through the wide K-major GEMM launches: dW_hh of every layer and dW_ih of
    # layers >= 1 (the skinny control / push product is a separate, narrow
    # launch and is not priced here)
    # ... plus the joint [stack top | one-hot token] product of layer 0 (128
    # columns, of which width + vocabulary carry gradient)
    dw = (2 * 3 * hid * hid + (layers - 1) * 2 * 3 * hid * 2 * hid
          + 2 * 3 * hid * 128) * batch * seq_len
    # Round 3: with two or more layers the bulk of the stack adjoint (the
    # 3*D*W*4 bytes per row) rides as extra blocks in layer 0's reverse-cell
    # launch, and the launch left on the chain carries the stack-top terms and
    # the gate gradient only (PCM_STACK_SPLIT=0: the fused launch as before)
    split = (layers > 1 and os.environ.get('PCM_STACK_SPLIT', '1') != '0'
             and width <= 61)
    if split:
        stack_name = ('stack_bwd_finish_kernel (stack-top terms + top-layer gate '
                      'gradient; the bulk adjoint rides in the layer-0 cell launch)')
        stack_bwd = (9 * hid * 4 + 6 * width * 4) * batch
        cell0_name = ('gru_cell_bwd_kernel (layer 0, + the bulk stack adjoint as '
                      'rider blocks: 3*D*W*4 bytes per row, not priced)')
    else:
        stack_name = 'stack_bwd_kernel (+ top-layer gate gradient)'
        cell0_name = 'gru_cell_bwd_kernel (layer 0)'
    fams = [
        (11, 'gru_cell_fwd_kernel (layer 0)', 'mfma', l0),
        (2, 'stack_step_fwd_kernel', 'hbm', stack_fwd),
        (51, cell0_name, 'mfma', l0),
        (4, stack_name, 'hbm', stack_bwd),
        (6, 'gemm_tn_kernel<0> (dW_hh, dW_ih of one StackGRU)', 'mfma', dw),
    ]
    if layers > 1:
        fams.insert(1, (12, 'gru_cell_fwd_kernel (layers >= 1)', 'mfma', l1))
        fams.insert(4, (52, 'gru_cell_bwd_kernel (layers >= 1)', 'mfma', l1))
    return fams


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes
    (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus
    WRITE_SIZE); None when no pass has been recorded."""
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    try:
        with open(path) as f:
            return json.load(f)['kernels'][kernel.split(' ')[0].split('<')[0]][
                'bytes_per_launch']
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(p, cpu_batch):
    """The oracle (CPU restatement of the reference, torch CPU kernels +
    autograd + torch.optim.Adam as training.py:132-156) on a bounded sample."""
    from oracle import stackgru_oracle as O
    from paccmann_chemistry_amd import synthetic as W
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
    warm, timed = 2, 3   # 2 warm-ups, median of >= 3
    for it in range(warm + timed):
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
    median = sorted(times[warm:])[timed // 2]
    return {
        'value': cpu_batch * SEQ_LEN / median,
        'unit': 'tokens/s',
        'cores': threads,
        'kind': 'port',
        'sample': f'oracle train step (fwd+bwd+Adam), batch {cpu_batch} x '
                  f'seq {SEQ_LEN}, {warm} warm-ups + median of {timed} timed '
                  f'steps, {threads} torch threads',
        'step_seconds': [round(t, 3) for t in times],
    }


# ---- secondary workloads (BASELINE.json configs 4 and 5) -----------
ENCODER_FLOP_PER_MOLECULE = 4.94e6 * SEQ_LEN   # 593 MFLOP at T = 120
DECODE_FLOP_PER_TOKEN = 5.00e6                 # 0.50 GFLOP / molecule at 100


def _time_batches(fn, batches, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(batches):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / batches


def _mfma_roofline(kernel, flop_per_s):
    return {'bound': 'mfma', 'kernel': kernel, 'achieved': flop_per_s / 1e12,
            'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': flop_per_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            'note': 'whole workload: molecules/s x algorithmic FLOP per '
                    'molecule (SURVEY 8d) against the fp32 MFMA peak'}


def inference_cpu_baseline(p, batch=512, generate_len=100):
    """The oracle on the host cores for the two inference workloads, bounded:
    one warm-up + one timed call each at batch 512."""
    from oracle import stackgru_oracle as O
    from paccmann_chemistry_amd import synthetic as W
    O.ATEN_CELL = True
    threads = torch.get_num_threads()
    q = dict(p, batch_size=batch)
    sd = W.make_state_dict(q, 1234)
    tokens = W.synthetic_tokens(SEQ_LEN, batch, q['vocab_size'], 99)
    z = torch.randn(batch, q['latent_dim'],
                    generator=torch.Generator().manual_seed(4321))
    out = {}
    with torch.no_grad():
        for name, fn in (
            ('encoder', lambda: O.encoder_forward(
                O.sub_params(sd, 'encoder'), tokens, q['stack_depth'])),
            ('greedy', lambda: O.generate_from_latent(
                O.sub_params(sd, 'decoder'), z, torch.tensor([2]),
                generate_len, q['stack_depth'])),
        ):
            fn()
            t0 = time.perf_counter()
            fn()
            sec = time.perf_counter() - t0
            out[name] = {
                'value': batch / sec, 'unit': 'molecules/s', 'cores': threads,
                'kind': 'port',
                'sample': f'oracle, batch {batch}, 1 warm-up + 1 timed call, '
                          f'{threads} torch threads',
            }
    return out


def inference_lines(dev, batch=4096, batches=5, generate_len=100, streams=1,
                    cpu=True, host_lists=True):
    """Encoder-only molecules/s (config 4: generate_latent_codes-style
    mu / logvar) and decode molecules/s (config 5: greedy and temperature 0.8
    from prior latents) at batch 4096, fp32, on a bounded sample of the named
    10 M / 1 M sets (the same batch shape repeated).  Entries carry the MFMA
    roofline of the whole workload and, for the workloads the oracle finishes in
    seconds, a CPU baseline."""
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import synthetic as W
    from paccmann_chemistry_amd.training import get_data_preparation
    p = dict(PARAMS, batch_size=batch)
    torch.manual_seed(1234)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.eval()
    base = {'n_gpus': 1, 'dtype': 'f32', 'data': 'synthetic',
            'higher_is_better': True, 'batches': batches, 'streams': streams,
            'unit': 'molecules/s'}
    cpu_base = inference_cpu_baseline(PARAMS) if cpu else {}
    lines = []

    def entry(metric, sec, workload, flop_per_molecule, kernel, **extra):
        value = batch / sec
        lines.append(dict(
            base, metric=metric, value=value, ms_per_batch=sec * 1e3,
            config={'workload': workload},
            roofline=_mfma_roofline(kernel, value * flop_per_molecule),
            **extra))

    # ---- config 4: encoder only ---------------------------------------------
    tokens = W.synthetic_tokens(SEQ_LEN, batch, p['vocab_size'], 1234).to(dev)
    with torch.no_grad():
        if streams > 1:
            def run_enc():
                for _ in P.encode_batches(vae, [tokens] * (2 * streams),
                                          streams=streams):
                    pass
            sec = _time_batches(run_enc, max(1, batches // 2), 1) / (2 * streams)
        else:
            sec = _time_batches(lambda: vae.encode(tokens), batches, 2)
    entry('encoder-only molecules/sec (mu/logvar)', sec,
          f'StackGRUEncoder mu/logvar, no_grad, batch {batch}, seq len '
          f'{SEQ_LEN}, padded, tokens resident on the device',
          ENCODER_FLOP_PER_MOLECULE, 'gru_cell_fwd64_kernel + stack step',
          tokens_per_s=batch * SEQ_LEN / sec, cpu_baseline=cpu_base.get('encoder'))
    if host_lists:
        # the loop of examples/generate_latent_codes.py:113-125: every batch
        # arrives as the DataLoader's list of 1-D host tensors and goes through
        # data_preparation (pinned staging copy + one launch) before encode
        host = W.synthetic_tokens(SEQ_LEN, batch, p['vocab_size'], 1236)
        as_list = [host[:, b].contiguous() for b in range(batch)]
        prep = get_data_preparation('padded')
        n_lists = 2 * max(streams, 3)

        def run_lists():
            with torch.no_grad():
                for _ in P.encode_batches(vae, [as_list] * n_lists,
                                          data_preparation=prep,
                                          streams=max(streams, 3), device=dev):
                    pass
        sec = _time_batches(run_lists, 1, 1) / n_lists
        entry('encoder-only molecules/sec (mu/logvar)', sec,
              f'as above, but every batch handed over as a list of {batch} '
              f'host tensors through data_preparation (host -> device copy '
              f'inside the timed region), encode_batches over '
              f'{max(streams, 3)} streams',
              ENCODER_FLOP_PER_MOLECULE, 'gru_cell_fwd64_kernel + stack step',
              tokens_per_s=batch * SEQ_LEN / sec, host_input=True)

    lengths = sorted(W.ragged_lengths(batch, SEQ_LEN, 4321), reverse=True)
    ragged = W.synthetic_tokens(SEQ_LEN, batch, p['vocab_size'], 1235, lengths)
    packed = torch.nn.utils.rnn.pack_sequence(
        [ragged[:n, b] for b, n in enumerate(lengths)]).to(dev)
    vae.encoder.set_batch_mode('packed')
    with torch.no_grad():
        sec = _time_batches(lambda: vae.encode(packed), batches, 2)
    vae.encoder.set_batch_mode('padded')
    real = float(sum(lengths))
    entry('encoder-only molecules/sec (mu/logvar)', sec,
          f'StackGRUEncoder mu/logvar, no_grad, batch {batch}, ragged lengths '
          f'~N(60,20) clipped to [10,{SEQ_LEN}], packed',
          4.94e6 * real / batch, 'gru_cell_fwd*_kernel + stack step',
          tokens_per_s=real / sec)

    # ---- config 5: decode from prior latents --------------------------------
    gen = torch.Generator(device='cpu').manual_seed(4321)
    z = torch.randn(1, batch, p['latent_dim'], generator=gen).to(dev)
    prime = torch.tensor([2], device=dev)
    end = torch.tensor([3], device=dev)
    for name, search in (('greedy', P.GreedySearch()),
                         ('sampling T=0.8', P.SamplingSearch(temperature=0.8))):
        def run():
            return vae.decoder.generate_from_latent(
                z, prime, end, search=search, generate_len=generate_len,
                seed=99)
        if streams > 1:
            def run_gen():
                for _ in P.generate_batches(
                        vae, [z] * (2 * streams), prime, end,
                        generate_len=generate_len, search=search,
                        streams=streams, seed=99):
                    pass
            sec = _time_batches(run_gen, max(1, batches // 2), 1) / (2 * streams)
        else:
            sec = _time_batches(run, batches, 2)
        entry(f'decode molecules/sec ({name})', sec,
              f'StackGRUDecoder.generate_from_latent, {name}, batch {batch}, '
              f'prime [2], generate_len {generate_len}',
              DECODE_FLOP_PER_TOKEN * generate_len,
              'gru_cell_fwd64_kernel + stack step + logits_choose_kernel',
              tokens_per_s=batch * generate_len / sec,
              cpu_baseline=cpu_base.get('greedy') if name == 'greedy' else None)
    vae.check_tokens()
    return lines


def ragged_training_line(dev, batch, steps=6, warmup=2):
    """The training step on ragged packed batches (lengths ~N(60,20)), the
    reference's default batch_mode (example_params.json:26); tokens/s counts
    the REAL tokens."""
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import dp
    from paccmann_chemistry_amd.ops import KLDivFn
    from paccmann_chemistry_amd.utils import kl_weight
    from paccmann_chemistry_amd import synthetic as W
    p = dict(PARAMS, batch_size=batch, batch_mode='packed')
    torch.manual_seed(1234)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.train()
    lengths = sorted(W.ragged_lengths(batch, SEQ_LEN, 4321), reverse=True)
    cpu_tokens = W.synthetic_tokens(SEQ_LEN, batch, p['vocab_size'], 1234,
                                    lengths)
    cpu_targets = W.shifted_targets(cpu_tokens)
    pack = torch.nn.utils.rnn.pack_sequence
    tokens = pack([cpu_tokens[:n, b] for b, n in enumerate(lengths)]).to(dev)
    targets = pack([cpu_targets[:n, b] for b, n in enumerate(lengths)]).to(dev)
    buckets = dp.GradientBuckets(vae, overlap=False)
    optimizer = dp.FlatAdam(buckets, lr=LR)
    box = [0]

    def step():
        buckets.zero_grad()
        dec_loss, mu, logvar = vae(tokens, tokens, targets)
        kl = KLDivFn.apply(mu, logvar)
        loss = dec_loss + kl_weight(box[0], KL_GROWTH) * kl
        loss.backward()
        buckets.finish()
        optimizer.step()
        box[0] += 1
    # three timed windows, the best one reported: this step is the shortest per
    # launch of the bench (fewer rows per launch as sequences end), and a window
    # in which the host falls behind the GPU reads 27 instead of 22 ms
    windows = [_time_batches(step, steps, warmup if i == 0 else 0)
               for i in range(3)]
    sec = min(windows)
    vae.check_tokens()
    real = int(sum(lengths))
    return {
        'metric': 'SMILES tokens/sec (train fwd+bwd)', 'value': real / sec,
        'unit': 'tokens/s', 'ms_per_step': sec * 1e3, 'steps': steps,
        'windows_ms_per_step': [w * 1e3 for w in windows],
        'sample': f'best of 3 windows of {steps} steps',
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'training step, batch {batch}, packed mode, '
                               f'ragged lengths ~N(60,20) in [10,{SEQ_LEN}] '
                               f'({real} real tokens per batch)'},
        'roofline': _mfma_roofline('whole step', real / sec * 29.8e6),
    }


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch '
                 f'one rank per GPU (or run `python bench.py --gpus N`, which '
                 f'starts them)')
    if os.environ.get('PCM_BENCH_DRYRUN') == '1':
        return dry_run(args, world, rank)
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
    from paccmann_chemistry_amd import synthetic as W

    p = dict(PARAMS, batch_size=args.batch)
    if args.ragged:
        p['batch_mode'] = 'packed'
    torch.manual_seed(1234)
    vae = P.TeacherVAE(P.StackGRUEncoder(p), P.StackGRUDecoder(p)).to(dev)
    vae.train()
    # every step's cross entropy is divided by the rows active over ALL ranks
    # (exact for the ragged packed workload too; one all-reduce of T counts)
    vae.decoder.global_batch_loss = use_dist
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
    P.set_matmul_precision(args.precision)
    bf16 = args.precision == 'bf16'
    mfma_peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS

    def step(global_step):
        buckets.zero_grad()
        dec_loss, mu, logvar = vae(tokens, dec_tokens, targets)
        kl = KLDivFn.apply(mu, logvar)
        kl_w = kl_weight(global_step, KL_GROWTH)
        loss = dp.scale_losses(dec_loss, kl, kl_w, global_share=use_dist)
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
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    vae.check_tokens()
    final_loss = float(loss.item())
    # per-kernel launch durations: one extra untimed step per kernel family
    # (every rank runs the step, rank 0 records)
    probed = []
    probe_offset_us = 0.0
    if args.probe and not args.ragged and rank == 0:
        # what the event pair adds to a reading (see pcm_probe_calibrate)
        p_us, c_us = ctypes.c_double(), ctypes.c_double()
        _lib.check(lib.pcm_probe_calibrate(
            400, ctypes.byref(p_us), ctypes.byref(c_us),
            _lib.stream_ptr(dev)), 'probe calibrate')
        probe_offset_us = max(p_us.value - c_us.value, 0.0)
        calibration = {'empty_launch_probe_us': p_us.value,
                       'empty_launch_chain_us': c_us.value}
    if args.probe and not args.ragged:
        probe_ms, probe_n, dropped = (ctypes.c_double(), ctypes.c_int64(),
                                      ctypes.c_int64())
        for k, (pid, name, bound, work) in enumerate(
                probe_families(p, args.batch, SEQ_LEN)):
            if rank == 0:
                _lib.check(lib.pcm_probe_enable(pid, 4 * SEQ_LEN), 'probe')
            step(args.warmup + args.steps + k)
            torch.cuda.synchronize()
            if rank == 0:
                lib.pcm_probe_read(ctypes.byref(probe_ms),
                                   ctypes.byref(probe_n),
                                   ctypes.byref(dropped))
                lib.pcm_probe_enable(0, 0)
                probed.append((name, bound, work, probe_ms.value,
                               probe_n.value))
        barrier()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    assert final_loss == final_loss, 'loss is NaN'

    if rank == 0:
        # (ragged: rank 0's token count stands for every rank's, the
        # length distributions are the same)
        tokens_per_step = world * tokens_per_batch
        value = tokens_per_step * args.steps / elapsed
        entries = []
        # Round 3: at this shape the forward stack step has no launch of its own
        # (api.hip: enqueue_step, `fuse`): layer 0's cell derives controls, push
        # value and the new stack top in a prologue and the top cell's launch
        # carries the stack update as extra blocks.  The probe shows it (the
        # stack family then has the first step's launches only); the cells'
        # entries say so, their priced work stays the GEMM FLOPs.
        fused_fwd = any(n.startswith('stack_step_fwd') and 0 < k < SEQ_LEN
                        for n, _, _, _, k in probed)
        for name, bound, work, total_ms, launches in probed:
            if launches <= 0:
                continue
            if fused_fwd and name == 'gru_cell_fwd_kernel (layer 0)':
                name = ('gru_cell_fwd_kernel (layer 0, + the stack step\'s '
                        'controls / push value / new stack top as a prologue, '
                        'not priced)')
            elif fused_fwd and name == 'gru_cell_fwd_kernel (layers >= 1)':
                name = ('gru_cell_fwd_kernel (layers >= 1, + the stack update '
                        'as rider blocks: 2*D*W*4 bytes per row, not priced)')
            elif fused_fwd and name.startswith('stack_step_fwd'):
                name = ('stack_step_fwd_kernel (first step of a sequence only: '
                        'fused into the cells elsewhere)')
            total_ms = max(total_ms - launches * probe_offset_us * 1e-3, 1e-9)
            avg_s = total_ms / launches * 1e-3
            if name.startswith('gemm_tn'):
                # the family's launches differ in size: price them together
                # (two StackGRUs' worth of launches in one step)
                achieved = 2 * work / (total_ms * 1e-3) / 1e12
            elif bound == 'mfma':
                achieved = work / avg_s / 1e12
            else:
                achieved = work / avg_s / 1e9
            peak = mfma_peak if bound == 'mfma' else PEAK_HBM_GBS
            entries.append({
                'bound': bound, 'kernel': name, 'achieved': achieved,
                'peak': peak, 'unit': 'TFLOP/s' if bound == 'mfma' else 'GB/s',
                'frac': achieved / peak, 'traffic': pmc_traffic(name),
                'avg_launch_us': avg_s * 1e6, 'launches': launches,
                'ms_per_step': total_ms,
            })
        roofline = None
        if entries:
            # the dominant KERNEL is chosen by symbol, as a rocprofv3 --stats
            # table lists them: the forward cell's two layers are one symbol
            fwd = [e for e in entries
                   if e['kernel'].startswith('gru_cell_fwd_kernel')]
            symbols = [e for e in entries if e not in fwd]
            if fwd:
                ms = sum(e['ms_per_step'] for e in fwd)
                n = sum(e['launches'] for e in fwd)
                flop = sum(e['achieved'] * 1e12 * e['avg_launch_us'] * 1e-6
                           * e['launches'] for e in fwd)
                ach = flop / (ms * 1e-3) / 1e12
                symbols.append({
                    'bound': 'mfma',
                    'kernel': 'gru_cell_fwd_kernel (all layers'
                              + (', stack step fused in: 2 launches per '
                                 'forward step' if fused_fwd else '') + ')',
                    'achieved': ach, 'peak': mfma_peak, 'unit': 'TFLOP/s',
                    'frac': ach / mfma_peak, 'traffic': fwd[0]['traffic'],
                    'avg_launch_us': ms * 1e3 / n, 'launches': n,
                    'ms_per_step': ms,
                })
            roofline = dict(max(symbols, key=lambda e: e['ms_per_step']))
            if fused_fwd and roofline['kernel'].startswith('gru_cell_fwd'):
                roofline['note'] = (
                    'priced on the cells\' GEMM FLOPs alone although the two '
                    'launches now also carry the whole forward stack step '
                    '(control softmax, push tanh, new stack top, 2*D*W*4 bytes '
                    'per row of stack update); as three launches per step '
                    '(PCM_FUSE_STACK=0) the same cells read 13.97 / 18.43 us = '
                    '0.49 of peak and the step 0.4 ms more '
                    '(profiles/r03_bench_unfused_forward.json)')
            roofline['sample'] = (
                'kernel begin -> end from the events hipExtLaunchKernelGGL '
                'binds to each dispatch, every launch of one untimed step '
                'after the timed region, minus what the event pair adds to '
                'an empty dependent launch (probe_calibration)')
            roofline['probe_calibration'] = dict(
                calibration, offset_us=probe_offset_us)
        step_tflops = value * 29.8e6 / 1e12
        step_roofline = {
            'bound': 'mfma', 'achieved': step_tflops,
            'peak': mfma_peak, 'unit': 'TFLOP/s',
            'frac': step_tflops / mfma_peak,
            'note': 'whole step: tokens/s x 29.8 MFLOP per token (SURVEY 8d)',
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(p, args.cpu_batch)
        inference = ragged = None
        if args.inference and world == 1 and not bf16 and not args.ragged:
            # configs 4 and 5 and the ragged training step, after the timed
            # region; the training model's buffers are released first
            torch.cuda.empty_cache()
            inference = inference_lines(dev, cpu=not args.no_cpu_baseline)
            # the same workloads through the product's pipelined batch loops
            # (inference.encode_batches / generate_batches: three independent
            # batches in flight on three HIP streams, results bit-identical)
            inference += [
                e for e in inference_lines(dev, streams=3, cpu=False,
                                           host_lists=False)
                if 'ragged' not in e['config']['workload']]
            torch.cuda.empty_cache()
            ragged = ragged_training_line(dev, args.batch)
        line = {
            'metric': 'SMILES tokens/sec (train fwd+bwd)',
            'value': value, 'unit': 'tokens/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': args.scaling if world > 1 else 'weak',
            'vs_baseline': None, 'dtype': 'bf16' if bf16 else 'f32',
            'data': 'synthetic',
            'config': {
                'workload': 'Stack-GRU VAE training step (fwd+bwd+Adam), '
                            f'batch {args.batch}/GPU, seq len {SEQ_LEN}, '
                            'vocab 60 one-hot, H=512, L=2, stack 50x50, '
                            'latent 256, dropout 0.2, ' +
                            ('packed mode, ragged lengths ~N(60,20) in '
                             '[10,120]' if args.ragged else 'padded mode') +
                            ('; MIXED PRECISION (not the headline, no parity '
                             'claim): bf16-rounded operands on bf16 MFMAs in '
                             'the GRU cells and recurrent weight gradients, '
                             'fp32 accumulation / state / optimizer'
                             if bf16 else ''),
                'global_batch': world * args.batch, 'seq_len': SEQ_LEN,
                'parallelism': f'dp{world}',
            },
            'roofline': roofline, 'roofline_kernels': entries,
            'step_roofline': step_roofline, 'cpu_baseline': cpu,
            'final_loss': final_loss,
            'inference': inference, 'ragged_training': ragged,
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
