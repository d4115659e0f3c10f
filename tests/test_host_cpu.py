"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol
the header declares, and the host-side mirror keeps the reference's
constructor / params schema / state_dict keys / error behaviour."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'paccmann_hip.h')


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pcm_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from paccmann_chemistry_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    raw = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(raw, name), f'{name} declared but not exported'
    # the ctypes prototypes cover the header one to one
    assert sorted(_lib.PROTOTYPES) == names
    lib = _lib.load()
    assert lib.pcm_version() == _lib.ABI_VERSION


def test_size_helpers_without_gpu():
    """Pure host arithmetic of the ABI (no device call)."""
    from paccmann_chemistry_amd import _lib
    lib = _lib.load()
    dims = _lib.make_dims(512, 60, 60, 512, 2, 50, 50, True)
    train = lib.pcm_stackgru_saved_bytes(ctypes.byref(dims), 120, 1)
    infer = lib.pcm_stackgru_saved_bytes(ctypes.byref(dims), 120, 0)
    assert train > infer > 0
    # training keeps hidden states, 4H gates and the [D,W] stack of every step
    floor = 120 * 512 * (2 * 512 + 2 * 4 * 512 + 2500) * 4
    assert floor < train < 2 * floor
    off, ld = ctypes.c_size_t(), ctypes.c_size_t()
    assert lib.pcm_stackgru_out_view(
        ctypes.byref(dims), 120, 1, ctypes.byref(off), ctypes.byref(ld)
    ) == 0
    assert ld.value == 512 and off.value > 0
    bad = _lib.make_dims(0, 60, 60, 512, 2, 50, 50, True)
    assert lib.pcm_stackgru_saved_bytes(ctypes.byref(bad), 120, 1) == 0
    assert b'must be positive' in lib.pcm_last_error()
    assert lib.pcm_kl_scratch_floats(131072) == 128


def test_argument_checks_reject_before_any_launch():
    """Bad arguments come back as error codes with a message, before anything
    is enqueued (so no device is needed to see them)."""
    from paccmann_chemistry_amd import _lib
    lib = _lib.load()
    dims = _lib.make_dims(4, 20, 20, 32, 2, 7, 5, True)
    buf = (ctypes.c_float * 64)()
    somewhere = ctypes.cast(buf, ctypes.c_void_p)
    weights = _lib.PcmStackGRUParams()
    for name, _ in weights._fields_:
        field = getattr(weights, name)
        if hasattr(field, '__len__'):
            for i in range(2):
                field[i] = somewhere.value
        else:
            setattr(weights, name, somewhere.value)
    need = lib.pcm_stackgru_saved_bytes(ctypes.byref(dims), 3, 0)

    def forward(active, saved_bytes=need, dropout=0.0, lengths=somewhere):
        act = None
        if active is not None:
            act = ctypes.cast((ctypes.c_int32 * 3)(*active), ctypes.c_void_p)
        return lib.pcm_stackgru_forward(
            ctypes.byref(dims), ctypes.byref(weights), somewhere, 3, lengths,
            act, None, 0, None, dropout, 0, 0, somewhere, saved_bytes, None,
            None, None, None)

    assert forward([4, 2, 3]) != 0            # grows again at step 2
    assert b'non-increasing' in lib.pcm_last_error()
    assert forward([5, 2, 1]) != 0            # more rows than the batch
    assert forward([4, 2, 1], lengths=None) != 0
    assert b'needs lengths' in lib.pcm_last_error()
    assert forward(None, saved_bytes=16) != 0
    assert b'workspace' in lib.pcm_last_error()
    assert forward(None, dropout=1.5) != 0
    assert b'dropout' in lib.pcm_last_error()
    assert lib.pcm_prepare_batch(None, None, 4, 3, 0, None, 1.0, 0, None, 1,
                                 None, None, None, None) != 0
    assert b'null pointer' in lib.pcm_last_error()
    assert lib.pcm_prepare_batch(somewhere, somewhere, 4, 3, 1, None, 1.0, 0,
                                 None, 1, somewhere, somewhere, somewhere,
                                 None) != 0
    assert b'step_base' in lib.pcm_last_error()


PARAMS = {
    'latent_dim': 16, 'rnn_cell_size': 32, 'embedding_size': 20,
    'vocab_size': 20, 'stack_width': 5, 'stack_depth': 7, 'n_layers': 2,
    'dropout': 0.1, 'batch_size': 4, 'batch_mode': 'padded',
    'embedding': 'one_hot'
}


def test_state_dict_keys_match_reference_layout():
    """The 34 keys listed in SURVEY.md section 8b (checkpoint interchange)."""
    import paccmann_chemistry_amd as P
    from oracle import weights as W
    vae = P.TeacherVAE(P.StackGRUEncoder(PARAMS), P.StackGRUDecoder(PARAMS))
    sd = vae.state_dict()
    expected = W.vae_shapes(dict(PARAMS, use_stack=True))
    assert len(sd) == 34
    assert {k: tuple(v.shape) for k, v in sd.items()} == expected
    # one-hot table is frozen and the identity (stack_rnn.py:88-97)
    assert not vae.encoder.embedding.weight.requires_grad
    assert torch.equal(vae.encoder.embedding.weight, torch.eye(20))
    bi = P.StackGRUEncoder(dict(PARAMS, bidirectional=True))
    assert any(k.startswith('backward_stackgru.') for k in bi.state_dict())
    assert bi.hidden_to_mu.in_features == 64


def test_same_default_init_as_reference_order():
    """Same constructor order => same draws under the same torch seed as
    nn.Linear / nn.Embedding / nn.GRU in the reference."""
    import paccmann_chemistry_amd as P
    p = dict(PARAMS, embedding='learned', embedding_size=12)
    torch.manual_seed(11)
    ours = P.StackGRU(p)
    torch.manual_seed(11)
    lin1 = torch.nn.Linear(32, 3)
    lin2 = torch.nn.Linear(32, 5)
    emb = torch.nn.Embedding(20, 12, padding_idx=0)
    gru = torch.nn.GRU(17, 32, 2, dropout=0.1)
    assert torch.equal(ours.stack_controls_layer.weight, lin1.weight)
    assert torch.equal(ours.stack_input_layer.bias, lin2.bias)
    assert torch.equal(ours.embedding.weight, emb.weight)
    for name, param in gru.named_parameters():
        assert torch.equal(getattr(ours.gru, name), param), name


def test_error_behaviour():
    import paccmann_chemistry_amd as P
    with pytest.raises(TypeError):
        P.StackGRU(dict(PARAMS, batch_mode=None))
    with pytest.raises(NotImplementedError):
        P.StackGRU(dict(PARAMS, batch_mode='ragged'))
    enc = P.StackGRUEncoder(PARAMS)
    packed = torch.nn.utils.rnn.pack_sequence([torch.tensor([2, 5, 3])])
    with pytest.raises(TypeError):
        enc._forward_pass_padded(packed, None, None)
    with pytest.raises(TypeError):
        enc._forward_pass_packed(torch.zeros(3, 4).long(), None, None)
    with pytest.raises(ValueError):
        enc._post_gru_reshape(torch.zeros(2, 5, 32))
    hidden = torch.arange(32.).repeat(2, 4, 1)
    assert torch.equal(enc._post_gru_reshape(hidden), hidden[-1])
    dec = P.StackGRUDecoder(dict(PARAMS, batch_mode='Packed'))
    with pytest.raises(TypeError):
        dec._forward_pass_packed(torch.zeros(3, 4).long(), None, None, None)
    vae = P.TeacherVAE(enc, dec)
    with pytest.raises(TypeError):
        vae._associate_language(object())


def test_search_golden_tokens():
    """The reference's golden vectors (utils/tests/test_search.py:7-122)."""
    from paccmann_chemistry_amd.search import (
        BeamSearch, GreedySearch, SamplingSearch
    )
    logits = torch.tensor(
        [[[0.1, 0.2, 0.3, 0.4, 0.5], [0.5, 0.4, 0.3, 0.2, 0.1]] * 5] * 2
    )
    greedy = GreedySearch()
    assert greedy(logits).tolist() == [[4, 0] * 5] * 2
    for step_logits, token in zip(logits.permute(1, 0, 2), [4, 0] * 5):
        assert greedy.step(step_logits).view(-1).tolist() == [token] * 2
    sampling = SamplingSearch()
    assert list(sampling(logits).shape) == [2, 10]
    for step_logits in logits.permute(1, 0, 2):
        assert list(sampling.step(step_logits).shape) == [2, 1]
    with pytest.raises(ValueError):
        greedy(logits[0])
    base = [[4, 4, 4], [0, 0, 0]] * 4
    tokens, _ = BeamSearch()(logits)
    assert tokens.tolist() == [base + [[4, 4, 3], [0, 1, 0]]] * 2
    tokens, _ = BeamSearch(temperature=0.001)(logits)
    assert tokens.tolist() == [base + [[4, 4, 4], [0, 1, 2]]] * 2
    tokens, _ = BeamSearch(temperature=3.)(logits)
    assert tokens.tolist() == [
        [[4, 4, 4], [0, 0, 0]] * 3 + [[4, 4, 4], [0, 0, 1], [4, 4, 4],
                                      [0, 1, 0]]
    ] * 2
    tokens, _ = BeamSearch(top_tokens=2)(logits)
    assert tokens.tolist() == [base + [[4, 4, 3], [0, 1, 0]]] * 2
    search = BeamSearch()
    beams = [[[list(), 0.0]]] * 2
    for step_logits in logits.permute(1, 0, 2):
        stacked = torch.stack([step_logits] * (search.beam_width + 1))
        _, beams = search.step(stacked, beams)
    got = [
        [[int(t) for t in col] for col in zip(*[beam[0] for beam in sample])]
        for sample in beams
    ]
    assert got == [base + [[4, 4, 3], [0, 1, 0]]] * 2


def test_batch_preparation():
    """utils/utils.py:29-168 semantics."""
    from paccmann_chemistry_amd import utils
    from paccmann_chemistry_amd.training import get_data_preparation
    batch = [torch.tensor([2, 7, 8, 9, 3]), torch.tensor([2, 5, 3]),
             torch.tensor([2, 6, 6, 3])]
    batch = utils.collate_fn(batch)
    assert [len(s) for s in batch] == [5, 4, 3]
    cpu = torch.device('cpu')
    padded = torch.nn.utils.rnn.pad_sequence(batch)
    enc, dec, tgt = utils.sequential_data_preparation(
        padded, input_keep=1, device=cpu)
    assert torch.equal(enc, dec) and enc.shape == (5, 3)
    assert torch.equal(tgt[:-1], enc[1:]) and tgt[-1].eq(0).all()
    torch.manual_seed(0)
    _, dec, _ = utils.sequential_data_preparation(
        padded, input_keep=0.5, device=cpu)
    dropped = (dec != enc).any(dim=1)
    for t in range(5):  # whole time rows become <UNK>=1
        if dropped[t]:
            assert dec[t].eq(1).all()
    enc_p, dec_p, tgt_p = utils.packed_sequential_data_preparation(
        batch, input_keep=1, device=cpu)
    tokens, lengths = utils.packed_to_lengths(enc_p)
    assert lengths.tolist() == [5, 4, 3] and torch.equal(tokens, padded)
    targets, _ = utils.packed_to_lengths(tgt_p)
    assert targets[:, 1].tolist() == [6, 6, 3, 0, 0]
    data, sizes = utils.perpare_packed_input(enc_p)
    assert [len(d) for d in data] == sizes.tolist() == [3, 3, 3, 2, 1]
    assert [len(s) for s in utils.unpack_sequence(enc_p)] == [5, 4, 3]
    with pytest.raises(TypeError):
        get_data_preparation(None)
    with pytest.raises(NotImplementedError):
        get_data_preparation('ragged')
    enc2, _, _ = get_data_preparation('padded')(
        batch, input_keep=1., start_index=2, end_index=3, device=cpu)
    assert torch.equal(enc2, padded)


def test_kl_weight_and_factories():
    from conftest import Golden
    from paccmann_chemistry_amd.hyperparams import (
        OPTIMIZER_FACTORY, SEARCH_FACTORY
    )
    from paccmann_chemistry_amd.utils import kl_weight
    g = Golden('kl_weight')
    for growth in (0.0015, 0.003, 0.004):
        got = np.array([kl_weight(int(s), growth) for s in g.data['steps']])
        assert np.array_equal(got, g.data[f'g{growth}'])
    assert sorted(SEARCH_FACTORY) == ['beam', 'greedy', 'sampling']
    assert sorted(OPTIMIZER_FACTORY) == [
        'adadelta', 'adagrad', 'adam', 'adamax', 'rmsprop', 'sgd'
    ]


def test_reference_import_paths_resolve():
    """Code written against the reference's package layout
    (examples/train_vae.py:9-16, generate_latent_codes.py:9-15) imports the
    MI355X implementation through the compatibility package."""
    import subprocess
    import sys
    code = (
        'from paccmann_chemistry.utils import (collate_fn, get_device, '
        'disable_rdkit_logging, sequential_data_preparation, '
        'packed_sequential_data_preparation, kl_weight, packed_to_padded)\n'
        'from paccmann_chemistry.models.vae import (StackGRUDecoder, '
        'StackGRUEncoder, TeacherVAE)\n'
        'from paccmann_chemistry.models import StackGRU\n'
        'from paccmann_chemistry.models.training import train_vae, test_vae\n'
        'from paccmann_chemistry.utils.hyperparams import SEARCH_FACTORY, '
        'OPTIMIZER_FACTORY\n'
        'from paccmann_chemistry.utils.search import GreedySearch\n'
        'from paccmann_chemistry.utils.loss_functions import '
        'vae_loss_function\n'
        'import paccmann_chemistry_amd as P\n'
        'assert TeacherVAE is P.TeacherVAE and StackGRU is P.StackGRU\n'
        'disable_rdkit_logging()\n'
        'print("ok")\n'
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, '-c', code], cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == 'ok'


def test_packed_to_padded_sorted_batch():
    from paccmann_chemistry_amd.utils import packed_to_padded
    seq = [torch.tensor([5, 6, 7]), torch.tensor([8, 9]), torch.tensor([4])]
    out = packed_to_padded(seq, [])
    assert out.tolist() == [[5.0, 8.0, 4.0], [6.0, 9.0, 0.0], [7.0, 0.0, 0.0]]


@pytest.mark.parametrize('name', ['prepare_padded', 'prepare_packed'])
def test_batch_preparation_against_reference_fixture(name, monkeypatch):
    """The host-side restatements of sequential_data_preparation /
    packed_sequential_data_preparation against what the REFERENCE produced for
    the same batch (tests/golden/prepare_*.npz, oracle/make_golden.py), with
    the reference's recorded Bernoulli draws injected."""
    from conftest import GOLDEN
    from paccmann_chemistry_amd import utils as U
    g = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    flat, offs = torch.from_numpy(g['flat']), g['offsets']
    batch = [flat[offs[i]:offs[i + 1]] for i in range(len(offs) - 1)]
    keep = torch.from_numpy(g['keep']).float()
    cursor = {'at': 0}

    def draws(rows, input_keep):
        at = cursor['at']
        cursor['at'] += rows
        return keep[at:at + rows]

    monkeypatch.setattr(U, '_dropout_rows', draws)
    cpu = torch.device('cpu')
    if int(g['packed']):
        enc, dec, tgt = U.packed_sequential_data_preparation(
            batch, input_keep=float(g['input_keep']), device=cpu)
        assert torch.equal(enc.batch_sizes, torch.from_numpy(g['batch_sizes']))
        enc, dec, tgt = enc.data, dec.data, tgt.data
    else:
        enc, dec, tgt = U.sequential_data_preparation(
            torch.nn.utils.rnn.pad_sequence(batch),
            input_keep=float(g['input_keep']), device=cpu)
    assert cursor['at'] == keep.numel()
    assert torch.equal(enc, torch.from_numpy(g['enc']))
    assert torch.equal(dec, torch.from_numpy(g['dec']))
    assert torch.equal(tgt, torch.from_numpy(g['tgt']))
    assert 0 < int((dec != enc).sum())     # the draws did drop tokens


def test_packed_to_padded_follows_stop_tokens():
    """utils/utils.py:171-203 derives the ended rows from STOP targets."""
    from paccmann_chemistry_amd.utils import packed_to_padded
    seqs = [torch.tensor([2, 5, 6, 3]), torch.tensor([2, 7, 8, 3]),
            torch.tensor([2, 9, 3]), torch.tensor([2, 3])]
    tgt = [torch.cat([q[1:], torch.zeros(1, dtype=torch.long)]) for q in seqs]
    packed = torch.nn.utils.rnn.pack_sequence(tgt)
    steps, prev = [], 0
    for n in packed.batch_sizes.tolist():
        steps.append(packed.data[prev:prev + n])
        prev += n
    outs = [torch.arange(10, 10 + n).float()
            for n in packed.batch_sizes.tolist()]
    want = torch.tensor([[10., 10, 10, 10], [11, 11, 11, 11],
                         [12, 12, 12, 0], [13, 13, 0, 0]])
    assert torch.equal(packed_to_padded(outs, steps), want)
    assert torch.equal(packed_to_padded(outs), want)
    with pytest.raises(ValueError):
        packed_to_padded([outs[0], outs[0][:2]], steps)   # no STOP seen yet


def _bench_dry_run(*flags, env_extra=None):
    """bench.py's launcher / rendezvous / reduction legs on the CPU
    (PCM_BENCH_DRYRUN=1: gloo, no GPU, no library call)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCM_BENCH_DRYRUN='1')
    for key in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(key, None)
    env.update(env_extra or {})
    return subprocess.run(
        [sys.executable, os.path.join(root, 'bench.py')] + list(flags),
        env=env, capture_output=True, text=True, timeout=300
    ), json


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no rendezvous in the environment) starts
    two ranks itself; the line's n_gpus equals --gpus."""
    done, json = _bench_dry_run('--gpus', '2', '--steps', '1', '--warmup', '0')
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [l for l in done.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, done.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['ranks_seen'] == 2
    assert line['scaling'] == 'weak' and line['global_batch'] == 1024
    assert line['max_elapsed'] == pytest.approx(0.002)   # MAX over ranks


def test_bench_strong_scaling_splits_the_global_batch():
    done, json = _bench_dry_run('--gpus', '2', '--scaling', 'strong')
    assert done.returncode == 0, done.stderr[-2000:]
    line = json.loads(
        [l for l in done.stdout.splitlines() if l.startswith('{')][0])
    assert line['scaling'] == 'strong'
    assert line['batch_per_gpu'] == 2048 and line['global_batch'] == 4096


def test_bench_refuses_a_world_size_other_than_gpus():
    done, _ = _bench_dry_run('--gpus', '4', env_extra={
        'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert done.returncode != 0
    assert '--gpus 4 but WORLD_SIZE=1' in done.stderr
