"""The callers of the hot path end to end on the GPU: the reference's epoch
loop (models/training.py:70-325 -- batch preparation, forward, loss, backward,
optimizer step, periodic generation / evaluation / checkpoints) over a
DataLoader of synthetic SMILES-shaped token sequences, as
examples/train_vae.py:166-263 drives it (pytoda's dataset and language are not
installed here; a list dataset and a two-attribute language stand in)."""
import json
import logging
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Language:
    start_index, stop_index, padding_index, unknown_index = 2, 3, 0, 1


def _dataset(n, vocab, seed):
    """START, tokens with a learnable regularity (each token determines the
    next), STOP; lengths 6..20."""
    rng = np.random.default_rng(seed)
    data = []
    for _ in range(n):
        length = int(rng.integers(4, 18))
        tok = [int(rng.integers(4, vocab))]
        for _ in range(length - 1):
            tok.append(4 + (tok[-1] * 3 + 1) % (vocab - 4))
        data.append(torch.tensor([2] + tok + [3]))
    return data


@pytest.mark.parametrize('batch_mode', ['packed', 'padded'])
def test_train_vae_epochs(tmp_path, batch_mode):
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd.training import train_vae
    from paccmann_chemistry_amd.training import test_vae as evaluate
    from paccmann_chemistry_amd.utils import collate_fn
    params = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 24,
        'vocab_size': 24, 'stack_width': 8, 'stack_depth': 10, 'n_layers': 2,
        'dropout': 0.1, 'batch_size': 32, 'batch_mode': batch_mode,
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    torch.manual_seed(3)
    train = torch.utils.data.DataLoader(
        _dataset(256, 24, 1), batch_size=32, shuffle=True, drop_last=True,
        collate_fn=collate_fn)
    val = torch.utils.data.DataLoader(
        _dataset(64, 24, 2), batch_size=32, shuffle=False, drop_last=True,
        collate_fn=collate_fn)
    vae = P.TeacherVAE(P.StackGRUEncoder(params), P.StackGRUDecoder(params))
    os.makedirs(tmp_path / 'weights')
    logger = logging.getLogger('test_train_loop')
    before = {k: v.detach().clone() for k, v in vae.state_dict().items()}
    first = evaluate(vae, val, logger, 1.0, batch_mode)
    tracker = None
    for epoch in range(12):
        tracker = train_vae(
            epoch, vae, train, val, _Language(), str(tmp_path),
            search=P.GreedySearch(), optimizer='adam', lr=5e-3,
            kl_growth=0.003, input_keep=0.9, test_input_keep=1.0,
            generate_len=12, log_interval=3, eval_interval=5, save_interval=6,
            loss_tracker=tracker, logger=logger, batch_mode=batch_mode)
    last = evaluate(vae, val, logger, 1.0, batch_mode)
    assert all(np.isfinite(v) for v in last)
    # the reconstruction term falls by a wide margin on this learnable set
    assert last[1] < 0.6 * first[1], (first, last)
    changed = [k for k, v in vae.state_dict().items()
               if not torch.equal(v.cpu(), before[k].cpu())]
    assert len(changed) >= 30, changed    # all but the frozen one-hot tables
    with open(tmp_path / 'loss_tracker.json') as fp:
        saved = json.load(fp)
    assert saved['test_rec_a'] < 10e4
    best = tmp_path / 'weights' / 'best_rec.pt'
    assert best.exists()
    again = P.TeacherVAE(P.StackGRUEncoder(params), P.StackGRUDecoder(params))
    again.load(str(best))
    assert (tmp_path / 'weights' / 'saved_model_epoch_0_iter_6.pt').exists()


def test_resume_reproduces_uninterrupted_run(tmp_path):
    """Two epochs with a persistent optimizer == one epoch, save_checkpoint,
    load_checkpoint into fresh objects, one more epoch: same weights bit for
    bit (dropout 0, no word dropout, fixed batch order)."""
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd.training import (
        load_checkpoint, save_checkpoint, train_vae
    )
    from paccmann_chemistry_amd.utils import collate_fn
    params = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 24,
        'vocab_size': 24, 'stack_width': 8, 'stack_depth': 10, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 32, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    loader = torch.utils.data.DataLoader(
        _dataset(128, 24, 5), batch_size=32, shuffle=False, drop_last=True,
        collate_fn=collate_fn)
    os.makedirs(tmp_path / 'weights')
    logger = logging.getLogger('test_train_loop')
    kw = dict(search=P.GreedySearch(), lr=5e-3, kl_growth=0.003, input_keep=1.,
              log_interval=10 ** 9, eval_interval=10 ** 9,
              save_interval=10 ** 9, logger=logger, batch_mode='padded')

    def fresh():
        torch.manual_seed(11)
        vae = P.TeacherVAE(P.StackGRUEncoder(params),
                           P.StackGRUDecoder(params)).to('cuda')
        return vae, torch.optim.Adam(vae.parameters(), lr=5e-3)

    def epoch(vae, opt, e):
        torch.manual_seed(100 + e)      # the reparameterisation draws
        train_vae(e, vae, loader, loader, _Language(), str(tmp_path),
                  persistent_optimizer=opt, **kw)

    vae, opt = fresh()
    epoch(vae, opt, 0)
    epoch(vae, opt, 1)
    straight = {k: v.clone() for k, v in vae.state_dict().items()}

    vae, opt = fresh()
    epoch(vae, opt, 0)
    save_checkpoint(str(tmp_path / 'ckpt.pt'), vae, opt, 1, {'ep_loss': 0})
    vae, opt = fresh()
    nxt, tracker = load_checkpoint(str(tmp_path / 'ckpt.pt'), vae, opt)
    assert nxt == 1 and tracker == {'ep_loss': 0}
    epoch(vae, opt, nxt)
    for key, val in vae.state_dict().items():
        assert torch.equal(val, straight[key]), key


def test_flat_adam_resume_and_bad_token_in_loop(tmp_path):
    """The bench's optimizer (dp.FlatAdam over flat gradient buckets) through
    train_vae with save_checkpoint / load_checkpoint: a resumed run equals the
    uninterrupted one bit for bit.  And the loop's single per-step
    device->host read (TeacherVAE.loss_item) reports out-of-vocabulary ids."""
    import paccmann_chemistry_amd as P
    from paccmann_chemistry_amd import dp
    from paccmann_chemistry_amd.training import (
        load_checkpoint, save_checkpoint, train_vae
    )
    from paccmann_chemistry_amd.utils import collate_fn
    params = {
        'latent_dim': 16, 'rnn_cell_size': 64, 'embedding_size': 24,
        'vocab_size': 24, 'stack_width': 8, 'stack_depth': 10, 'n_layers': 2,
        'dropout': 0.0, 'batch_size': 32, 'batch_mode': 'padded',
        'embedding': 'one_hot', 'use_stack': True, 'bidirectional': False
    }
    loader = torch.utils.data.DataLoader(
        _dataset(128, 24, 5), batch_size=32, shuffle=False, drop_last=True,
        collate_fn=collate_fn)
    os.makedirs(tmp_path / 'weights')
    logger = logging.getLogger('test_train_loop')
    kw = dict(search=P.GreedySearch(), lr=5e-3, kl_growth=0.003, input_keep=1.,
              log_interval=10 ** 9, eval_interval=10 ** 9,
              save_interval=10 ** 9, logger=logger, batch_mode='padded')

    def fresh():
        torch.manual_seed(11)
        vae = P.TeacherVAE(P.StackGRUEncoder(params),
                           P.StackGRUDecoder(params)).to('cuda')
        buckets = dp.GradientBuckets(vae, overlap=False)
        return vae, dp.FlatAdam(buckets, lr=5e-3)

    def epoch(vae, opt, e, data=loader):
        torch.manual_seed(100 + e)
        train_vae(e, vae, data, data, _Language(), str(tmp_path),
                  persistent_optimizer=opt, **kw)

    vae, opt = fresh()
    start = {k: v.clone() for k, v in vae.state_dict().items()}
    epoch(vae, opt, 0)
    epoch(vae, opt, 1)
    straight = {k: v.clone() for k, v in vae.state_dict().items()}
    assert opt.step_count == 8
    moved = [k for k in straight if not torch.equal(straight[k], start[k])]
    assert len(moved) >= 30

    vae, opt = fresh()
    epoch(vae, opt, 0)
    save_checkpoint(str(tmp_path / 'flat.pt'), vae, opt, 1, {'ep_loss': 0})
    vae, opt = fresh()
    nxt, _ = load_checkpoint(str(tmp_path / 'flat.pt'), vae, opt)
    assert nxt == 1 and opt.step_count == 4
    epoch(vae, opt, nxt)
    for key, val in vae.state_dict().items():
        assert torch.equal(val, straight[key]), key

    bad = [seq.clone() for seq in _dataset(32, 24, 6)]
    bad[5][2] = 999
    bad_loader = torch.utils.data.DataLoader(
        bad, batch_size=32, shuffle=False, collate_fn=collate_fn)
    with pytest.raises(RuntimeError, match='index out of range'):
        epoch(vae, opt, 2, bad_loader)
    vae.check_tokens()   # the status word was cleared by the raise
