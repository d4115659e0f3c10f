"""Train / test loops: thin mirror of the reference's
``paccmann_chemistry/models/training.py`` -- the callers of the hot path.
Same function names and arguments; TensorBoard / rdkit image logging (off the
path, optional in the reference) is not carried over.
"""
import json
import logging
import os
from time import time

import numpy as np
import torch

from .hyperparams import OPTIMIZER_FACTORY
from .loss_functions import vae_loss_function
from .search import SamplingSearch
from .utils import (
    crop_start_stop, get_device, packed_sequential_data_preparation,
    prepare_batch_on_device, sequential_data_preparation, unpack_sequence
)


def _on_hip(device):
    device = get_device() if device is None else torch.device(device)
    return device.type == 'cuda'


def _prepare_packed(batch, input_keep, start_index, end_index, device=None):
    """training.py:328-336; on a HIP device the three PackedSequences are
    built by one launch from one staged copy (utils.prepare_batch_on_device)."""
    if _on_hip(device) and isinstance(batch, (list, tuple)):
        return prepare_batch_on_device(
            batch, True, input_keep=input_keep, device=device
        )
    return packed_sequential_data_preparation(
        batch, input_keep=input_keep, start_index=start_index,
        end_index=end_index, device=device
    )


def _prepare_padded(batch, input_keep, start_index, end_index, device):
    """training.py:339-348 (one launch on a HIP device, as above)."""
    if _on_hip(device) and isinstance(batch, (list, tuple)):
        return tuple(prepare_batch_on_device(
            batch, False, input_keep=input_keep, device=device
        ))
    padded_batch = torch.nn.utils.rnn.pad_sequence(batch).to(device)
    return sequential_data_preparation(
        padded_batch, input_keep=input_keep, start_index=start_index,
        end_index=end_index, device=device
    )


def get_data_preparation(mode):
    """training.py:351-367."""
    if not isinstance(mode, str):
        raise TypeError('Argument `mode` should be a string.')
    mode = mode.capitalize()
    MODES = {'Padded': _prepare_padded, 'Packed': _prepare_packed}
    if mode not in MODES:
        raise NotImplementedError(
            f'Unknown mode: {mode}. Available modes: {MODES.keys()}'
        )
    return MODES[mode]


def save_checkpoint(path, model, optimizer, epoch, loss_tracker=None):
    """Everything a resumed run needs (the reference saves the model's
    state_dict only, vae.py:674-676 / training.py:208-213, and cannot resume):
    model and optimizer state, the next epoch, the best-loss bookkeeping."""
    torch.save({
        'model': model.state_dict(), 'optimizer': optimizer.state_dict(),
        'epoch': int(epoch), 'loss_tracker': loss_tracker
    }, path)


def load_checkpoint(path, model, optimizer=None):
    """Restore what save_checkpoint wrote; returns (epoch, loss_tracker)."""
    state = torch.load(path, map_location=get_device(), weights_only=True)
    model.load_state_dict(state['model'])
    if optimizer is not None:
        optimizer.load_state_dict(state['optimizer'])
    return state['epoch'], state['loss_tracker']


def test_vae(model, dataloader, logger, input_keep, batch_mode):
    """training.py:20-67: average (loss, reconstruction, KL) over the data."""
    device = get_device()
    data_preparation = get_data_preparation(batch_mode)
    vae_model = model.to(device)
    vae_model.eval()
    test_loss, test_rec, test_kl_div = 0, 0, 0
    with torch.no_grad():
        for _iter, batch in enumerate(dataloader):
            if (_iter + 1) % 500 == 0:
                logger.info(
                    f'**TESTING**\t Processing batch {_iter}/{len(dataloader)}'
                )
            encoder_seq, decoder_seq, target_seq = data_preparation(
                batch, input_keep=input_keep, start_index=2, end_index=3,
                device=device
            )
            decoder_loss, mu, logvar = vae_model(
                encoder_seq, decoder_seq, target_seq
            )
            loss, kl_div = vae_loss_function(
                decoder_loss, mu, logvar, eval_mode=True
            )
            test_loss += loss.item()
            test_rec += decoder_loss.item()
            test_kl_div += kl_div.item()
    vae_model.check_tokens()
    test_loss /= len(dataloader)
    test_rec /= len(dataloader)
    test_kl_div /= len(dataloader)
    vae_model.train()
    return test_loss, test_rec, test_kl_div


test_vae.__test__ = False  # not a pytest test


def _reconstruction_example(outputs, target_seq, smiles_language):
    """Token-level version of print_example_reconstruction
    (utils/utils.py:302-343)."""
    idx = np.random.randint(len(outputs))
    pred = crop_start_stop(outputs[idx].long(), smiles_language)
    if isinstance(target_seq, torch.Tensor):
        target = target_seq.permute(1, 0)[idx].tolist()
    else:
        target = target_seq[idx].tolist()
    to_text = getattr(smiles_language, 'token_indexes_to_smiles', None)
    if to_text is None:
        return target, pred
    return to_text(target), to_text(pred)


def train_vae(
    epoch, model, train_dataloader, val_dataloader, smiles_language,
    model_dir, search=SamplingSearch(), optimizer='adam', lr=1e-3,
    kl_growth=0.0015, input_keep=1., test_input_keep=0., start_index=2,
    end_index=3, generate_len=100, log_interval=100, eval_interval=200,
    save_interval=200, loss_tracker=None, logger=None, batch_mode='padded',
    writer=None, persistent_optimizer=None
):  # yapf: disable
    """training.py:70-325: one epoch.  The optimizer is re-created at the top
    of every epoch exactly as the reference does (training.py:132), unless the
    caller passes ``persistent_optimizer`` (not in the reference: an optimizer
    instance that keeps its moments across epochs and can be checkpointed with
    ``save_checkpoint`` for a resume)."""
    if loss_tracker is None:
        loss_tracker = {
            'test_loss_a': 10e4, 'test_rec_a': 10e4, 'test_kld_a': 10e4,
            'ep_loss': 0, 'ep_rec': 0, 'ep_kld': 0
        }
    logger = logger or logging.getLogger('train_vae')
    device = get_device()
    data_preparation = get_data_preparation(batch_mode)
    vae_model = model.to(device)
    vae_model.train()
    train_loss = 0
    if persistent_optimizer is not None:
        optimizer = persistent_optimizer
    else:
        optimizer = OPTIMIZER_FACTORY[optimizer](vae_model.parameters(), lr=lr)
    t = time()
    loss = decoder_loss = None
    # the per-step host read of training.py:154 (loss.item()), with the
    # bad-token status of the whole model folded into the same read, is
    # collected one step late: the next step is already queued on the GPU when
    # the host blocks, so the device never waits for the DataLoader, the batch
    # preparation or the launch of the next forward
    pending = None
    for _iter, batch in enumerate(train_dataloader):
        global_step = epoch * len(train_dataloader) + _iter
        encoder_seq, decoder_seq, target_seq = data_preparation(
            batch, input_keep=input_keep, start_index=start_index,
            end_index=end_index, device=device
        )
        optimizer.zero_grad()
        decoder_loss, mu, logvar = vae_model(
            encoder_seq, decoder_seq, target_seq
        )
        loss, kl_div = vae_loss_function(
            decoder_loss, mu, logvar, kl_growth=kl_growth, step=global_step
        )
        loss.backward()
        ticket = vae_model.loss_ticket(loss)
        optimizer.step()
        if pending is not None:
            train_loss += pending.result()
        pending = ticket

        if _iter and _iter % log_interval == 0:
            train_loss += pending.result()
            pending = None
            logger.info(
                f'***TRAINING***\t Epoch: {epoch}, '
                f'step {_iter}/{len(train_dataloader)}.\t'
                f'Loss: {train_loss/log_interval:2.4f}, time spent: {time()-t}'
            )
            t = time()
            train_loss = 0
            if batch_mode.lower() == 'packed':
                target_seq = unpack_sequence(target_seq)
            target, pred = _reconstruction_example(
                vae_model.decoder.outputs.cpu(), target_seq, smiles_language
            )
            logger.info(
                f'Sample input:\n\t {target}, model reconstructed:\n\t {pred}'
            )
        saves = _iter and _iter % save_interval == 0
        evaluates = _iter and _iter % eval_interval == 0
        if (saves or evaluates) and pending is not None:
            # a checkpoint (or an evaluation that may write best_*.pt) must not
            # be written from weights that a step with clamped, out-of-range
            # token ids has updated: collect this step's status first.  The
            # reference raises inside the embedding lookup, before the update
            # (stack_rnn.py:150); here the step that raises has already been
            # applied to the in-memory model, but nothing is saved from it.
            train_loss += pending.result()
            pending = None
        if _iter and _iter % save_interval == 0:
            save_dir = os.path.join(
                model_dir, f'weights/saved_model_epoch_{epoch}_iter_{_iter}.pt'
            )
            vae_model.save(save_dir)
            logger.info(f'***SAVING***\t Epoch {epoch}, saved model.')
        if _iter and _iter % eval_interval == 0:
            vae_model.eval()
            latent_z = torch.randn(1, mu.shape[0], mu.shape[1]).to(device)
            molecule_iter = vae_model.generate(
                latent_z,
                prime_input=torch.tensor([smiles_language.start_index]
                                         ).to(device),
                end_token=torch.tensor([smiles_language.stop_index]
                                       ).to(device),
                generate_len=generate_len, search=search
            )
            mol = next(molecule_iter)
            logger.info(f'\nSample Generated Molecule:\n{mol.tolist()}')
            test_loss, test_rec, test_kld = test_vae(
                vae_model, val_dataloader, logger, test_input_keep, batch_mode
            )
            logger.info(
                f'***TESTING*** \t Epoch {epoch}, test loss = '
                f'{test_loss:.4f}, reconstruction = {test_rec:.4f}, '
                f'KL = {test_kld:.4f}.'
            )
            vae_model.train()
            for key, val, name in (
                ('test_loss_a', test_loss, 'loss'),
                ('test_rec_a', test_rec, 'rec'), ('test_kld_a', test_kld, 'kld')
            ):
                if val < loss_tracker[key]:
                    loss_tracker.update({key: val, f'ep_{name}': epoch})
                    vae_model.save(
                        os.path.join(model_dir, f'weights/best_{name}.pt')
                    )
            with open(os.path.join(model_dir, 'loss_tracker.json'), 'w') as fp:
                json.dump(loss_tracker, fp)
    if pending is not None:
        # (a bad token in the last step raises here)
        train_loss += pending.result()
    if loss is not None:
        logger.info(
            f'Epoch {epoch} finished, \t Training Loss = {loss.item():.4f},'
            f'Reconstruction = {decoder_loss.item():.4f}, KL = '
            f'{(loss - decoder_loss).item():.8f}'
        )
    return loss_tracker
