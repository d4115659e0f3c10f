#!/usr/bin/env python3
"""BASELINE.json config 1 on this build: the call sequence of the reference's
``examples/train_vae.py`` (language -> datasets -> params augmentation ->
DataLoaders -> encoder / decoder / TeacherVAE -> ``train_vae`` per epoch,
reference lines 60-264) over a synthetic SMILES-shaped ``.smi`` set, through
the reference's import paths (``paccmann_chemistry.*``, ``pytoda.*``).

    python examples/train_vae_synthetic.py OUT_DIR [--lines 10000]
        [--params examples/example_params_synthetic.json] [--epochs N]

The model runs on the MI355X (there is no CPU path); pytoda is the stand-in of
``standins/`` unless the real package is installed."""
import argparse
import json
import logging
import os
import sys
from time import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _path in (os.path.join(ROOT, 'standins'), ROOT):
    if _path not in sys.path:
        sys.path.insert(0, _path)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from paccmann_chemistry.models.training import train_vae  # noqa: E402
from paccmann_chemistry.models.vae import (  # noqa: E402
    StackGRUDecoder, StackGRUEncoder, TeacherVAE
)
from paccmann_chemistry.utils import (  # noqa: E402
    collate_fn, disable_rdkit_logging, get_device
)
from paccmann_chemistry.utils.hyperparams import SEARCH_FACTORY  # noqa: E402
from pytoda.datasets import SMILESDataset  # noqa: E402
from pytoda.smiles.smiles_language import SMILESLanguage  # noqa: E402

logging.basicConfig(stream=sys.stdout, level=logging.INFO)
logger = logging.getLogger('train_vae_synthetic')

ATOMS = ['C', 'C', 'C', 'c', 'c', 'N', 'O', 'n', 'S', 'F', 'Cl', 'Br', '[nH]',
         '[C@H]', '[C@@H]', '[N+]', '[O-]', 'o', 's', 'P', 'I']
BONDS = ['', '', '', '', '=', '#', '/', '\\']


def synthetic_smiles(rng, max_tokens=100):
    """A SMILES-shaped string (atoms, bonds, branches, ring closures): the
    token statistics of a molecule line, not valid chemistry."""
    target = int(np.clip(rng.normal(45, 18), 6, max_tokens))
    out, open_branches, ring, tokens = [], 0, 1, 0
    while tokens < target:
        roll = rng.random()
        if roll < 0.08 and tokens > 2 and open_branches < 3:
            out.append('(')
            open_branches += 1
        elif roll < 0.14 and open_branches:
            out.append(')')
            open_branches -= 1
        elif roll < 0.22 and ring < 9:
            out.append(str(ring))
            if rng.random() < 0.5:
                ring += 1
        else:
            bond = BONDS[int(rng.integers(len(BONDS)))]
            if bond:
                out.append(bond)
                tokens += 1
            out.append(ATOMS[int(rng.integers(len(ATOMS)))])
        tokens += 1
    out.append(')' * open_branches)
    return ''.join(out)


def write_smi(path, lines, seed):
    rng = np.random.default_rng(seed)
    with open(path, 'w') as fp:
        for index in range(lines):
            fp.write(f'{synthetic_smiles(rng)}\tmol{index}\n')


def make_dataset(path, params, language, device):
    return SMILESDataset(
        path, smiles_language=language, padding=False,
        selfies=params.get('selfies', False),
        add_start_and_stop=params.get('add_start_stop_token', True),
        augment=params.get('augment_smiles', False),
        canonical=params.get('canonical', False),
        kekulize=params.get('kekulize', False),
        all_bonds_explicit=params.get('all_bonds_explicit', False),
        all_hs_explicit=params.get('all_hs_explicit', False),
        remove_bonddir=params.get('remove_bonddir', False),
        remove_chirality=params.get('remove_chirality', False),
        backend='lazy', device=device
    )


def make_loader(dataset, params):
    return torch.utils.data.DataLoader(
        dataset, batch_size=params.get('batch_size', 64),
        collate_fn=collate_fn, drop_last=True, shuffle=True,
        pin_memory=params.get('pin_memory', True),
        num_workers=params.get('num_workers', 8)
    )


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('out_dir')
    ap.add_argument('--lines', type=int, default=10000)
    ap.add_argument('--test-lines', type=int, default=1000)
    ap.add_argument('--params', default=os.path.join(
        ROOT, 'examples', 'example_params_synthetic.json'))
    ap.add_argument('--epochs', type=int, default=None,
                    help='overrides params["epochs"]; the reference runs '
                         'range(epochs + 1) epochs (examples/train_vae.py:240)')
    ap.add_argument('--batch-size', type=int, default=None)
    ap.add_argument('--training-name', default='synthetic')
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)

    device = get_device()
    disable_rdkit_logging()
    torch.manual_seed(args.seed)
    with open(args.params) as fp:
        params = json.load(fp)
    if args.epochs is not None:
        params['epochs'] = args.epochs
    if args.batch_size is not None:
        params['batch_size'] = args.batch_size

    os.makedirs(args.out_dir, exist_ok=True)
    train_path = os.path.join(args.out_dir, 'train.smi')
    test_path = os.path.join(args.out_dir, 'test.smi')
    write_smi(train_path, args.lines, args.seed + 1)
    write_smi(test_path, args.test_lines, args.seed + 2)

    model_dir = os.path.join(args.out_dir, args.training_name)
    for sub in ('weights', 'results', os.path.join('logs', 'val_logs')):
        os.makedirs(os.path.join(model_dir, sub), exist_ok=True)

    # the language comes from the training set and is saved beside the model
    train_data = make_dataset(train_path, params, None, device)
    language = train_data.smiles_language
    language_path = os.path.join(args.out_dir, f'{args.training_name}.lang')
    language.save(language_path)
    language = SMILESLanguage.load(language_path)   # the round trip is the point
    train_data = make_dataset(train_path, params, language, device)
    test_data = make_dataset(test_path, params, language, device)

    vocab = language.index_to_token
    params.update({
        'vocab_size': language.number_of_tokens,
        'pad_index': language.padding_index,
        'start_index': next(i for i, t in vocab.items() if t == '<START>'),
        'end_index': next(i for i, t in vocab.items() if t == '<STOP>'),
    })
    if params.get('embedding', 'learned') == 'one_hot':
        params['embedding_size'] = params['vocab_size']
    with open(os.path.join(model_dir, 'model_params.json'), 'w') as fp:
        json.dump(params, fp)

    train_loader = make_loader(train_data, params)
    test_loader = make_loader(test_data, params)
    gru_vae = TeacherVAE(
        StackGRUEncoder(params).to(device), StackGRUDecoder(params).to(device)
    ).to(device)
    logger.info('Total # params: %d',
                sum(p.numel() for p in gru_vae.parameters()))
    search = SEARCH_FACTORY[params.get('decoder_search', 'sampling')](
        temperature=params.get('temperature', 1.),
        beam_width=params.get('beam_width', 3),
        top_tokens=params.get('top_tokens', 5)
    )
    loss_tracker = {
        'test_loss_a': 10e4, 'test_rec_a': 10e4, 'test_kld_a': 10e4,
        'ep_loss': 0, 'ep_rec': 0, 'ep_kld': 0
    }
    tokens = sum(len(train_data[i]) for i in range(
        0, len(train_data), max(1, len(train_data) // 200)))
    mean_len = tokens / len(range(
        0, len(train_data), max(1, len(train_data) // 200)))
    for epoch in range(params['epochs'] + 1):
        t0 = time()
        loss_tracker = train_vae(
            epoch, gru_vae, train_loader, test_loader, language, model_dir,
            search=search, optimizer=params.get('optimizer', 'adadelta'),
            lr=params['learning_rate'], kl_growth=params['kl_growth'],
            input_keep=params['input_keep'],
            test_input_keep=params['test_input_keep'],
            generate_len=params['generate_len'],
            log_interval=params['log_interval'],
            save_interval=params['save_interval'],
            eval_interval=params['eval_interval'],
            loss_tracker=loss_tracker, logger=logger,
            batch_mode=params.get('batch_mode')
        )
        took = time() - t0
        steps = len(train_loader)
        logger.info(
            'Epoch %d, took %.1f s: %d steps, ~%.0f tokens/s (logging, '
            'evaluation and checkpoints included)', epoch, took, steps,
            steps * params['batch_size'] * mean_len / took)
    gru_vae.save(os.path.join(model_dir, 'weights', 'last.pt'))
    logger.info('Training done: %s', json.dumps(loss_tracker))
    return loss_tracker


if __name__ == '__main__':
    main()
