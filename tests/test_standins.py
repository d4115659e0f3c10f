"""The build-owned stand-ins (standins/: pytoda's dataset / language surface,
a no-op tensorboard writer) and the example script that follows the reference's
examples/train_vae.py call sequence on a synthetic SMILES set (BASELINE.json
config 1)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

STANDINS = os.path.join(ROOT, 'standins')
REFERENCE = '/root/reference'


@pytest.fixture
def standins_on_path(monkeypatch):
    monkeypatch.syspath_prepend(STANDINS)
    for name in [m for m in sys.modules if m.split('.')[0] == 'pytoda']:
        monkeypatch.delitem(sys.modules, name)
    yield
    for name in [m for m in sys.modules if m.split('.')[0] == 'pytoda']:
        sys.modules.pop(name, None)


def _write_smi(path):
    lines = ['CC(=O)Oc1ccccc1C(=O)O\taspirin', 'C[C@H](N)C(=O)O ala',
             'ClCCBr', 'c1cc[nH]c1', '[O-][N+](=O)c1ccc(Cl)cc1 x', '']
    with open(path, 'w') as fp:
        fp.write('\n'.join(lines))
    return [l.split()[0] for l in lines if l]


def test_language_and_dataset_surface(tmp_path, standins_on_path):
    """Exactly what the reference's scripts touch: keyword
    arguments of SMILESDataset, the language's index attributes, the
    <START>/<STOP> lookup of examples/train_vae.py:147-157, save / load, text
    round trip, ``dataset._dataset.selfies`` (models/training.py:127)."""
    from pytoda.datasets import SMILESDataset
    from pytoda.smiles.smiles_language import SMILESLanguage
    import pytoda
    assert pytoda.smiles.SMILESLanguage is SMILESLanguage
    assert hasattr(pytoda.proteins, 'ProteinLanguage')
    assert pytoda.transforms.Compose([lambda x: x + 1, lambda x: x * 2])(1) == 4
    smiles = _write_smi(tmp_path / 'a.smi')
    data = SMILESDataset(
        str(tmp_path / 'a.smi'), smiles_language=None, padding=False,
        selfies=False, add_start_and_stop=True, augment=False, canonical=False,
        kekulize=False, all_bonds_explicit=False, all_hs_explicit=False,
        remove_bonddir=False, remove_chirality=False, backend='lazy',
        device=torch.device('cpu'))
    lang = data.smiles_language
    assert len(data) == len(smiles) and data._dataset.selfies is False
    assert (lang.padding_index, lang.unknown_index, lang.start_index,
            lang.stop_index) == (0, 1, 2, 3)
    vocab = lang.index_to_token
    assert list(vocab.keys())[list(vocab.values()).index('<START>')] == 2
    assert list(vocab.keys())[list(vocab.values()).index('<STOP>')] == 3
    assert lang.number_of_tokens == len(vocab) > 4
    for index, text in enumerate(smiles):
        ids = data[index]
        assert ids.dtype == torch.int64 and ids.dim() == 1
        assert ids[0] == 2 and ids[-1] == 3 and (ids[1:-1] > 3).all()
        assert lang.token_indexes_to_smiles(ids.tolist()) == text
    # multi-character tokens are single ids
    assert len(data[2]) == 2 + 4                       # Cl C C Br
    assert '[C@H]' in lang.token_to_index and '[nH]' in lang.token_to_index
    lang.save(str(tmp_path / 'l.lang'))
    again = SMILESLanguage.load(str(tmp_path / 'l.lang'))
    assert again.token_to_index == lang.token_to_index
    assert again.number_of_tokens == lang.number_of_tokens
    # a language built elsewhere maps unseen tokens to <UNK>
    other = SMILESDataset(str(tmp_path / 'a.smi'), smiles_language=again,
                          padding=False, add_start_and_stop=True)
    assert torch.equal(other[0], data[0])
    assert again.smiles_to_token_indexes('C[Se]C')[2] == again.unknown_index
    # transforms that need rdkit / selfies are refused, not skipped
    with pytest.raises(NotImplementedError, match='canonical'):
        SMILESDataset(str(tmp_path / 'a.smi'), canonical=True)
    # pickles (pytoda's own format) are never loaded
    with open(tmp_path / 'p.pkl', 'wb') as fp:
        fp.write(b'\x80\x04N.')
    with pytest.raises(ValueError, match='not a JSON'):
        SMILESLanguage.load(str(tmp_path / 'p.pkl'))


def test_dataset_feeds_the_reference_batch_preparation(tmp_path,
                                                       standins_on_path):
    """DataLoader(collate_fn) -> get_data_preparation, as
    examples/train_vae.py:166-174 and models/training.py:138-144 chain them."""
    from pytoda.datasets import SMILESDataset
    from paccmann_chemistry.utils import collate_fn
    from paccmann_chemistry.models.training import get_data_preparation
    _write_smi(tmp_path / 'a.smi')
    data = SMILESDataset(str(tmp_path / 'a.smi'), padding=False,
                         add_start_and_stop=True, backend='lazy')
    loader = torch.utils.data.DataLoader(data, batch_size=4, drop_last=True,
                                         collate_fn=collate_fn)
    batch = next(iter(loader))
    lens = [len(x) for x in batch]
    assert lens == sorted(lens, reverse=True)
    for mode in ('padded', 'packed'):
        enc, dec, tgt = get_data_preparation(mode)(
            batch, input_keep=1.0, start_index=2, end_index=3,
            device=torch.device('cpu'))
        if mode == 'padded':
            assert enc.shape[1] == 4 and torch.equal(enc, dec)
        else:
            assert int(enc.batch_sizes[0]) == 4


def test_tensorboard_stub_only_when_missing():
    code = (
        'import sys; sys.path.insert(0, %r)\n'
        'import tensorboard_stub as t\n'
        'installed = t.install()\n'
        'from torch.utils.tensorboard import SummaryWriter\n'
        'w = SummaryWriter("logs/x"); w.add_hparams(hparam_dict={}, '
        'metric_dict={}); w.add_scalar("a", 1.0, 0); w.close()\n'
        'print("installed", installed)\n' % STANDINS)
    done = subprocess.run([sys.executable, '-c', code], capture_output=True,
                          text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-2000:]
    assert 'installed' in done.stdout


@pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, 'examples')),
                    reason='the reference checkout exists in the build '
                           'container only')
@pytest.mark.parametrize('script', ['train_vae.py', 'generate_latent_codes.py'])
def test_reference_scripts_import_against_this_build(script):
    """The reference CI's smoke test (.github/workflows/build.yml:59-60:
    ``python3 examples/train_vae.py -h``) with THIS repository and the
    stand-ins first on the path: every import of the script resolves
    (paccmann_chemistry.* to the MI355X implementation) and argparse prints
    its usage.  Nothing of the reference is copied: the script is run where it
    lies."""
    done = subprocess.run(
        [sys.executable, os.path.join(STANDINS, 'run_script.py'),
         os.path.join(REFERENCE, 'examples', script), '-h'],
        capture_output=True, text=True, timeout=300, cwd=ROOT,
        env={k: v for k, v in os.environ.items() if k != 'PYTHONPATH'})
    assert done.returncode == 0, done.stderr[-3000:]
    if script == 'train_vae.py':
        assert 'usage:' in done.stdout
    # (generate_latent_codes.py has no __main__ guard: run as a script it only
    # executes its imports and definitions, which is what is checked)
    probe = (
        'import sys; sys.path[:0] = [%r, %r]\n'
        'import paccmann_chemistry, paccmann_chemistry.models.vae as v\n'
        'print(paccmann_chemistry.__file__); print(v.TeacherVAE.__module__)\n'
        % (STANDINS, ROOT))
    where = subprocess.run([sys.executable, '-c', probe], capture_output=True,
                           text=True, timeout=300).stdout
    assert where.splitlines()[0].startswith(ROOT)
    assert 'paccmann_chemistry_amd' in where.splitlines()[1]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_example_script_trains_two_epochs_on_synthetic_smiles(tmp_path):
    """examples/train_vae_synthetic.py end to end on the GPU: 10 k synthetic
    lines, the example_params model (H=512, L=2, 50x50 stack, latent 256,
    packed mode, one-hot over the set's own vocabulary, batch 256), epochs = 1
    (the reference's loop runs range(epochs + 1) = 2 epochs), with logging,
    generation, evaluation and checkpoints at the configured intervals."""
    done = subprocess.run(
        [sys.executable, os.path.join(ROOT, 'examples',
                                      'train_vae_synthetic.py'),
         str(tmp_path), '--lines', '10000', '--test-lines', '512',
         '--epochs', '1'],
        capture_output=True, text=True, timeout=580, cwd=ROOT)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    out = done.stdout
    assert 'Epoch 0, took' in out and 'Epoch 1, took' in out
    assert '***TESTING***' in out and '***SAVING***' in out
    assert 'Training done' in out
    model_dir = tmp_path / 'synthetic'
    with open(model_dir / 'model_params.json') as fp:
        params = json.load(fp)
    assert params['vocab_size'] == params['embedding_size'] > 20
    assert params['start_index'] == 2 and params['end_index'] == 3
    assert (model_dir / 'weights' / 'best_loss.pt').exists()
    assert (model_dir / 'weights' / 'last.pt').exists()
    with open(model_dir / 'loss_tracker.json') as fp:
        tracker = json.load(fp)
    assert tracker['test_loss_a'] < 10e4
    # the loss went down over the run: first and last logged training loss
    import re
    losses = [float(x) for x in re.findall(r'Loss: ([0-9.]+),', out)]
    assert len(losses) >= 4 and losses[-1] < losses[0]
    # checkpoint interchange: the saved state_dict has the reference's 34 keys
    sd = torch.load(model_dir / 'weights' / 'last.pt', weights_only=True,
                    map_location='cpu')
    assert len(sd) == 34
