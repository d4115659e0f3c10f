"""`pytoda.datasets.SMILESDataset` stand-in: a `.smi` file (one molecule per
line, ``SMILES[<tab or space>name]``) as a map-style dataset of 1-D int64
token-id tensors, built with the keyword arguments the reference's scripts
pass (examples/train_vae.py:96-111, examples/generate_latent_codes.py:57-72).

Samples are CPU tensors whatever ``device`` says: the reference's loaders use
worker processes and pinned memory, and its batch preparation moves every batch
to the device itself (models/training.py:328-348).  The chemistry transforms
(augment / canonical / kekulize / explicit bonds and hydrogens / bond direction
/ chirality / SELFIES) need rdkit or selfies, which this image does not have:
asking for one raises, it is never silently skipped."""
import torch

from ..smiles.smiles_language import SMILESLanguage


class _Lines:
    """What the reference reaches through ``dataset._dataset`` (only
    ``.selfies``, models/training.py:127)."""

    def __init__(self, lines, selfies):
        self.lines = lines
        self.selfies = selfies

    def __len__(self):
        return len(self.lines)


class SMILESDataset(torch.utils.data.Dataset):

    def __init__(
        self, *smi_filepaths, smiles_language=None, padding=True,
        padding_length=None, add_start_and_stop=False, augment=False,
        canonical=False, kekulize=False, all_bonds_explicit=False,
        all_hs_explicit=False, remove_bonddir=False, remove_chirality=False,
        selfies=False, sanitize=True, randomize=False, device=None,
        backend='eager', name='smiles-dataset'
    ):
        wanted = {
            'augment': augment, 'canonical': canonical, 'kekulize': kekulize,
            'all_bonds_explicit': all_bonds_explicit,
            'all_hs_explicit': all_hs_explicit,
            'remove_bonddir': remove_bonddir,
            'remove_chirality': remove_chirality, 'selfies': selfies,
            'randomize': randomize,
        }
        unavailable = [key for key, on in wanted.items() if on]
        if unavailable:
            raise NotImplementedError(
                'pytoda stand-in: ' + ', '.join(unavailable) +
                ' need rdkit / selfies, which are not installed'
            )
        lines = []
        for path in smi_filepaths:
            with open(path) as fp:
                for line in fp:
                    fields = line.split()
                    if fields:
                        lines.append(fields[0])
        self._dataset = _Lines(lines, selfies)
        self.device = device
        self.name = name
        self.padding = padding
        if smiles_language is None:
            smiles_language = SMILESLanguage(
                add_start_and_stop=add_start_and_stop
            )
            smiles_language.add_smis(lines)
        else:
            smiles_language.add_start_and_stop = add_start_and_stop
        self.smiles_language = smiles_language
        self.padding_length = padding_length
        if padding and padding_length is None:
            extra = 2 if add_start_and_stop else 0
            self.padding_length = max(
                [len(smiles_language.tokenize(s)) + extra for s in lines] or [0]
            )

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, index):
        ids = self.smiles_language.smiles_to_token_indexes(
            self._dataset.lines[index]
        )
        if self.padding:   # left padding, as pytoda pads
            pad = self.smiles_language.padding_index
            ids = [pad] * max(0, self.padding_length - len(ids)) + ids
        return torch.tensor(ids, dtype=torch.int64)
