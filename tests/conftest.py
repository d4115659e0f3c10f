import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950)')


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) on a box without a GPU."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


class Golden:
    """One tests/golden/*.npz fixture split into params / weights / grads /
    data (see oracle/make_golden.py for how each was produced)."""

    def __init__(self, name):
        raw = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        self.name = name
        self.params, self.weights, self.grads = {}, {}, {}
        self.grad_norms, self.grad_samples, self.data = {}, {}, {}
        for key in raw.files:
            val = raw[key]
            if key.startswith('p::'):
                self.params[key[3:]] = val.item()
            elif key.startswith('w::'):
                self.weights[key[3:]] = torch.from_numpy(val)
            elif key.startswith('g::'):
                self.grads[key[3:]] = torch.from_numpy(val)
            elif key.startswith('gn::'):
                self.grad_norms[key[4:]] = float(val)
            elif key.startswith('gs::'):
                self.grad_samples[key[4:]] = torch.from_numpy(val)
            else:
                self.data[key] = val
        if not self.weights and 'seed' in self.data and self.params:
            from oracle import weights as W
            full = W.make_state_dict(self.params, int(self.data['seed']))
            self.weights = full

    def tensor(self, key):
        return torch.from_numpy(self.data[key])

    @property
    def lengths(self):
        if 'lengths' in self.data:
            return torch.from_numpy(self.data['lengths'])
        return None


def golden_names(prefix):
    return sorted(
        f[:-4] for f in os.listdir(GOLDEN)
        if f.startswith(prefix) and f.endswith('.npz')
    )


@pytest.fixture
def load_golden():
    return Golden
