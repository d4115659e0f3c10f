"""Factories: same keys as the reference's ``utils/hyperparams.py``."""
import torch.optim as optim

from .search import BeamSearch, GreedySearch, SamplingSearch

SEARCH_FACTORY = {
    'sampling': SamplingSearch,
    'greedy': GreedySearch,
    'beam': BeamSearch,
}

OPTIMIZER_FACTORY = {
    'adadelta': optim.Adadelta,
    'adagrad': optim.Adagrad,
    'adam': optim.Adam,
    'adamax': optim.Adamax,
    'rmsprop': optim.RMSprop,
    'sgd': optim.SGD
}
