"""Minimal stand-in for the `pytoda` surface the reference's example scripts
and training loop touch (see standins/README.md).  NOT pytoda."""
from . import datasets, proteins, smiles, transforms  # noqa: F401

__version__ = '0.1.1+standin'
