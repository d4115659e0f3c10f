"""Import-path compatibility for code written against the reference package:
``paccmann_chemistry.models.*`` and ``paccmann_chemistry.utils.*`` resolve to the
MI355X implementation in ``paccmann_chemistry_amd``."""
name = 'paccmann_chemistry'
__version__ = '0.0.4+amd'
