from paccmann_chemistry_amd.search import (  # noqa: F401
    BeamSearch, GreedySearch, SamplingSearch, Search
)
