from paccmann_chemistry_amd.hyperparams import (  # noqa: F401
    OPTIMIZER_FACTORY, SEARCH_FACTORY
)
