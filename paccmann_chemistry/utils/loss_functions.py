from paccmann_chemistry_amd.loss_functions import (  # noqa: F401
    vae_loss_function
)
