from paccmann_chemistry_amd.training import (  # noqa: F401
    get_data_preparation, test_vae, train_vae
)
