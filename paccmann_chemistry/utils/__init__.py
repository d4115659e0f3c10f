from paccmann_chemistry_amd.utils import (  # noqa: F401
    collate_fn, crop_start_stop, cuda, disable_rdkit_logging, get_device,
    kl_weight, manage_step_packed_vars, packed_sequential_data_preparation,
    packed_to_padded, perpare_packed_input, print_example_reconstruction,
    repack_sequence, sequential_data_preparation, to_np, unpack_sequence
)
