"""MI355X-native stack-augmented GRU VAE hot path of paccmann_chemistry.

Host-side mirror of the reference's module interface over libpaccmann_hip.so
(C ABI: include/paccmann_hip.h).  Reference module -> module here:

    paccmann_chemistry.models.stack_rnn      -> paccmann_chemistry_amd.stack_rnn
    paccmann_chemistry.models.vae            -> paccmann_chemistry_amd.vae
    paccmann_chemistry.utils.loss_functions  -> paccmann_chemistry_amd.loss_functions
    paccmann_chemistry.utils.search          -> paccmann_chemistry_amd.search
    paccmann_chemistry.utils.hyperparams     -> paccmann_chemistry_amd.hyperparams
    paccmann_chemistry.utils (utils.py)      -> paccmann_chemistry_amd.utils
"""
from .stack_rnn import StackGRU  # noqa: F401
from .vae import StackGRUDecoder, StackGRUEncoder, TeacherVAE  # noqa: F401
from .loss_functions import vae_loss_function  # noqa: F401
from .search import BeamSearch, GreedySearch, SamplingSearch  # noqa: F401
from .ops import get_matmul_precision, set_matmul_precision  # noqa: F401
from .inference import encode_batches, generate_batches  # noqa: F401

__version__ = '0.1.0'
