from paccmann_chemistry_amd.stack_rnn import StackGRU  # noqa: F401
from paccmann_chemistry_amd.vae import (  # noqa: F401
    StackGRUDecoder, StackGRUEncoder, TeacherVAE
)
