from paccmann_chemistry_amd.vae import *  # noqa: F401,F403
from paccmann_chemistry_amd.vae import (  # noqa: F401
    StackGRUDecoder, StackGRUEncoder, TeacherVAE
)
