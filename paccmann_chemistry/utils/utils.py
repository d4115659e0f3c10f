from paccmann_chemistry_amd.utils import *  # noqa: F401,F403
