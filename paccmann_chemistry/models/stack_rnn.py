from paccmann_chemistry_amd.stack_rnn import StackGRU  # noqa: F401
