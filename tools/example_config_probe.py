"""bench.py's training step at the reference's examples/example_params.json
shape (batch 256, vocabulary 380 one-hot, packed mode, dropout 0.2) instead of
BASELINE.json's configs[1]: which kernels the larger vocabulary sends down the
generic paths (output layer + CE in two launches, token products on their own).

    python tools/example_config_probe.py [bench.py flags]   # e.g. --ragged
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

bench.PARAMS.update(vocab_size=380, embedding_size=380)
if '--batch' not in sys.argv:
    sys.argv += ['--batch', '256']
sys.argv += ['--no-cpu-baseline', '--probe', '0']
bench.main()
