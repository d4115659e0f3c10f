"""Diagnostics: the piecewise loss-curve run of tests/test_loss_curve.py with
both curves written to gpurun_out/ (segment length and optimizer from argv)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import test_loss_curve as T  # noqa: E402
from oracle import weights as W  # noqa: E402

flat = sys.argv[1] == 'flat'
segment = int(sys.argv[2])
hip, ref, early = T._piecewise_curves(W.LOSS_CURVES['loss_curve_small'],
                                      flat, segment=segment)
print('early re-seats', early)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.savez(os.path.join(ROOT, 'gpurun_out', f'piecewise_{sys.argv[1]}_{segment}.npz'),
         hip=hip, ref=ref)
err = np.maximum(T._rel(hip[:, 0], ref[:, 0]), T._rel(hip[:, 1], ref[:, 1]))
for a in range(0, len(err), segment):
    print(a, f'{err[a:a + segment].max():.2e}', flush=True)
