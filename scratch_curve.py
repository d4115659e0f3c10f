import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from oracle import weights as W
import test_loss_curve as T
for name in sys.argv[1:]:
    c = T._gpu_curve(W.LOSS_CURVES[name])
    np.save(f'gpurun_out/curve_{name}.npy', c)
    print(name, c[-1])
