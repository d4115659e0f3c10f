"""No-op `torch.utils.tensorboard.SummaryWriter` for images without
tensorboard (the reference's examples/train_vae.py:19,76 builds a writer and
only calls ``add_hparams`` on it: the ``writer=`` argument of ``train_vae`` is
commented out, examples/train_vae.py:261)."""
import importlib
import sys
import types


class SummaryWriter:

    def __init__(self, log_dir=None, *args, **kwargs):
        self.log_dir = log_dir

    def __getattr__(self, name):
        if name.startswith('add_') or name in ('flush', 'close'):
            return lambda *args, **kwargs: None
        raise AttributeError(name)


def install():
    """Register the stub unless the real module imports."""
    try:
        importlib.import_module('torch.utils.tensorboard')
        return False
    except ImportError:
        pass
    import torch.utils
    module = types.ModuleType('torch.utils.tensorboard')
    module.SummaryWriter = SummaryWriter
    module.__all__ = ['SummaryWriter']
    sys.modules['torch.utils.tensorboard'] = module
    torch.utils.tensorboard = module
    return True
