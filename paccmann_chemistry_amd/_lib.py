"""ctypes binding of libpaccmann_hip.so (the C ABI in include/paccmann_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (PCM_LIBRARY: a diagnostics build of the same library, e.g. tools/bin/*.so)
LIB_PATH = os.environ.get(
    'PCM_LIBRARY', os.path.join(_HERE, 'libpaccmann_hip.so')
)
PCM_MAX_LAYERS = 8
ABI_VERSION = 4

SEARCH_GREEDY = 0
SEARCH_SAMPLING = 1
STATUS_BAD_TOKEN = 1
STATUS_TIMEOUT = 2

c_float_p = ctypes.c_void_p  # device pointers travel as plain addresses


class PcmDims(ctypes.Structure):
    _fields_ = [
        ('batch', ctypes.c_int32), ('vocab', ctypes.c_int32),
        ('emb', ctypes.c_int32), ('hidden', ctypes.c_int32),
        ('layers', ctypes.c_int32), ('stack_depth', ctypes.c_int32),
        ('stack_width', ctypes.c_int32), ('use_stack', ctypes.c_int32)
    ]


class PcmStackGRUParams(ctypes.Structure):
    _fields_ = [
        ('ctrl_w', ctypes.c_void_p), ('ctrl_b', ctypes.c_void_p),
        ('push_w', ctypes.c_void_p), ('push_b', ctypes.c_void_p),
        ('emb', ctypes.c_void_p),
        ('w_ih', ctypes.c_void_p * PCM_MAX_LAYERS),
        ('w_hh', ctypes.c_void_p * PCM_MAX_LAYERS),
        ('b_ih', ctypes.c_void_p * PCM_MAX_LAYERS),
        ('b_hh', ctypes.c_void_p * PCM_MAX_LAYERS)
    ]


_I32, _I64, _U64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64
_F32, _SZ, _VP = ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p
_DIMS_P = ctypes.POINTER(PcmDims)
_PARAMS_P = ctypes.POINTER(PcmStackGRUParams)

# name -> (restype, argtypes); mirrors include/paccmann_hip.h one to one
PROTOTYPES = {
    'pcm_version': (ctypes.c_int, []),
    'pcm_last_error': (ctypes.c_char_p, []),
    'pcm_device_check': (ctypes.c_int, []),
    'pcm_probe_enable': (ctypes.c_int, [_I32, _I32]),
    'pcm_set_matmul_precision': (ctypes.c_int, [_I32]),
    'pcm_get_matmul_precision': (ctypes.c_int, []),
    'pcm_probe_calibrate': (
        ctypes.c_int, [
            _I32, ctypes.POINTER(ctypes.c_double),
            ctypes.POINTER(ctypes.c_double), _VP
        ]
    ),
    'pcm_probe_read': (
        ctypes.c_int, [
            ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64),
            ctypes.POINTER(_I64)
        ]
    ),
    'pcm_stackgru_saved_bytes': (_SZ, [_DIMS_P, _I32, _I32]),
    'pcm_stackgru_out_view': (
        ctypes.c_int,
        [_DIMS_P, _I32, _I32, ctypes.POINTER(_SZ), ctypes.POINTER(_SZ)]
    ),
    'pcm_stackgru_scratch_bytes': (_SZ, [_DIMS_P, _I32]),
    'pcm_stackgru_forward': (
        ctypes.c_int, [
            _DIMS_P, _PARAMS_P, _VP, _I32, _VP, _VP, _VP, _I64, _VP, _F32,
            _U64, _I32, _VP, _SZ, _VP, _VP, _VP, _VP
        ]
    ),
    'pcm_stackgru_backward': (
        ctypes.c_int, [
            _DIMS_P, _PARAMS_P, _VP, _I32, _VP, _VP, _F32, _U64, _VP, _SZ,
            _VP, _I64, _VP, _VP, _VP, _VP, _PARAMS_P, _I32, _VP, _SZ, _VP
        ]
    ),
    'pcm_linear_forward': (
        ctypes.c_int, [_VP, _I64, _VP, _VP, _VP, _I64, _I32, _I32, _I32, _VP]
    ),
    'pcm_linear_scratch_bytes': (_SZ, [_I32, _I32, _I32]),
    'pcm_linear_backward': (
        ctypes.c_int, [
            _VP, _I64, _VP, _VP, _VP, _I64, _VP, _I64, _VP, _VP, _I32, _I32,
            _I32, _I32, _VP, _SZ, _VP
        ]
    ),
    'pcm_softmax_ce': (
        ctypes.c_int,
        [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP]
    ),
    'pcm_output_ce': (
        ctypes.c_int,
        [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _VP, _VP,
         _VP, _VP, _VP, _VP, _VP]
    ),
    'pcm_softmax_ce_static': (
        ctypes.c_int,
        [_VP, _VP, _VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP]
    ),
    'pcm_reparam_forward': (ctypes.c_int, [_VP, _VP, _VP, _VP, _I64, _VP]),
    'pcm_reparam_backward': (
        ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _VP]
    ),
    'pcm_kl_scratch_floats': (_SZ, [_I64]),
    'pcm_kl_forward': (ctypes.c_int, [_VP, _VP, _VP, _VP, _I64, _VP]),
    'pcm_kl_backward': (ctypes.c_int, [_VP, _VP, _VP, _VP, _VP, _I64, _VP]),
    'pcm_generate_workspace_bytes': (_SZ, [_DIMS_P]),
    'pcm_stackgru_generate': (
        ctypes.c_int, [
            _DIMS_P, _PARAMS_P, _VP, _VP, _VP, _I64, _VP, _I32, _I32, _I32,
            _F32, _U64, _VP, _VP, _VP, _SZ, _VP, _VP
        ]
    ),
    'pcm_search_step': (
        ctypes.c_int, [_VP, _I64, _I32, _I32, _F32, _U64, _I32, _VP, _VP]
    ),
    'pcm_generate_beam_workspace_bytes': (_SZ, [_DIMS_P, _I32, _I32]),
    'pcm_stackgru_generate_beam': (
        ctypes.c_int, [
            _DIMS_P, _PARAMS_P, _VP, _VP, _VP, _I64, _VP, _I32, _I32, _I32,
            _I32, _F32, _VP, _VP, _VP, _VP, _SZ, _VP, _VP
        ]
    ),
    'pcm_prepare_batch': (
        ctypes.c_int, [
            _VP, _VP, _I32, _I32, _I32, _VP, _F32, _U64, _VP, _I64, _VP, _VP,
            _VP, _VP
        ]
    ),
    'pcm_adam_step': (
        ctypes.c_int,
        [_VP, _VP, _VP, _VP, _I64, _F32, _F32, _F32, _F32, _I64, _F32, _VP]
    ),
}

_lib = None


class PaccmannHipError(RuntimeError):
    """A libpaccmann_hip call failed (or the library is not built)."""


def load():
    """Load the shared object once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PaccmannHipError(
            f'{LIB_PATH} is missing: build it with '
            '`make -C paccmann_chemistry_amd/csrc` (or '
            '`python -c "import __graft_entry__ as g; g.build()"`). '
            'There is no CPU fallback for this path.'
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pcm_version() != ABI_VERSION:
        raise PaccmannHipError(
            f'ABI mismatch: library {lib.pcm_version()}, binding '
            f'{ABI_VERSION}'
        )
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().pcm_last_error().decode('utf-8', 'replace')
        raise PaccmannHipError(f'{what} failed (code {rc}): {msg}')


def require_device(tensor, what):
    """The product path runs on the GPU only."""
    if not tensor.is_cuda:
        raise PaccmannHipError(
            f'{what}: tensor lives on {tensor.device}; the MI355X kernels '
            'need GPU tensors and there is no CPU fallback'
        )


def ptr(tensor):
    return None if tensor is None else tensor.data_ptr()


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def make_dims(batch, vocab, emb, hidden, layers, depth, width, use_stack):
    return PcmDims(
        int(batch), int(vocab), int(emb), int(hidden), int(layers),
        int(depth), int(width), 1 if use_stack else 0
    )


def make_params(tensors, n_layers):
    """tensors: list in state_dict order -- ctrl_w, ctrl_b, push_w, push_b,
    emb, then (w_ih, w_hh, b_ih, b_hh) per layer; entries may be None."""
    p = PcmStackGRUParams()
    p.ctrl_w, p.ctrl_b = ptr(tensors[0]), ptr(tensors[1])
    p.push_w, p.push_b = ptr(tensors[2]), ptr(tensors[3])
    p.emb = ptr(tensors[4])
    for layer in range(n_layers):
        base = 5 + 4 * layer
        p.w_ih[layer] = ptr(tensors[base])
        p.w_hh[layer] = ptr(tensors[base + 1])
        p.b_ih[layer] = ptr(tensors[base + 2])
        p.b_hh[layer] = ptr(tensors[base + 3])
    return p
