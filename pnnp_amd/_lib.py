"""ctypes loader for libpnnp_hip.so -- the only way the product reaches a kernel.

There is NO CPU fallback: if the library is missing or a tensor is not on a HIP
device, the call raises.  (The CPU oracle under ``oracle/`` is test infrastructure
and is never imported from here.)
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PNNP_LIB: another build of the SAME library (A/B experiments: tools/build_variant.sh); never a different implementation
LIB_PATH = os.environ.get('PNNP_LIB') or os.path.join(_HERE, 'libpnnp_hip.so')
_lib = None

ERRORS = {-1: 'invalid argument', -2: 'unsupported configuration', -3: 'kernel launch failed', -4: 'workspace too small'}


class PnnpError(RuntimeError):
    pass


def lib():
    """Load libpnnp_hip.so (built in-tree by ``__graft_entry__.build()``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PnnpError(f'{LIB_PATH} not found: run `python -c "import __graft_entry__ as g; g.build()"` '
                            f'(hipcc --offload-arch=gfx950). pnnp_amd has no CPU fallback.')
        _lib = C.CDLL(LIB_PATH)
        _lib.pnnp_error_string.restype = C.c_char_p
    return _lib


def check(code, what=''):
    if code != 0:
        try:
            msg = lib().pnnp_error_string(code).decode()
        except Exception:
            msg = ERRORS.get(code, 'unknown')
        raise PnnpError(f'libpnnp_hip: {what} failed with code {code} ({msg})')


def require_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PnnpError('pnnp_amd runs on the GPU only: got a CPU tensor (no CPU fallback; '
                            'use oracle/ for a CPU reference in tests)')


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
