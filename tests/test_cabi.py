"""The C-ABI library loads and exports every symbol include/pnnp_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(REPO, 'include', 'pnnp_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(pnnp_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_exported():
    so = os.path.join(REPO, 'pnnp_amd', 'libpnnp_hip.so')
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    names = _declared()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.pnnp_error_string.restype = ctypes.c_char_p
    assert lib.pnnp_version() >= 100
    assert lib.pnnp_error_string(0) == b'ok'
    assert lib.pnnp_error_string(-2) == b'unsupported configuration'


def test_product_has_no_cpu_fallback():
    import numpy as np
    import torch
    from pnnp_amd import _lib, process
    with pytest.raises(_lib.PnnpError):
        process.generate_noisy_torch(torch.zeros(4, 8, 8), noise_code='p',
                                     param=dict(K=1., sigGs=1., sigR=1., q=1., ratio=1., wp=1023, bl=64, bias=0))
    # nothing under pnnp_amd may import the oracle
    import pathlib
    for f in pathlib.Path(REPO, 'pnnp_amd').rglob('*.py'):
        src = f.read_text()
        assert 'import oracle' not in src and 'from oracle' not in src, f
