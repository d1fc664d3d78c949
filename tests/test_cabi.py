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


def test_abi_version_and_shape_policies_without_a_gpu():
    """Round 6: the ABI version / PnnpPackJob size the Python bindings check, and the pure shape policies of the fp16x2 family (which tile width a layer runs on,
    how many K slices a small grid is cut into, which pointwise shapes have a tile) -- host-side functions that need no device (without one the library assumes
    the MI355X's 256 compute units)."""
    so = os.path.join(REPO, 'pnnp_amd', 'libpnnp_hip.so')
    lib = ctypes.CDLL(so)
    from pnnp_amd import ops
    assert lib.pnnp_abi_version() == ops.ABI_VERSION
    assert lib.pnnp_pack_job_bytes() == ctypes.sizeof(ops.PackJob)
    if lib.pnnp_device_cus() != 256:
        pytest.skip('policies below are stated for 256 compute units')
    tc = lambda B, H, W, N, pool=0: lib.pnnp_h2_tile_columns(B, H, W, N, pool)
    assert tc(16, 512, 512, 32) == 32 and tc(16, 256, 256, 64) == 64 and tc(16, 32, 32, 512) == 64      # config 3's layers
    assert tc(1, 32, 32, 512) == 32 and tc(1, 64, 64, 256) == 32                                          # one crop: 64-column tiles would leave CUs idle
    assert tc(1, 32, 32, 512, 1) == 64                                                                     # the pooled forward keeps 64 whenever the layer has them
    sk = lambda B, H, W, chunks, N: lib.pnnp_h2_splitk(B, H, W, chunks, N)
    assert sk(16, 32, 32, 32, 512) == 1                                                                    # a full batch fills the chip: no split
    assert sk(1, 32, 32, 32, 512) == 8 and sk(1, 64, 64, 16, 256) == 4 and sk(1, 128, 128, 8, 128) == 2    # one crop: 32 / 64 / 128 tiles -> 256 workgroups
    assert sk(1, 256, 256, 4, 64) == 1 and sk(1, 32, 32, 2, 512) == 1                                      # enough tiles / too few chunks
    for chunks in (4, 6, 8, 12, 16, 32):
        s = sk(1, 32, 32, chunks, 512)
        assert chunks % s == 0 and chunks // s >= 2                                                        # whole slices of at least two chunks
    assert lib.pnnp_gemm_h2_supported(64, 32) and lib.pnnp_gemm_h2_supported(32, 128) and not lib.pnnp_gemm_h2_supported(48, 64) and not lib.pnnp_gemm_h2_supported(64, 48)
    X3G_PW = ops.X3G_PW
    assert lib.pnnp_h2g_wgrad_supported(X3G_PW, 32, 64) and not lib.pnnp_x3g_wgrad_supported(X3G_PW, 32, 64)     # sc9's 32 x 64 tile exists on fp16x2 only


def test_build_flags_have_one_source():
    """Round 6: tools/build_variant.sh (the A/B builds) added -fno-slp-vectorize to csrc/wgrad_h2s.hip while tools/build.py (the product) did not, and the
    adopted rolling refill was 2.1 % SLOWER in the product than the source it replaced (profiles/r6/ab_wgrad_rolling.txt, part 4).  The per-file flags live in
    tools/build.py's EXTRA table only; the file that needs the flag has it; a stamp beside every object names the flags it was compiled with."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('pnnp_build', os.path.join(REPO, 'tools', 'build.py'))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    assert '-fno-slp-vectorize' in b.EXTRA['wgrad_h2s.hip'] and '-fno-slp-vectorize' in b.EXTRA['wgrad_h2g.hip'] and '-fno-slp-vectorize' in b.EXTRA['conv_h2s.hip']
    variant = open(os.path.join(REPO, 'tools', 'build_variant.sh')).read()
    assert 'build.EXTRA.get' in variant and 'wgrad_h2s.hip|' not in variant
    cmd = [b.HIPCC] + b.COMMON + ['-c', os.path.join(b.SRC, 'misc.hip'), '-o', os.path.join(b.OBJ, 'misc.o')]
    assert REPO not in b.flags_of(cmd)                               # (the GPU box sees the repository under another path and must not rebuild)
