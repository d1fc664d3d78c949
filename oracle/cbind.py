"""ctypes binding of oracle/_build/libpnnp_oracle.so (ORACLE, test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, '_build', 'libpnnp_oracle.so')
NPARAM = 16
FLAG = dict(p=1, g=2, r=4, q=8, d=16, b=32)
F_ORI, F_CLIP, F_TORCH = 0x100, 0x200, 0x1000
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, 'pnnp_oracle.c')):
            build()
        _lib = C.CDLL(SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def pack(raw, black4, wp, norm=True, clip=False):
    raw = np.ascontiguousarray(raw)
    is_f32 = raw.dtype == np.float32
    if not is_f32:
        raw = raw.astype(np.uint16, copy=False)
    H, W = raw.shape
    out = np.empty((4, H // 2, W // 2), np.float32)
    b = np.ascontiguousarray(black4, np.float64)
    lib().pnnp_oracle_pack(_p(raw), C.c_int(is_f32), H, W, _p(out), _p(b), C.c_double(wp), int(norm), int(clip))
    return out


def unpack(packed, wp, bl):
    p = np.ascontiguousarray(packed, np.float32)
    _, h, w = p.shape
    out = np.empty((2 * h, 2 * w), np.uint16)
    lib().pnnp_oracle_unpack(_p(p), h, w, _p(out), int(wp), int(bl))
    return out


def param_rows(plist):
    """list of param dicts -> float32 [B][NPARAM] in the order of include/pnnp_hip.h."""
    rows = np.zeros((len(plist), NPARAM), np.float32)
    for i, p in enumerate(plist):
        bias = np.broadcast_to(np.asarray(p.get('bias', 0), np.float64).reshape(-1), (4,)) if np.ndim(p.get('bias', 0)) == 0 \
            else np.asarray(p['bias'], np.float64).reshape(-1)[:4]
        rows[i, :9] = [p['K'], p['sigGs'], p.get('sigTL', 0), p.get('lam', 0), p['sigR'], p['q'], p['ratio'], p['wp'], p['bl']]
        rows[i, 9:13] = bias
    return rows


def noise_flags(code, ori=False, clip=False, torch_mode=False):
    f = 0
    for ch in code.lower():
        f |= FLAG.get(ch, 0)
    return f | (F_ORI if ori else 0) | (F_CLIP if clip else 0) | (F_TORCH if torch_mode else 0)


def noise_sample(y, params, flags, mfm=1.0, seed=0, offset=0, crop_base=0):
    y = np.ascontiguousarray(y, np.float32)
    B, Cc, H, W = y.shape
    out = np.empty_like(y)
    pr = np.ascontiguousarray(params, np.float32)
    assert pr.shape == (B, NPARAM)
    lib().pnnp_oracle_noise_sample(_p(y), _p(out), B, Cc, H, W, _p(pr), C.c_uint(flags), C.c_float(mfm),
                                   C.c_uint64(seed), C.c_uint64(offset), C.c_uint32(crop_base))
    return out


def sna(gt, aug, K, wp, bl, ratio, black_lr, ori, seed=0, offset=0, crop=0):
    gt = np.ascontiguousarray(gt, np.float32)
    Cc, H, W = gt.shape
    dn = np.empty_like(gt); dy = np.empty_like(gt)
    a = np.ascontiguousarray(aug, np.float32)
    lib().pnnp_oracle_sna(_p(gt), _p(dn), _p(dy), Cc, H, W, _p(a), C.c_float(K), C.c_float(wp), C.c_float(bl), C.c_float(ratio),
                          int(black_lr), int(ori), C.c_uint64(seed), C.c_uint64(offset), C.c_uint32(crop))
    return dn, dy


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32); k = np.asarray(key, np.uint32); o = np.zeros(4, np.uint32)
    lib().pnnp_oracle_philox(_p(c), _p(k), _p(o))
    return o
