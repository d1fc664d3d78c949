"""Bayer pack / unpack with the reference's names and signatures, executed by the
HIP kernels in csrc/pack.hip (bit-exact with the reference).

Mirrors utils/isp_ops.py:57-112 of the reference.  Inputs may be numpy arrays
(like the reference's DataLoader workers pass) or CUDA tensors; numpy inputs are
staged to the GPU, processed there and returned as numpy, CUDA tensors stay on the
device.  There is no CPU implementation here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_NP2T = {np.dtype('uint16'): torch.uint16, np.dtype('float32'): torch.float32, np.dtype('float64'): torch.float64,
         np.dtype('int16'): torch.int16, np.dtype('int32'): torch.int32, np.dtype('uint8'): torch.uint8,
         np.dtype('int64'): torch.int64}


def _to_dev(a):
    """-> (cuda tensor, was_numpy)."""
    if torch.is_tensor(a):
        if a.is_cuda:
            return a.contiguous(), False
        return a.contiguous().cuda(), True
    a = np.ascontiguousarray(a)
    if a.dtype not in _NP2T:
        raise TypeError(f'unsupported dtype {a.dtype}')
    return torch.from_numpy(a).cuda(), True


def _ret(t, was_host):
    return t.cpu().numpy() if was_host else t


def raw2bayer(raw, wp=1023, bl=64, norm=True, clip=False, bias=np.array([0, 0, 0, 0])):
    """utils/isp_ops.py:84-96: [H,W] (or batched [B,H,W]) u16/f32 -> f32 [4,H/2,W/2]
    ([B,4,H/2,W/2]), planes R,G1,B,G2; (x-(bias+bl))/(wp-(bias+bl)) in float64."""
    x, host = _to_dev(raw)
    if x.dtype not in (torch.uint16, torch.float32):
        x = x.to(torch.float32)          # reference: raw.astype(np.float32)
    batched = x.dim() == 3
    if not batched:
        x = x[None]
    B, H, W = x.shape
    out = torch.empty((B, 4, H // 2, W // 2), dtype=torch.float32, device=x.device)
    black = (C.c_double * 4)(*[float(b) + float(bl) for b in np.broadcast_to(np.asarray(bias, np.float64).reshape(-1), (4,))])
    fn = _lib.lib().pnnp_pack_bayer_u16 if x.dtype == torch.uint16 else _lib.lib().pnnp_pack_bayer_f32
    _lib.check(fn(_lib.ptr(x), B, H, W, C.c_int64(W), C.c_int64(H * W), _lib.ptr(out), black,
                  C.c_double(float(wp)), int(bool(norm)), int(bool(clip)), _lib.stream()), 'pack_bayer')
    if not batched:
        out = out[0]
    return _ret(out, host)


def bayer2raw(packed_raw, wp=16383, bl=512, device_out=False):
    """utils/isp_ops.py:98-112: f32 [4,h,w] / [1,4,h,w] -> u16 [2h,2w].  Like the
    reference the result is a numpy array unless ``device_out``."""
    x, _ = _to_dev(packed_raw.detach() if torch.is_tensor(packed_raw) else packed_raw)
    x = x.float()
    if x.dim() == 4:
        x = x[0].contiguous()
    _, h, w = x.shape
    out = torch.empty((2 * h, 2 * w), dtype=torch.uint16, device=x.device)
    _lib.check(_lib.lib().pnnp_unpack_bayer_u16(_lib.ptr(x), 1, h, w, _lib.ptr(out), int(wp), int(bl), _lib.stream()),
               'unpack_bayer')
    return out if device_out else out.cpu().numpy()


def _move(fn_name, a, out_shape):
    x, host = _to_dev(a)
    out = torch.empty(out_shape, dtype=x.dtype, device=x.device)
    return x, out, host, getattr(_lib.lib(), fn_name)


def bayer2rggb(bayer):
    """utils/isp_ops.py:57-59: [H,W] -> [H/2,W/2,4] in order (0,0),(0,1),(1,0),(1,1)."""
    H, W = bayer.shape
    x, out, host, fn = _move('pnnp_bayer_to_rggb', bayer, (H // 2, W // 2, 4))
    _lib.check(fn(_lib.ptr(x), _lib.ptr(out), H, W, x.element_size(), _lib.stream()), 'bayer2rggb')
    return _ret(out, host)


def rggb2bayer(rggb):
    """utils/isp_ops.py:61-63."""
    h, w, _ = rggb.shape
    x, out, host, fn = _move('pnnp_rggb_to_bayer', rggb, (2 * h, 2 * w))
    _lib.check(fn(_lib.ptr(x), _lib.ptr(out), h, w, x.element_size(), _lib.stream()), 'rggb2bayer')
    return _ret(out, host)


def bayer2rows(bayer):
    """utils/isp_ops.py:65-68: [H,W] -> [2,H/2,W]."""
    H, W = bayer.shape
    x, out, host, fn = _move('pnnp_bayer_to_rows', bayer, (2, H // 2, W))
    _lib.check(fn(_lib.ptr(x), _lib.ptr(out), H, W, x.element_size(), _lib.stream()), 'bayer2rows')
    return _ret(out, host)


def rows2bayer(rows):
    """utils/isp_ops.py:76-81: [2,h,W] -> float64 [2h,W] (np.empty default dtype)."""
    _, h, W = rows.shape
    if not torch.is_tensor(rows):
        rows = np.asarray(rows, np.float64)
    x, host = _to_dev(rows)
    x = x.to(torch.float64)
    out = torch.empty((2 * h, W), dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().pnnp_rows_to_bayer(_lib.ptr(x), _lib.ptr(out), h, W, 8, _lib.stream()), 'rows2bayer')
    return _ret(out, host)
