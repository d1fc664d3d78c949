"""ORACLE (test infrastructure, not product): numpy restatement of the Bayer
pack / unpack index maps of the reference.

Reference: utils/isp_ops.py:57-112.  Pinned bit-exactly by
tests/golden/pack_*.npz (tests/test_oracle_isp.py).
"""
import numpy as np

# plane order of raw2bayer / bayer2raw: R, G1, B, G2 = Bayer offsets
# (row, col) = (0,0), (0,1), (1,1), (1,0)           utils/isp_ops.py:87-90,108-111
PACK_OFFSETS = ((0, 0), (0, 1), (1, 1), (1, 0))


def raw2bayer(raw, wp=1023, bl=64, norm=True, clip=False, bias=None):
    """utils/isp_ops.py:84-96.  u16/f32 [H,W] -> f32 [4,H/2,W/2].

    The reference subtracts an int64/float64 black-level array from a float32
    stack, which numpy promotes to float64; the quotient is rounded to float32
    once at the end.  Restated with explicit float64 arithmetic.
    """
    bias = np.zeros(4) if bias is None else np.asarray(bias)
    r = np.asarray(raw).astype(np.float32)
    planes = np.empty((4, r.shape[0] // 2, r.shape[1] // 2), np.float32)
    for k, (dy, dx) in enumerate(PACK_OFFSETS):
        planes[k] = r[dy::2, dx::2]
    if not norm:
        out = planes
    else:
        black = (bias.astype(np.float64) + float(bl)).reshape(4, 1, 1)
        out = (planes.astype(np.float64) - black) / (float(wp) - black)
    if clip:
        out = np.minimum(np.maximum(out, 0), 1)
    return out.astype(np.float32)


def bayer2raw(packed, wp=16383, bl=512):
    """utils/isp_ops.py:98-112.  f32 [4,h,w] (or [1,4,h,w]) -> u16 [2h,2w].

    float32 multiply then float32 add (python-int scalars stay float32 under
    numpy-2 promotion), then C-cast truncation into uint16.
    """
    p = np.asarray(packed, dtype=np.float32)
    if p.ndim == 4:
        p = p[0]
    p = np.minimum(np.maximum(p, np.float32(0)), np.float32(1))
    v = p * np.float32(wp - bl)
    v = v + np.float32(bl)
    _, h, w = v.shape
    raw = np.empty((2 * h, 2 * w), np.uint16)
    for k, (dy, dx) in enumerate(PACK_OFFSETS):
        raw[dy::2, dx::2] = v[k].astype(np.uint16)
    return raw


def bayer2rggb(bayer):
    """utils/isp_ops.py:57-59: [H,W] -> [H/2,W/2,4], order (0,0),(0,1),(1,0),(1,1)."""
    H, W = bayer.shape
    out = np.empty((H // 2, W // 2, 4), bayer.dtype)
    out[..., 0] = bayer[0::2, 0::2]
    out[..., 1] = bayer[0::2, 1::2]
    out[..., 2] = bayer[1::2, 0::2]
    out[..., 3] = bayer[1::2, 1::2]
    return out


def rggb2bayer(rggb):
    """utils/isp_ops.py:61-63: inverse of bayer2rggb."""
    h, w, _ = rggb.shape
    out = np.empty((2 * h, 2 * w), rggb.dtype)
    out[0::2, 0::2] = rggb[..., 0]
    out[0::2, 1::2] = rggb[..., 1]
    out[1::2, 0::2] = rggb[..., 2]
    out[1::2, 1::2] = rggb[..., 3]
    return out


def bayer2rows(bayer):
    """utils/isp_ops.py:65-68: [H,W] -> [2,H/2,W] (even rows, odd rows)."""
    return np.stack((bayer[0::2], bayer[1::2]))


def rows2bayer(rows):
    """utils/isp_ops.py:76-81: [2,h,W] -> float64 [2h,W] (np.empty default dtype)."""
    _, h, W = rows.shape
    out = np.empty((2 * h, W), np.float64)
    out[0::2] = rows[0]
    out[1::2] = rows[1]
    return out


def pack_raw_bayer(raw, wp=1023, clip=True):
    """data_process/process.py:40-64 (CFA-pattern-aware pack of a rawpy object; float32 arithmetic)."""
    im = np.asarray(raw.raw_image_visible).astype(np.float32)
    pat = np.asarray(raw.raw_pattern)
    planes = []
    for c in range(4):
        r, q = np.where(pat == c)
        planes.append(im[r[0]::2, q[0]::2])
    out = np.stack(planes, axis=0).astype(np.float32)
    black = np.array(raw.black_level_per_channel)[:, None, None].astype(np.float32)
    out = (out - black) / (wp - black)
    return np.clip(out, 0.0, 1.0) if clip else out
