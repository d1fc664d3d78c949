"""GPU parity: HIP Bayer pack/unpack (through the C ABI) vs the oracle and the golden
vectors captured from the reference.  Bit-exact."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_golden_small_bit_exact(golden_dir):
    from pnnp_amd import isp_ops as I
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'pack_meta.json')))
    for c in meta['cases']:
        t = c['tag']
        got = I.raw2bayer(g[t + '_raw'], wp=c['wp'], bl=c['bl'], norm=c['norm'], clip=c['clip'], bias=g[t + '_bias'])
        assert got.dtype == np.float32
        assert np.array_equal(got.view(np.uint32), g[t + '_packed'].view(np.uint32)), t
    for (H, W) in [(16, 24), (64, 64)]:
        for (wp, bl) in [(16383, 512), (1023, 64)]:
            got = I.bayer2raw(g[f'unpack_H{H}W{W}wp{wp}_in'], wp=wp, bl=bl)
            assert got.dtype == np.uint16 and np.array_equal(got, g[f'unpack_H{H}W{W}wp{wp}_out'])
            raw = g[f'rt_H{H}W{W}wp{wp}_in']
            assert np.array_equal(I.bayer2raw(I.raw2bayer(raw, wp=wp, bl=bl), wp=wp, bl=bl), raw)
    b = g['maps_bayer']
    assert np.array_equal(I.bayer2rggb(b), g['maps_rggb'])
    assert np.array_equal(I.rggb2bayer(g['maps_rggb']), b)
    assert np.array_equal(I.bayer2rows(b), g['maps_rows'])
    back = I.rows2bayer(g['maps_rows'])
    assert back.dtype == np.float64 and np.array_equal(back, g['maps_rows_back'])


def test_full_crop_hash_and_roundtrip(golden_dir):
    """BASELINE size: 1024x1024 Bayer tile <-> 4x512x512, hash of the reference's output."""
    from pnnp_amd import isp_ops as I
    meta = json.load(open(os.path.join(golden_dir, 'pack_meta.json')))
    for k, m in meta['big'].items():
        raw = np.random.default_rng(m['seed']).integers(0, m['wp'] + 1, size=(1024, 1024), dtype=np.uint16)
        p = I.raw2bayer(raw, wp=m['wp'], bl=m['bl'], norm=True, clip=True)
        assert _sha(p) == m['packed_sha']
        assert _sha(I.bayer2raw(p, wp=m['wp'], bl=m['bl'])) == m['unpack_sha']


@pytest.mark.parametrize('shape', [(2, 2), (2, 6), (6, 10), (34, 70), (130, 258), (2848, 4256)])
@pytest.mark.parametrize('dtype', ['u16', 'f32'])
def test_vs_oracle_ragged(shape, dtype):
    """Ragged widths (W/2 not a multiple of 4), tiny images, full Sony frame; device tensors in/out."""
    from oracle import cbind, isp_np
    from pnnp_amd import isp_ops as I
    H, W = shape
    rng = np.random.default_rng(H * 7919 + W)
    raw = rng.integers(0, 16384, size=(H, W), dtype=np.uint16)
    if dtype == 'f32':
        raw = raw.astype(np.float32) + rng.random((H, W), dtype=np.float32)
    bias = np.array([0.25, -1.5, 3.0, 0.0])
    for norm, clip in [(True, False), (True, True), (False, False)]:
        ref = isp_np.raw2bayer(raw, wp=16383, bl=512, norm=norm, clip=clip, bias=bias)
        ref_c = cbind.pack(raw, bias + 512, 16383, norm=norm, clip=clip)
        assert np.array_equal(ref.view(np.uint32), ref_c.view(np.uint32))     # numpy oracle == C oracle
        dev = torch.from_numpy(raw).cuda()
        got = I.raw2bayer(dev, wp=16383, bl=512, norm=norm, clip=clip, bias=bias)
        assert got.is_cuda and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy().view(np.uint32), ref.view(np.uint32)), (shape, dtype, norm, clip)
    packed = rng.random((4, H // 2, W // 2), dtype=np.float32) * 1.4 - 0.2
    u = I.bayer2raw(torch.from_numpy(packed).cuda(), wp=16383, bl=512)
    assert np.array_equal(u, isp_np.bayer2raw(packed, wp=16383, bl=512))
    assert np.array_equal(u, cbind.unpack(packed, 16383, 512))


def test_batched_and_empty():
    from oracle import isp_np
    from pnnp_amd import isp_ops as I
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 1024, size=(3, 64, 96), dtype=np.uint16)
    got = I.raw2bayer(torch.from_numpy(raw).cuda(), wp=1023, bl=64).cpu().numpy()
    for b in range(3):
        assert np.array_equal(got[b], isp_np.raw2bayer(raw[b], wp=1023, bl=64))
    e = I.raw2bayer(torch.zeros((0, 8), dtype=torch.uint16).cuda(), wp=1023, bl=64)
    assert tuple(e.shape) == (4, 0, 4)


def test_pack_raw_bayer_pattern(golden_dir):
    """process.py:40-64 pack_raw_bayer on a rawpy-like object, RGGB and GBRG patterns: bit-exact."""
    import types
    from pnnp_amd import process as P
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    for name in ('rggb', 'gbrg'):
        raw = types.SimpleNamespace(raw_image_visible=g[f'prb_{name}_im'], raw_pattern=g[f'prb_{name}_pat'],
                                    black_level_per_channel=list(g[f'prb_{name}_bl']))
        for clip in (True, False):
            got = P.pack_raw_bayer(raw, wp=16383, clip=clip)
            assert np.array_equal(got.view(np.uint32), g[f'prb_{name}_c{int(clip)}'].view(np.uint32)), (name, clip)


def test_every_uint16_code_is_bit_exact():
    """The kernel divides by the per-plane constant (wp - black) with a 3-operation correctly-rounded sequence instead of the
    hardware's fp64 division: exhaustive check over all 65536 input codes in every Bayer position, against numpy's float64
    division (the oracle), for the two cameras' levels, fractional biases and both clip settings; float32 inputs as well."""
    from oracle import isp_np as O
    from pnnp_amd import isp_ops as I
    codes = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    raw = np.empty((512, 512), np.uint16)
    raw[0::2, 0::2] = codes; raw[0::2, 1::2] = codes[::-1]; raw[1::2, 0::2] = codes.T; raw[1::2, 1::2] = codes[:, ::-1]
    rng = np.random.default_rng(5)
    for (wp, bl) in ((16383, 512), (1023, 64), (4095, 240)):
        for bias in (np.zeros(4), rng.normal(0, 3, 4), np.array([0.1, -0.7, 1e-3, 2.5])):
            for clip in (False, True):
                ref = O.raw2bayer(raw, wp, bl, True, clip, bias)
                got = I.raw2bayer(raw, wp, bl, True, clip, bias)
                assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (wp, bl, bias, clip)
    f = (rng.random((256, 256)) * 20000 - 1000).astype(np.float32)
    for bias in (np.zeros(4), np.array([0.3, -1.2, 7.7, 0.01])):
        ref = O.raw2bayer(f, 16383, 512, True, False, bias)
        got = I.raw2bayer(f, 16383, 512, True, False, bias)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
