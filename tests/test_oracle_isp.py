"""Pin oracle/isp_np.py to the reference's outputs (tests/golden/pack_*; bit-exact)."""
import hashlib
import json
import os

import numpy as np

from oracle import isp_np as O


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_raw2bayer_small_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    meta = json.load(open(os.path.join(golden_dir, 'pack_meta.json')))
    assert len(meta['cases']) == 32
    for c in meta['cases']:
        t = c['tag']
        got = O.raw2bayer(g[t + '_raw'], wp=c['wp'], bl=c['bl'], norm=c['norm'], clip=c['clip'], bias=g[t + '_bias'])
        ref = g[t + '_packed']
        assert got.dtype == np.float32 and got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), t


def test_bayer2raw_and_roundtrip(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    for (H, W) in [(16, 24), (64, 64)]:
        for (wp, bl) in [(16383, 512), (1023, 64)]:
            got = O.bayer2raw(g[f'unpack_H{H}W{W}wp{wp}_in'], wp=wp, bl=bl)
            assert got.dtype == np.uint16
            assert np.array_equal(got, g[f'unpack_H{H}W{W}wp{wp}_out'])
            raw = g[f'rt_H{H}W{W}wp{wp}_in']
            rt = O.bayer2raw(O.raw2bayer(raw, wp=wp, bl=bl), wp=wp, bl=bl)
            assert np.array_equal(rt, g[f'rt_H{H}W{W}wp{wp}_out'])
            assert np.array_equal(rt, raw)        # encode -> decode is the identity on in-range data


def test_index_maps(golden_dir):
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    b = g['maps_bayer']
    assert np.array_equal(O.bayer2rggb(b), g['maps_rggb'])
    assert np.array_equal(O.rggb2bayer(O.bayer2rggb(b)), b)
    assert np.array_equal(O.bayer2rows(b), g['maps_rows'])
    back = O.rows2bayer(O.bayer2rows(b))
    assert back.dtype == g['maps_rows_back'].dtype == np.float64
    assert np.array_equal(back, g['maps_rows_back'])


def test_full_crop_hash(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, 'pack_meta.json')))
    for k, m in meta['big'].items():
        raw = np.random.default_rng(m['seed']).integers(0, m['wp'] + 1, size=(1024, 1024), dtype=np.uint16)
        assert _sha(raw) == m['in_sha']
        p = O.raw2bayer(raw, wp=m['wp'], bl=m['bl'], norm=True, clip=True)
        assert _sha(p) == m['packed_sha']
        assert _sha(O.bayer2raw(p, wp=m['wp'], bl=m['bl'])) == m['unpack_sha']


def test_pack_raw_bayer(golden_dir):
    import types
    g = np.load(os.path.join(golden_dir, 'pack_small.npz'))
    for name in ('rggb', 'gbrg'):
        raw = types.SimpleNamespace(raw_image_visible=g[f'prb_{name}_im'], raw_pattern=g[f'prb_{name}_pat'],
                                    black_level_per_channel=list(g[f'prb_{name}_bl']))
        for clip in (True, False):
            got = O.pack_raw_bayer(raw, wp=16383, clip=clip)
            assert got.dtype == np.float32
            assert np.array_equal(got.view(np.uint32), g[f'prb_{name}_c{int(clip)}'].view(np.uint32)), (name, clip)
