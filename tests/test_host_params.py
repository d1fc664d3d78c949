"""Host logic: noise-parameter tables and scalar samplers vs goldens captured from the
reference under np.random.seed (pins values AND the order of host RNG draws)."""
import json
import os

import numpy as np
import pytest

from pnnp_amd import process as P


def _close(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a, b)


def test_tables(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'params.json')))
    for cam, tab in g['tables'].items():
        mine = P.get_camera_noisy_params(cam)
        assert set(mine) == set(tab), cam
        for k in tab:
            assert _close(mine[k], tab[k]), (cam, k)
    for key, tab in g['specific'].items():
        cam, iso = key.split(':')
        mine = P.get_specific_noise_params(cam, int(iso))
        assert set(mine) == set(tab), key
        for k in tab:
            assert _close(mine[k], tab[k]), (key, k)
    assert P.get_specific_noise_params('NikonD850', 100) is None
    assert P.get_camera_noisy_params('NoSuchCamera') == P.get_camera_noisy_params('NikonD850')


def test_seeded_samplers(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'params.json')))
    assert len(g['samples']) == 27
    for tag, c in g['samples'].items():
        seed = int(tag[1])
        np.random.seed(seed)
        fn = P.sample_params_max if c['kind'] == 'max' else P.sample_params
        a = fn(**c['kw']); b = fn(**c['kw'])
        for got, ref in ((a, c['first']), (b, c['second'])):
            assert set(got) == set(ref), tag
            for k in ref:
                assert _close(got[k], ref[k]), (tag, k, got[k], ref[k])


def test_reference_quirk_keyerror(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'params.json')))
    for cam, err in g['errors'].items():
        assert err == 'KeyError'
        with pytest.raises(KeyError):
            P.sample_params(camera_type=cam)


def test_flags_and_param_rows():
    assert P.noise_flags('PRq') == 0x01 | 0x04 | 0x08
    assert P.noise_flags('pb', ori=True, clip=True, torch_mode=True) == 0x01 | 0x20 | 0x100 | 0x200 | 0x1000
    import torch
    p = dict(K=1.5, sigGs=2.0, sigTL=3.0, lam=-0.1, sigR=0.5, q=1 / 2 ** 14, ratio=100.0, wp=16383, bl=512,
             bias=np.array([1., 2., 3., 4.]))
    rows = P.pack_params([p, dict(p, bias=0)], torch.device('cpu'))
    assert rows.shape == (2, 16)
    assert rows[0, :13].tolist() == pytest.approx([1.5, 2.0, 3.0, -0.1, 0.5, 1 / 2 ** 14, 100.0, 16383, 512, 1, 2, 3, 4])
    assert rows[1, 9:13].tolist() == [0, 0, 0, 0]


def test_load_weights_by_name():
    import torch
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.utils import load_weights, pkl_convert, tensor_dim5to4
    net = UNetSeeInDark(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4))
    sd = {k: torch.full_like(v, 0.5) for k, v in net.state_dict().items()}
    sd['conv1_1.weight'] = torch.zeros(3, 3)              # wrong shape -> dropped
    sd['not_there.weight'] = torch.zeros(1)                # unknown key -> dropped
    before = net.state_dict()['conv1_1.weight'].clone()
    load_weights(net, {'module.' + k: v for k, v in sd.items()} and sd, by_name=True)
    got = net.state_dict()
    assert torch.equal(got['conv1_1.weight'], before)
    assert float(got['conv5_2.bias'].mean()) == 0.5
    assert set(pkl_convert({'module.a': 1, 'b': 2})) == {'a'}
    assert tensor_dim5to4(torch.zeros(2, 8, 4, 16, 16)).shape == (16, 4, 16, 16)


def test_high_bit_recovery_lut_matches_reference(golden_dir):
    """HighBitRecovery.get_lut / HB2LB_LUT (process.py:686-716) on the host: same numpy draws, same scipy cdf per integer."""
    import json, os
    import numpy as np
    from pnnp_amd import process as P
    g = np.load(os.path.join(golden_dir, 'hbr.npz')); meta = json.load(open(os.path.join(golden_dir, 'hbr.json')))
    for tag, m in meta.items():
        np.random.seed(m['seed'])
        hbr = P.HighBitRecovery(camera_type=m['camera_type'], noise_code=m['noise_code'])
        hbr.get_lut([m['iso']], blc_mean=None)
        L = hbr.lut[m['iso']]
        assert (L['low'], L['high'], L['bias'], L['sigma']) == (m['low'], m['high'], m['bias'], m['sigma'])
        np.testing.assert_allclose(L['cdf'], g[tag + '_cdf'], rtol=1e-13)
        np.testing.assert_allclose(L['range'], g[tag + '_range'], rtol=1e-10, atol=1e-300)
