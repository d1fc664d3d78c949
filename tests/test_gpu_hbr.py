"""GPU parity of HighBitRecovery (process.py:675-751): the LUT reproduces the reference's scipy / numpy
draws; map() with the reference's uniform draw injected matches its output (float32 result of a float64 quantile:
rtol 1e-5 / atol 2e-6 of the DN range); with the device RNG the re-drawn values keep the bin they came from."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'hbr.npz')), json.load(open(os.path.join(golden_dir, 'hbr.json')))


@pytest.mark.parametrize('tag', ['imx_gauss', 'sony_tukey'])
def test_hbr_lut_and_map(G, tag):
    from pnnp_amd import process as P
    g, meta = G
    m = meta[tag]
    np.random.seed(m['seed'])
    hbr = P.HighBitRecovery(camera_type=m['camera_type'], noise_code=m['noise_code'])
    hbr.get_lut([m['iso']], blc_mean=None)
    L = hbr.lut[m['iso']]
    assert (L['low'], L['high']) == (m['low'], m['high']) and L['bias'] == m['bias'] and L['sigma'] == m['sigma']
    np.testing.assert_allclose(L['cdf'], g[tag + '_cdf'], rtol=1e-13, atol=0)
    np.testing.assert_allclose(L['range'], g[tag + '_range'], rtol=1e-10, atol=1e-300)
    data = torch.from_numpy(g[tag + '_data']).cuda(); u = torch.from_numpy(g[tag + '_u']).cuda()
    span = m['wp'] - m['bl']
    res = hbr.map(data, iso=m['iso'], norm=True, rand=u).cpu().numpy()
    np.testing.assert_allclose(res, g[tag + '_res'], rtol=1e-5, atol=2e-6)
    res_dn = hbr.map(data, iso=m['iso'], norm=False, rand=u).cpu().numpy()
    np.testing.assert_allclose(res_dn, g[tag + '_res_dn'], rtol=1e-5, atol=2e-6 * span)
    # device RNG: every re-drawn value stays inside the bin of the integer it replaced; out-of-range pixels untouched
    P.manual_seed(5)
    own = hbr.map(data, iso=m['iso'], norm=False).cpu().numpy() - m['bl']
    d = g[tag + '_data'] * span
    x = np.round(d); delta = d - x
    inside = (x >= m['low']) & (x < m['high'])
    assert np.all(np.abs((own - delta)[inside] - x[inside]) <= 0.5 + 1e-3)
    np.testing.assert_allclose(own[~inside], d[~inside], rtol=0, atol=1e-3)
    assert np.abs((own - delta)[inside] - x[inside]).mean() > 0.1           # and really re-drawn
