"""GPU parity of SNA_torch (shot-noise augmentation, process.py:562-588; SURVEY 8f row f4 / Trainer.preprocess Mix_Dataset branch):
* the gain draw K and the deterministic signal term dy: exact against the reference's golden (tests/golden/sna.*);
* the Poisson term dn: tier A element-wise against the C oracle on the same counter RNG, tier B moments against the
  reference's own draw (mean within 5 sigma of the sampling error, variance within 10 %)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def G(golden_dir):
    return np.load(os.path.join(golden_dir, 'sna.npz')), json.load(open(os.path.join(golden_dir, 'sna.json')))


@pytest.mark.parametrize('tag', ['imx', 'imx_black', 'sony'])
def test_sna(G, tag):
    from oracle import cbind
    from pnnp_amd import process as P
    g, meta = G
    m = meta[tag]
    gt = torch.from_numpy(g['gt']).cuda()
    aug = np.array(m['aug'], np.float32)
    np.random.seed(9)
    P.manual_seed(77)
    dn, dy = P.SNA_torch(gt, aug.copy(), camera_type=m['camera_type'], ratio=m['ratio'], black_lr=m['black_lr'], ori=m['ori'], iso=m['iso'])
    assert np.array_equal(dy.cpu().numpy(), g[tag + '_dy'])                     # deterministic part: bit-exact
    # tier A: same counter RNG in the C oracle (K from the golden = the same numpy draw)
    seed, off = P.get_rng_state()
    rn, ry = cbind.sna(g['gt'], aug, m['K'], m['wp'], m['bl'], m['ratio'], m['black_lr'], m['ori'], seed=seed, offset=off - 1)
    assert np.array_equal(ry, g[tag + '_dy'])
    d = np.abs(dn.cpu().numpy() - rn)
    scale = m['K'] / (m['wp'] - m['bl']) * (1.0 if m['ori'] else m['ratio'])
    assert float((d <= 1e-5 * scale + 1e-6 * np.abs(rn)).mean()) >= 0.999
    # tier B: moments against the reference's torch.poisson draw
    x = dn.cpu().numpy().reshape(4, -1).astype(np.float64)
    n = x.shape[1]
    for c in range(4):
        mu, var = m['dn_mean'][c], m['dn_var'][c]
        if var == 0.0:
            assert np.all(x[c] == 0.0)
            continue
        assert abs(x[c].mean() - mu) < 5 * np.sqrt(2 * var / n)
        assert abs(x[c].var() - var) < 0.10 * var


def test_sna_errors():
    from pnnp_amd import process as P
    gt = torch.rand(4, 8, 8, device='cuda')
    with pytest.raises(KeyError):
        P.SNA_torch(gt, np.ones(4, np.float32), camera_type='IMX686', iso=123)      # not a calibrated ISO: dict lookup, like the reference
