"""Pin oracle/noise_np.py to the reference: seeded numpy / torch streams reproduce the
reference's generate_noisy_obs / generate_noisy_torch bit for bit."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import noise_np as O


def _params(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, 'noise_seeded.json')))
    P = []
    for p, ty in zip(meta['params'], meta['ptypes']):
        d = {}
        for k, v in p.items():     # rebuild the exact scalar types (NEP-50 promotion depends on them)
            d[k] = np.array(v) if ty[k] == 'ndarray' else (np.float64(v) if ty[k] == 'float64' else v)
        P.append(d)
    return meta['cases'], P


def test_seeded_numpy_path_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'noise_seeded.npz'))
    cases, P = _params(golden_dir)
    y = g['y']
    n = 0
    for c in cases:
        if c['kind'] != 'np':
            continue
        np.random.seed(11)
        z = O.generate_noisy_obs(y.copy(), noise_code=c['code'], param=dict(P[c['p']]), MultiFrameMean=c['mfm'],
                                 ori=c['ori'], clip=c['clip'])
        assert z.dtype == np.float32
        assert np.array_equal(z, g[c['tag']]), c['tag']
        n += 1
    assert n >= 150


def test_seeded_torch_path_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, 'noise_seeded.npz'))
    cases, P = _params(golden_dir)
    y = g['y']
    n = 0
    for c in cases:
        if c['kind'] != 'th':
            continue
        pt = {k: torch.from_numpy(np.array(v, np.float32)) for k, v in P[c['p']].items()}
        torch.manual_seed(11)
        z = O.generate_noisy_torch(torch.from_numpy(y.copy()), noise_code=c['code'], param=pt, ori=c['ori'], clip=c['clip'])
        assert np.array_equal(z.numpy(), g[c['tag']]), c['tag']
        n += 1
    assert n == 60


def test_torch_path_error_behaviour():
    y = torch.zeros(4, 8, 8)
    p = dict(K=1.0, sigGs=1.0, sigR=1.0, q=1 / 2 ** 14, ratio=100.0, wp=16383, bl=512, bias=np.zeros(4))
    with pytest.raises(NotImplementedError):      # process.py:654
        O.generate_noisy_torch(y, noise_code='pg', param=p)
    with pytest.raises(TypeError):                # process.py:651 (Normal without scale)
        O.generate_noisy_torch(y, noise_code='r', param=p)


def test_kl_definition():
    a = np.random.default_rng(0).normal(size=100000)
    b = np.random.default_rng(1).normal(size=100000)
    edges = np.linspace(-5, 5, 101)
    f, i, s = O.kl_div_hist(a, b, edges)
    assert 0 <= s < 2e-3
    f2, _, _ = O.kl_div_hist(a, b + 0.5, edges)
    assert f2 > 0.1
