"""Shared statistical acceptance test of a sampler against the reference's own draws
(tests/golden/noise_stats.*).  `sample(y[1,C,H,W], param_dict, flags, seed, offset)`
returns the noisy image (ori=True, clip=False)."""
import json
import os

import numpy as np

from oracle.noise_np import kl_from_hist


def check_against_reference(golden_dir, sample, flags_fn, subset=None):
    meta = json.load(open(os.path.join(golden_dir, 'noise_stats.json')))
    g = np.load(os.path.join(golden_dir, 'noise_stats.npz'))
    C, H, W = meta['shape']
    checked = 0
    for idx, c in enumerate(meta['cases']):
        if subset is not None and idx % subset[1] != subset[0]:
            continue
        base = {k: (np.array(v) if isinstance(v, list) else v) for k, v in meta['cams'][c['cam']].items()}
        p = dict(base, ratio=c['ratio'])
        y = np.full((1, C, H, W), c['y'], np.float32)
        th = c['kind'] == 'th'
        flags = flags_fn(c['code'], ori=True, clip=False, torch_mode=th)
        z = sample(y, p, flags, 99, idx)[0]
        dn = z.astype(np.float64) * (p['wp'] - p['bl'])
        mean, var, rowvar = g[c['tag'] + '_mom'][:3]
        n = dn.size
        use_row = 'r' in c['code'] and (th or 'b' not in c['code']) and p['sigR'] > 0
        # mean: 5 sigma of the difference of two independent sample means
        sig_mean = np.sqrt(2 * (var / n + (p['sigR'] ** 2 / (C * H) if use_row else 0)))
        assert abs(dn.mean() - mean) <= 5 * sig_mean + 1e-6, (c['tag'], dn.mean(), mean)
        if var > 1e-12:
            # variance: 5 sigma of the difference of two independent variance estimates,
            # sd(s^2) = sqrt((m4 - s^4)/n) (+ the 1024-draw row part), floor 1 %
            m4 = ((dn - dn.mean()) ** 4).mean()
            sd = np.sqrt(max(m4 - dn.var() ** 2, 0) / n + (2 * p['sigR'] ** 4 / (C * H - 1) if use_row else 0))
            tol = max(0.01, 5 * np.sqrt(2) * sd / var)
            assert abs(dn.var() / var - 1) < tol, (c['tag'], dn.var(), var, tol)
            if use_row:
                # both sides estimate the row variance from 1024 draws: sd of the ratio = 6.3 %; 5 sigma
                assert abs(dn.mean(axis=2).var() / rowvar - 1) < 0.32, (c['tag'], dn.mean(axis=2).var(), rowvar)
            else:
                hist, _ = np.histogram(dn, bins=g[c['tag'] + '_edges'])
                _, _, sym = kl_from_hist(hist, g[c['tag'] + '_hist'])
                assert sym < 2e-3, (c['tag'], sym)      # integer-DN histogram KL (kl_div_norm definition)
        else:
            assert dn.var() < 1e-9, c['tag']
        checked += 1
    return checked


def check_row_variance_analytic(sample, flags_fn, sig_r=1.7, sig_gs=3.0, crops=16):
    """Precise check of the row term: 16 x 4 x 256 = 16384 row draws (sd of the variance
    estimate 1.1 %): Var(row mean) = sigR^2 + sigGs^2 / W within 5 %; per-pixel variance
    = sigR^2 + sigGs^2 within 2 %."""
    C, H, W = 4, 256, 256
    p = dict(K=1.0, sigGs=sig_gs, sigTL=0.0, lam=0.0, sigR=sig_r, q=1 / 2 ** 14, ratio=1.0, wp=16383, bl=512,
             bias=np.zeros(4))
    y = np.zeros((crops, C, H, W), np.float32)
    for th in (False, True):
        z = sample(y, p, flags_fn('pr', ori=True, clip=False, torch_mode=th), 7, 11)
        dn = z.astype(np.float64) * (p['wp'] - p['bl'])
        rv = dn.mean(axis=3).var()
        assert abs(rv / (sig_r ** 2 + sig_gs ** 2 / W) - 1) < 0.05, rv
        assert abs(dn.var() / (sig_r ** 2 + sig_gs ** 2) - 1) < 0.02, dn.var()
