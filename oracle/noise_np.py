"""ORACLE (test infrastructure, not product): the reference's physics-based
noise sampler restated on the reference's own RNG streams.

* ``generate_noisy_obs``   numpy stream   -- data_process/process.py:591-631
* ``generate_noisy_torch`` torch-CPU stream -- data_process/process.py:634-673

Because they draw from ``np.random`` / ``torch`` global generators in the same
order as the reference, seeding those generators reproduces the reference's
output bit for bit (pinned by tests/golden/noise_seeded.npz, generated with
numpy 2.2.6 / torch 2.10.0).  The HIP sampler uses its own counter-based RNG
(see oracle/pnnp_oracle.c); it is compared with these functions statistically.
"""
import numpy as np
import torch
from scipy import stats

HALF_CLIP = 2   # data_process/process.py:19


def _flags(noise_code):
    c = noise_code.lower()
    return {k: (k in c) for k in "rqgpdb"}


def generate_noisy_obs(y, camera_type=None, wp=16383, noise_code='p', param=None,
                       MultiFrameMean=1, ori=False, clip=False):
    """process.py:591-631 (numpy; what a DataLoader worker runs)."""
    p = param
    f = _flags(noise_code)
    span = p['wp'] - p['bl']
    y = y * span
    y = y / p['ratio']
    mfm = MultiFrameMean ** 0.5
    if f['p']:
        shot = np.random.poisson(mfm * y / p['K']).astype(np.float32) * p['K'] / mfm
    else:
        g = np.random.randn(*y.shape).astype(np.float32)
        shot = y + g * np.sqrt(np.maximum(y / p['K'], 1e-10)) * p['K'] / mfm
    read = row = quant = dark = 0
    if not f['b']:
        if f['g']:
            read = stats.tukeylambda.rvs(p['lam'], scale=p['sigTL'] / mfm, size=y.shape).astype(np.float32)
        else:
            read = stats.norm.rvs(scale=p['sigGs'] / mfm, size=y.shape).astype(np.float32)
        if f['r']:
            row = np.random.randn(y.shape[-3], y.shape[-2], 1).astype(np.float32) * p['sigR'] / mfm
        if f['q']:
            quant = np.random.uniform(low=-0.5, high=0.5, size=y.shape)
        if f['d']:
            dark = p['bias'].reshape(-1, 1, 1)
    z = (shot + read + row + quant + dark) / span
    z = np.clip(z, 0, 1) if clip else np.clip(z, -p['bl'] / p['wp'], 1)
    if ori is False:
        z = z * p['ratio']
    return z.astype(np.float32)


def generate_noisy_torch(y, camera_type=None, noise_code='p', param=None,
                         MultiFrameMean=1, ori=False, clip=False):
    """process.py:634-673 (torch; the trainer's per-crop device path).

    Quirks kept: row / quant / bias are applied even with 'b'; quant is scaled by
    q*(wp-bl); 'g' raises NotImplementedError; codes without 'p' hit the
    reference's broken ``tdist.Normal(y)`` call (TypeError).
    """
    p = param
    f = _flags(noise_code)
    span = p['wp'] - p['bl']
    y = y * span
    y = y / p['ratio']
    mfm = MultiFrameMean ** 0.5
    if not f['p']:
        raise TypeError("Normal.__init__() missing 1 required positional argument: 'scale'")
    shot = torch.poisson(mfm * y / p['K']) * p['K'] / mfm
    read = 0
    if not f['b']:
        if f['g']:
            raise NotImplementedError
        # tdist.Normal(loc, scale).sample() == torch.normal(loc.expand, scale.expand)
        loc = torch.zeros_like(y)
        scale = torch.as_tensor(p['sigGs'] / mfm, dtype=y.dtype).expand_as(y)
        read = torch.normal(loc, scale)
    row = torch.randn(y.shape[-3], y.shape[-2], 1) * p['sigR'] / mfm if f['r'] else 0
    quant = (torch.rand(y.shape) - 0.5) * p['q'] * span if f['q'] else 0
    dark = torch.from_numpy(p['bias'].reshape(-1, 1, 1)) if f['d'] else 0
    z = (shot + read + row + quant + dark) / span
    z = torch.clamp(z, 0, 1) if clip else torch.clamp(z, -p['bl'] / p['wp'], 1)
    if ori is False:
        z = z * p['ratio']
    return z


def kl_div_hist(p_samples, q_samples, bin_edges):
    """Histogram KL used as the statistical acceptance metric.  Definition as in
    utils/kld_div.py:163-200 (``kl_div_norm``): histograms over common bin edges
    normalised by the sample count, restricted to bins where both are non-zero;
    returns (kl_fwd, kl_inv, kl_sym)."""
    hp, _ = np.histogram(p_samples, bins=bin_edges)
    hq, _ = np.histogram(q_samples, bins=bin_edges)
    return kl_from_hist(hp, hq)


def kl_from_hist(hp, hq):
    p = hp.astype(np.float64) / max(hp.sum(), 1)
    q = hq.astype(np.float64) / max(hq.sum(), 1)
    both = (p > 0) & (q > 0)
    p, q = p[both], q[both]
    lp, lq = np.log(p), np.log(q)
    fwd = float(np.sum(p * (lp - lq)))
    inv = float(np.sum(q * (lq - lp)))
    return fwd, inv, 0.5 * (fwd + inv)
