"""GPU parity of the HIP noise sampler (through the C ABI):
tier A  HIP vs the C oracle on the same counter-based RNG -- element-wise;
tier B  HIP vs the reference's own draws (golden moments / histograms) -- statistical.
Tolerances are stated inline."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SONY = dict(K=1.5256, sigGs=6.3, sigTL=3.2, lam=-0.026, sigR=0.98, q=1 / 2 ** 14, ratio=150.0, wp=16383, bl=512,
            bias=np.array([0.5, -0.25, 0.125, 1.0]))
IMX = dict(K=8.74253, sigGs=14.30362, sigTL=12.8901, lam=0.015, sigR=0.9, q=1 / 2 ** 10, ratio=2.0, wp=1023, bl=64,
           bias=np.array([-0.08113494, -0.04906388, -0.9408157, -1.2048522]))


def _hip(y, plist, flags, mfm=1.0, seed=1997, offset=3, crop_base=0):
    from pnnp_amd import process as P
    yd = torch.from_numpy(y).cuda()
    rows = P.pack_params(plist, yd.device)
    return P.noise_sample(yd, rows, flags, mfm=mfm, seed=seed, offset=offset, crop_base=crop_base).cpu().numpy()


@pytest.mark.parametrize('code,torch_mode', [('p', True), ('pr', True), ('prq', True), ('pb', True), ('pbrq', True),
                                             ('p', False), ('pr', False), ('pgrq', False), ('prqd', False), ('pb', False),
                                             ('rq', False), ('g', False)])
@pytest.mark.parametrize('ori,clip', [(False, False), (True, True)])
def test_tier_a_vs_c_oracle(code, torch_mode, ori, clip):
    """Same uniforms on both sides; differences come only from libm (log/exp/cos/lgamma)
    rounding, which can flip a Poisson accept/reject at a handful of pixels.
    A flipped rejection draws a fresh, independent Poisson sample for that pixel.
    Bar: >= 99.9 % of pixels equal to 1e-5 relative; the rest within 10 sqrt(lam)+2 counts."""
    from oracle import cbind
    from pnnp_amd import process as P
    rng = np.random.default_rng(42)
    B, C, H, W = 3, 4, 64, 100        # W/4 not a multiple of 64: waves straddle rows
    y = (rng.random((B, C, H, W), dtype=np.float32) ** 3).astype(np.float32)
    y[0, :, :8] = 0.0                 # lam = 0 and tiny-lam regions
    y[1, :, :8] *= 1e-3
    plist = [SONY, IMX, dict(SONY, ratio=100.0, K=0.3)]      # crop 2: lam up to ~500 (deep PTRS)
    flags = P.noise_flags(code, ori=ori, clip=clip, torch_mode=torch_mode)
    assert flags == cbind.noise_flags(code, ori=ori, clip=clip, torch_mode=torch_mode)
    for mfm in (1.0, 2.0):
        ref = cbind.noise_sample(y, cbind.param_rows(plist), flags, mfm=mfm, seed=1997, offset=3, crop_base=5)
        got = _hip(y, plist, flags, mfm=mfm, seed=1997, offset=3, crop_base=5)
        assert got.shape == ref.shape and np.isfinite(got).all()
        for b, p in enumerate(plist):
            scale = (1.0 if ori else p['ratio']) / (p['wp'] - p['bl'])
            d = np.abs(got[b] - ref[b])
            tol = 1e-5 * np.maximum(np.abs(ref[b]), scale)
            frac = float((d <= tol).mean())
            assert frac >= 0.999, (code, b, mfm, frac)
            lam = mfm * y[b] * (p['wp'] - p['bl']) / p['ratio'] / p['K']
            bound = (10 * np.sqrt(np.maximum(lam, 1)) + 2) * p['K'] / mfm * scale + tol
            assert (d <= bound).all(), (code, b, float((d / bound).max()))


def test_determinism_and_counter_semantics():
    from pnnp_amd import process as P
    rng = np.random.default_rng(1)
    y = rng.random((4, 4, 32, 48), dtype=np.float32)
    plist = [SONY] * 4
    f = P.noise_flags('prq', torch_mode=True)
    a = _hip(y, plist, f, offset=7)
    b = _hip(y, plist, f, offset=7)
    assert np.array_equal(a, b)                               # same (seed, offset) => same bits
    assert not np.array_equal(a, _hip(y, plist, f, offset=8))  # next call => new noise
    assert not np.array_equal(a, _hip(y, plist, f, seed=1998, offset=7))
    # crop_base: sharding the batch over ranks reproduces the single-process result
    lo = _hip(y[:2], plist[:2], f, offset=7, crop_base=0)
    hi = _hip(y[2:], plist[2:], f, offset=7, crop_base=2)
    assert np.array_equal(np.concatenate([lo, hi]), a)


def test_grid_stride_second_trip_and_ragged_geometry():
    """The kernel's index arithmetic is 32-bit multiply-shift division (csrc/noise.hip FastDiv) and its row-noise normals are drawn per
    BLOCK and re-drawn in every trip of the grid-stride loop.  (a) 20 crops of 4x512x512 are more quads than the capped grid holds (second
    trip): the whole batch must equal its two halves sampled separately with crop_base -- which run one trip each.  (b) odd sizes (W % 4 != 0,
    C = 3, rows shorter than a wave, a single row) against the C oracle at tier A's bar."""
    from oracle import cbind
    from pnnp_amd import process as P
    g = torch.Generator(device='cuda').manual_seed(3)
    y = torch.rand(20, 4, 512, 512, device='cuda', generator=g) * 0.3
    rows = P.pack_params([SONY] * 20, y.device)
    f = P.noise_flags('prq', torch_mode=True)
    whole = P.noise_sample(y, rows, f, seed=5, offset=11)
    lo = P.noise_sample(y[:10], rows[:10], f, seed=5, offset=11, crop_base=0)
    hi = P.noise_sample(y[10:], rows[10:], f, seed=5, offset=11, crop_base=10)
    assert torch.equal(whole, torch.cat([lo, hi]))
    rng = np.random.default_rng(9)
    for B, C, H, W in ((7, 3, 5, 37), (2, 4, 1, 1030), (1, 1, 300, 6), (3, 4, 33, 4)):
        yy = (rng.random((B, C, H, W), dtype=np.float32) ** 2).astype(np.float32)
        plist = [dict(SONY, ratio=100.0 + 50 * b) for b in range(B)]
        for code in ('pr', 'prq'):
            fl = P.noise_flags(code, torch_mode=True)
            ref = cbind.noise_sample(yy, cbind.param_rows(plist), fl, mfm=1.0, seed=1997, offset=3, crop_base=2)
            got = _hip(yy, plist, fl, crop_base=2)
            for b, pp in enumerate(plist):
                scale = pp['ratio'] / (pp['wp'] - pp['bl'])
                d = np.abs(got[b] - ref[b])
                frac = float((d <= 1e-5 * np.maximum(np.abs(ref[b]), scale)).mean())
                assert frac >= 0.995, ((B, C, H, W), code, b, frac)       # (small maps: one flipped pixel of 555 is already 0.2 %)


def test_row_noise_structure():
    """'r': one draw per (packed channel, row), constant along W, independent across
    channels, rows and crops (process.py:615,660)."""
    from pnnp_amd import process as P
    y = np.zeros((2, 4, 128, 200), np.float32)
    p = dict(SONY, sigGs=0.0, sigR=2.0, ratio=1.0)
    # code 'pr' with y=0 and sigGs=0 leaves only the row term (+0 poisson)
    z = _hip(y, [p, p], P.noise_flags('pr', ori=True, torch_mode=True)) * (p['wp'] - p['bl'])
    assert np.allclose(z, z[..., :1], atol=0)                 # constant along W
    r = z[..., 0]
    assert abs(r.std() - 2.0) < 0.25 and abs(r.mean()) < 0.3   # 1024 draws: sd(sd)~0.045
    assert abs(np.corrcoef(r[0, 0], r[0, 1])[0, 1]) < 0.3     # channels independent
    assert abs(np.corrcoef(r[0, 0], r[1, 0])[0, 1]) < 0.3     # crops independent


def test_deterministic_parts_exact():
    """With every noise source off (lam=0 => Poisson 0, sig=0) the arithmetic skeleton is
    exact: clip bounds -bl/wp (not -bl/(wp-bl)), x ratio unless ori, dark bias per channel."""
    from pnnp_amd import process as P
    y = np.zeros((1, 4, 8, 16), np.float32)
    p = dict(SONY, sigGs=0.0, sigR=0.0, ratio=200.0, bias=np.array([-600.0, 10.0, 20000.0, 0.0]))
    z = _hip(y, [p], P.noise_flags('pd', torch_mode=False))
    span = np.float32(16383 - 512)
    exp = np.clip(np.float32(p['bias']) / span, -np.float32(512) / np.float32(16383), 1).astype(np.float32) * np.float32(200.0)
    for c in range(4):
        assert np.all(z[0, c] == exp[c]), (c, z[0, c, 0, 0], exp[c])
    z = _hip(y, [p], P.noise_flags('pd', ori=True, clip=True, torch_mode=False))
    exp = np.clip(np.float32(p['bias']) / span, 0, 1).astype(np.float32)
    for c in range(4):
        assert np.all(z[0, c] == exp[c])
    # OBS mode: 'b' removes bias too; TORCH mode keeps it (process.py:660-663)
    assert np.all(_hip(y, [p], P.noise_flags('pdb', torch_mode=False)) == 0)
    assert np.any(_hip(y, [p], P.noise_flags('pdb', torch_mode=True)) != 0)


def test_tier_b_vs_reference_statistics(golden_dir):
    """Distribution parity with the reference's own draws (4x256x256 flat patches, numpy and
    torch paths).  Bars (tests/_noise_stats.py): mean within 5 sigma, variance within 3 %
    (12 % when 1024 row draws dominate), row-mean variance within 15 %, integer-DN histogram
    KL (kl_div_norm definition, utils/kld_div.py:163-200) < 2e-3."""
    from _noise_stats import check_against_reference
    from pnnp_amd import process as P

    def sample(y, p, flags, seed, offset):
        return _hip(y, [p], flags, seed=seed, offset=offset)
    assert check_against_reference(golden_dir, sample, P.noise_flags) >= 150


def test_row_variance_analytic():
    from _noise_stats import check_row_variance_analytic
    from pnnp_amd import process as P

    def sample(y, p, flags, seed, offset):
        return _hip(y, [p] * y.shape[0], flags, seed=seed, offset=offset)
    check_row_variance_analytic(sample, P.noise_flags)


def test_reference_api_mirror():
    """generate_noisy_torch / generate_noisy_obs signatures, parameter forms and errors."""
    from pnnp_amd import process as P
    P.manual_seed(1997)
    y = torch.rand(4, 64, 64, device='cuda') * 0.5
    host = P.sample_params_max('SonyA7S2')
    dev = {k: torch.from_numpy(np.array(v, np.float32)).cuda() for k, v in host.items()}   # trainer_SID.py:455-459
    P.manual_seed(5); a = P.generate_noisy_torch(y, param=dev, noise_code='pr', ori=False, clip=P.HALF_CLIP)
    P.manual_seed(5); b = P.generate_noisy_torch(y, param=host, noise_code='pr', ori=False, clip=P.HALF_CLIP)
    assert a.shape == y.shape and a.is_cuda and torch.equal(a, b)
    c = P.generate_noisy_torch(y, param=host, noise_code='pr')
    assert not torch.equal(a, c)                    # offset advanced
    with pytest.raises(NotImplementedError):
        P.generate_noisy_torch(y, param=host, noise_code='pg')
    with pytest.raises(TypeError):
        P.generate_noisy_torch(y, param=host, noise_code='r')
    z = P.generate_noisy_obs(y.cpu().numpy(), noise_code='pgrq', param=host)
    assert isinstance(z, np.ndarray) and z.dtype == np.float32 and z.shape == (4, 64, 64)


def test_tukey_extension_in_torch_mode():
    """Row f3: with tukey=True the torch-mode sampler accepts 'g' and draws the Tukey-lambda read noise of
    generate_noisy_obs.  (i) tier A against the C oracle with the same flag; (ii) for a code without r/q/d/b the
    two modes are the same pipeline, so torch-mode 'pg' must equal obs-mode 'pg' bit for bit; (iii) the
    C ABI still refuses TORCH+G without the flag (process.py:654)."""
    from oracle import cbind
    from pnnp_amd import _lib, process as P
    rng = np.random.default_rng(7)
    y = (rng.random((2, 4, 48, 72), dtype=np.float32) ** 2).astype(np.float32)
    plist = [SONY, IMX]
    flags = P.noise_flags('pgrq', clip=True, torch_mode=True) | P.F_TORCH_TUKEY
    ref = cbind.noise_sample(y, cbind.param_rows(plist), flags, seed=11, offset=2)
    got = _hip(y, plist, flags, seed=11, offset=2)
    for b, p in enumerate(plist):
        tol = 1e-5 * np.maximum(np.abs(ref[b]), p['ratio'] / (p['wp'] - p['bl']))
        assert float((np.abs(got[b] - ref[b]) <= tol).mean()) >= 0.999
    a = _hip(y, plist, P.noise_flags('pg', torch_mode=True) | P.F_TORCH_TUKEY, seed=11, offset=2)
    b = _hip(y, plist, P.noise_flags('pg', torch_mode=False), seed=11, offset=2)
    assert np.array_equal(a, b)
    with pytest.raises(_lib.PnnpError):
        _hip(y, plist, P.noise_flags('pg', torch_mode=True))
    yd = torch.from_numpy(y[0]).cuda()
    P.manual_seed(3); z = P.generate_noisy_torch(yd, param=SONY, noise_code='pgrq', tukey=True)
    P.manual_seed(3); z2 = P.generate_noisy_torch(yd, param=SONY, noise_code='prq')
    assert z.shape == yd.shape and not torch.equal(z, z2)
    # heavier tails than the Gaussian of the same scale would give: the read term uses sigTL and lam
    dark = torch.zeros(4, 256, 256, device='cuda')
    P.manual_seed(4); n = P.generate_noisy_torch(dark, param=dict(SONY, lam=-0.2), noise_code='pg', tukey=True, ori=True)
    n = n.cpu().numpy().astype(np.float64) * (SONY['wp'] - SONY['bl'])
    n = n[n > -SONY['bl'] / SONY['wp'] * (SONY['wp'] - SONY['bl']) + 1e-3]
    from scipy import stats
    q = np.quantile(n, [0.05, 0.25, 0.5, 0.75, 0.95])
    want = stats.tukeylambda.ppf([0.05, 0.25, 0.5, 0.75, 0.95], -0.2) * SONY['sigTL']
    assert np.abs(q - want).max() < 0.08 * SONY['sigTL']


@pytest.mark.parametrize('clip', [1, 2])
@pytest.mark.parametrize('ori', [False, True])
def test_trainer_preprocess_clamp_is_the_oracle_sampler_plus_the_trainers_clamp(clip, ori):
    """HipTrainStep.make_noisy -- the path every bench step runs -- fuses the clamp of Trainer.preprocess
    (trainer_SID.py:481-485: lr.clamp(lb, 1) with lb = -inf for clip == HALF_CLIP (2), else 0) into the sampler's store
    (F_POST_MAX1 / F_POST_MIN0).  Checked against the C oracle's sampler WITHOUT the post flags followed by that clamp in
    numpy, on crops bright and dark enough for both bounds to bite; tier-A bar (99.9 % of pixels to 1e-5)."""
    from oracle import cbind
    from pnnp_amd import process as P
    from pnnp_amd.trainer import HipTrainStep

    class _Net:                                    # make_noisy never touches the network
        engine = None
    rng = np.random.default_rng(7)
    B, C, H, W = 3, 4, 48, 72
    y = rng.random((B, C, H, W), dtype=np.float32)
    y[0] *= 1.2                                    # values above 1: the upper clamp bites after x ratio
    y[1] *= 0.002                                  # dark: read/row noise drives pixels below 0
    plist = [SONY, dict(SONY, ratio=250.0), dict(SONY, ratio=100.0, K=0.3)]
    ts = HipTrainStep(_Net(), camera_type='SonyA7S2', noise_code='pr', ori=ori, clip=clip, seed=11, rank=2)
    ts.step_count = 5
    got, rows = ts.make_noisy(torch.from_numpy(y).cuda(), plist)
    got = got.cpu().numpy()
    flags = cbind.noise_flags('pr', ori=ori, clip=True, torch_mode=True)          # clip: 1 and 2 are both truthy (process.py:668)
    ref = cbind.noise_sample(y, cbind.param_rows(plist), flags, seed=11, offset=5, crop_base=2 * B)
    ref = np.clip(ref, -np.inf if clip == 2 else 0.0, 1.0)
    assert got.max() <= 1.0 and got.min() >= 0.0  # (the sampler's own clip already floors at 0 before x ratio, process.py:668)
    assert (ref == 0.0).mean() > 0.01 and (ori or (ref == 1.0).mean() > 0.01)          # the bounds really are exercised (un-brightened crops never reach 1)
    for b, p in enumerate(plist):
        scale = (1.0 if ori else p['ratio']) / (p['wp'] - p['bl'])
        tol = 1e-5 * np.maximum(np.abs(ref[b]), scale)
        assert float((np.abs(got[b] - ref[b]) <= tol).mean()) >= 0.999, (clip, ori, b)
    # the same flags through the oracle's own post-clamp bits agree with the numpy clamp exactly
    pf = flags | P.F_POST_MAX1 | (0 if clip == 2 else P.F_POST_MIN0)
    ref2 = cbind.noise_sample(y, cbind.param_rows(plist), pf, seed=11, offset=5, crop_base=2 * B)
    assert np.array_equal(ref2, ref)
