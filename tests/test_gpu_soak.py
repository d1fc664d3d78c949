"""300-step soak of the fp16x2 kernel family on SID-like data (trainer_SID.py:86-102).

The family's precision is relative to each TENSOR's maximum (csrc/h2.h); every reference-golden and float64 test uses well-conditioned data, and
the trajectory test of tests/test_gpu_unet.py runs 40 steps on uniform crops.  This one trains the nf = 32 UNet for 300 optimiser steps on crops
shaped like SID short exposures -- hr = rand^2.2 x 0.1 (most pixels dark), 0.1 % of the pixels saturated at 1.0, the physics sampler with the
SonyA7S2 parameters (ratio ~ U(100, 300) per crop, `clip: 2`) -- three times from the same weights: on the fp32 `direct` family, on `direct`
started one float32 ulp away, and on fp16x2.  It checks that
  (a) fp16x2 follows the direct family's loss curve as closely as the one-ulp twin does (worst window, 3x + 1e-5: the trajectory test's bar);
  (b) the fp16x2-trained network denoises a held-out frame to the same PSNR on the device and through the CPU oracle (oracle/net_torch.py) within 0.002 dB
      (north_star's bar: 0.02 dB), and the fp16x2-trained and direct-trained networks agree as closely as direct and its one-ulp twin do (3x + 0.002 dB);
  (c) no tensor the kernels split -- activations and gradients, sampled at steps 1, 100, 200, 300 -- has more than `frac` of its non-zero
      elements' sum of squares below 2^-18 of its maximum; log2(amax / median) is recorded per tensor and the report names every tensor beyond
      18 bits with the share of its energy that sits in the absolute-error regime (what it costs).
PNNP_SOAK_OUT=<file>: the report is also written there (profiles/r6/soak.txt comes from that)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sid_like(B, S, gen):
    hr = torch.rand(B, 4, S, S, device='cuda', generator=gen) ** 2.2 * 0.1
    sat = torch.rand(B, 4, S, S, device='cuda', generator=gen) < 1e-3
    return torch.where(sat, torch.ones_like(hr), hr)


def _psnr(a, b):
    return float(-10.0 * torch.log10(((a.double().clamp(0, 1) - b.double().clamp(0, 1)) ** 2).mean()))


def test_h2_soak_300_steps_on_sid_like_crops():
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.trainer import HipTrainStep
    STEPS, B, S, NPOOL = 300, 4, 256, 6
    sd = O.init_state_he(O.unet_param_shapes(nf=32), seed=5, res_scale=0.5, head_scale=0.05, head_bias=0.1)
    gen = torch.Generator(device='cuda').manual_seed(21)
    pool = [_sid_like(B, S, gen) for _ in range(NPOOL)]
    held = _sid_like(1, S, gen)
    lines = []

    def run(pol, perturb=0.0, ranges=False):
        net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
        net.load_state_dict({k: v.clone() * (1.0 + perturb) for k, v in sd.items()})
        net = net.cuda()
        net.engine.set_policy(**pol)
        ts = HipTrainStep(net, lr=1e-4, camera_type='SonyA7S2', noise_code='pr', ori=False, clip=2, seed=1997)
        losses, rep = [], {}
        for s in range(STEPS):
            np.random.seed(1997 + s)                     # the same per-crop parameter draws (ratio ~ U(100, 300), K, sigmas) for every family
            losses.append(float(ts.step(pool[s % NPOOL])[0]))
            if ranges and (s + 1) in (1, 100, 200, 300):
                rep[s + 1] = net.engine.h2_range_report()
        return net, ts, losses, rep

    direct = dict(x3=False, wino=False, thin=False)
    _, _, ref, _ = run(direct)
    net_t, _, twin, _ = run(direct, perturb=1e-7)
    net_d, _, _, _ = run(direct)                          # (the direct network itself, for the held-out frame; bit-identical to the first run)
    net_h, ts_h, h2, rep = run(dict(x3=True, wino=True, thin=True, h2=True), ranges=True)
    assert np.isfinite(h2).all() and np.isfinite(ref).all()
    assert ref[-1] < 0.8 * ref[0], (ref[0], ref[-1])      # the run really trains

    def worst(a, lo, hi):
        return max(abs(p - q) / q for p, q in zip(a[lo:hi], ref[lo:hi]))
    lines.append(f'soak: {STEPS} Adam steps (lr 1e-4), UNet nf=32, {B} crops of 4x{S}x{S} per step from a pool of {NPOOL} batches; hr = rand^2.2 * 0.1 with 0.1 % saturated pixels; '
                 f"physics sampler 'pr', SonyA7S2, clip 2; loss {ref[0]:.5f} -> {ref[-1]:.5f} (direct), {h2[0]:.5f} -> {h2[-1]:.5f} (fp16x2)")
    lines.append('worst relative loss difference against the direct family, per window of 50 steps:   fp16x2   |   direct started one ulp away')
    ok = True
    for lo in range(0, STEPS, 50):
        d, u = worst(h2, lo, lo + 50), worst(twin, lo, lo + 50)
        lines.append(f'  steps {lo + 1:3d}-{lo + 50:3d}: {d:.2e} | {u:.2e}')
        ok = ok and d <= 3 * u + 1e-5
    d_all, u_all = worst(h2, 0, STEPS), worst(twin, 0, STEPS)
    lines.append(f'  whole run: fp16x2 {d_all:.2e}, one-ulp twin {u_all:.2e} (bar: 3x + 1e-5 per window)')

    # (b) held-out frame: noisy by the same sampler, denoised by both final networks and by the CPU oracle with the fp16x2 network's weights
    np.random.seed(4242)
    noisy, _ = ts_h.make_noisy(held)
    with torch.no_grad():
        out_h, out_d, out_t = net_h(noisy), net_d(noisy), net_t(noisy)
        sd_h = {k: v.detach().cpu().clone() for k, v in net_h.state_dict().items()}
        out_o = O.unet_forward(sd_h, noisy.cpu())
    p_h, p_d, p_t, p_o, p_in = _psnr(out_h, held), _psnr(out_d, held), _psnr(out_t, held), _psnr(out_o.cuda(), held), _psnr(noisy, held)
    lines.append(f'held-out frame 4x{S}x{S} (PSNR vs the clean frame, dB): noisy input {p_in:.4f}; fp16x2-trained net on the device {p_h:.5f}, the SAME weights through the CPU oracle '
                 f'{p_o:.5f} (|diff| {abs(p_h - p_o):.1e}; bar 0.002, north_star 0.02); direct-trained net {p_d:.5f} (|diff| to fp16x2-trained {abs(p_h - p_d):.5f}); '
                 f'direct started one ulp away {p_t:.5f} (|diff| to direct {abs(p_t - p_d):.5f}: what 300 steps of training make of ONE ulp)')

    # (c) dynamic range of every tensor the kernels split
    lines.append('per tensor, worst over steps 1 / 100 / 200 / 300: log2(amax / median|x|), share of non-zero elements below 2^-18 amax, share of the sum of squares they carry')
    worst_rows = {}
    for step, rows in rep.items():
        for r in rows:
            k = (r['kind'], r['name'])
            if k not in worst_rows or (r['log2_ratio'] == r['log2_ratio'] and r['log2_ratio'] > worst_rows[k][1]['log2_ratio']):
                worst_rows[k] = (step, r)
    beyond, worst_l2 = [], 0.0
    for (kind, name), (step, r) in sorted(worst_rows.items()):
        flag = ' <-- beyond 18 bits' if r['log2_ratio'] > 18 else ''
        lines.append(f"  {kind:4s} {name:10s} step {step:3d}: log2 {r['log2_ratio']:5.1f}  small {r['frac_small']:.2e}  l2 {r['l2_small']:.2e}  amax {r['amax']:.3e}{flag}")
        worst_l2 = max(worst_l2, r['l2_small'])
        if r['log2_ratio'] > 18:
            beyond.append(name)
    lines.append(f'tensors with log2(amax / median) > 18: {beyond if beyond else "none"}; largest share of a tensor\'s energy in the absolute-error regime: {worst_l2:.2e}')
    report = '\n'.join(lines)
    print(report)
    if os.environ.get('PNNP_SOAK_OUT'):
        with open(os.environ['PNNP_SOAK_OUT'], 'w') as f:
            f.write(report + '\n')
    assert ok, 'fp16x2 left the one-ulp band of the direct family'
    assert abs(p_h - p_o) < 0.002, (p_h, p_o)                  # the same network on the device and on the CPU oracle
    assert abs(p_h - p_d) < 3 * abs(p_t - p_d) + 0.002, (p_h, p_d, p_t)      # two trainings: no closer than the one-ulp twin gets (chaos, not precision)
    # elements below 2^-18 amax keep an absolute accuracy of 2^-40 amax: with a share e of the energy there, an output fed by them alone is off by
    # <= 2^-22 / sqrt(e) relative; 1e-6 of the energy keeps that at 2.4e-4 of such an output and 2.4e-10 of the tensor's scale
    assert worst_l2 < 1e-6, worst_l2
