"""GPU parity of NoiseFlow.sample (archs/noise_flow.py:173-188) against the golden outputs of the
reference (injected prior draw z) and the torch oracle; plus the statistical sanity of the
counter-based prior draw."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ARCH = 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'


def _close_chain(got, ref):
    """Eight chained couplings (x1 = (z1 - shift) * exp(-s*tanh(.))) and 4x4 inverses in fp32: device
    expf/tanhf differ from the CPU's by ulps and the (z1 - shift) cancellation amplifies that at a few
    pixels.  Bar: >= 99.9 % of elements within rtol 2e-4 (+2e-4 of the output scale), all within 5 %."""
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    tight = err <= 2e-4 * np.abs(ref) + 2e-4 * scale
    assert tight.mean() >= 0.999, tight.mean()
    assert (err <= 5e-2 * np.abs(ref) + 5e-3 * scale).all(), err.max()


def _net(g):
    from pnnp_amd.archs import NoiseFlow
    net = NoiseFlow({'x_shape': (4, 32, 32), 'arch': ARCH})
    assert list(net.state_dict().keys()) == list(g['keys'])           # 222-key state_dict contract
    net.load_state_dict({k: torch.from_numpy(g['sd:' + k]) for k in net.state_dict().keys()})
    return net.cuda().eval()


def test_sample_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    clean = torch.from_numpy(g['clean']).cuda(); z = torch.from_numpy(g['z']).cuda()
    for iso in (100, 1600, 3000, 6400):
        x = net.sample(clean=clean, iso=torch.tensor(float(iso)).cuda(), z=z)
        ref = g[f'out_iso{iso}']
        _close_chain(x.cpu().numpy(), ref)
    assert torch.equal(z.cpu(), torch.from_numpy(g['z']))             # the injected draw is not clobbered


def test_sample_ragged_shape_vs_oracle(golden_dir):
    from oracle import noiseflow_torch as N
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('sd:')}
    gen = torch.Generator().manual_seed(9)
    clean = torch.rand(3, 4, 50, 70, generator=gen) * 0.02; z = torch.randn(3, 4, 50, 70, generator=gen)
    ref = N.sample(sd, clean, torch.tensor(800.0), z).numpy()
    x = net.sample(clean=clean.cuda(), iso=800.0, z=z.cuda()).cpu().numpy()
    _close_chain(x, ref)


def test_prior_draw_statistics_and_api(golden_dir):
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    clean = torch.rand(4, 4, 128, 128, device='cuda') * 0.01
    a = net.sample(clean=clean, iso=1600.0)
    b = net.sample(mode='sample', clean=clean, iso=1600.0)
    assert a.shape == clean.shape and not torch.equal(a, b) and torch.isfinite(a).all()
    # same model, injected standard-normal z: the sample std must agree with the self-drawn one
    z = torch.randn(clean.shape, device='cuda')
    c = net.sample(clean=clean, iso=1600.0, z=z)
    assert abs(float(a.std()) / float(c.std()) - 1) < 0.05 and abs(float(a.mean()) - float(c.mean())) < 0.05 * float(a.std())
    nll, sdz = net.loss(noise=a, clean=clean, iso=1600.0)             # evaluation-only NLL of its own samples: finite
    assert torch.isfinite(nll) and float(sdz) > 0


def test_config5_resunet_with_noiseflow_proxy_step(golden_dir):
    """BASELINE config 5 plumbing: ResUnet denoiser + NoiseFlow.sample proxy (IMX686 constants:
    ratio from {1,2,4,8,16}), fused train step: finite, reproducible noise statistics, loss decreases."""
    from pnnp_amd.archs import ResUnet, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    proxy = _net(g)
    torch.manual_seed(1); np.random.seed(1)
    net = ResUnet(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda()
    ts = HipTrainStep(net, lr=2e-3, clip=2)
    hr = torch.rand(3, 4, 64, 64, device='cuda') * 0.01
    noisy, ratio, iso = ts.make_noisy_proxy(hr, proxy, ratio_choices=(1, 2, 4, 8, 16))
    assert noisy.shape == hr.shape and torch.isfinite(noisy).all() and float(noisy.max()) <= 1.0
    assert float(torch.as_tensor(ratio).flatten()[0]) in (1, 2, 4, 8, 16) and iso in ts.LEGAL_ISO
    losses = []
    for _ in range(10):
        noisy, _, _ = ts.make_noisy_proxy(hr, proxy, ratio=ratio, iso=1600)
        losses.append(float(ts.step(hr, noisy=noisy)[0]))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses


def test_train_step_raises_the_scale_assertion_on_its_first_proxy_step_and_at_epoch_end(golden_dir):
    """signal_dependant.py:50 asserts `scale >= 0` inside every proxy sample.  The fused step defers it to a device flag: it must be
    read on the FIRST proxy step (a run shorter than `proxy_check_every` steps still fails, before Adam has seen NaN noise twice) and
    by `check_proxy()`, which the epoch loop of runfile.main calls at every epoch end."""
    from pnnp_amd.archs import ResUnet, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    proxy = _net(g)
    net = ResUnet(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda()
    bad = torch.full((2, 4, 32, 32), -1e4, device='cuda')                    # a * clean + b < 0
    good = torch.rand(2, 4, 32, 32, device='cuda') * 0.01
    ts = HipTrainStep(net, lr=1e-4, clip=2, proxy_net=proxy, proxy_ratio_choices=(1,), proxy_iso=1600)
    with pytest.raises(AssertionError):
        ts.step(bad)                                                         # step 1 reads the flag
    ts.check_proxy()                                                         # cleared by the raise
    for _ in range(3):
        ts.step(good)
    ts.step(bad)                                                             # step 5: not a check step -- the flag stays set on the device
    with pytest.raises(AssertionError):
        ts.check_proxy()                                                     # ... until the epoch end
    ts.check_proxy()


def test_fused_preprocess_equals_the_tensor_ops_of_the_reference(golden_dir):
    """NoiseFlow.sample_mixed = the proxy branch of preprocess (trainer_SID.py:463-472,481-485; trainer_LRID.py:419-427) in one chain
    of kernels: with the prior draw injected it must equal, BIT FOR BIT, the reference's sequence of tensor ops

        clean = hr / ratio;  noise = sample(clean=clean, iso=iso) * ratio;  lr = (hr + noise).clamp(lb, 1)

    for a per-batch scalar ratio (LRID) and per-crop ratios (SID), half clip and full clip; and the deferred `scale >= 0` check
    raises the reference's AssertionError when (and only when) a sample saw a negative scale."""
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    gen = torch.Generator(device='cuda').manual_seed(3)
    hr = torch.rand(3, 4, 48, 80, device='cuda', generator=gen) * 0.02
    hr[0, :, :4] = 0.999                                                     # pixels that the upper clamp catches
    z = torch.randn(hr.shape, device='cuda', generator=gen)
    for ratio in (4.0, torch.tensor([100.0, 237.5, 300.0], device='cuda')):
        rt = ratio.view(-1, 1, 1, 1) if torch.is_tensor(ratio) else ratio
        for lo in (-float('inf'), 0.0):
            # the elementwise ops of the reference on the CPU (IEEE division / multiply / add, one rounding each), sample() on the device
            rc = rt.cpu() if torch.is_tensor(rt) else rt
            clean = (hr.cpu() / rc).cuda()
            ref = (hr.cpu() + net.sample(clean=clean, iso=1600.0, z=z).cpu() * rc).clamp(lo, 1.0).cuda()
            got = net.sample_mixed(hr, ratio, 1600.0, lo, 1.0, z=z)
            same = (got == ref) | (got.isnan() & ref.isnan())
            assert bool(same.all()), (ratio, lo, int((~same).sum()), float((got - ref).abs().nan_to_num().max()), got[~same][:4].tolist(), ref[~same][:4].tolist())
    net.check_scale_flag()                                                   # nothing negative so far: no error, no flag
    bad = torch.full_like(hr, -1e4)                                          # a * clean + b < 0
    net.sample_mixed(bad, 1.0, 1600.0, z=z)
    with pytest.raises(AssertionError):
        net.check_scale_flag()
    net.check_scale_flag()                                                   # the flag was cleared by the raise
    with pytest.raises(AssertionError):                                      # the plain API still asserts at once, like the reference
        net.sample(clean=bad, iso=1600.0, z=z)


def test_forward_and_loss_match_reference_golden(golden_dir):
    """Density direction (evaluation only): z, the log-det objective and the NLL of the reference's
    forward()/loss() in eval mode.  z: the chain bar of _close_chain; objective / NLL: sums over 4096 pixels of eight
    couplings' log-scales, rtol 2e-5."""
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    noise = torch.from_numpy(g['fw_noise']).cuda(); clean = torch.from_numpy(g['clean']).cuda()
    for iso in (100, 3000):
        z, obj = net.forward(noise=noise, clean=clean, iso=torch.tensor(float(iso)).cuda())
        _close_chain(z.cpu().numpy(), g[f'fw_z_iso{iso}'])
        np.testing.assert_allclose(obj.cpu().numpy(), g[f'fw_obj_iso{iso}'], rtol=2e-5)
        nll, sdz = net.loss(noise=noise, clean=clean, iso=torch.tensor(float(iso)).cuda())
        assert abs(float(nll) - g[f'fw_nll_iso{iso}'][0]) < 1e-4 * abs(g[f'fw_nll_iso{iso}'][0])
        assert abs(float(sdz) - g[f'fw_nll_iso{iso}'][1]) < 1e-6
        nll2, _ = net(noise=noise, clean=clean, iso=float(iso), mode='loss')           # the reference's mode dispatch
        assert float(nll2) == float(nll)


def test_forward_ragged_vs_oracle_and_inverse_round_trip(golden_dir):
    from oracle import noiseflow_torch as N
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    net = _net(g)
    sd = {k: torch.from_numpy(g['sd:' + k]) for k in [str(x) for x in g['keys']]}
    gen = torch.Generator().manual_seed(3)
    noise = torch.randn(3, 4, 40, 40, generator=gen) * 0.03        # square (the reference's log-det assumes it), not a multiple of 32
    clean = torch.rand(3, 4, 40, 40, generator=gen) * 0.02
    iso = 800.0
    z, obj = net.forward(noise=noise.cuda(), clean=clean.cuda(), iso=iso)
    zr, objr = N.forward(sd, noise, clean, torch.tensor(iso))
    _close_chain(z.cpu().numpy(), zr.numpy())
    np.testing.assert_allclose(obj.cpu().numpy(), objr.numpy(), rtol=2e-5)
    back = net.inverse(noise=z, clean=clean.cuda(), iso=iso).cpu().numpy()         # x -> z -> x: 16 couplings in fp32
    err = np.abs(back - noise.numpy()); scale = float(noise.abs().max())
    assert (err <= 1e-3 * scale).mean() >= 0.99 and (err <= 5e-2 * scale).all(), (float(err.max()), float((err <= 1e-3 * scale).mean()))


# ------------------------------------------------------------------------------------------------ fitting (row f4)
def _grad_close(got, ref, name, rtol=2e-4):
    """Gradients are sums over B*H*W pixels of fp32 products; bar: rtol of the tensor's largest reference entry, plus an
    absolute floor for gradients that are mathematically zero (conv biases in front of a training-mode BatchNorm)."""
    tol = rtol * float(np.abs(ref).max()) + 2e-6
    err = float(np.abs(got - ref).max())
    assert err <= tol, (name, err, tol, float(np.abs(ref).max()))


def test_train_mode_loss_backward_matches_reference_golden(golden_dir):
    """net.train(); nll, sd = net.loss(...); nll.backward() (trainer_NF_SID.py:102,116-126): NLL, all 125 parameter
    gradients and the BatchNorm buffers after the step, against the reference's own autograd on CPU."""
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    noise = torch.from_numpy(g['tr_noise']).cuda(); clean = torch.from_numpy(g['tr_clean']).cuda()
    for iso in (1600, 3000):
        net = _net(g).train()
        net.zero_grad()
        nll, sdz = net.loss(noise=noise, clean=clean, iso=torch.tensor(float(iso)).cuda())
        ref = g[f'tr_nll_iso{iso}']
        assert abs(float(nll.detach()) - ref[0]) < 2e-5 * abs(ref[0]) and abs(float(sdz) - ref[1]) < 1e-6
        nll.backward()
        named = dict(net.named_parameters())
        names = [k.split(':', 1)[1] for k in g.files if k.startswith(f'tr_grad_iso{iso}:')]
        assert len(names) == 125
        for k in names:
            assert named[k].grad is not None, k
            _grad_close(named[k].grad.cpu().numpy(), g[f'tr_grad_iso{iso}:' + k], k)
        assert named['model.0.cam_param'].grad is None                       # frozen in the reference (signal_dependant.py:25)
        sd = net.state_dict()
        for k in g.files:
            if k.startswith(f'tr_buf_iso{iso}:'):
                np.testing.assert_allclose(sd[k.split(':', 1)[1]].cpu().numpy(), g[k], rtol=2e-5, atol=1e-7, err_msg=k)


def test_train_mode_ragged_batch_vs_oracle(golden_dir):
    """Shapes that are not multiples of the 32x32 tile, several tiles per crop, batch of 5: against autograd on the oracle."""
    from oracle import noiseflow_torch as O
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    sd = {k: torch.from_numpy(g['sd:' + k]) for k in [str(x) for x in g['keys']]}
    gen = torch.Generator().manual_seed(11)
    for (B, H, W) in ((5, 48, 48), (2, 40, 72)):
        noise = torch.randn(B, 4, H, W, generator=gen) * 0.03
        clean = torch.rand(B, 4, H, W, generator=gen) * 0.02
        iso = 800.0
        rn, rs, rg, rb = O.loss_and_grads(sd, noise, clean, torch.tensor(iso))
        net = _net(g).train()
        nll, sdz = net(noise=noise.cuda(), clean=clean.cuda(), iso=iso, mode='loss')
        assert abs(float(nll) - float(rn)) < 2e-5 * abs(float(rn))
        nll.backward()
        named = dict(net.named_parameters())
        for k, v in rg.items():
            _grad_close(named[k].grad.cpu().numpy(), v.numpy(), k)
        cur = net.state_dict()
        for k, v in rb.items():
            np.testing.assert_allclose(cur[k].cpu().numpy(), v.numpy(), rtol=2e-5, atol=1e-7, err_msg=k)


def test_fit_steps_follow_oracle_trajectory(golden_dir):
    """Three Adam steps (lr 2e-3, runfiles/SonyA7S2/NoiseFlow.yml:56) on a fixed batch: the NLL sequence follows the
    CPU oracle's (autograd + torch.optim.Adam) and decreases."""
    from oracle import noiseflow_torch as O
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    noise = torch.from_numpy(g['tr_noise']); clean = torch.from_numpy(g['tr_clean'])
    sd = {k: torch.from_numpy(g['sd:' + k]).clone() for k in [str(x) for x in g['keys']]}
    leaves = {k: sd[k].clone().requires_grad_(True) for k in sd if O.trainable(k)}
    opt_ref = torch.optim.Adam(list(leaves.values()), lr=2e-3)
    ref = []
    for _ in range(3):
        cur = dict(sd); cur.update(leaves)
        for k in list(cur.keys()):                                         # the net.0 / net.3 aliases follow conv2d_1 / conv2d_2
            if '.net.0.' in k: cur[k] = cur[k.replace('.net.0.', '.conv2d_1.')]
            if '.net.3.' in k: cur[k] = cur[k.replace('.net.3.', '.conv2d_2.')]
        opt_ref.zero_grad()
        nll, _ = O.loss(cur, noise, clean, torch.tensor(1600.0), training=True)
        nll.backward(); opt_ref.step(); ref.append(float(nll))
        for k in sd:                                                       # carry the updated BatchNorm buffers
            if 'running_' in k or 'num_batches' in k: sd[k] = cur[k].detach()
    net = _net(g).train()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    got = []
    for _ in range(3):
        opt.zero_grad()
        nll, _ = net.loss(noise=noise.cuda(), clean=clean.cuda(), iso=1600.0)
        nll.backward(); opt.step(); got.append(float(nll))
    assert got[-1] < got[0]
    np.testing.assert_allclose(got, ref, rtol=5e-4)


def test_train_mode_edge_shapes_and_input_gradient(golden_dir):
    """One crop, odd non-square sizes (33x47: partial tiles in both directions), and the gradient w.r.t. the noise input
    (the backward's last dx), against autograd on the oracle.  (Non-square: both sides keep the reference's W*W log-det quirk.)"""
    from oracle import noiseflow_torch as O
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    sd = {k: torch.from_numpy(g['sd:' + k]) for k in [str(x) for x in g['keys']]}
    gen = torch.Generator().manual_seed(21)
    for (B, H, W) in ((1, 33, 47), (2, 16, 16)):
        noise = torch.randn(B, 4, H, W, generator=gen) * 0.03
        clean = torch.rand(B, 4, H, W, generator=gen) * 0.02
        cur = {k: v.clone() for k, v in sd.items()}
        xin = noise.clone().requires_grad_(True)
        rn, _ = O.loss(cur, xin, clean, torch.tensor(1000.0), training=True)
        (rgx,) = torch.autograd.grad(rn, xin)
        net = _net(g).train()
        xg = noise.cuda().requires_grad_(True)
        nll, _ = net.loss(noise=xg, clean=clean.cuda(), iso=1000.0)
        assert abs(float(nll.detach()) - float(rn)) < 2e-5 * abs(float(rn))
        nll.backward()
        err = (xg.grad.cpu() - rgx).abs().max()
        assert float(err) <= 2e-4 * float(rgx.abs().max()), (float(err), float(rgx.abs().max()))


def test_train_pair_c_abi_rejects_bad_arguments():
    import ctypes as C
    from pnnp_amd import _lib
    L = _lib.lib()
    t = torch.zeros(4 * 32 * 32, device='cuda')
    p = _lib.ptr(t)
    nul = C.c_void_p(0)
    bad = L.pnnp_nf_train_fwd_pair_f32(p, nul, nul, p, p, p, p, p, p, p, p, p, 0, 32, 32, _lib.stream())      # B = 0
    assert bad != 0
    bad = L.pnnp_nf_train_fwd_pair_f32(p, p, nul, p, p, p, p, p, p, p, p, p, 1, 32, 32, _lib.stream())        # clean without ab
    assert bad != 0
    z = torch.zeros_like(t)
    bad = L.pnnp_nf_train_fwd_pair_f32(p, nul, nul, p, p, p, p, p, p, p, _lib.ptr(z), p, 1, 32, 32, _lib.stream())   # z aliases x
    assert bad != 0
    assert L.pnnp_nf_train_tiles(3, 33, 64) == 3 * 2 * 2 and L.pnnp_nf_train_pblocks(1, 32, 33) == 2


def test_train_mode_sampling_uses_batch_statistics_like_the_lrid_trainer(golden_dir):
    """trainer_LRID.py:34-39 never calls .eval() on the proxy it samples from (:420-427): every BatchNorm of the couplings
    normalises with the statistics of the batch being sampled and moves its running buffers.  Golden: the reference in
    train() mode with an injected prior draw (2 ISOs): sample within the chained-coupling bar, buffers rtol 2e-5."""
    g = np.load(os.path.join(golden_dir, 'noiseflow.npz'))
    clean = torch.from_numpy(g['tr_clean']).cuda(); z = torch.from_numpy(g['ts_z']).cuda()
    for iso in (1600, 3000):
        net = _net(g).train()
        x = net.sample(clean=clean, iso=float(iso), z=z)
        _close_chain(x.cpu().numpy(), g[f'ts_out_iso{iso}'])
        sd = net.state_dict()
        for k in sd:
            if 'running_' in k:
                np.testing.assert_allclose(sd[k].cpu().numpy(), g[f'ts_buf_iso{iso}:' + k], rtol=2e-5, atol=1e-7, err_msg=k)
            elif 'num_batches' in k:
                assert int(sd[k]) == int(g[f'ts_buf_iso{iso}:' + k]), k
        # and it is not the eval-mode result
        xe = _net(g).sample(clean=clean, iso=float(iso), z=z)
        assert float((xe - x).abs().max()) > 1e-3 * float(x.abs().max())
