"""GPU parity of the whole denoiser path (through the drop-in module and the fused train
step) against the oracle and the golden vectors captured from the reference modules.
fp32 MFMA vs fp32 CPU: only the summation order differs; tolerances are stated inline."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _load(net, sd):
    net.load_state_dict({k: v.clone() for k, v in sd.items()})
    return net.cuda()


def _probe(a, idx):
    return np.asarray(a, np.float32).reshape(-1)[idx]


@pytest.mark.parametrize('res', [False, True])
def test_unet_nf8_golden_forward_backward_autograd(golden_dir, res):
    """Golden from the reference module (nf=8, 2x4x64x48): forward, loss, parameter gradients
    via autograd (loss.backward()) as the reference trainer uses the module."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    g = np.load(os.path.join(golden_dir, f'unet_nf8_res{int(res)}.npz'))
    sd = O.init_state(O.unet_param_shapes(nf=8), seed=42)
    net = _load(UNetSeeInDark(dict(nframes=1, res=res, nf=8, in_nc=4, out_nc=4)), sd)
    x = torch.from_numpy(g['x']).cuda(); t = torch.from_numpy(g['t']).cuda()
    with torch.no_grad():
        y0 = net(x)
    np.testing.assert_allclose(y0.cpu().numpy(), g['y'], rtol=1e-4, atol=2e-6)
    y = net(x)
    np.testing.assert_allclose(y.detach().cpu().numpy(), g['y'], rtol=1e-4, atol=2e-6)
    loss = torch.nn.functional.l1_loss(y.clamp(0, 1), t)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    loss.backward()
    for k, p in net.named_parameters():
        got = _probe(p.grad.cpu().numpy(), g['g:' + k + ':idx'])
        ref = g['g:' + k + ':val']
        tol = 1e-3 * np.abs(ref).max() + 1e-9
        assert np.abs(got - ref).max() <= tol, (k, np.abs(got - ref).max(), tol)
        s, l2 = g['g:' + k + ':sum']
        assert abs(float(p.grad.double().norm()) - l2) <= 1e-3 * l2 + 1e-9, k
    # reference-style optimisation: 3 torch.optim.Adam steps through autograd
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    for it in range(3):
        opt.zero_grad()
        pred = net(x)
        l = torch.nn.functional.l1_loss(pred.clamp(0, 1), t)
        l.backward(); opt.step()
        assert abs(l.item() - g['train_losses'][it, 0]) < 5e-6, (it, l.item(), g['train_losses'][it, 0])


def test_fused_train_step_matches_golden_and_oracle(golden_dir):
    """HipTrainStep (fused loss / backward / Adam on flat buffers) on the golden's fixed
    (noisy, clean) pair: losses of 3 steps vs the reference (<5e-6) and final weights vs the
    oracle's Adam (rtol 2e-3 on the update)."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.trainer import HipTrainStep
    g = np.load(os.path.join(golden_dir, 'unet_nf8_res0.npz'))
    sd = O.init_state(O.unet_param_shapes(nf=8), seed=42)
    net = _load(UNetSeeInDark(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)), sd)
    x = torch.from_numpy(g['x']).cuda(); t = torch.from_numpy(g['t']).cuda()
    ts = HipTrainStep(net, lr=1e-4, clip=0)
    sd_o = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in sd.items()}; v = {k: torch.zeros_like(vv) for k, vv in sd.items()}
    for it in range(3):
        lo = ts.step(t, noisy=x)
        l_ref, ps_ref, _ = O.train_step(sd_o, m, v, it + 1, torch.from_numpy(g['x']), torch.from_numpy(g['t']), lr=1e-4)
        assert abs(float(lo[0]) - g['train_losses'][it, 0]) < 5e-6
        assert abs(float(lo[0]) - l_ref) < 5e-6
        ps = HipTrainStep.psnr_from(lo, x[0].numel())
        assert abs(ps - g['train_losses'][it, 1]) < 2e-3 and abs(ps - ps_ref) < 2e-3
    for k, p in net.named_parameters():
        upd_ref = (sd_o[k] - sd[k]).numpy(); upd = p.detach().cpu().numpy() - sd[k].numpy()
        assert np.abs(upd - upd_ref).max() <= 0.05 * np.abs(upd_ref).max() + 1e-9, k   # Adam's g/sqrt(v) amplifies tiny grads
        got = _probe(p.detach().cpu().numpy(), np.linspace(0, p.numel() - 1, min(32, p.numel())).astype(np.int64))
        np.testing.assert_allclose(got, g['w3:' + k + ':val'], rtol=2e-3, atol=5e-6, err_msg=k)


def test_unet_nf32_full_crop_golden(golden_dir):
    """BASELINE configs 1/2: nf=32 UNet on one 4x512x512 crop vs the reference's output probes."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    g = np.load(os.path.join(golden_dir, 'unet_nf32_512.npz'))
    sd = O.init_state(O.unet_param_shapes(nf=32), seed=7)
    net = _load(UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)), sd).eval()
    x = torch.rand(1, 4, 512, 512, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        y = net(x)
    np.testing.assert_allclose(_probe(y.cpu().numpy(), g['idx']), g['val'], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(y.double().sum(dim=(0, 2, 3)).cpu().numpy(), g['chan_sum'], rtol=1e-4)


def test_unet_nf32_train_vs_oracle_and_eval_shapes():
    """nf=32, B=2 96x160 crops: loss + all parameter gradients vs the torch-fp32 oracle
    (rel. L2 error < 2e-3 per tensor: the L1 sign, max-pool argmax and LeakyReLU masks are
    discontinuous, so last-bit forward differences flip a few of them); then a ragged eval shape (not a multiple of 32 wide)."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    torch.manual_seed(3)
    net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    initialize_weights(net)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.cuda()
    x = torch.rand(2, 4, 96, 160); t = torch.rand(2, 4, 96, 160)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_ref = O.l1_clamp_loss(O.unet_forward(leaves, x), t)
    loss_ref.backward()
    ts = HipTrainStep(net, lr=0.0, clip=0)
    lo = ts.step(t.cuda(), noisy=x.cuda())
    assert abs(float(lo[0]) - loss_ref.item()) < 2e-6
    for k, p in net.named_parameters():
        got = net.engine.params.grad_view(k, p.shape).cpu()
        ref = leaves[k].grad
        rel = float((got - ref).norm() / (ref.norm() + 1e-12))
        assert rel < 2e-3, (k, rel)
    xe = torch.rand(1, 4, 48, 112)
    with torch.no_grad():
        ye = net(xe.cuda())
        yr = O.unet_forward(sd, xe)
    np.testing.assert_allclose(ye.cpu().numpy(), yr.numpy(), rtol=1e-4, atol=2e-6)


def test_train_step_with_sampler_runs_and_learns():
    """Whole hot path: sampler -> UNet -> L1 -> backward -> Adam on random crops; the loss must
    go down over a few steps on a fixed batch and stay finite; noise is reproducible."""
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    torch.manual_seed(0); np.random.seed(1997)
    net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda()
    ts = HipTrainStep(net, lr=1e-3, camera_type='SonyA7S2', noise_code='pr', clip=2)
    hr = (torch.rand(2, 4, 64, 64, device='cuda') * 0.5)
    plist = ts.sample_noise_params(2)
    n1, rows = ts.make_noisy(hr, plist)
    n2, _ = ts.make_noisy(hr, plist)
    assert torch.equal(n1, n2) and float(n1.max()) <= 1.0 and torch.isfinite(n1).all()
    losses = [float(ts.step(hr, rows=rows)[0]) for _ in range(12)]
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], losses


def test_rccl_bucketed_reducer_on_one_gpu():
    """The data-parallel step's collective path (side stream, events, RCCL all-reduce per bucket,
    Adam waiting on it) exercised with a 1-rank `nccl` group: must equal the plain step bit for bit."""
    import socket
    import torch.distributed as dist
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.trainer import HipTrainStep
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=torch.device('cuda', 0))
    try:
        sd = O.init_state(O.unet_param_shapes(nf=8), seed=11)
        x = torch.rand(2, 4, 64, 64).cuda(); t = torch.rand(2, 4, 64, 64).cuda()
        outs = []
        for force in (False, True):
            net = _load(UNetSeeInDark(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)), sd)
            ts = HipTrainStep(net, lr=1e-3, clip=0, force_reducer=force, bucket_bytes=64 << 10)
            losses = [float(ts.step(t, noisy=x)[0]) for _ in range(3)]
            if force:
                assert ts.reducer is not None and len(ts.reducer.buckets) > 3
            outs.append((losses, net.engine.params.flat.clone()))
        assert outs[0][0] == outs[1][0]
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        dist.destroy_process_group()


def test_ori_multiplies_prediction_by_ratio_before_the_loss():
    """dst.ori = True (trainer_SID.py:97-99): pred = pred * ratio; loss(pred.clamp(0,1), hr).  Loss and every parameter
    gradient vs autograd on the oracle with per-crop ratios; bars as in the nf=32 oracle test."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.trainer import HipTrainStep
    from pnnp_amd._lib import PnnpError
    sd = O.init_state_he(O.unet_param_shapes(nf=8), seed=5)
    net = _load(UNetSeeInDark(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)), sd)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(3, 4, 64, 80, generator=g) * 0.02; t = torch.rand(3, 4, 64, 80, generator=g)
    ratio = torch.tensor([120.0, 1.0, 37.5])
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_ref = O.l1_clamp_loss(O.unet_forward(leaves, x) * ratio.view(-1, 1, 1, 1), t)
    loss_ref.backward()
    ts = HipTrainStep(net, lr=0.0, clip=0, ori=True)
    with pytest.raises(PnnpError):
        ts.step(t.cuda(), noisy=x.cuda())                       # ori without a ratio must not silently drop the scaling
    loss_ori = float(ts.step(t.cuda(), noisy=x.cuda(), ratio=ratio.cuda())[0])          # (the returned tensor is a reused buffer)
    assert abs(loss_ori - loss_ref.item()) < 2e-6
    for k, p in net.named_parameters():
        got = net.engine.params.grad_view(k, p.shape).cpu()
        ref = leaves[k].grad
        assert float((got - ref).norm() / (ref.norm() + 1e-12)) < 2e-3, k
    # and it is not the un-scaled objective
    lo1 = HipTrainStep(net, lr=0.0, clip=0, ori=False).step(t.cuda(), noisy=x.cuda())
    assert abs(float(lo1[0]) - loss_ori) > 1e-3


def test_autograd_path_refuses_overwritten_activations():
    """The engine keeps ONE saved training forward per input shape (activation buffers are reused): a backward whose
    activations were overwritten -- second training forward, or a same-shape no_grad forward in between -- must raise
    instead of returning gradients of the wrong input; gradient accumulation step by step works."""
    from oracle import net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd._lib import PnnpError
    sd = O.init_state(O.unet_param_shapes(nf=8), seed=1)
    net = _load(UNetSeeInDark(dict(nframes=1, res=False, nf=8, in_nc=4, out_nc=4)), sd)
    a = torch.rand(1, 4, 32, 32).cuda(); b = torch.rand(1, 4, 32, 32).cuda()
    la = net(a).abs().mean(); lb = net(b).abs().mean()
    with pytest.raises(PnnpError):
        (la + lb).backward()
    la = net(a).abs().mean()
    with torch.no_grad():
        net(b)
    with pytest.raises(PnnpError):
        la.backward()
    # sequential accumulation is fine and equals the oracle's sum of gradients
    net.zero_grad()
    net(a).abs().mean().backward(); net(b).abs().mean().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    (O.unet_forward(leaves, a.cpu()).abs().mean() + O.unet_forward(leaves, b.cpu()).abs().mean()).backward()
    for k, p in net.named_parameters():
        assert float((p.grad.cpu() - leaves[k].grad).norm() / (leaves[k].grad.norm() + 1e-12)) < 2e-3, k


def test_kernel_families_follow_the_same_training_trajectory():
    """40 Adam steps (the reference's lr 1e-4) from the same initial weights on the same crops: the bf16x3 family (float32 operands
    split into three bf16 pieces), the Winograd and the direct fp32-MFMA families must trace the same loss curve.  Training amplifies
    any difference (Adam normalises every gradient component, each step's rounding feeds the next); the yardstick is therefore the
    direct family itself started from weights scaled by (1 + 1e-7), about one float32 ulp: measured divergence of the loss curves
    over 40 steps 4e-4 relative for that perturbation, 1.5e-4 for bf16x3 and 1.4e-4 for Winograd against direct.  Bars: a family
    stays within 3x the one-ulp divergence (+1e-5), the direct family re-run is bit-identical, and the run really trains
    (loss down by 30 %)."""
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.trainer import HipTrainStep
    from oracle import net_torch as O
    sd = O.init_state_he(O.unet_param_shapes(nf=32), seed=5, res_scale=0.5, head_scale=0.05, head_bias=0.1)
    g = torch.Generator(device='cuda').manual_seed(9)
    hr = torch.rand(2, 4, 128, 128, device='cuda', generator=g)
    noisy = (hr + 0.1 * torch.randn(2, 4, 128, 128, device='cuda', generator=g)).clamp(0, 1)

    def run(pol, perturb=0.0):
        net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
        net.load_state_dict({k: v.clone() * (1.0 + perturb) for k, v in sd.items()})
        net = net.cuda()
        net.engine.set_policy(**pol)
        ts = HipTrainStep(net, lr=1e-4, clip=0)
        return [float(ts.step(hr, noisy=noisy)[0]) for _ in range(40)]

    direct = dict(x3=False, wino=False, thin=False)
    ref = run(direct)
    assert run(direct) == ref                                   # deterministic: bit-identical re-run
    assert ref[-1] < 0.7 * ref[0], (ref[0], ref[-1])
    rel = lambda a: max(abs(p - q) / q for p, q in zip(a, ref))
    ulp = rel(run(direct, perturb=1e-7))
    for fam, pol in (('x3', dict(x3=True, wino=True, thin=True, h2=False)), ('wino', dict(x3=False, wino=True, thin=True)),
                     ('h2', dict(x3=True, wino=True, thin=True, h2=True))):
        d = rel(run(pol))
        print(f'{fam} vs direct over 40 steps: worst relative loss difference {d:.2e}; one-ulp perturbation of direct: {ulp:.2e} (loss {ref[0]:.4f} -> {ref[-1]:.4f})')
        assert d <= 3 * ulp + 1e-5, (fam, d, ulp)


@pytest.mark.parametrize('res', [False, True])
def test_fused_head_equals_conv_then_head_kernel(res):
    """Round 6: conv10_1 runs inside conv9_2's epilogue (csrc/conv_h2s.hip EK_HEAD, archs/Unet.py:93-94).  Against the two-kernel path
    (`set_policy(head_fused=False)`: conv9_2, then pnnp_head_fwd_f32) on a frame whose width does not fill the 32-pixel tiles: the output
    planes agree to float32 summation order (the head's 32-term sums: 1e-6 of the largest output), the stored 32-channel map, its sign bits and amax
    slot are BIT-identical in a training forward, the parameter gradients of a backward pass agree, and an eval forward (which stores no
    32-channel map at all) gives the training forward's output bit for bit."""
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    torch.manual_seed(3)
    net = UNetSeeInDark(dict(nframes=1, res=res, nf=32, in_nc=4, out_nc=4)); initialize_weights(net)
    with torch.no_grad():
        net.conv10_1.bias.uniform_(-0.1, 0.1)
    net = net.cuda()
    e = net.engine
    g = torch.Generator(device='cuda').manual_seed(4)
    x = torch.rand(2, 4, 48, 80, device='cuda', generator=g)
    go = torch.randn(2, 48, 80, 8, device='cuda', generator=g); go[..., 4:] = 0

    def run(fused):
        # (split-K off: this small frame's EVAL forward would otherwise cut its deep layers along K -- another partition of the same sums -- and the
        #  eval-vs-training comparison below is about the head, bit for bit)
        e.set_policy(head_fused=fused, splitk=False)
        out = e.forward(x, train=True).clone()
        a = e.saved[0]
        c9, bits = a['c9'].clone(), a['bits:conv9_2'].clone()
        e.backward(go.clone())
        with torch.no_grad():
            ev = net(x).clone()
        return out, c9, bits, e.params.grad.clone(), ev
    o1, c1, b1, g1, ev1 = run(True)
    o0, c0, b0, g0, ev0 = run(False)
    assert e._head_fusable() is False
    scale = float(o0.abs().max())
    print(f'fused head vs conv + head kernel: max |diff| / max |out| {float((o1 - o0).abs().max()) / scale:.2e}')
    assert float((o1 - o0).abs().max()) <= 1e-6 * scale
    assert torch.equal(c1, c0) and torch.equal(b1, b0)
    assert float((g1 - g0).norm() / g0.norm()) < 1e-6
    assert torch.equal(ev1, o1)                                  # eval forward (no 32-channel map stored) == training forward
    assert float((ev0 - o0).abs().max()) <= 1e-6 * scale
