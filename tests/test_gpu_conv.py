"""GPU parity of the convolution-stack kernels (through the C ABI) against plain
torch-fp32 CPU references of the same op.  fp32 MFMA accumulates in a different order than
the CPU, so the bar is rtol 1e-4 / atol 1e-5 scaled by the magnitude of the sums."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, ref, rtol=1e-4, atol=1e-5, what=''):
    got = got.detach().cpu().float(); ref = ref.detach().cpu().float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(1.0, float(ref.abs().max()))
    err = (got - ref).abs()
    bad = err > atol * scale + rtol * ref.abs()
    assert not bad.any(), (what, float(err.max()), int(bad.sum()), got.numel(), np.argwhere(bad.numpy())[:5].tolist())


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


CASES = [  # B, H, W, C1, C2 (0 = no concat), Cout
    (1, 8, 32, 8, 0, 32), (2, 16, 48, 32, 0, 32), (1, 12, 40, 16, 0, 64), (1, 6, 70, 64, 0, 128),
    (1, 9, 33, 8, 8, 16), (2, 8, 32, 32, 32, 32), (1, 8, 36, 64, 64, 64), (1, 5, 17, 128, 128, 128),
    (1, 4, 4, 256, 0, 256), (1, 34, 66, 8, 0, 8), (1, 8, 32, 8, 0, 4), (1, 16, 32, 32, 0, 256),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('taps', [9, 1])
def test_conv_fwd(case, taps):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    k = 3 if taps == 9 else 1
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, k, k, seed=3, scale=0.2); b = _rand(Co, seed=4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    for act in (0, 1, 2):
        ref = F.conv2d(xin, w, b, padding=k // 2)
        ref = F.leaky_relu(ref, 0.2) if act == 1 else (F.relu(ref) if act == 2 else ref)
        wd = w.cuda()
        fwd = torch.empty(w.numel(), device='cuda')
        ops.pack_conv_weight(wd, fwd, None)
        y = torch.full((B, H, W, Co), float('nan'), device='cuda')
        ops.conv_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, fwd, b.cuda(), y, Co, taps, act)
        close(nchw(y), ref, what=f'fwd {case} taps{taps} act{act}')
    # residual add before the activation
    r = _rand(B, Co, H, W, seed=9)
    y = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, fwd, b.cuda(), y, Co, taps, 2, residual=nhwc(r).cuda())
    close(nchw(y), F.relu(F.conv2d(xin, w, b, padding=k // 2) + r), what='residual')


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('taps', [9, 1])
def test_conv_bwd_data(case, taps):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    if Co % 8:
        pytest.skip('K side needs channel multiples of 8 (the 4-channel output travels padded to 8)')
    k = 3 if taps == 9 else 1
    w = _rand(Co, C1 + C2, k, k, seed=3, scale=0.2)
    g = _rand(B, Co, H, W, seed=5)
    xin = _rand(B, C1 + C2, H, W, seed=6).requires_grad_(True)
    F.conv2d(xin, w, None, padding=k // 2).backward(g)
    ref = xin.grad
    wd = w.cuda()
    dg = torch.empty(w.numel(), device='cuda')
    ops.pack_conv_weight(wd, None, dg)
    m1 = _rand(B, C1, H, W, seed=7); m2 = _rand(B, max(C2, 1), H, W, seed=8)
    d1 = torch.full((B, H, W, C1), float('nan'), device='cuda')
    d2 = torch.full((B, H, W, C2), float('nan'), device='cuda') if C2 else None
    ops.conv_bwd_data(nhwc(g).cuda(), dg, d1, dx2=d2, taps=taps)
    close(nchw(d1), ref[:, :C1], what=f'dgrad {case}')
    if C2:
        close(nchw(d2), ref[:, C1:], what=f'dgrad2 {case}')
    # masks (LeakyReLU' on dst1, ReLU' on dst2) and accumulation into dst2
    base2 = _rand(B, max(C2, 1), H, W, seed=10)
    d1 = torch.empty((B, H, W, C1), device='cuda')
    d2 = nhwc(base2).cuda().clone() if C2 else None
    ops.conv_bwd_data(nhwc(g).cuda(), dg, d1, mask1=nhwc(m1).cuda(), mode1=1, dx2=d2,
                      mask2=nhwc(m2).cuda() if C2 else None, mode2=2, accum2=1, taps=taps)
    close(nchw(d1), ref[:, :C1] * torch.where(m1 > 0, 1.0, 0.2), what='mask1')
    if C2:
        close(nchw(d2), base2 + ref[:, C1:] * (m2 > 0).float(), what='mask2+accum')


@pytest.mark.parametrize('case', CASES + [(3, 32, 64, 32, 0, 32), (2, 16, 32, 64, 64, 64)])
@pytest.mark.parametrize('taps', [9, 1])
def test_conv_bwd_weight(case, taps):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    k = 3 if taps == 9 else 1
    w = _rand(Co, C1 + C2, k, k, seed=3, scale=0.2).requires_grad_(True)
    b = _rand(Co, seed=4).requires_grad_(True)
    g = _rand(B, Co, H, W, seed=5)
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    F.conv2d(xin, w, b, padding=k // 2).backward(g)
    ws = torch.empty(ops.wgrad_workspace_floats(B, H, W, Co, C1 + C2, taps), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, taps, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'bgrad {case}')
    ops.conv_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, taps, ws, accumulate=1)
    close(dW, 2 * w.grad, rtol=2e-4, atol=2e-5, what='wgrad accumulate')


def test_padded_boundary_channels():
    """4-channel network input / output travel as zero-padded 8-channel NHWC tensors."""
    from pnnp_amd import ops
    B, H, W, Co = 2, 16, 32, 32
    x = _rand(B, 4, H, W, seed=1); w = _rand(Co, 4, 3, 3, seed=2, scale=0.3); b = _rand(Co, seed=3)
    x8 = torch.empty((B, H, W, 8), device='cuda'); ops.nchw_to_nhwc(x.cuda(), x8, 8)
    assert torch.equal(x8[..., :4].cpu(), nhwc(x)) and float(x8[..., 4:].abs().max()) == 0
    fwd = torch.empty(9 * 8 * Co, device='cuda'); ops.pack_conv_weight(w.cuda(), fwd, None, cin_pad=8)
    y = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_fwd(x8, None, fwd, b.cuda(), y, Co, 9, 1)
    close(nchw(y), F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2), what='conv1_1')
    # wgrad against the padded input: N = 4 real channels
    g = _rand(B, Co, H, W, seed=4)
    wr = w.clone().requires_grad_(True); F.conv2d(x, wr, None, padding=1).backward(g)
    ws = torch.empty(ops.wgrad_workspace_floats(B, H, W, Co, 4, 9), device='cuda')
    dW = torch.empty(w.shape, device='cuda'); db = torch.empty(Co, device='cuda')
    ops.conv_bwd_weight(nhwc(g).cuda(), Co, x8, 4, None, dW, db, 9, ws)
    close(dW, wr.grad, rtol=2e-4, atol=2e-5, what='conv1_1 wgrad')
    # 1x1 head 32 -> 4 : forward, dgrad from the padded gradient, wgrad with M = 4 of 8
    w10 = _rand(4, 32, 1, 1, seed=5, scale=0.3); b10 = _rand(4, seed=6)
    c9 = _rand(B, 32, H, W, seed=7).requires_grad_(True)
    w10r = w10.clone().requires_grad_(True); b10r = b10.clone().requires_grad_(True)
    out = F.conv2d(c9, w10r, b10r); g4 = _rand(B, 4, H, W, seed=8); out.backward(g4)
    f10 = torch.empty(32 * 4, device='cuda'); d10 = torch.empty(8 * 32, device='cuda')
    ops.pack_conv_weight(w10.cuda(), f10, d10, cout_pad=8)
    o = torch.empty((B, H, W, 4), device='cuda')
    ops.conv_fwd(nhwc(c9.detach()).cuda(), None, f10, b10.cuda(), o, 4, 1, 0)
    close(nchw(o), out, what='conv10 fwd')
    g8 = torch.empty((B, H, W, 8), device='cuda'); ops.nchw_to_nhwc(g4.cuda(), g8, 8)
    dx = torch.empty((B, H, W, 32), device='cuda')
    ops.conv_bwd_data(g8, d10, dx, taps=1)
    close(nchw(dx), c9.grad, what='conv10 dgrad')
    ws = torch.empty(ops.wgrad_workspace_floats(B, H, W, 4, 32, 1), device='cuda')
    dW = torch.empty(w10.shape, device='cuda'); db = torch.empty(4, device='cuda')
    ops.conv_bwd_weight(g8, 4, nhwc(c9.detach()).cuda(), 32, None, dW, db, 1, ws)
    close(dW, w10r.grad, rtol=2e-4, atol=2e-5, what='conv10 wgrad'); close(db, b10r.grad, rtol=2e-4, atol=2e-5, what='conv10 bgrad')
    res = _rand(B, 4, H, W, seed=9)
    outn = torch.empty((B, 4, H, W), device='cuda'); ops.nhwc_to_nchw(o, outn, residual=res.cuda())
    close(outn, out + res, what='nhwc_to_nchw + residual')


@pytest.mark.parametrize('case', [(1, 4, 32, 16, 8), (2, 8, 16, 64, 32), (1, 3, 35, 128, 64), (1, 2, 2, 512, 256), (1, 16, 32, 64, 32)])
def test_convt(case):
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    x = _rand(B, Ci, H, W, seed=1).requires_grad_(True)
    w = _rand(Ci, Co, 2, 2, seed=2, scale=0.2).requires_grad_(True); b = _rand(Co, seed=3).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=2)
    g = _rand(B, Co, 2 * H, 2 * W, seed=4)
    y.backward(g)
    f = torch.empty(w.numel(), device='cuda'); d = torch.empty(w.numel(), device='cuda')
    ops.pack_convt_weight(w.detach().cuda(), f, d)
    yo = torch.full((B, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    ops.convt_fwd(nhwc(x.detach()).cuda(), f, b.detach().cuda(), yo, Co)
    close(nchw(yo), y, what=f'convT fwd {case}')
    m = _rand(B, Ci, H, W, seed=5)
    dx = torch.full((B, H, W, Ci), float('nan'), device='cuda')
    ops.convt_bwd_data(nhwc(g).cuda(), d, dx, mask=nhwc(m).cuda(), mode=1)
    close(nchw(dx), x.grad * torch.where(m > 0, 1.0, 0.2), what=f'convT dgrad {case}')
    ws = torch.empty(max(ops.wgrad_workspace_floats(B, H, W, Ci, Co, 4), 1024 * Co), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.convt_bwd_weight(nhwc(x.detach()).cuda(), nhwc(g).cuda(), dW, ws)
    ops.channel_sum(nhwc(g).cuda(), db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'convT wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'convT bgrad {case}')
    # the bias gradient fused into the weight-gradient kernel (what the engines use), with and without accumulation
    dW2 = torch.full(w.shape, float('nan'), device='cuda'); db2 = torch.full((Co,), float('nan'), device='cuda')
    ops.convt_bwd_weight(nhwc(x.detach()).cuda(), nhwc(g).cuda(), dW2, ws, dbias=db2)
    close(dW2, w.grad, rtol=2e-4, atol=2e-5, what=f'convT wgrad+bias {case}')
    close(db2, b.grad, rtol=2e-4, atol=2e-5, what=f'convT fused bgrad {case}')
    ops.convt_bwd_weight(nhwc(x.detach()).cuda(), nhwc(g).cuda(), dW2, ws, accumulate=1, dbias=db2)
    close(db2, 2 * b.grad, rtol=2e-4, atol=4e-5, what=f'convT fused bgrad accumulate {case}')


@pytest.mark.parametrize('case', [(1, 8, 32, 8, 16), (2, 16, 64, 32, 64), (1, 12, 40, 64, 128), (1, 6, 70, 128, 256),
                                  (1, 4, 4, 256, 512), (1, 34, 66, 16, 8)])
def test_conv3x3_stride2(case):
    """ResUnet down-sampling conv (archs/modules.py:130-138): forward, backward-data (4 parity-class
    GEMMs), backward-weight, vs F.conv2d(stride=2, padding=1)."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    x = _rand(B, Ci, H, W, seed=1).requires_grad_(True)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=0.2).requires_grad_(True); b = _rand(Co, seed=3).requires_grad_(True)
    y = F.conv2d(x, w, b, stride=2, padding=1)
    g = _rand(B, Co, H // 2, W // 2, seed=4)
    y.backward(g)
    f = torch.empty(w.numel(), device='cuda'); d = torch.empty(w.numel(), device='cuda')
    ops.pack_conv_weight(w.detach().cuda(), f, None)
    ops.pack_conv_s2_dgrad(w.detach().cuda(), d)
    yo = torch.full((B, H // 2, W // 2, Co), float('nan'), device='cuda')
    ops.conv_s2_fwd(nhwc(x.detach()).cuda(), f, b.detach().cuda(), yo, Co)
    close(nchw(yo), y, what=f's2 fwd {case}')
    dx = torch.full((B, H, W, Ci), float('nan'), device='cuda')
    ops.conv_s2_bwd_data(nhwc(g).cuda(), d, dx)
    close(nchw(dx), x.grad, what=f's2 dgrad {case}')
    m = _rand(B, Ci, H, W, seed=5); base = _rand(B, Ci, H, W, seed=6)
    dx = nhwc(base).cuda().clone()
    ops.conv_s2_bwd_data(nhwc(g).cuda(), d, dx, mask=nhwc(m).cuda(), mode=2, accum=1)
    close(nchw(dx), base + x.grad * (m > 0).float(), what='s2 dgrad mask+accum')
    ws = torch.empty(ops.wgrad_workspace_floats(B, H // 2, W // 2, Co, Ci, 18), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_s2_bwd_weight(nhwc(g).cuda(), nhwc(x.detach()).cuda(), dW, db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f's2 wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f's2 bgrad {case}')


def test_maxpool_loss_adam():
    from pnnp_amd import ops
    B, H, W, Cc = 2, 12, 20, 16
    x = _rand(B, Cc, H, W, seed=1)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = x[0, 0, 1, 0] = x[0, 0, 1, 1] = 0.5      # tie: first max wins
    xr = x.clone().requires_grad_(True)
    act = F.leaky_relu(xr, 0.2)
    y = F.max_pool2d(act, 2)
    gy = _rand(B, Cc, H // 2, W // 2, seed=2)
    y.backward(gy)
    actd = nhwc(act.detach()).cuda()
    yo = torch.empty((B, H // 2, W // 2, Cc), device='cuda'); ops.maxpool_fwd(actd, yo)
    assert torch.equal(nchw(yo).cpu(), y.detach())
    base = _rand(B, Cc, H, W, seed=3)
    gx = nhwc(base).cuda().clone()
    ops.maxpool_bwd(actd, nhwc(gy).cuda(), gx, 1, 1)
    close(nchw(gx), base + xr.grad, what='maxpool bwd (+lrelu derivative, accumulate)')
    # the same through the one-byte argmax / sign codes written by the forward pass (no second read of the activation)
    codes = torch.empty((B, H // 2, W // 2, Cc), dtype=torch.uint8, device='cuda')
    yo2 = torch.empty_like(yo); ops.maxpool_fwd(actd, yo2, codes=codes)
    assert torch.equal(yo2, yo)
    for mode, accum in ((1, 1), (1, 0), (2, 1), (0, 0)):
        ga = nhwc(base).cuda().clone(); gb_ = nhwc(base).cuda().clone()
        ops.maxpool_bwd(actd, nhwc(gy).cuda(), ga, mode, accum)
        ops.maxpool_bwd(actd, nhwc(gy).cuda(), gb_, mode, accum, codes=codes)
        assert torch.equal(ga, gb_), (mode, accum)
    # L1(clamp) loss + gradient + per-crop SSE
    pred = (_rand(3, 4, 16, 24, seed=4) * 0.8 + 0.5).requires_grad_(True); hr = _rand(3, 4, 16, 24, seed=5) * 0.5 + 0.5
    loss = F.l1_loss(pred.clamp(0, 1), hr); loss.backward()
    g8 = torch.full((3, 16, 24, 8), float('nan'), device='cuda'); lo = torch.empty(4, device='cuda'); ws = torch.empty(128 * 3, device='cuda')
    ops.l1_clamp_loss(pred.detach().cuda(), hr.cuda(), g8, lo, ws)
    assert abs(float(lo[0]) - loss.item()) < 1e-6
    close(nchw(g8[..., :4].contiguous()), pred.grad, rtol=1e-6, atol=1e-9, what='l1 grad')
    assert float(g8[..., 4:].abs().max()) == 0
    sse = ((pred.detach().clamp(0, 1) - hr.clamp(0, 1)) ** 2).sum(dim=(1, 2, 3))
    close(lo[1:], sse, rtol=1e-5, what='sse')
    # Adam vs torch.optim.Adam, 3 steps
    p = _rand(1003, seed=6); gs = [_rand(1003, seed=7 + i) * 0.1 for i in range(3)]
    pt = p.clone().requires_grad_(True); opt = torch.optim.Adam([pt], lr=1e-3)
    n = 1004
    pd = torch.zeros(n, device='cuda'); pd[:1003] = p.cuda(); m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    for i, gk in enumerate(gs):
        pt.grad = gk.clone(); opt.step()
        gd = torch.zeros(n, device='cuda'); gd[:1003] = gk.cuda() * 2
        ops.adam_step(pd, gd, m, v, 1e-3, i + 1, grad_scale=0.5)
    close(pd[:1003], pt.detach(), rtol=1e-5, atol=1e-7, what='adam')
