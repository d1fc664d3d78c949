"""Pointwise layers on the fp16x2 scheme (csrc/gemm_h2s.hip): ConvTranspose2d(2, 2) forward / backward-data against float64 and against the exact
bf16x3 kernels (csrc/gemm_x3s.hip), at the bars of tests/test_gpu_h2.py (the 3x3 kernels of the same scheme)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _slot(t):
    from pnnp_amd import ops
    return ops.amax(t, torch.zeros(1, dtype=torch.int32, device=t.device))


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 16, 32, 64, 32), (1, 8, 40, 512, 256), (3, 13, 21, 128, 64), (2, 32, 32, 256, 128), (2, 20, 44, 32, 32)])      # (last: 32-column backward-data tile, round 6)
def test_convt_h2_forward_and_backward_data_vs_float64_and_bf16x3(B, H, W, Cin, Cout):
    from pnnp_amd import ops
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + Cin)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    w = torch.randn(Cin, Cout, 2, 2, device='cuda', generator=g) * 0.05
    bias = torch.randn(Cout, device='cuda', generator=g)
    assert ops.gemm_h2_supported(Cin, 4 * Cout) and ops.gemm_h2_supported(Cout, Cin)
    jobs = ops.PackJobs()
    wf = torch.zeros(ops.h2mat_bytes(Cin, 4 * Cout), dtype=torch.uint8, device='cuda')
    wd = torch.zeros(ops.h2mat_bytes(4 * Cout, Cin), dtype=torch.uint8, device='cuda')
    sw = jobs.add_h2_convt(w, wf, wd)
    xf = torch.zeros(ops.x3mat_bytes(Cin, 4 * Cout), dtype=torch.uint8, device='cuda')
    xd = torch.zeros(ops.x3mat_bytes(4 * Cout, Cin), dtype=torch.uint8, device='cuda')
    jobs.add_x3_convt(w, xf, xd)
    jobs.run()
    # forward
    y = torch.empty(B, 2 * H, 2 * W, Cout, device='cuda'); y3 = torch.empty_like(y)
    sy = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.convt_h2_fwd(x, _slot(x), wf, sw, bias, y, Cout, amax_y=sy)
    ops.convt_x3_fwd(x, xf, bias, y3, Cout)
    ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=2).permute(0, 2, 3, 1)
    e_h2, e_x3 = _rel(y, ref), _rel(y3, ref)
    print(f'convT fwd {Cin}->{Cout}: rel L2 vs float64 h2 {e_h2:.2e}, bf16x3 {e_x3:.2e}')
    assert e_h2 < 6e-7 and e_h2 < 3 * e_x3 + 1e-7
    amax = torch.tensor(sy.item(), dtype=torch.int32).view(torch.float32).item()
    assert float(y.abs().max()) <= amax <= 1.0001 * float(y.abs().max())
    # backward-data with a LeakyReLU' mask
    gy = torch.randn(B, 2 * H, 2 * W, Cout, device='cuda', generator=g)
    mask = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    dx = torch.empty(B, H, W, Cin, device='cuda'); dx3 = torch.empty_like(dx)
    sd = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.convt_h2_bwd_data(gy, _slot(gy), wd, sw, dx, mask=mask, mode=1, amax_dx=sd)
    ops.convt_x3_bwd_data(gy, xd, dx3, mask=mask, mode=1)
    refd = F.conv2d(gy.permute(0, 3, 1, 2).double(), w.double(), stride=2).permute(0, 2, 3, 1)
    refd = torch.where(mask > 0, refd, 0.2 * refd)
    e_h2, e_x3 = _rel(dx, refd), _rel(dx3, refd)
    print(f'convT dgrad {Cout}->{Cin}: rel L2 vs float64 h2 {e_h2:.2e}, bf16x3 {e_x3:.2e}')
    assert e_h2 < 8e-7 and e_h2 < 3 * e_x3 + 1e-7
    amax = torch.tensor(sd.item(), dtype=torch.int32).view(torch.float32).item()
    assert float(dx.abs().max()) <= amax <= 1.0001 * float(dx.abs().max())


def test_convt_h2_scales_and_refusals():
    """Tensor magnitudes far from 1 (the per-tensor power-of-two scale), and shapes the kernel refuses (the engines then keep bf16x3)."""
    from pnnp_amd import ops
    g = torch.Generator(device='cuda').manual_seed(7)
    B, H, W, Cin, Cout = 1, 16, 32, 64, 32
    for sx, sw_ in ((1e-20, 1.0), (1e12, 1e-6)):
        x = torch.randn(B, H, W, Cin, device='cuda', generator=g) * sx
        w = torch.randn(Cin, Cout, 2, 2, device='cuda', generator=g) * sw_
        jobs = ops.PackJobs()
        wf = torch.zeros(ops.h2mat_bytes(Cin, 4 * Cout), dtype=torch.uint8, device='cuda')
        sw = jobs.add_h2_convt(w, wf, None); jobs.run()
        y = torch.empty(B, 2 * H, 2 * W, Cout, device='cuda')
        ops.convt_h2_fwd(x, _slot(x), wf, sw, None, y, Cout)
        ref = F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride=2).permute(0, 2, 3, 1)
        assert _rel(y, ref) < 6e-7, (sx, sw_, _rel(y, ref))
    assert not ops.gemm_h2_supported(48, 128) and not ops.gemm_h2_supported(64, 48) and ops.gemm_h2_supported(64, 32)      # (round 6: 32-column tiles)


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 32, 64, 64, 128), (1, 24, 40, 128, 256), (2, 48, 80, 32, 64)])      # (last: ResUnet's pool1 -- backward-data on the 32-column tile)
def test_conv_s2_h2_forward_and_backward_data_vs_float64_and_bf16x3(B, H, W, Cin, Cout):
    """Conv2d 3x3 stride 2 (ResUnet's pool layers): forward as 9 strided taps, backward-data as four parity-class GEMMs accumulating into dx."""
    from pnnp_amd import ops
    g = torch.Generator(device='cuda').manual_seed(B * 77 + Cin)
    x = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda', generator=g) * 0.05
    bias = torch.randn(Cout, device='cuda', generator=g)
    jobs = ops.PackJobs()
    wf = torch.zeros(ops.h2mat_bytes(9 * Cin, Cout), dtype=torch.uint8, device='cuda'); wd = torch.zeros(9 * ops.h2mat_bytes(Cout, Cin), dtype=torch.uint8, device='cuda')
    sw = jobs.add_h2_s2(w, wf, wd)
    xf = torch.zeros(ops.x3mat_bytes(9 * Cin, Cout), dtype=torch.uint8, device='cuda'); xd = torch.zeros(9 * ops.x3mat_bytes(Cout, Cin), dtype=torch.uint8, device='cuda')
    jobs.add_x3_s2(w, xf, xd); jobs.run()
    y = torch.empty(B, H // 2, W // 2, Cout, device='cuda'); y3 = torch.empty_like(y)
    ops.conv_s2_h2_fwd(x, _slot(x), wf, sw, bias, y, Cout, 0)
    ops.conv_s2_x3_fwd(x, xf, bias, y3, Cout)
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), bias.double(), stride=2, padding=1).permute(0, 2, 3, 1)
    e_h2, e_x3 = _rel(y, ref), _rel(y3, ref)
    print(f's2 fwd {Cin}->{Cout}: rel L2 vs float64 h2 {e_h2:.2e}, bf16x3 {e_x3:.2e}')
    assert e_h2 < 6e-7 and e_h2 < 3 * e_x3 + 1e-7
    gy = torch.randn(B, H // 2, W // 2, Cout, device='cuda', generator=g)
    base = torch.randn(B, H, W, Cin, device='cuda', generator=g)
    dx = base.clone(); dx3 = base.clone()
    sd = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.conv_s2_h2_bwd_data(gy, _slot(gy), wd, sw, dx, accum=1, amax_dx=sd)
    ops.conv_s2_x3_bwd_data(gy, xd, dx3, accum=1)
    refd = base.double() + F.conv_transpose2d(gy.permute(0, 3, 1, 2).double(), w.double(), stride=2, padding=1, output_padding=1).permute(0, 2, 3, 1)
    e_h2, e_x3 = _rel(dx, refd), _rel(dx3, refd)
    print(f's2 dgrad {Cout}->{Cin}: rel L2 vs float64 h2 {e_h2:.2e}, bf16x3 {e_x3:.2e}')
    assert e_h2 < 8e-7 and e_h2 < 3 * e_x3 + 1e-7
    amax = torch.tensor(sd.item(), dtype=torch.int32).view(torch.float32).item()
    assert float(dx.abs().max()) <= amax <= 1.0001 * float(dx.abs().max())


@pytest.mark.parametrize('C', [64, 32])                              # (32: ResUnet's sc9 -- forward on the 32-column tile, round 6)
def test_conv1x1_h2_two_inputs_and_two_accumulating_outputs(C):
    """ResidualBlock's 1x1 shortcut on cat([up, skip]): forward over two K segments with their own amax slots, backward-data ACCUMULATING into the
    two gradients (the general epilogue), the first one's slot raised to max |sum|."""
    from pnnp_amd import ops
    g = torch.Generator(device='cuda').manual_seed(11)
    B, H, W = 2, 24, 40
    u = torch.randn(B, H, W, C, device='cuda', generator=g); skip = torch.randn(B, H, W, C, device='cuda', generator=g) * 3
    w = torch.randn(C, 2 * C, 1, 1, device='cuda', generator=g) * 0.1
    jobs = ops.PackJobs()
    wf = torch.zeros(ops.h2mat_bytes(2 * C, C), dtype=torch.uint8, device='cuda'); wd = torch.zeros(ops.h2mat_bytes(C, 2 * C), dtype=torch.uint8, device='cuda')
    sw = jobs.add_h2_1x1(w, wf, wd); jobs.run()
    y = torch.empty(B, H, W, C, device='cuda')
    ops.conv1x1_h2_fwd(u, _slot(u), skip, _slot(skip), wf, sw, None, y, C, 0)
    ref = F.conv2d(torch.cat([u, skip], 3).permute(0, 3, 1, 2).double(), w.double()).permute(0, 2, 3, 1)
    assert _rel(y, ref) < 6e-7, _rel(y, ref)
    gy = torch.randn(B, H, W, C, device='cuda', generator=g)
    b1 = torch.randn(B, H, W, C, device='cuda', generator=g); b2 = torch.randn(B, H, W, C, device='cuda', generator=g)
    d1, d2 = b1.clone(), b2.clone()
    s1 = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.conv1x1_h2_bwd_data(gy, _slot(gy), wd, sw, d1, accum1=1, amax_dx1=s1, dx2=d2, accum2=1)
    full = torch.einsum('bhwo,oi->bhwi', gy.double(), w[:, :, 0, 0].double())
    assert _rel(d1, b1.double() + full[..., :C]) < 8e-7 and _rel(d2, b2.double() + full[..., C:]) < 8e-7
    amax = torch.tensor(s1.item(), dtype=torch.int32).view(torch.float32).item()
    assert float(d1.abs().max()) <= amax <= 1.0001 * float(d1.abs().max())


def _wgrad_check(name, got, ref, e_cap=1.2e-6):
    e = _rel(got, ref)
    print(f'{name}: rel L2 vs float64 {e:.2e}')
    assert e < e_cap, (name, e)


def test_pointwise_weight_gradients_h2_vs_float64():
    """Backward-weights of the three pointwise geometries on csrc/wgrad_h2g.hip (two fp16 pieces per operand, three MFMAs per block and tap):
    dW and dbias against float64, at the bar of the bf16x3 kernels' test (K = 10^4 .. 10^5 pixels per weight)."""
    from pnnp_amd import ops
    g = torch.Generator(device='cuda').manual_seed(21)
    # ConvTranspose2d(128 -> 64): x [B,H,W,128], gy [B,2H,2W,64]
    B, H, W, Ci, Co = 2, 32, 64, 128, 64
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g); gy = torch.randn(B, 2 * H, 2 * W, Co, device='cuda', generator=g) * 1e-3
    assert ops.x3g_wgrad_supported(ops.X3G_CT, Ci, Co)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, H, W, Ci, Co), device='cuda')
    dW = torch.empty(Ci, Co, 2, 2, device='cuda'); db = torch.empty(Co, device='cuda')
    ops.convt_h2_bwd_weight(x, _slot(x), gy, _slot(gy), dW, ws, dbias=db)
    ref = torch.einsum('bhwi,bhawco->ioac', x.double(), gy.double().reshape(B, H, 2, W, 2, Co))
    _wgrad_check('convT dW', dW, ref); _wgrad_check('convT dbias', db, gy.double().sum((0, 1, 2)))
    dW3 = torch.empty_like(dW); ops.convt_x3_bwd_weight(x, gy, dW3, ws)
    assert _rel(dW, ref) < 3 * _rel(dW3, ref) + 1e-7
    # Conv2d 3x3 stride 2 (64 -> 128)
    B, H, W, Ci, Co = 2, 32, 64, 64, 128
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g); gy = torch.randn(B, H // 2, W // 2, Co, device='cuda', generator=g)
    assert ops.x3g_wgrad_supported(ops.X3G_S2, Co, Ci)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_S2, B, H // 2, W // 2, Co, Ci), device='cuda')
    dW = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
    ops.conv_s2_h2_bwd_weight(gy, _slot(gy), x, _slot(x), dW, db, ws)
    xd = x.permute(0, 3, 1, 2).double().requires_grad_(False)
    wref = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
    F.conv2d(xd, wref, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).double())
    _wgrad_check('s2 dW', dW, wref.grad); _wgrad_check('s2 dbias', db, gy.double().sum((0, 1, 2)))
    # ... and the 64 x 32 tile only the fp16x2 kernel has (ResUnet's pool1: 32 -> 64 at 512^2)
    B, H, W, Ci, Co = 2, 64, 96, 32, 64
    x = torch.randn(B, H, W, Ci, device='cuda', generator=g); gy = torch.randn(B, H // 2, W // 2, Co, device='cuda', generator=g)
    assert ops.h2g_wgrad_supported(ops.X3G_S2, Co, Ci) and not ops.x3g_wgrad_supported(ops.X3G_S2, Co, Ci)
    ws = torch.empty(ops.h2g_wgrad_workspace_floats(ops.X3G_S2, B, H // 2, W // 2, Co, Ci), device='cuda')
    dW = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
    ops.conv_s2_h2_bwd_weight(gy, _slot(gy), x, _slot(x), dW, db, ws)
    wref = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
    F.conv2d(x.permute(0, 3, 1, 2).double(), wref, stride=2, padding=1).backward(gy.permute(0, 3, 1, 2).double())
    _wgrad_check('s2 dW (64 x 32 tile)', dW, wref.grad); _wgrad_check('s2 dbias (64 x 32 tile)', db, gy.double().sum((0, 1, 2)))
    # Conv2d 1x1 on cat([x1, x2]) (2 x 64 -> 64), accumulating into an existing gradient
    B, H, W, C = 2, 32, 64, 64
    x1 = torch.randn(B, H, W, C, device='cuda', generator=g); x2 = torch.randn(B, H, W, C, device='cuda', generator=g) * 5
    gy = torch.randn(B, H, W, C, device='cuda', generator=g)
    assert ops.x3g_wgrad_supported(ops.X3G_PW, C, 2 * C)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_PW, B, H, W, C, 2 * C), device='cuda')
    base = torch.randn(C, 2 * C, device='cuda', generator=g)
    dW = base.clone()
    ops.conv1x1_h2_bwd_weight(gy, _slot(gy), C, x1, _slot(x1), C, x2, _slot(x2), dW, None, ws, accumulate=1)
    ref = base.double() + torch.einsum('bhwo,bhwi->oi', gy.double(), torch.cat([x1, x2], 3).double())
    _wgrad_check('1x1 dW (accumulated)', dW, ref)
    # ... and the 32 x 64 tile only the fp16x2 kernel has (ResUnet's sc9: cat(2 x 32) -> 32 at 512^2; round 6), ragged map, with the bias gradient
    B, H, W, C = 2, 33, 70, 32
    x1 = torch.randn(B, H, W, C, device='cuda', generator=g); x2 = torch.randn(B, H, W, C, device='cuda', generator=g) * 5
    gy = torch.randn(B, H, W, C, device='cuda', generator=g)
    assert ops.h2g_wgrad_supported(ops.X3G_PW, C, 2 * C) and not ops.x3g_wgrad_supported(ops.X3G_PW, C, 2 * C)
    ws = torch.empty(ops.h2g_wgrad_workspace_floats(ops.X3G_PW, B, H, W, C, 2 * C), device='cuda')
    dW = torch.full((C, 2 * C), float('nan'), device='cuda'); db = torch.full((C,), float('nan'), device='cuda')
    ops.conv1x1_h2_bwd_weight(gy, _slot(gy), C, x1, _slot(x1), C, x2, _slot(x2), dW, db, ws)
    _wgrad_check('1x1 dW (32 x 64 tile)', dW, torch.einsum('bhwo,bhwi->oi', gy.double(), torch.cat([x1, x2], 3).double()))
    _wgrad_check('1x1 dbias (32 x 64 tile)', db, gy.double().sum((0, 1, 2)))


@pytest.mark.parametrize('shape', [(2, 24, 40, 64, 64), (1, 16, 32, 128, 128), (2, 9, 70, 64, 32)])
def test_convt_h2_bwd_data_with_sign_bits_equals_float_masks(shape):
    """Round 6: ConvTranspose2d backward-data takes the LeakyReLU' mask of its input map from the SIGN BITS the 3x3 forward kernel stored for that map (tile-private
    layout, csrc/h2.h) instead of reading the float32 activation: bit-identical gradients, ragged maps included (rows / columns that do not fill the 16 x 32 tiles)."""
    from pnnp_amd import ops
    B, H, W, Cin, Cout = shape          # ConvTranspose2d(Cin -> Cout): dx [B,H,W,Cin] from g [B,2H,2W,Cout]; mask = an activation [B,H,W,Cin] written by a 3x3 layer
    gen = torch.Generator(device='cuda').manual_seed(5)
    # the map `below` and its sign bits come out of a 3x3 fp16x2 forward layer, as in the network
    xin = torch.randn(B, H, W, 32, device='cuda', generator=gen)
    w3 = torch.randn(Cin, 32, 3, 3, device='cuda', generator=gen) * 0.1
    jobs = ops.PackJobs(); f3 = torch.zeros(ops.h2_weight_bytes(32, Cin), dtype=torch.uint8, device='cuda'); s3 = jobs.add_h2(w3, f3, None, cin_pad=32); jobs.run()
    slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device='cuda'))
    below = torch.empty(B, H, W, Cin, device='cuda'); bits = torch.zeros(ops.h2_bits_words(B, H, W, Cin), dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd(xin, None, f3, s3, None, below, Cin, 1, slot(xin), bits_y=bits)
    wt = torch.randn(Cin, Cout, 2, 2, device='cuda', generator=gen) * 0.1
    g = torch.randn(B, 2 * H, 2 * W, Cout, device='cuda', generator=gen)
    jobs = ops.PackJobs(); fd = torch.zeros(ops.h2mat_bytes(4 * Cout, Cin), dtype=torch.uint8, device='cuda'); ff = torch.zeros(ops.h2mat_bytes(Cin, 4 * Cout), dtype=torch.uint8, device='cuda')
    sw = jobs.add_h2_convt(wt, ff, fd); jobs.run()
    sg = slot(g)
    dx_f = torch.full((B, H, W, Cin), float('nan'), device='cuda'); dx_b = torch.full_like(dx_f, float('nan'))
    a_f = torch.zeros(1, dtype=torch.int32, device='cuda'); a_b = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.convt_h2_bwd_data(g, sg, fd, sw, dx_f, mask=below, mode=1, amax_dx=a_f)
    ops.convt_h2_bwd_data(g, sg, fd, sw, dx_b, mask=None, mode=1, amax_dx=a_b, bits=bits)
    assert torch.isfinite(dx_b).all()
    assert torch.equal(dx_b, dx_f) and torch.equal(a_b, a_f)
    frac_neg = float((below <= 0).float().mean())
    assert 0.2 < frac_neg < 0.8                                      # the mask really selects


@pytest.mark.parametrize('B', [1, 2, 3, 5, 9])
def test_convt_h2_bwd_weight_tiles_per_workgroup(B):
    """Round 6: wgrad_h2g_kernel re-requests a staging slot for the tile after next behind its last staging step (rolling refill).  128 -> 64 channels on a
    32 x 64 map: the batch sets the pixel tiles per workgroup (one, two, a few, uneven shares) -- against float64 sums."""
    from pnnp_amd import ops
    Ci, Co, H, W = 128, 64, 32, 64
    gen = torch.Generator(device='cuda').manual_seed(B * 3 + 1)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen); g = torch.randn(B, 2 * H, 2 * W, Co, device='cuda', generator=gen)
    # dW[ci][co][a][b] = sum over pixels of x[p][ci] g[2 y + a][2 x + b][co]
    gd = g.double().view(B, H, 2, W, 2, Co)
    ref = torch.einsum('bhwi,bhawco->ioac', x.double(), gd)
    slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device='cuda'))
    ws = torch.zeros(ops.h2g_wgrad_workspace_floats(ops.X3G_CT, B, H, W, Ci, Co), device='cuda')
    dW = torch.full((Ci, Co, 2, 2), float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.convt_h2_bwd_weight(x, slot(x), g, slot(g), dW, ws, dbias=db)
    err = (dW.double() - ref).norm() / ref.norm()
    assert err < 2e-6, (B, float(err))
    assert (db.double() - g.double().sum((0, 1, 2))).abs().max() < 1e-5 * g.double().abs().sum((0, 1, 2)).max()
