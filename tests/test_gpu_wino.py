"""GPU parity of the Winograd F(2x2,3x3) convolution kernels (forward and backward-data) against torch-fp32
CPU references.  Winograd adds the rounding of the input/weight/output transforms to the summation-order
differences of the direct kernel: bar rtol 2e-4 / atol 2e-5 x magnitude."""
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import _rand, close, nchw, nhwc

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, C1, C2, Cout
    (1, 16, 16, 8, 0, 64), (2, 32, 48, 32, 0, 64), (1, 20, 36, 64, 0, 128), (1, 9, 33, 64, 64, 64),
    (1, 5, 7, 128, 128, 128), (2, 16, 16, 256, 0, 256), (1, 89, 133, 16, 0, 64), (1, 2, 2, 64, 0, 64),
    (3, 160, 176, 8, 0, 64),          # many workgroups per channel block
]


@pytest.mark.parametrize('case', CASES)
def test_wino_fwd(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2); b = _rand(Co, seed=4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    assert ops.wino_supported(C1 + C2, Co)
    u = torch.empty(16 * w.shape[0] * w.shape[1], device='cuda')
    ops.pack_conv_weight_wino(w.cuda(), u, None)
    for act in (0, 1, 2):
        ref = F.conv2d(xin, w, b, padding=1)
        ref = F.leaky_relu(ref, 0.2) if act == 1 else (F.relu(ref) if act == 2 else ref)
        y = torch.full((B, H, W, Co), float('nan'), device='cuda')
        ops.conv_wino_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, u, b.cuda(), y, Co, act)
        close(nchw(y), ref, rtol=2e-4, atol=2e-5, what=f'wino fwd {case} act{act}')


@pytest.mark.parametrize('case', CASES)
def test_wino_bwd_data(case):
    """dx of conv3x3 on cat[x1,x2]; second destination masked by LeakyReLU' of a saved activation."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    if C1 % 64 or (C2 and C2 % 64) or Co % 8:
        pytest.skip('dgrad writes C1(+C2) channels: multiples of 64 only')
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2)
    g = _rand(B, Co, H, W, seed=5)
    ref = F.conv_transpose2d(g, w, padding=1)
    saved = _rand(B, C2 or C1, H, W, seed=6)
    u = torch.empty(16 * w.shape[0] * w.shape[1], device='cuda')
    ops.pack_conv_weight_wino(w.cuda(), None, u)
    dx1 = torch.full((B, H, W, C1), float('nan'), device='cuda')
    if C2:
        dx2 = torch.full((B, H, W, C2), float('nan'), device='cuda')
        ops.conv_wino_bwd_data(nhwc(g).cuda(), u, dx1, dx2=dx2, mask2=nhwc(saved).cuda(), mode2=1)
        close(nchw(dx1), ref[:, :C1], rtol=2e-4, atol=2e-5, what=f'wino dgrad dx1 {case}')
        slope = torch.where(saved > 0, torch.ones_like(saved), torch.full_like(saved, 0.2))
        close(nchw(dx2), ref[:, C1:] * slope, rtol=2e-4, atol=2e-5, what=f'wino dgrad dx2 {case}')
    else:
        ops.conv_wino_bwd_data(nhwc(g).cuda(), u, dx1, mask1=nhwc(saved).cuda(), mode1=1)
        slope = torch.where(saved > 0, torch.ones_like(saved), torch.full_like(saved, 0.2))
        close(nchw(dx1), ref * slope, rtol=2e-4, atol=2e-5, what=f'wino dgrad {case}')
        base = torch.ones((B, H, W, C1), device='cuda')
        ops.conv_wino_bwd_data(nhwc(g).cuda(), u, base, accum1=1)
        close(nchw(base), ref + 1, rtol=2e-4, atol=2e-5, what=f'wino dgrad accum {case}')


def test_wino_unsupported_shapes():
    from pnnp_amd import _lib, ops
    assert not ops.wino_supported(4, 64) and not ops.wino_supported(64, 32)
    w = torch.zeros(32, 64, 3, 3, device='cuda')
    with pytest.raises(_lib.PnnpError):
        ops.pack_conv_weight_wino(w, torch.empty(16 * 32 * 64, device='cuda'), None)


WG_CASES = [  # B, H, W, C1, C2, Cout
    (1, 4, 8, 64, 0, 64), (2, 16, 24, 64, 0, 128), (1, 12, 40, 128, 0, 64), (2, 8, 16, 64, 64, 64),
    (1, 20, 32, 128, 128, 128), (3, 32, 32, 64, 0, 64), (1, 64, 64, 64, 0, 64),
]


@pytest.mark.parametrize('case', WG_CASES)
def test_wino_bwd_weight(case):
    """dW and dbias of conv3x3 on cat[x1,x2] against torch autograd (CPU, fp32).  Sums run over up to B*H*W pixels in a
    different order (and through the 4x4 transforms): rtol 3e-4 / atol 3e-5 x magnitude."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    assert ops.wino_wgrad_supported(H, W, Co, C1, C2)
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    g = _rand(B, Co, H, W, seed=5)
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2).requires_grad_(True)
    b = _rand(Co, seed=4).requires_grad_(True)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    (F.conv2d(xin, w, b, padding=1) * g).sum().backward()
    ws = torch.empty(ops.wino_wgrad_workspace_floats(B, H, W, Co, C1 + C2), device='cuda')
    dW = torch.full((Co, C1 + C2, 3, 3), float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_wino_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, ws)
    close(dW, w.grad, rtol=3e-4, atol=3e-5, what=f'wino wgrad {case}')
    close(db, b.grad, rtol=3e-4, atol=3e-5, what=f'wino bgrad {case}')
    ops.conv_wino_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, ws, accumulate=1)
    close(dW, 2 * w.grad, rtol=3e-4, atol=3e-5, what='wino wgrad accumulate')
    close(db, 2 * b.grad, rtol=3e-4, atol=3e-5, what='wino bgrad accumulate')


def test_wino_bwd_weight_unsupported():
    from pnnp_amd import ops
    assert not ops.wino_wgrad_supported(6, 8, 64, 64) and not ops.wino_wgrad_supported(8, 12, 64, 64)
    assert not ops.wino_wgrad_supported(8, 8, 32, 64) and not ops.wino_wgrad_supported(8, 8, 64, 64, 32)


def test_wino_residual_and_shortcut_gradient():
    """ResidualBlock forms (archs/modules.py:176-197): forward y = conv(x) + residual; backward through the identity
    shortcut dx = (conv_bwd_data(g) + g_skip) * relu'(saved)."""
    from pnnp_amd import ops
    B, H, W, C = 2, 20, 24, 64
    x = _rand(B, C, H, W, seed=1); r = _rand(B, C, H, W, seed=2); w = _rand(C, C, 3, 3, seed=3, scale=0.2)
    uf = torch.empty(16 * C * C, device='cuda'); ud = torch.empty(16 * C * C, device='cuda')
    ops.pack_conv_weight_wino(w.cuda(), uf, ud)
    y = torch.full((B, H, W, C), float('nan'), device='cuda')
    ops.conv_wino_fwd(nhwc(x).cuda(), None, uf, None, y, C, 0, residual=nhwc(r).cuda())
    close(nchw(y), F.conv2d(x, w, None, padding=1) + r, rtol=2e-4, atol=2e-5, what='wino fwd + residual')
    g = _rand(B, C, H, W, seed=5); gs = _rand(B, C, H, W, seed=6); saved = _rand(B, C, H, W, seed=7)
    ref = (F.conv_transpose2d(g, w, padding=1) + gs) * (saved > 0).float()
    dx = torch.full((B, H, W, C), float('nan'), device='cuda')
    ops.conv_wino_bwd_data_res(nhwc(g).cuda(), ud, dx, addsrc=nhwc(gs).cuda(), mask=nhwc(saved).cuda(), mode=2)
    close(nchw(dx), ref, rtol=2e-4, atol=2e-5, what='wino dgrad + shortcut')
    ops.conv_wino_bwd_data_res(nhwc(g).cuda(), ud, dx, addsrc=nhwc(gs).cuda())
    close(nchw(dx), F.conv_transpose2d(g, w, padding=1) + gs, rtol=2e-4, atol=2e-5, what='wino dgrad + shortcut, no mask')
