"""GPU parity of the bf16x3 convolution kernels (csrc/conv_x3.hip: float32 operands split into three bf16 pieces on the
bf16 matrix cores) against torch-fp32 CPU references of the same op -- at the SAME tolerances as the fp32-MFMA kernels
(tests/test_gpu_conv.py: rtol 1e-4 / atol 1e-5 of the largest sum) -- and, to show the split is not a precision loss,
against a float64 reference next to the fp32-MFMA kernel's own error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import close, nchw, nhwc, _rand

pytestmark = pytest.mark.gpu

CASES = [  # B, H, W, C1, C2 (0 = no concat), Cout
    (1, 8, 32, 8, 0, 32), (2, 16, 48, 32, 0, 32), (1, 12, 40, 16, 0, 64), (1, 6, 70, 64, 0, 128),
    (2, 8, 32, 32, 32, 32), (1, 8, 36, 64, 64, 64), (1, 5, 17, 128, 128, 128), (1, 4, 4, 256, 0, 256),
    (1, 16, 32, 32, 0, 256), (1, 9, 33, 24, 0, 96), (3, 19, 50, 48, 48, 32),
]


def _packs(w, fwd=True, dgrad=True, cin_pad=None):
    from pnnp_amd import ops
    co, ci = w.shape[:2]
    jobs = ops.PackJobs()
    f = torch.zeros(ops.x3_weight_bytes(cin_pad or ci, co), dtype=torch.uint8, device='cuda') if fwd else None
    d = torch.zeros(ops.x3_weight_bytes(co, ci), dtype=torch.uint8, device='cuda') if dgrad else None
    jobs.add_x3(w, f, d, cin_pad=cin_pad)
    jobs.run()
    return f, d


def test_x3_pack_reconstructs_the_float32_weights_exactly():
    """hi + mid + lo == w bit for bit (the split is exact), and the pack order is the one the kernel streams."""
    w = (_rand(64, 24, 3, 3, seed=3) * torch.logspace(-6, 3, 64 * 24 * 9).reshape(64, 24, 3, 3)).cuda()
    f, d = _packs(w)
    def unpack(buf, K, N):                       # [N/32][K16][tap][oct][piece][32][8] uint16 -> float32 [piece][K][N][tap]
        K16 = (K + 15) // 16
        a = buf.view(torch.int16).cpu().numpy().view(np.uint16).reshape(N // 32, K16, 9, 2, 3, 32, 8)
        a = (a.astype(np.uint32) << 16).view(np.float32)
        return a.transpose(4, 1, 3, 6, 0, 5, 2).reshape(3, K16 * 16, N, 9)
    wn = w.cpu().numpy()
    pf = unpack(f, 24, 64)
    rec = (pf[0].astype(np.float64) + pf[1] + pf[2]).astype(np.float32)
    assert np.array_equal(rec[:24], wn.transpose(1, 0, 2, 3).reshape(24, 64, 9)) and not rec[24:].any()
    pd = unpack(d, 64, 32)                       # dgrad: K = Cout, N = Cin padded to 32, taps flipped
    rec = (pd[0].astype(np.float64) + pd[1] + pd[2]).astype(np.float32)
    assert np.array_equal(rec[:, :24], wn.reshape(64, 24, 9)[:, :, ::-1]) and not rec[:, 24:].any()
    assert np.abs(pf[1]).max() <= np.abs(pf[0]).max() * 2.0 ** -8 and np.abs(pf[2]).max() <= np.abs(pf[0]).max() * 2.0 ** -16


@pytest.mark.parametrize('case', CASES)
def test_x3_fwd(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2); b = _rand(Co, seed=4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    f, _ = _packs(w.cuda(), dgrad=False)
    for act in (0, 1, 2):
        ref = F.conv2d(xin, w, b, padding=1)
        ref = F.leaky_relu(ref, 0.2) if act == 1 else (F.relu(ref) if act == 2 else ref)
        y = torch.full((B, H, W, Co), float('nan'), device='cuda')
        ops.conv_x3_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, f, b.cuda(), y, Co, act)
        close(nchw(y), ref, what=f'x3 fwd {case} act{act}')
    r = _rand(B, Co, H, W, seed=9)
    y = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_x3_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, f, b.cuda(), y, Co, 2, residual=nhwc(r).cuda())
    close(nchw(y), F.relu(F.conv2d(xin, w, b, padding=1) + r), what='x3 residual')


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 32 == 0])
def test_x3_bwd_data(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2)
    g = _rand(B, Co, H, W, seed=5)
    xin = _rand(B, C1 + C2, H, W, seed=6).requires_grad_(True)
    F.conv2d(xin, w, None, padding=1).backward(g)
    ref = xin.grad
    _, dg = _packs(w.cuda(), fwd=False)
    m1 = _rand(B, C1, H, W, seed=7); m2 = _rand(B, max(C2, 1), H, W, seed=8)
    d1 = torch.full((B, H, W, C1), float('nan'), device='cuda')
    d2 = torch.full((B, H, W, C2), float('nan'), device='cuda') if C2 else None
    ops.conv_x3_bwd_data(nhwc(g).cuda(), dg, d1, dx2=d2)
    close(nchw(d1), ref[:, :C1], what=f'x3 dgrad {case}')
    if C2:
        close(nchw(d2), ref[:, C1:], what=f'x3 dgrad2 {case}')
    base2 = _rand(B, max(C2, 1), H, W, seed=10)
    d1 = torch.empty((B, H, W, C1), device='cuda')
    d2 = nhwc(base2).cuda().clone() if C2 else None
    ops.conv_x3_bwd_data(nhwc(g).cuda(), dg, d1, mask1=nhwc(m1).cuda(), mode1=1, dx2=d2,
                         mask2=nhwc(m2).cuda() if C2 else None, mode2=2, accum2=1)
    close(nchw(d1), ref[:, :C1] * torch.where(m1 > 0, 1.0, 0.2), what='x3 mask1')
    if C2:
        close(nchw(d2), base2 + ref[:, C1:] * (m2 > 0).float(), what='x3 mask2+accum')
    if not C2:
        add = _rand(B, C1, H, W, seed=11)
        dx = torch.empty((B, H, W, C1), device='cuda')
        ops.conv_x3_bwd_data_res(nhwc(g).cuda(), dg, dx, addsrc=nhwc(add).cuda(), mask=nhwc(m1).cuda(), mode=2)
        close(nchw(dx), (ref + add) * (m1 > 0).float(), what='x3 dgrad res')


def test_x3_padded_network_input():
    """conv1_1: the 4-channel input travels as 8-channel NHWC; the x3 pack pads the reduction to 16 with zeros."""
    from pnnp_amd import ops
    B, H, W, Co = 2, 16, 64, 32
    x = _rand(B, 4, H, W, seed=1); w = _rand(Co, 4, 3, 3, seed=2, scale=0.3); b = _rand(Co, seed=3)
    x8 = torch.empty((B, H, W, 8), device='cuda'); ops.nchw_to_nhwc(x.cuda(), x8, 8)
    f, _ = _packs(w.cuda(), dgrad=False, cin_pad=16)
    y = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_x3_fwd(x8, None, f, b.cuda(), y, Co, 1)
    close(nchw(y), F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2), what='x3 conv1_1')


def test_x3_is_as_accurate_as_the_fp32_mfma_kernel():
    """Both kernels against a float64 reference on a deep reduction (K = 9 x 512) with operands spanning 8 decades: the
    bf16x3 error must stay within 2x the fp32-MFMA kernel's (measured: ~1x), and both at float32 level."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 16, 32, 512, 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g) * torch.logspace(-4, 4, Ci, base=10.0).reshape(1, Ci, 1, 1).roll(1, 1)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    f3, _ = _packs(w.cuda(), dgrad=False)
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    y3 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, None, y32, Co, 9, 0)
    e3 = float((nchw(y3).cpu().double() - ref).norm() / ref.norm())
    e32 = float((nchw(y32).cpu().double() - ref).norm() / ref.norm())
    print(f'relative L2 error vs float64: bf16x3 {e3:.2e}, fp32 MFMA {e32:.2e}')
    assert e3 < 2.0 * e32 + 1e-8 and e3 < 5e-7


@pytest.mark.parametrize('case', [(2, 8, 32, 32, 32, 64), (1, 9, 33, 64, 64, 32), (1, 16, 64, 128, 128, 64)])
def test_x3_bwd_data_mask_on_the_second_destination_only(case):
    """Backward-data of a decoder's first conv (archs/Unet.py:74-93: cat([up, skip])): the `up` half has no activation behind it, the skip half
    does -- one launch, act' mask on destination 2 only (the masked-epilogue kernel requests the missing mask out of range)."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2)
    g = _rand(B, Co, H, W, seed=5)
    xin = _rand(B, C1 + C2, H, W, seed=6).requires_grad_(True)
    F.conv2d(xin, w, None, padding=1).backward(g)
    ref = xin.grad
    _, dg = _packs(w.cuda(), fwd=False)
    m2 = _rand(B, C2, H, W, seed=8)
    d1 = torch.full((B, H, W, C1), float('nan'), device='cuda'); d2 = torch.full((B, H, W, C2), float('nan'), device='cuda')
    ops.conv_x3_bwd_data(nhwc(g).cuda(), dg, d1, dx2=d2, mask2=nhwc(m2).cuda(), mode2=1)
    close(nchw(d1), ref[:, :C1], what=f'x3 dgrad dst1 (no mask) {case}')
    close(nchw(d2), ref[:, C1:] * torch.where(m2 > 0, 1.0, 0.2), what=f'x3 dgrad dst2 (LeakyReLU mask) {case}')


def test_x3_more_output_channels_than_the_bias_buffer_holds():
    """The matrix-core 3x3 kernels keep the layer's bias vector in LDS (1024 channels): a wider layer is refused up front (pnnp_x3_supported /
    pnnp_h2_supported, PNNP_E_UNSUPPORTED from the entry point) and runs on the fp32-MFMA kernel -- same results, same contract.  (Until round 5
    round 3's kernel took these layers; it is gone: DESIGN Appendix A.)"""
    from pnnp_amd import ops
    from pnnp_amd._lib import PnnpError
    B, H, W, Ci, Co = 1, 16, 32, 32, 1088
    assert not ops.x3_supported(Ci, Co) and not ops.h2_supported(Ci, Co) and ops.x3_supported(Ci, 1024)
    x = _rand(B, Ci, H, W, seed=1); w = _rand(Co, Ci, 3, 3, seed=3, scale=0.2); b = _rand(Co, seed=4)
    f, _ = _packs(w.cuda(), dgrad=False)
    y = torch.full((B, H, W, Co), float('nan'), device='cuda')
    with pytest.raises(PnnpError):
        ops.conv_x3_fwd(nhwc(x).cuda(), None, f, b.cuda(), y, Co, 1)
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, b.cuda(), y, Co, 9, 1)
    close(nchw(y), F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2), what='fp32-MFMA fwd 1088 output channels')


@pytest.mark.parametrize('case', [(2, 16, 48, 32, 0, 32), (1, 12, 40, 32, 0, 64), (1, 6, 70, 64, 0, 128), (2, 8, 32, 32, 32, 32),
                                  (1, 8, 36, 64, 64, 64), (1, 5, 17, 128, 128, 128), (1, 4, 4, 256, 0, 256), (1, 16, 32, 32, 0, 256),
                                  (3, 32, 64, 32, 0, 32), (2, 16, 32, 64, 64, 64), (3, 19, 50, 64, 32, 96), (1, 9, 33, 96, 0, 32)])
def test_x3_bwd_weight(case):
    """dW and dbias vs autograd on the CPU, at the fp32-MFMA weight-gradient kernel's bars (rtol 2e-4 / atol 2e-5 of the largest
    sum), with and without accumulation."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2).requires_grad_(True)
    b = _rand(Co, seed=4).requires_grad_(True)
    g = _rand(B, Co, H, W, seed=5)
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    F.conv2d(xin, w, b, padding=1).backward(g)
    assert ops.x3_wgrad_supported(H, W, Co, C1, C2)
    ws = torch.empty(ops.x3_wgrad_workspace_floats(B, H, W, Co, C1 + C2), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_x3_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'x3 wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'x3 bgrad {case}')
    ops.conv_x3_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, ws, accumulate=1)
    close(dW, 2 * w.grad, rtol=2e-4, atol=2e-5, what='x3 wgrad accumulate')
    close(db, 2 * b.grad, rtol=2e-4, atol=4e-5, what='x3 bgrad accumulate')
    dW2 = torch.full(w.shape, float('nan'), device='cuda')
    ops.conv_x3_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW2, None, ws)       # no bias gradient asked for
    close(dW2, w.grad, rtol=2e-4, atol=2e-5, what='x3 wgrad without dbias')


@pytest.mark.parametrize('case', [(2, 8, 16, 64, 32), (1, 3, 35, 128, 64), (1, 2, 2, 512, 256), (1, 16, 32, 64, 32), (2, 9, 40, 256, 128), (1, 8, 64, 32, 32)])
def test_x3_convt(case):
    """ConvTranspose2d(k2, s2) forward and backward-data on the pointwise bf16x3 GEMM kernel vs F.conv_transpose2d."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    x = _rand(B, Ci, H, W, seed=1).requires_grad_(True)
    w = _rand(Ci, Co, 2, 2, seed=2, scale=0.2); b = _rand(Co, seed=3)
    y = F.conv_transpose2d(x, w, b, stride=2)
    g = _rand(B, Co, 2 * H, 2 * W, seed=4)
    y.backward(g)
    jobs = ops.PackJobs()
    f = torch.zeros(ops.x3mat_bytes(Ci, 4 * Co), dtype=torch.uint8, device='cuda'); d = torch.zeros(ops.x3mat_bytes(4 * Co, Ci), dtype=torch.uint8, device='cuda')
    jobs.add_x3_convt(w.cuda(), f, d); jobs.run()
    yo = torch.full((B, 2 * H, 2 * W, Co), float('nan'), device='cuda')
    ops.convt_x3_fwd(nhwc(x.detach()).cuda(), f, b.cuda(), yo, Co)
    close(nchw(yo), y, what=f'x3 convT fwd {case}')
    m = _rand(B, Ci, H, W, seed=5)
    dx = torch.full((B, H, W, Ci), float('nan'), device='cuda')
    ops.convt_x3_bwd_data(nhwc(g).cuda(), d, dx, mask=nhwc(m).cuda(), mode=1)
    close(nchw(dx), x.grad * torch.where(m > 0, 1.0, 0.2), what=f'x3 convT dgrad {case}')


@pytest.mark.parametrize('case', [(2, 8, 32, 32, 32, 64), (1, 12, 40, 64, 64, 64), (1, 5, 17, 128, 128, 128), (1, 16, 33, 64, 0, 128), (2, 9, 20, 32, 0, 32)])
def test_x3_conv1x1(case):
    """Conv2d 1x1 (ResidualBlock shortcuts: concat inputs, accumulate into existing gradients) on the pointwise kernel."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, 1, 1, seed=3, scale=0.2); b = _rand(Co, seed=4)
    xin = (torch.cat([x1, x2], 1) if C2 else x1).requires_grad_(True)
    y = F.conv2d(xin, w, b)
    g = _rand(B, Co, H, W, seed=5); y.backward(g)
    jobs = ops.PackJobs()
    f = torch.zeros(ops.x3mat_bytes(C1 + C2, Co), dtype=torch.uint8, device='cuda'); d = torch.zeros(ops.x3mat_bytes(Co, C1 + C2), dtype=torch.uint8, device='cuda')
    jobs.add_x3_1x1(w.cuda(), f, d); jobs.run()
    yo = torch.full((B, H, W, Co), float('nan'), device='cuda')
    ops.conv1x1_x3_fwd(nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None, f, b.cuda(), yo, Co, 2)
    close(nchw(yo), F.relu(y), what=f'x3 1x1 fwd {case}')
    base1 = _rand(B, C1, H, W, seed=6); base2 = _rand(B, max(C2, 1), H, W, seed=7)
    d1 = nhwc(base1).cuda().clone(); d2 = nhwc(base2).cuda().clone() if C2 else None
    ops.conv1x1_x3_bwd_data(nhwc(g).cuda(), d, d1, accum1=1, dx2=d2, accum2=1)
    close(nchw(d1), base1 + xin.grad[:, :C1], what='x3 1x1 dgrad accum')
    if C2:
        close(nchw(d2), base2 + xin.grad[:, C1:], what='x3 1x1 dgrad2 accum')


@pytest.mark.parametrize('case', [(2, 16, 64, 32, 64), (1, 12, 40, 64, 128), (1, 6, 70, 128, 256), (1, 4, 4, 256, 512), (1, 34, 66, 32, 32)])
def test_x3_conv3x3_stride2(case):
    """ResUnet's down-sampling conv (archs/modules.py:130-138) forward and backward-data on the pointwise kernel."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    x = _rand(B, Ci, H, W, seed=1).requires_grad_(True)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=0.2); b = _rand(Co, seed=3)
    y = F.conv2d(x, w, b, stride=2, padding=1)
    g = _rand(B, Co, H // 2, W // 2, seed=4); y.backward(g)
    jobs = ops.PackJobs()
    f = torch.zeros(ops.x3mat_bytes(9 * Ci, Co), dtype=torch.uint8, device='cuda'); d = torch.zeros(9 * ops.x3mat_bytes(Co, Ci), dtype=torch.uint8, device='cuda')
    jobs.add_x3_s2(w.cuda(), f, d); jobs.run()
    yo = torch.full((B, H // 2, W // 2, Co), float('nan'), device='cuda')
    ops.conv_s2_x3_fwd(nhwc(x.detach()).cuda(), f, b.cuda(), yo, Co)
    close(nchw(yo), y, what=f'x3 s2 fwd {case}')
    base = _rand(B, Ci, H, W, seed=6)
    dx = nhwc(base).cuda().clone()
    ops.conv_s2_x3_bwd_data(nhwc(g).cuda(), d, dx, accum=1)
    close(nchw(dx), base + x.grad, what=f'x3 s2 dgrad {case}')


@pytest.mark.parametrize('cin,cout,shape', [(32, 32, (2, 32, 64)), (32, 64, (1, 48, 96)), (64, 64, (2, 16, 32)), (16, 128, (1, 32, 32))])
def test_fused_maxpool_equals_conv_then_pool_kernel(cin, cout, shape):
    """conv3x3 + LeakyReLU + MaxPool2d(2) in one kernel (archs/Unet.py:33-35): y, the pooled map and the argmax/sign codes are
    bit-identical to the un-fused conv followed by the pool kernel (same accumulation, same first-maximum rule)."""
    from pnnp_amd import ops
    B, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin, device='cuda', generator=g)
    x[0, :4, :8] = 0.0                                                  # ties: equal values inside windows (first maximum must win)
    w = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * 0.1
    b = torch.randn(cout, device='cuda', generator=g) * 0.1
    b[: cout // 2] = 0.0
    wx = torch.empty(ops.x3_weight_bytes(cin, cout), dtype=torch.uint8, device='cuda')
    jobs = ops.PackJobs(); jobs.add_x3(w, wx, None, cin_pad=(cin + 15) // 16 * 16); jobs.run()
    y0 = torch.empty(B, H, W, cout, device='cuda'); p0 = torch.empty(B, H // 2, W // 2, cout, device='cuda')
    c0 = torch.empty(B, H // 2, W // 2, cout, dtype=torch.uint8, device='cuda')
    ops.conv_x3_fwd(x, None, wx, b, y0, cout, 1)
    ops.maxpool_fwd(y0, p0, codes=c0)
    y1 = torch.full_like(y0, float('nan')); p1 = torch.full_like(p0, float('nan')); c1 = torch.full_like(c0, 255)
    ops.conv_x3_fwd_pool(x, None, wx, b, y1, p1, c1, cout, 1)
    assert torch.equal(y1, y0) and torch.equal(p1, p0) and torch.equal(c1, c0)
    assert int((c0 & 3 != 0).sum()) > 0 and int((c0 >> 2 == 0).sum()) > 0      # the case has non-trivial argmax and all-negative windows


# ---------------------------------------------------------------------------------------------------------------------
# Precision case of the bf16x3 family, hardened (round 3): float64 yardsticks for backward-data and backward-weight next to
# their fp32-MFMA twins, a max-element bound under cancellation, and the dynamic range the 3-way split supports
# (include/pnnp_hip.h, "bf16x3 dynamic range").
def _f64_wgrad(g, x):
    """dW[co][ci][ky][kx] = sum_{b,y,x} g[b,y,x,co] xpad[b,y+ky,x+kx,ci] in float64 on the GPU (NHWC inputs), one batch
    element at a time (a plain matmul per tap: the yardstick, not a product path)."""
    B, H, W, Co = g.shape
    Ci = x.shape[3]
    dW = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device=g.device)
    for b in range(B):
        xp = F.pad(x[b].double(), (0, 0, 1, 1, 1, 1))
        gb = g[b].double().reshape(H * W, Co)
        for ky in range(3):
            for kx in range(3):
                dW[:, :, ky, kx] += gb.t() @ xp[ky:ky + H, kx:kx + W].reshape(H * W, Ci)
    return dW


def test_x3_dgrad_is_as_accurate_as_the_fp32_mfma_kernel():
    """conv_x3_bwd_data and the fp32-MFMA backward-data kernel against float64 on a K = 9 x 512 reduction with gradients spanning
    8 decades: the bf16x3 error stays within 2x the fp32-MFMA kernel's, both at float32 level."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 16, 32, 64, 512
    gen = torch.Generator().manual_seed(1)
    g = torch.randn(B, Co, H, W, generator=gen) * torch.logspace(-4, 4, Co, base=10.0).reshape(1, Co, 1, 1).roll(3, 1)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.05
    ref = F.conv_transpose2d(g.double(), w.double(), None, padding=1)              # = d/dx of conv2d(x, w, padding=1)
    _, d3 = _packs(w.cuda(), fwd=False)
    d32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), None, d32)
    y3 = torch.empty((B, H, W, Ci), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_bwd_data(nhwc(g).cuda(), d3, y3)
    ops.conv_bwd_data(nhwc(g).cuda(), d32, y32)
    e3 = float((nchw(y3).cpu().double() - ref).norm() / ref.norm())
    e32 = float((nchw(y32).cpu().double() - ref).norm() / ref.norm())
    m3 = float((nchw(y3).cpu().double() - ref).abs().max() / ref.abs().max())
    m32 = float((nchw(y32).cpu().double() - ref).abs().max() / ref.abs().max())
    print(f'dgrad vs float64: rel L2 bf16x3 {e3:.2e} fp32-MFMA {e32:.2e}; max-element / max|ref| bf16x3 {m3:.2e} fp32-MFMA {m32:.2e}')
    assert e3 < 2.0 * e32 + 1e-8 and e3 < 5e-7
    assert m3 < 2.0 * m32 + 1e-8


@pytest.mark.parametrize('shape', [(16, 512, 512, 32, 32), (4, 128, 128, 64, 128)])
def test_x3_wgrad_is_as_accurate_as_the_fp32_mfma_kernel(shape):
    """wgrad_x3 splits BOTH operands on the fly; its reduction runs over pixels -- K = 16 x 512 x 512 = 4.2 M terms per weight at
    the benchmark's top level.  Against a float64 evaluation of the same sums (GPU matmuls in double) next to the fp32-MFMA
    backward-weight kernel: relative L2 AND max-element error within 2x the fp32 kernel's."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = shape
    gen = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen) * torch.logspace(-2, 2, Ci, device='cuda').roll(5)
    g = torch.randn(B, H, W, Co, device='cuda', generator=gen) * torch.logspace(-3, 1, Co, device='cuda')
    ref = _f64_wgrad(g, x)
    assert ops.x3_wgrad_supported(H, W, Co, Ci, 0)
    ws = torch.empty(max(ops.x3_wgrad_workspace_floats(B, H, W, Co, Ci), ops.wgrad_workspace_floats(B, H, W, Co, Ci, 9)), device='cuda')
    d3 = torch.empty(Co, Ci, 3, 3, device='cuda'); d32 = torch.empty_like(d3)
    ops.conv_x3_bwd_weight(g, Co, x, Ci, None, d3, None, ws)
    ops.conv_bwd_weight(g, Co, x, Ci, None, d32, None, 9, ws)
    rn = ref.norm()
    # per-weight condition: sum |g||x| is ~sqrt(K) x the result for random signs, so scale the max-element error by the row scale
    e3, e32 = float((d3.double() - ref).norm() / rn), float((d32.double() - ref).norm() / rn)
    scale = ref.abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-30)
    m3, m32 = float(((d3.double() - ref).abs() / scale).max()), float(((d32.double() - ref).abs() / scale).max())
    print(f'wgrad {shape} vs float64: rel L2 bf16x3 {e3:.2e} fp32-MFMA {e32:.2e}; max-element (per (co,ci) scale) bf16x3 {m3:.2e} fp32-MFMA {m32:.2e}')
    assert e3 < 2.0 * e32 + 1e-8, (e3, e32)
    assert m3 < 2.0 * m32 + 1e-7, (m3, m32)


def test_x3_max_element_error_under_cancellation():
    """Sums that cancel: every output is sum_k (w_k x_k - w_k x_k') with x' = x (1 + 2^-12 r), O(1) terms whose sum is ~2^-12 of
    their magnitude.  An fp32 dot product is accurate to ~K 2^-24 max|term|, not to the (tiny) result; the bf16x3 kernel must obey
    the SAME absolute bound element by element (max, not L2) and stay within 2x the fp32-MFMA kernel's worst element."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 16, 32, 256, 64
    gen = torch.Generator().manual_seed(4)
    xa = torch.randn(B, Ci // 2, H, W, generator=gen)
    xb = xa * (1 + 2.0 ** -12 * torch.randn(B, Ci // 2, H, W, generator=gen))
    x = torch.stack([xa, xb], 2).reshape(B, Ci, H, W)                       # channel 2k: x, 2k+1: x'
    wa = torch.randn(Co, Ci // 2, 3, 3, generator=gen) * 0.1
    w = torch.stack([wa, -wa], 2).reshape(Co, Ci, 3, 3)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    terms = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1)   # sum of |terms| per output
    assert float(ref.abs().mean() / terms.mean()) < 1e-3                     # the case does cancel
    f3, _ = _packs(w.cuda(), dgrad=False)
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    y3 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, None, y32, Co, 9, 0)
    r3 = ((nchw(y3).cpu().double() - ref).abs() / terms).max()
    r32 = ((nchw(y32).cpu().double() - ref).abs() / terms).max()
    print(f'cancellation: max |err| / sum|terms|: bf16x3 {float(r3):.2e}, fp32-MFMA {float(r32):.2e} (2^-24 = {2.0 ** -24:.2e})')
    assert float(r3) < 2.0 * float(r32) + 2.0 ** -26 and float(r3) < 8 * 2.0 ** -24


@pytest.mark.parametrize('xs,wsc', [(1e-30, 1.0), (1e+30, 1e-3), (1e-15, 1e-15), (1.0, 1e-30)])
def test_x3_dynamic_range(xs, wsc):
    """The supported range (include/pnnp_hip.h): a piece is a bf16 with the float32 exponent range, so the three pieces of |a| >=
    2^-110 (7.7e-34) are all normal numbers and the split is exact; products and sums obey float32's own range.  Operands scaled
    to 1e-30 / 1e+30 / both 1e-15 give results as accurate (vs float64) as at scale 1, and within 2x of the fp32-MFMA kernel."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 8, 32, 128, 32
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, H, W, generator=gen) * xs
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.05 * wsc
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    f3, _ = _packs(w.cuda(), dgrad=False)
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    y3 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, None, y32, Co, 9, 0)
    e3 = float((nchw(y3).cpu().double() - ref).norm() / ref.norm())
    e32 = float((nchw(y32).cpu().double() - ref).norm() / ref.norm())
    print(f'scale x {xs:g} w {wsc:g}: rel L2 vs float64 bf16x3 {e3:.2e}, fp32-MFMA {e32:.2e}')
    assert torch.isfinite(y3).all()
    assert e3 < 2.0 * e32 + 1e-8 and e3 < 1e-6          # (uniform-scale data at K = 1152: ~5e-7 for both kernels, tools/x3_bias_probe.py)


def test_x3_below_the_supported_range_degrades_gracefully():
    """Below 2^-110 the `lo` (then `mid`) piece is a bf16 subnormal: whether the matrix core keeps or flushes it, the result loses
    at most those pieces -- the error stays below 2^-8 relative (one piece) and the output finite.  Documented limit, not a
    float32-accuracy claim."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 8, 32, 64, 32
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1e-36
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.5
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    f3, _ = _packs(w.cuda(), dgrad=False)
    y3 = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    e3 = float((nchw(y3).cpu().double() - ref).norm() / ref.norm())
    print(f'x at 1e-36 (below the supported 7.7e-34): rel L2 vs float64 {e3:.2e}')
    assert torch.isfinite(y3).all() and e3 < 2.0 ** -7


@pytest.mark.parametrize('positive', [False, True])
def test_x3_forward_signed_mean_error_is_bounded(positive):
    """The bf16 matrix core rounds an inexact accumulation toward MINUS INFINITY (csrc/wgrad_x3.hip, WX3_ALT_SIGN) and
    rounds each small product to 1/8 ulp of the accumulator first.  `wgrad_x3` cancels the resulting drift with alternating signs;
    the forward / backward-data kernel (v_mfma_f32_16x16x32_bf16, two pieces concatenated along K) keeps it.  This bounds it: on a
    LONG uniform-scale reduction (K = 9 x 512 = 4608, every term the same size -- the worst case for a per-accumulation rounding
    bias) the mean SIGNED relative error against float64 stays within 4e-6 (measured -1.4e-6; the fp32-MFMA kernel's is sign-random
    at +-4e-7) and the L2 error within 2x of the fp32-MFMA kernel's.  All-positive data: the sum grows with K, each rounding is
    relative to the running sum, and the bias must stay below 5e-7."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 16, 32, 512, 64
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, Ci, H, W, generator=g) + 0.5
    w = (torch.rand(Co, Ci, 3, 3, generator=g) + 0.5) * 0.05
    if not positive:
        x = x * (torch.randint(0, 2, x.shape, generator=g) * 2 - 1)
        w = w * (torch.randint(0, 2, w.shape, generator=g) * 2 - 1)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    f3, _ = _packs(w.cuda(), dgrad=False)
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    y3 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, None, y32, Co, 9, 0)

    def stats(y):
        d = nchw(y).cpu().double() - ref
        # bias relative to the typical output size (a per-element ratio would be dominated by outputs that cancel to ~0)
        return float(d.norm() / ref.norm()), float(d.mean() / ref.abs().mean())
    (l3, b3), (l32, b32) = stats(y3), stats(y32)
    print(f'K=4608 {"positive" if positive else "rnd-sign"}: bf16x3 L2 {l3:.2e} signed mean {b3:+.2e} | fp32-MFMA L2 {l32:.2e} signed mean {b32:+.2e}')
    assert l3 < 2.0 * l32 + 1e-8
    assert abs(b3) < (5e-7 if positive else 4e-6)


# ---- round 4: weight gradients of the pointwise / strided layers on the bf16 matrix cores (csrc/wgrad_x3g.hip)
@pytest.mark.parametrize('case', [(2, 8, 16, 256, 64), (1, 5, 19, 512, 256), (2, 6, 24, 128, 64), (1, 3, 35, 128, 64), (1, 16, 32, 64, 32),
                                  (2, 9, 40, 64, 32), (1, 7, 9, 256, 128), (1, 4, 34, 64, 64)])
def test_x3_convt_bwd_weight(case):
    """dW [Cin][Cout][2][2] and dbias of ConvTranspose2d(k2, s2) (archs/Unet.py:35-47) vs autograd, for every tile configuration
    (256 x 64, 128 x 64, 64 x 32 output tiles), maps that do not fill their pixel tiles, and `accumulate`."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    assert ops.x3g_wgrad_supported(ops.X3G_CT, Ci, Co)
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Ci, Co, 2, 2, seed=2, scale=0.2).requires_grad_(True); b = _rand(Co, seed=3).requires_grad_(True)
    g = _rand(B, Co, 2 * H, 2 * W, seed=4)
    F.conv_transpose2d(x, w, b, stride=2).backward(g)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, H, W, Ci, Co), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.convt_x3_bwd_weight(nhwc(x).cuda(), nhwc(g).cuda(), dW, ws, dbias=db)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'x3 convT wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'x3 convT dbias {case}')
    ops.convt_x3_bwd_weight(nhwc(x).cuda(), nhwc(g).cuda(), dW, ws, accumulate=1, dbias=db)
    close(dW, 2 * w.grad, rtol=2e-4, atol=4e-5, what='x3 convT wgrad accumulate')
    close(db, 2 * b.grad, rtol=2e-4, atol=4e-5, what='x3 convT dbias accumulate')
    dW2 = torch.full(w.shape, float('nan'), device='cuda')
    ops.convt_x3_bwd_weight(nhwc(x).cuda(), nhwc(g).cuda(), dW2, ws)
    close(dW2, w.grad, rtol=2e-4, atol=2e-5, what='x3 convT wgrad without dbias')


@pytest.mark.parametrize('case', [(2, 16, 64, 32, 128), (1, 12, 40, 64, 128), (1, 6, 70, 128, 256), (1, 4, 4, 256, 512), (1, 34, 66, 32, 128), (2, 2, 6, 64, 256)])
def test_x3_conv3x3_stride2_bwd_weight(case):
    """dW [Cout][Cin][3][3] and dbias of the ResUnet's down-sampling conv (archs/modules.py:130-138) vs autograd: Cout in multiples of
    128 run on the bf16x3 kernel (halo rows / columns outside the map, odd tile counts); others are refused (the fp32 kernel takes them)."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = case
    assert ops.x3g_wgrad_supported(ops.X3G_S2, Co, Ci) and not ops.x3g_wgrad_supported(ops.X3G_S2, 64, 32)
    x = _rand(B, Ci, H, W, seed=1)
    w = _rand(Co, Ci, 3, 3, seed=2, scale=0.2).requires_grad_(True); b = _rand(Co, seed=3).requires_grad_(True)
    g = _rand(B, Co, H // 2, W // 2, seed=4)
    F.conv2d(x, w, b, stride=2, padding=1).backward(g)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_S2, B, H // 2, W // 2, Co, Ci), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_s2_x3_bwd_weight(nhwc(g).cuda(), nhwc(x).cuda(), dW, db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'x3 s2 wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'x3 s2 dbias {case}')
    ops.conv_s2_x3_bwd_weight(nhwc(g).cuda(), nhwc(x).cuda(), dW, None, ws, accumulate=1)
    close(dW, 2 * w.grad, rtol=2e-4, atol=4e-5, what='x3 s2 wgrad accumulate')


@pytest.mark.parametrize('case', [(2, 8, 32, 64, 64, 128), (1, 12, 40, 128, 128, 128), (1, 5, 17, 128, 0, 64), (1, 16, 33, 64, 64, 64), (2, 9, 20, 256, 0, 128)])
def test_x3_conv1x1_bwd_weight(case):
    """dW [Cout][C1 + C2] of the ResidualBlock shortcut (archs/modules.py:184-187), input = the un-materialised cat of two tensors."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    assert ops.x3g_wgrad_supported(ops.X3G_PW, Co, C1 + C2)
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, 1, 1, seed=3, scale=0.2).requires_grad_(True); b = _rand(Co, seed=4).requires_grad_(True)
    g = _rand(B, Co, H, W, seed=5)
    F.conv2d(torch.cat([x1, x2], 1) if C2 else x1, w, b).backward(g)
    ws = torch.empty(ops.x3g_wgrad_workspace_floats(ops.X3G_PW, B, H, W, Co, C1 + C2), device='cuda')
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv1x1_x3_bwd_weight(nhwc(g).cuda(), Co, nhwc(x1).cuda(), C1, nhwc(x2).cuda() if C2 else None, dW, db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'x3 1x1 wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'x3 1x1 dbias {case}')


def test_x3_convt_wgrad_is_as_accurate_as_the_fp32_mfma_kernel():
    """float64 yardstick at the benchmark's second decoder level (upv7: 64 x 64 -> 128 x 128, 256 -> 128 channels, B = 4: K = 16 384 pixels
    per weight): relative L2 error of the bf16x3 kernel within 2x of the fp32-MFMA kernel's (alternating signs over the pixel splits)."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 4, 64, 64, 256, 128
    gen = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen); g = torch.randn(B, 2 * H, 2 * W, Co, device='cuda', generator=gen)
    ref = torch.zeros(Ci, Co, 2, 2, dtype=torch.float64, device='cuda')
    xd = x.double().reshape(-1, Ci)
    for a_ in range(2):
        for c_ in range(2):
            ref[:, :, a_, c_] = xd.t() @ g[:, a_::2, c_::2].double().reshape(-1, Co)
    ws = torch.empty(max(ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, H, W, Ci, Co), ops.wgrad_workspace_floats(B, H, W, Ci, Co, 4)), device='cuda')
    d3 = torch.empty(Ci, Co, 2, 2, device='cuda'); d32 = torch.empty_like(d3)
    ops.convt_x3_bwd_weight(x, g, d3, ws)
    ops.convt_bwd_weight(x, g, d32, ws)
    e3 = float((d3.double() - ref).norm() / ref.norm()); e32 = float((d32.double() - ref).norm() / ref.norm())
    print(f'convT wgrad K = {B * H * W}: rel L2 vs float64: bf16x3 {e3:.2e}, fp32-MFMA {e32:.2e}')
    assert e3 < 2.0 * e32 + 1e-8 and e3 < 2e-6
