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


@pytest.mark.parametrize('B,H,W,Cin,Cout', [(2, 16, 32, 64, 32), (1, 8, 40, 512, 256), (3, 13, 21, 128, 64), (2, 32, 32, 256, 128)])
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
    assert not ops.gemm_h2_supported(48, 128) and not ops.gemm_h2_supported(64, 32)
