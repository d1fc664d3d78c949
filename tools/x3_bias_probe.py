#!/usr/bin/env python3
"""Where does the bf16x3 family lose accuracy on LONG uniform reductions?  (round 3: the fp64-yardstick test of wgrad_x3 at
K = 16 x 512 x 512 pixels came out at 5e-6 relative L2 against 8e-7 for the fp32-MFMA kernel.)  Probes, each against float64:
  * forward conv, K = 9 x Cin, uniform-scale random-sign data and all-POSITIVE data (a truncating accumulator shows up as a
    systematic negative bias that grows with the number of accumulating MFMAs);
  * backward-weight at growing pixel counts.
Prints relative L2 error, max error and the mean SIGNED relative error (bias) for bf16x3 and fp32-MFMA."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from pnnp_amd import ops


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def stats(y, ref):
    d = y.double() - ref
    return float(d.norm() / ref.norm()), float(d.abs().max() / ref.abs().max()), float((d / ref.abs().clamp_min(1e-30)).mean())


def fwd(Ci, positive):
    B, H, W, Co = 1, 16, 32, 64
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, Ci, H, W, generator=g) + 0.5
    w = (torch.rand(Co, Ci, 3, 3, generator=g) + 0.5) * 0.05
    if not positive:
        x = x * (torch.randint(0, 2, x.shape, generator=g) * 2 - 1)
        w = w * (torch.randint(0, 2, w.shape, generator=g) * 2 - 1)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    jobs = ops.PackJobs()
    f3 = torch.zeros(ops.x3_weight_bytes(Ci, Co), dtype=torch.uint8, device='cuda')
    jobs.add_x3(w.cuda(), f3, None, cin_pad=Ci); jobs.run()
    f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
    y3 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y3)
    ops.conv_x3_fwd(nhwc(x).cuda(), None, f3, None, y3, Co, 0)
    ops.conv_fwd(nhwc(x).cuda(), None, f32, None, y32, Co, 9, 0)
    s3, s32 = stats(nchw(y3).cpu(), ref), stats(nchw(y32).cpu(), ref)
    print(f'fwd Cin={Ci:4d} K={9 * Ci:5d} {"positive" if positive else "rnd-sign"}: bf16x3 L2 {s3[0]:.2e} max {s3[1]:.2e} bias {s3[2]:+.2e} | '
          f'fp32-MFMA L2 {s32[0]:.2e} max {s32[1]:.2e} bias {s32[2]:+.2e}')


def wgrad(B, H, positive):
    Ci = Co = 32
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.rand(B, H, H, Ci, device='cuda', generator=g) + 0.5
    gg = torch.rand(B, H, H, Co, device='cuda', generator=g) + 0.5
    if not positive:
        x = x * (torch.randint(0, 2, x.shape, device='cuda', generator=g) * 2 - 1)
        gg = gg * (torch.randint(0, 2, gg.shape, device='cuda', generator=g) * 2 - 1)
    ref = torch.zeros(Co, Ci, 3, 3, dtype=torch.float64, device='cuda')
    for b in range(B):
        xp = F.pad(x[b].double(), (0, 0, 1, 1, 1, 1)); gb = gg[b].double().reshape(H * H, Co)
        for ky in range(3):
            for kx in range(3):
                ref[:, :, ky, kx] += gb.t() @ xp[ky:ky + H, kx:kx + H].reshape(H * H, Ci)
    ws = torch.empty(max(ops.x3_wgrad_workspace_floats(B, H, H, Co, Ci), ops.wgrad_workspace_floats(B, H, H, Co, Ci, 9)), device='cuda')
    d3 = torch.empty(Co, Ci, 3, 3, device='cuda'); d32 = torch.empty_like(d3)
    ops.conv_x3_bwd_weight(gg, Co, x, Ci, None, d3, None, ws)
    ops.conv_bwd_weight(gg, Co, x, Ci, None, d32, None, 9, ws)
    s3, s32 = stats(d3, ref), stats(d32, ref)
    print(f'wgrad B={B:2d} H={H:3d} K={B * H * H:8d} {"positive" if positive else "rnd-sign"}: bf16x3 L2 {s3[0]:.2e} max {s3[1]:.2e} bias {s3[2]:+.2e} | '
          f'fp32-MFMA L2 {s32[0]:.2e} max {s32[1]:.2e} bias {s32[2]:+.2e}')


if __name__ == '__main__':
    for pos in (False, True):
        for ci in (32, 128, 512, 1024):
            fwd(ci, pos)
    for pos in (False, True):
        for B, H in ((1, 64), (1, 256), (4, 512), (16, 512)):
            wgrad(B, H, pos)
