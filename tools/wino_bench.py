#!/usr/bin/env python3
"""Per-layer timing: Winograd F(2x2,3x3) kernel vs the direct implicit-GEMM kernel on the UNet's 3x3 layers (B=16, 512^2 crops)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B = 16
LAYERS = [(256, 64, 0, 64), (128, 64, 0, 128), (128, 128, 0, 128), (64, 128, 0, 256), (64, 256, 0, 256),
          (32, 256, 0, 512), (32, 512, 0, 512), (64, 256, 256, 256), (128, 128, 128, 128), (256, 64, 64, 64), (512, 32, 0, 64)]
for (S, C1, C2, Co) in LAYERS:
    x1 = torch.randn(B, S, S, C1, device='cuda'); x2 = torch.randn(B, S, S, C2, device='cuda') if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    y = torch.empty(B, S, S, Co, device='cuda'); y2 = torch.empty_like(y)
    pk = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w, pk, None)
    u = torch.empty(16 * Co * (C1 + C2), device='cuda'); ops.pack_conv_weight_wino(w, u, None)
    td = t(lambda: ops.conv_fwd(x1, x2, pk, b, y, Co, 9, 1))
    tw = t(lambda: ops.conv_wino_fwd(x1, x2, u, b, y2, Co, 1))
    fl = 2.0 * B * S * S * Co * (C1 + C2) * 9
    err = float((y - y2).abs().max() / y.abs().max())
    print(f'{S:4d}^2 {C1}+{C2}->{Co}: direct {td:7.3f} ms {fl/td/1e9:7.1f} TF | wino {tw:7.3f} ms {fl/tw/1e9:7.1f} TF(alg)  x{td/tw:5.2f}  relerr {err:.2e}', flush=True)
print('--- backward-data (act\' mask on the written tensor; concat layers split into two destinations)')
for (S, C1, C2, Co) in LAYERS:
    if C1 % 64 or (C2 and C2 % 64):
        continue
    g = torch.randn(B, S, S, Co, device='cuda'); w = torch.randn(Co, C1 + C2, 3, 3, device='cuda') * 0.05
    pk = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w, None, pk)
    u = torch.empty(16 * Co * (C1 + C2), device='cuda'); ops.pack_conv_weight_wino(w, None, u)
    d1 = torch.empty(B, S, S, C1, device='cuda'); e1 = torch.empty_like(d1); m1 = torch.randn_like(d1)
    d2 = torch.empty(B, S, S, C2, device='cuda') if C2 else None; e2 = torch.empty_like(d2) if C2 else None
    m2 = torch.randn_like(d2) if C2 else None
    kw = dict(mask1=None if C2 else m1, mode1=0 if C2 else 1, mask2=m2, mode2=1 if C2 else 0)
    td = t(lambda: ops.conv_bwd_data(g, pk, d1, dx2=d2, **kw))
    tw = t(lambda: ops.conv_wino_bwd_data(g, u, e1, dx2=e2, **kw))
    fl = 2.0 * B * S * S * Co * (C1 + C2) * 9
    err = float((d1 - e1).abs().max() / d1.abs().max())
    print(f'{S:4d}^2 {Co}->{C1}+{C2}: direct {td:7.3f} ms {fl/td/1e9:7.1f} TF | wino {tw:7.3f} ms {fl/tw/1e9:7.1f} TF(alg)  x{td/tw:5.2f}  relerr {err:.2e}', flush=True)
print('--- backward-weight')
for (S, C1, C2, Co) in LAYERS:
    if not ops.wino_wgrad_supported(S, S, Co, C1, C2):
        continue
    g = torch.randn(B, S, S, Co, device='cuda'); x1 = torch.randn(B, S, S, C1, device='cuda')
    x2 = torch.randn(B, S, S, C2, device='cuda') if C2 else None
    dW = torch.empty(Co, C1 + C2, 3, 3, device='cuda'); dW2 = torch.empty_like(dW); db = torch.empty(Co, device='cuda')
    ws = torch.empty(ops.wgrad_workspace_floats(B, S, S, Co, C1 + C2, 9), device='cuda')
    ws2 = torch.empty(ops.wino_wgrad_workspace_floats(B, S, S, Co, C1 + C2), device='cuda')
    td = t(lambda: ops.conv_bwd_weight(g, Co, x1, C1, x2, dW, db, 9, ws))
    tw = t(lambda: ops.conv_wino_bwd_weight(g, Co, x1, C1, x2, dW2, db, ws2))
    fl = 2.0 * B * S * S * Co * (C1 + C2) * 9
    err = float((dW - dW2).abs().max() / dW.abs().max())
    print(f'{S:4d}^2 {C1}+{C2}->{Co}: direct {td:7.3f} ms {fl/td/1e9:7.1f} TF | wino {tw:7.3f} ms {fl/tw/1e9:7.1f} TF(alg)  x{td/tw:5.2f}  relerr {err:.2e}', flush=True)
