#!/usr/bin/env python3
"""Per-level timing of the ConvTranspose2d(k2,s2) kernels (forward, backward-data, backward-weight) at the UNet shapes."""
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
for (S, Ci, Co) in ((32, 512, 256), (64, 256, 128), (128, 128, 64), (256, 64, 32)):
    x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Ci, Co, 2, 2, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    f = torch.empty(w.numel(), device='cuda'); d = torch.empty(w.numel(), device='cuda'); ops.pack_convt_weight(w, f, d)
    y = torch.empty(B, 2 * S, 2 * S, Co, device='cuda'); g = torch.randn_like(y); dx = torch.empty_like(x)
    dW = torch.empty_like(w); db = torch.empty(Co, device='cuda')
    ws = torch.empty(ops.wgrad_workspace_floats(B, S, S, Ci, Co, 4), device='cuda')
    fl = 8.0 * B * S * S * Ci * Co
    byt = 4.0 * B * S * S * (Ci + 4 * Co)
    tf = t(lambda: ops.convt_fwd(x, f, b, y, Co)); td = t(lambda: ops.convt_bwd_data(g, d, dx, mask=x, mode=1)); tw = t(lambda: ops.convt_bwd_weight(x, g, dW, ws, dbias=db))
    jobs = ops.PackJobs()
    f3 = torch.zeros(ops.x3mat_bytes(Ci, 4 * Co), dtype=torch.uint8, device='cuda'); d3 = torch.zeros(ops.x3mat_bytes(4 * Co, Ci), dtype=torch.uint8, device='cuda')
    jobs.add_x3_convt(w, f3, d3); jobs.run()
    tf3 = t(lambda: ops.convt_x3_fwd(x, f3, b, y, Co)); td3 = t(lambda: ops.convt_x3_bwd_data(g, d3, dx, mask=x, mode=1))
    print(f'{S:4d}^2 {Ci}->{Co}: bf16x3 fwd {tf3*1e3:7.1f} us {fl/tf3/1e9:6.1f} TF | dgrad {td3*1e3:7.1f} us {fl/td3/1e9:6.1f} TF', flush=True)
    print(f'{S:4d}^2 {Ci}->{Co}: fwd {tf*1e3:7.1f} us {fl/tf/1e9:6.1f} TF ({byt/tf/1e6:5.0f} GB/s) | dgrad {td*1e3:7.1f} us {fl/td/1e9:6.1f} TF | wgrad {tw*1e3:7.1f} us {fl/tw/1e9:6.1f} TF', flush=True)
