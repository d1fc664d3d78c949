#!/usr/bin/env python3
"""Weight gradients of the pointwise / strided layers, bf16x3 (csrc/wgrad_x3g.hip) next to fp32-MFMA (csrc/wgrad.hip), at the shapes of
config 3 (UNet: ConvTranspose2d upv6..9, B = 16) and config 5 (ResUnet: + stride-2 pool1..4 and 1x1 shortcuts sc6..9, B = 12)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops


def timeit(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def main():
    dev = 'cuda'
    rows = []
    for B, tag in ((16, 'UNet B=16'), (12, 'ResUnet B=12')):
        for lvl, (h, ci, co) in enumerate([(32, 512, 256), (64, 256, 128), (128, 128, 64), (256, 64, 32)]):
            x = torch.randn(B, h, h, ci, device=dev); g = torch.randn(B, 2 * h, 2 * h, co, device=dev)
            dW = torch.empty(ci, co, 2, 2, device=dev); db = torch.empty(co, device=dev)
            ws = torch.empty(max(ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, h, h, ci, co), ops.wgrad_workspace_floats(B, h, h, ci, co, 4)), device=dev)
            fl = 8.0 * B * h * h * ci * co
            t3 = timeit(lambda: ops.convt_x3_bwd_weight(x, g, dW, ws, dbias=db)); t32 = timeit(lambda: ops.convt_bwd_weight(x, g, dW, ws, dbias=db))
            rows.append((f'{tag} convT upv{6 + lvl} {ci}->{co} @{h}', fl, t3, t32))
        if B == 12:
            for l, (h, ci, co) in enumerate([(512, 32, 64), (256, 64, 128), (128, 128, 256), (64, 256, 512)]):
                x = torch.randn(B, h, h, ci, device=dev); g = torch.randn(B, h // 2, h // 2, co, device=dev)
                dW = torch.empty(co, ci, 3, 3, device=dev); db = torch.empty(co, device=dev)
                ws = torch.empty(max(ops.x3g_wgrad_workspace_floats(ops.X3G_S2, B, h // 2, h // 2, co, ci), ops.wgrad_workspace_floats(B, h // 2, h // 2, co, ci, 18)), device=dev)
                fl = 2.0 * B * (h // 2) ** 2 * ci * co * 9
                t32 = timeit(lambda: ops.conv_s2_bwd_weight(g, x, dW, db, ws))
                t3 = timeit(lambda: ops.conv_s2_x3_bwd_weight(g, x, dW, db, ws)) if ops.x3g_wgrad_supported(ops.X3G_S2, co, ci) else float('nan')
                rows.append((f'{tag} s2 pool{l + 1} {ci}->{co} @{h}', fl, t3, t32))
            for i, (h, c) in enumerate([(64, 256), (128, 128), (256, 64), (512, 32)]):
                x1 = torch.randn(B, h, h, c, device=dev); x2 = torch.randn(B, h, h, c, device=dev); g = torch.randn(B, h, h, c, device=dev)
                dW = torch.empty(c, 2 * c, 1, 1, device=dev)
                ws = torch.empty(max(ops.x3g_wgrad_workspace_floats(ops.X3G_PW, B, h, h, c, 2 * c), ops.wgrad_workspace_floats(B, h, h, c, 2 * c, 1)), device=dev)
                fl = 2.0 * B * h * h * c * 2 * c
                t32 = timeit(lambda: ops.conv_bwd_weight(g, c, x1, c, x2, dW, None, 1, ws))
                t3 = timeit(lambda: ops.conv1x1_x3_bwd_weight(g, c, x1, c, x2, dW, None, ws)) if ops.x3g_wgrad_supported(ops.X3G_PW, c, 2 * c) else float('nan')
                rows.append((f'{tag} 1x1 sc{6 + i} {2 * c}->{c} @{h}', fl, t3, t32))
    print(f'{"layer":44s} {"x3 ms":>8s} {"TF":>7s} {"fp32 ms":>8s} {"TF":>7s}')
    for name, fl, t3, t32 in rows:
        print(f'{name:44s} {t3:8.3f} {fl / t3 / 1e9:7.1f} {t32:8.3f} {fl / t32 / 1e9:7.1f}')


if __name__ == '__main__':
    main()
