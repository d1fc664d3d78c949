#!/usr/bin/env python3
"""Where a Winograd workgroup spends its cycles (prologue / main loop / epilogue), from the kernel's own clock64() stamps
(pnnp_wino_set_debug, present only in a profiling build:  PNNP_HIPCC_EXTRA=-DPNNP_WINO_DEBUG=1 python tools/build.py --force).
Also cycles per K-chunk iteration against the 64 x 64 = 4096 MFMA cycles it contains."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops, _lib

L = _lib.lib()
B = 16
LAYERS = [(256, 64, 0, 64), (128, 128, 0, 128), (64, 256, 0, 256), (32, 512, 0, 512), (64, 256, 256, 256), (256, 64, 64, 64)]
for (S, C1, C2, Co) in LAYERS:
    x1 = torch.randn(B, S, S, C1, device='cuda'); x2 = torch.randn(B, S, S, C2, device='cuda') if C2 else None
    w = torch.randn(Co, C1 + C2, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    y = torch.empty(B, S, S, Co, device='cuda')
    u = torch.empty(16 * Co * (C1 + C2), device='cuda'); ops.pack_conv_weight_wino(w, u, None)
    wgs = B * (S // 16) ** 2 * (Co // 64)
    dbg = torch.zeros(wgs * 4, dtype=torch.int64, device='cuda')
    for _ in range(3): ops.conv_wino_fwd(x1, x2, u, b, y, Co, 1)
    torch.cuda.synchronize()
    L.pnnp_wino_set_debug(C.c_void_p(dbg.data_ptr()))
    ops.conv_wino_fwd(x1, x2, u, b, y, Co, 1)
    torch.cuda.synchronize()
    L.pnnp_wino_set_debug(C.c_void_p(0))
    d = dbg.view(wgs, 4).double()
    pro, loop, epi = (d[:, 1] - d[:, 0]), (d[:, 2] - d[:, 1]), (d[:, 3] - d[:, 2])
    tot = d[:, 3] - d[:, 0]
    nch = (C1 + C2) // 8
    span = float(d[:, 3].max() - d[:, 0].min())
    print(f'{S:4d}^2 {C1}+{C2}->{Co}: WGs {wgs:6d} ({wgs/256:.1f}/CU)  per-WG cycles: prologue {pro.mean():7.0f}  loop {loop.mean():8.0f} '
          f'({loop.mean()/nch:6.0f}/chunk vs 4096 MFMA)  epilogue {epi.mean():7.0f}  total {tot.mean():8.0f} | '
          f'share pro {100*pro.sum()/tot.sum():4.1f}% loop {100*loop.sum()/tot.sum():4.1f}% epi {100*epi.sum()/tot.sum():4.1f}% | '
          f'kernel span {span:9.0f} ticks, sum(WG)/256/span = {float(tot.sum())/256/span:.2f}', flush=True)
