#!/usr/bin/env python3
"""Cycles per step inside one K-chunk of the Winograd kernel (needs a build with -DWINO_STEPTIME=1):
    PNNP_HIPCC_EXTRA="-DWINO_STEPTIME=1 -DPNNP_WINO_DEBUG=1" python tools/build.py && python tools/wino_steps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops, _lib
L = _lib.lib()
B, S, C1, Co = 16, 64, 256, 256
x1 = torch.randn(B, S, S, C1, device='cuda'); w = torch.randn(Co, C1, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
y = torch.empty(B, S, S, Co, device='cuda'); u = torch.empty(16 * Co * C1, device='cuda'); ops.pack_conv_weight_wino(w, u, None)
wgs = B * (S // 16) ** 2 * (Co // 64)
dbg = torch.zeros(wgs * 21, dtype=torch.int64, device='cuda')
for _ in range(3): ops.conv_wino_fwd(x1, None, u, b, y, Co, 1)
torch.cuda.synchronize()
L.pnnp_wino_set_debug(C.c_void_p(dbg.data_ptr())); ops.conv_wino_fwd(x1, None, u, b, y, Co, 1); torch.cuda.synchronize(); L.pnnp_wino_set_debug(C.c_void_p(0))
st = dbg[wgs * 4:].view(wgs, 17).double()
d = (st[:, 1:] - st[:, :-1])
print('mean cycles per step (16 steps of chunk 2; 4 MFMAs = 256 cycles each):')
print(' '.join('%5.0f' % v for v in d.mean(0).tolist()), ' total %.0f' % float((st[:, 16] - st[:, 0]).mean()))
print('median:'); print(' '.join('%5.0f' % v for v in d.median(0).values.tolist()))
