"""With a library built with -DWHS_STAMPS (tools/build_variant.sh whst wgrad_h2s.hip -DWHS_STAMPS; PNNP_LIB=...): per-wave cycle sums of wgrad_h2s_kernel
on one 3x3 layer (B = 16), per pixel tile.   usage: whs_stamps.py S Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
S, Ci, Co = (int(v) for v in sys.argv[1:4])
B = 16
x = torch.randn(B, S, S, Ci, device='cuda'); g = torch.randn(B, S, S, Co, device='cuda')
slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device='cuda'))
ws = torch.zeros(ops.x3_wgrad_workspace_floats(B, S, S, Co, Ci), device='cuda')
dW = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
sg, sx = slot(g), slot(x)
for _ in range(3):
    ops.conv_h2_bwd_weight(g, sg, Co, x, sx, Ci, None, None, dW, db, ws)
torch.cuda.synchronize()
d = ws[:256 * 16 * 8].reshape(256, 16, 8).cpu()
for wv, names in ((0, ['mfma', 'entry->first tile (total)', 'barrier', 'epilogue (total)']), (6, None), (12, ['stage(+wait)', 'issue', 'barrier']), (13, None)):
    if names: cn = names
    m = d[:, wv].mean(0); n = max(float(m[5]), 1.0)
    print(('consumer' if wv < 12 else 'producer'), wv, ' '.join(f'{k}={float(v) / (1.0 if "total" in k else n):.0f}' for k, v in zip(cn, m)), f'loop/tile={float(m[4]) / n:.0f} tiles={n:.0f}')
th = {(1, 1): 4, (1, 0): 3, (0, 1): 3, (0, 0): 2}[(Co % 64 != 0, Ci % 64 != 0)] if False else None
print(f'wgrad {S} {Ci}->{Co}')
