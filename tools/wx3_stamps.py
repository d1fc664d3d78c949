"""With a library built with -DWX3_STAMPS: per-wave cycle sums of wgrad_x3_kernel on one 3x3 backward-weight (B=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
S, Ci, Co = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 256, 256)))
B = 16
x = torch.randn(B, S, S, Ci, device='cuda'); g = torch.randn(B, S, S, Co, device='cuda')
dW = torch.empty(Co, Ci, 3, 3, device='cuda'); db = torch.empty(Co, device='cuda')
ws = torch.empty(ops.x3_wgrad_workspace_floats(B, S, S, Co, Ci), device='cuda')
for _ in range(2): ops.conv_x3_bwd_weight(g, Co, x, Ci, None, dW, db, ws)
torch.cuda.synchronize()
d = ws[:256 * 8 * 8].reshape(256, 8, 8).cpu()
names = ['barrier', 'mfma', 'loop', 'total', 'tiles', 'loadwait']
for wv in (0, 4, 1, 5):
    m = d[:, wv].mean(0)
    print('wave', wv, ' '.join(f'{n}={float(v):.0f}' for n, v in zip(names, m)))
m = d.mean((0, 1))
print('per tile:', ' '.join(f'{n}={float(v / m[4]):.0f}' for n, v in zip(names, m) if n != 'tiles'))
e = ws[256 * 8 * 8: 256 * 8 * 8 + 8 * 32].reshape(8, 32).cpu()
for wv in (0, 4):
    print('wave', wv, 'group end times in tile 5 (cycles after its barrier):', ' '.join(f'{float(v):.0f}' for v in e[wv][:19]))
