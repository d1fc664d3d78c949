"""With a library built with -DGX_STAMPS: per-wave cycle sums of gemm_x3_kernel on one ConvTranspose2d forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
B, S, Ci, Co = 16, int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 512, int(sys.argv[3]) if len(sys.argv) > 3 else 256
x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Ci, Co, 2, 2, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
y = torch.empty(B, 2 * S, 2 * S, Co, device='cuda')
jobs = ops.PackJobs()
f3 = torch.zeros(ops.x3mat_bytes(Ci, 4 * Co), dtype=torch.uint8, device='cuda'); d3 = torch.zeros(ops.x3mat_bytes(4 * Co, Ci), dtype=torch.uint8, device='cuda')
jobs.add_x3_convt(w, f3, d3); jobs.run()
for _ in range(3): ops.convt_x3_fwd(x, f3, b, y, Co)
torch.cuda.synchronize()
d = y.reshape(-1)[:256 * 8 * 8].reshape(256, 8, 8).cpu()
names = ['wait', 'barrier', 'mfma', 'epi', 'other', 'total', 'items']
for wv in (0, 4, 1, 5):
    m = d[:, wv].mean(0)
    print('wave', wv, ' '.join(f'{n}={float(v):.0f}' for n, v in zip(names, m)))
m = d.mean((0, 1))
print('all   ', ' '.join(f'{n}={float(v):.0f}' for n, v in zip(names, m)), ' per item:', ' '.join(f'{n}={float(v / m[6]):.0f}' for n, v in zip(names[:6], m[:6])))
