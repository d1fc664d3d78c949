"""With a library built with -DX3_STAMPS: per-wave cycle sums of igemm_x3_kernel<64> on one 3x3 forward layer (B=16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
S, Ci, Co = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 256, 256)))
B = 16
x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
wx = torch.empty(ops.x3_weight_bytes(Ci, Co), dtype=torch.uint8, device='cuda')
jobs = ops.PackJobs(); jobs.add_x3(w, wx, None, cin_pad=(Ci + 15) // 16 * 16); jobs.run()
y = torch.empty(B, S, S, Co, device='cuda')
for _ in range(3): ops.conv_x3_fwd(x, None, wx, b, y, Co, 1)
torch.cuda.synchronize()
d = y.reshape(-1)[:256 * 8 * 16].reshape(256, 8, 16)[:, :, :14].cpu()
names = ['wait', 'barrier', 'row0', 'top', 'total', 'row1', 'row2', 'epiA', 'epiB', 'e_setup', 'e_r0', 'e_r1', 'e_r2', 'e_r3']
for wv in (0, 4, 1, 5, 2, 6, 3, 7):
    m = d[:, wv].mean(0)
    print('wave', wv, ' '.join(f'{n}={float(v):.0f}' for n, v in zip(names, m)))
bn = 64 if (Co >= 64 and (S // 32) * (S // 16) * B * (Co // 64) * 4 >= 256 * 3) else 32
tiles = (S // 32) * (S // 16) * B * (Co // bn) / 256.0
items = tiles * (Ci // 16) * 3
m = d.mean((0, 1))
print(f'items per CU {items:.0f};  per item:', ' '.join(f'{n}={float(v) / items:.0f}' for n, v in zip(names, m)), ' (MFMA cycles per item and SIMD: %d; BN = %d)' % (4608 * bn // 64, bn))
