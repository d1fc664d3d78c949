"""With a library built with -DX3S_STAMPS: per-wave cycle sums of igemm_x3s_kernel on one 3x3 layer (B = 16), per chunk (= 3 items).
usage: x3s_stamps.py S Cin Cout [fwd|dgrad]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
S, Ci, Co = (int(v) for v in sys.argv[1:4]); mode = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
B = 16
x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
wx = torch.empty(ops.x3_weight_bytes(Ci, Co), dtype=torch.uint8, device='cuda'); wd = torch.empty(ops.x3_weight_bytes(Co, Ci), dtype=torch.uint8, device='cuda')
jobs = ops.PackJobs(); jobs.add_x3(w, wx, wd, cin_pad=(Ci + 15) // 16 * 16); jobs.run()
if mode == 'fwd':
    y = torch.empty(B, S, S, Co, device='cuda')
    run = lambda: ops.conv_x3_fwd(x, None, wx, b, y, Co, 1); out = y; N = Co
else:
    g = torch.randn(B, S, S, Co, device='cuda'); dx = torch.empty(B, S, S, Ci, device='cuda'); mask = torch.randn(B, S, S, Ci, device='cuda')
    run = lambda: ops.conv_x3_bwd_data(g, wd, dx, mask1=mask, mode1=1); out = dx; N = Ci
for _ in range(3): run()
torch.cuda.synchronize()
d = out.reshape(-1)[:256 * 12 * 8].reshape(256, 12, 8).cpu()
for wv, names in ((0, ['mfma', 'epilogue', 'barrier']), (4, None), (8, ['work', 'vmwait', 'barrier']), (9, None)):
    if names: cn = names
    m = d[:, wv].mean(0); n = float(m[5])
    print(('consumer' if wv < 8 else 'producer'), wv, ' '.join(f'{k}={float(v) / n:.0f}' for k, v in zip(cn, m)), f'total/chunk={float(m[4]) / n:.0f} chunks={n:.0f}')
bn = 64 if (N >= 64 and (S // 32) * (S // 16) * B * (N // 64) * 4 >= 256 * 3) else 32
print(f'{mode} {S} {Ci}->{Co}: BN={bn}; MFMA cycles per chunk and SIMD: {3 * 4608 * bn // 64}')
