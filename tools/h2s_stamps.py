"""With a library built with -DH2S_STAMPS (tools/build_variant.sh h2st conv_h2s.hip -DH2S_STAMPS; PNNP_LIB=...): per-wave cycle sums of
igemm_h2s_kernel on one 3x3 layer (B = 16), per chunk.   usage: h2s_stamps.py S Cin Cout [fwd|fwdres|dgrad|dgradf]   (fwdres: forward with a residual = the general epilogue; dgradf: float32 masks)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
S, Ci, Co = (int(v) for v in sys.argv[1:4]); mode = sys.argv[4] if len(sys.argv) > 4 else 'fwd'
B = 16
x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
wx = torch.empty(ops.h2_weight_bytes(Ci, Co), dtype=torch.uint8, device='cuda'); wd = torch.empty(ops.h2_weight_bytes(Co, Ci), dtype=torch.uint8, device='cuda')
jobs = ops.PackJobs(); sw = jobs.add_h2(w, wx, wd, cin_pad=(Ci + 15) // 16 * 16); jobs.run()
slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device='cuda'))
if mode in ('fwd', 'fwdres'):
    y = torch.empty(B, S, S, Co, device='cuda'); sx = slot(x); sy = torch.zeros(1, dtype=torch.int32, device='cuda')
    bits = torch.zeros(ops.h2_bits_words(B, S, S, Co), dtype=torch.int32, device='cuda')
    res = torch.randn(B, S, S, Co, device='cuda') if mode == 'fwdres' else None
    run = lambda: ops.conv_h2_fwd(x, None, wx, sw, b, y, Co, 1 if res is None else 0, sx, amax_y=sy, bits_y=bits if res is None else None, residual=res); out = y; N = Co
else:
    g = torch.randn(B, S, S, Co, device='cuda'); dx = torch.empty(B, S, S, Ci, device='cuda'); mask = torch.randn(B, S, S, Ci, device='cuda')
    sg = slot(g); sd = torch.zeros(1, dtype=torch.int32, device='cuda')
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (ops.h2_bits_words(B, S, S, Ci),), dtype=torch.int32, device='cuda')
    if mode == 'dgradf':
        run = lambda: ops.conv_h2_bwd_data(g, sg, wd, sw, dx, mask1=mask, mode1=1, amax_dx1=sd)
    else:
        run = lambda: ops.conv_h2_bwd_data(g, sg, wd, sw, dx, bits1=bits, mode1=1, amax_dx1=sd)
    out = dx; N = Ci
for _ in range(3): run()
torch.cuda.synchronize()
d = out.reshape(-1)[:256 * 12 * 8].reshape(256, 12, 8).cpu()
for wv, names in ((0, ['mfma', 'epilogue', 'barrier']), (4, None), (8, ['work', 'vmwait', 'barrier']), (9, None)):
    if names: cn = names
    m = d[:, wv].mean(0); n = float(m[5])
    print(('consumer' if wv < 8 else 'producer'), wv, ' '.join(f'{k}={float(v) / n:.0f}' for k, v in zip(cn, m)), f'total/chunk={float(m[4]) / n:.0f} chunks={n:.0f}')
bn = 64 if (N >= 64 and (S // 32) * (S // 16) * B * (N // 64) * 4 >= 256 * 3) else 32
print(f'{mode} {S} {Ci}->{Co}: BN={bn}; MFMA cycles per chunk and SIMD: {2 * 14 * 16 * 16 * bn // 64}')
