"""With a library built with -DGX_STAMPS: per-wave cycle sums of gemm_x3_kernel on one layer.  usage: gx_stamps.py convt|s2|pw  H Cin Cout [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops
kind = sys.argv[1]; h, ci, co = (int(v) for v in sys.argv[2:5]); B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
dev = 'cuda'
u8 = lambda n: torch.empty(n, device=dev, dtype=torch.uint8)
if kind == 'convt':
    w = torch.randn(ci, co, 2, 2, device=dev) * 0.02; f = u8(ops.x3mat_bytes(ci, 4 * co)); j = ops.PackJobs(); j.add_x3_convt(w, f, None); j.run()
    x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, 2 * h, 2 * h, co, device=dev)
    run = lambda: ops.convt_x3_fwd(x, f, None, y, co); K = ci; N = 4 * co; px = B * h * h
elif kind == 's2':
    w = torch.randn(co, ci, 3, 3, device=dev) * 0.02; f = u8(ops.x3mat_bytes(9 * ci, co)); d = u8(9 * ops.x3mat_bytes(co, ci)); j = ops.PackJobs(); j.add_x3_s2(w, f, d); j.run()
    x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, h // 2, h // 2, co, device=dev)
    run = lambda: ops.conv_s2_x3_fwd(x, f, None, y, co); K = 9 * ci; N = co; px = B * h * h // 4
else:
    w = torch.randn(co, ci, 1, 1, device=dev) * 0.02; f = u8(ops.x3mat_bytes(ci, co)); j = ops.PackJobs(); j.add_x3_1x1(w, f, None); j.run()
    x = torch.randn(B, h, h, ci // 2, device=dev); x2 = torch.randn(B, h, h, ci // 2, device=dev); y = torch.empty(B, h, h, co, device=dev)
    run = lambda: ops.conv1x1_x3_fwd(x, x2, f, None, y, co, 0); K = ci; N = co; px = B * h * h
SPEC = os.environ.get('GXS', '0') == '1'
for _ in range(3): run()
torch.cuda.synchronize()
if SPEC:
    torch.cuda.synchronize()
    wgs = 256
    d = y.reshape(-1)[:wgs * 12 * 8].reshape(wgs, 12, 8).cpu()
    for wv, names in ((0, ['mfma', 'epilogue', 'barrier', 'top', 'total', 'items']), (4, None), (8, ['request', 'split', 'vmwait', 'barrier', 'total', 'items']), (9, None)):
        if names: cur_names = names
        m = d[:, wv].mean(0)
        items = float(m[5])
        print(('consumer' if wv < 8 else 'producer'), wv, ' '.join(f'{n}={float(v) / items:.0f}' for n, v in zip(cur_names[:4], m)), f'total/item={float(m[4]) / items:.0f} items={items:.0f}')
    print(f'{kind} {h} {ci}->{co} B={B}')
    sys.exit(0)
bn = 128 if (N % 128 == 0 and (px // 256) * (N // 128) * 4 >= 256 * 3) else 64
tiles = (px // 256) * ((N + bn - 1) // bn)
wgs = min(tiles, 256)
d = y.reshape(-1)[:wgs * 8 * 8].reshape(wgs, 8, 8).cpu()
names = ['wait', 'barrier', 'mfma', 'epilogue', 'other', 'total', 'items']
for wv in (0, 4, 1, 5):
    m = d[:, wv].mean(0)
    print('wave', wv, ' '.join(f'{n}={float(v):.0f}' for n, v in zip(names, m)))
m = d.mean((0, 1)); items = float(m[6]); tl = items / (K // 32)
print(f'{kind} {h} {ci}->{co} B={B}: BN={bn} tiles/CU {tl:.2f} items/CU {items:.0f}; per item:', ' '.join(f'{n}={float(v) / items:.0f}' for n, v in zip(names[:5], m)),
      f'| per tile: epilogue={float(m[3]) / tl:.0f} total={float(m[5]) / tl:.0f}  (MFMA cycles per item and SIMD: {3072 * bn // 128})')
