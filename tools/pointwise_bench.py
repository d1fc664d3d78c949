#!/usr/bin/env python3
"""Forward / backward-data of the layers that run on csrc/gemm_x3.hip, per layer: ConvTranspose2d upv6..9 (config 3, B = 16; config 5, B = 12),
ResUnet's stride-2 convs pool1..4 and 1x1 shortcuts sc6..9 (B = 12).  ms, algorithmic TFLOP/s and the HBM floor of the layer at 5 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops


REPS = int(os.environ.get("PW_REPS", "9"))


def timeit(fn, reps=None):
    reps = reps or REPS
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
    u8 = lambda n: torch.empty(n, device=dev, dtype=torch.uint8)
    for B, tag in ((16, 'UNet B=16'), (12, 'ResUnet B=12')):
        for lvl, (h, ci, co) in enumerate([(32, 512, 256), (64, 256, 128), (128, 128, 64), (256, 64, 32)]):
            w = torch.randn(ci, co, 2, 2, device=dev) * 0.02; b = torch.randn(co, device=dev)
            f = u8(ops.x3mat_bytes(ci, 4 * co)); d = u8(ops.x3mat_bytes(4 * co, ci))
            j = ops.PackJobs(); j.add_x3_convt(w, f, d); j.run()
            x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, 2 * h, 2 * h, co, device=dev); dx = torch.empty_like(x)
            fl = 8.0 * B * h * h * ci * co; by = 4.0 * B * h * h * (ci + 4 * co)
            rows.append((f'{tag} convT upv{6 + lvl} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.convt_x3_fwd(x, f, b, y, co)),
                         timeit(lambda: ops.convt_x3_bwd_data(y, d, dx))))
        if B != 12:
            continue
        for l, (h, ci, co) in enumerate([(512, 32, 64), (256, 64, 128), (128, 128, 256), (64, 256, 512)]):
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.02; b = torch.randn(co, device=dev)
            f = u8(ops.x3mat_bytes(9 * ci, co)); d = u8(9 * ops.x3mat_bytes(co, ci))
            j = ops.PackJobs(); j.add_x3_s2(w, f, d); j.run()
            x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, h // 2, h // 2, co, device=dev); dx = torch.empty_like(x)
            fl = 2.0 * B * (h // 2) ** 2 * ci * co * 9; by = 4.0 * B * (h * h * ci + (h // 2) ** 2 * co)
            rows.append((f'{tag} s2 pool{l + 1} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.conv_s2_x3_fwd(x, f, b, y, co)),
                         timeit(lambda: ops.conv_s2_x3_bwd_data(y, d, dx))))
        for i, (h, c) in enumerate([(64, 256), (128, 128), (256, 64), (512, 32)]):
            w = torch.randn(c, 2 * c, 1, 1, device=dev) * 0.02
            f = u8(ops.x3mat_bytes(2 * c, c)); d = u8(ops.x3mat_bytes(c, 2 * c))
            j = ops.PackJobs(); j.add_x3_1x1(w, f, d); j.run()
            x1 = torch.randn(B, h, h, c, device=dev); x2 = torch.randn(B, h, h, c, device=dev); y = torch.empty(B, h, h, c, device=dev)
            d1 = torch.empty_like(x1); d2 = torch.empty_like(x2)
            fl = 2.0 * B * h * h * c * 2 * c; by = 4.0 * B * h * h * 3 * c
            rows.append((f'{tag} 1x1 sc{6 + i} {2 * c}->{c} @{h}', fl, by, timeit(lambda: ops.conv1x1_x3_fwd(x1, x2, f, None, y, c, 0)),
                         timeit(lambda: ops.conv1x1_x3_bwd_data(y, d, d1, dx2=d2))))
    print(f'{"layer":40s} {"fwd ms":>8s} {"TF":>7s} {"dgrad ms":>8s} {"TF":>7s} {"HBM floor ms":>12s}')
    tf = td = 0.0
    for name, fl, by, t1, t2 in rows:
        print(f'{name:40s} {t1:8.3f} {fl / t1 / 1e9:7.1f} {t2:8.3f} {fl / t2 / 1e9:7.1f} {by / 5e9:12.3f}')
    for key in ('UNet B=16 convT', 'ResUnet B=12 convT', 's2', '1x1'):
        s1 = sum(r[3] for r in rows if key in r[0]); s2 = sum(r[4] for r in rows if key in r[0])
        print(f'total {key:34s} {s1:8.3f} {"":7s} {s2:8.3f}')


if __name__ == '__main__':
    main()
