#!/usr/bin/env python3
"""Forward / backward-data of the layers that run on csrc/gemm_x3.hip, per layer: ConvTranspose2d upv6..9 (config 3, B = 16; config 5, B = 12),
ResUnet's stride-2 convs pool1..4 and 1x1 shortcuts sc6..9 (B = 12).  ms, algorithmic TFLOP/s and the HBM floor of the layer at 5 TB/s.
--h2: the same layers on the fp16x2 kernel (csrc/gemm_h2s.hip); layers it does not take (N % 64) print nan."""
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
    H2 = '--h2' in sys.argv
    nan = float('nan')
    slot = lambda t: ops.amax(t, torch.zeros(1, dtype=torch.int32, device=dev))
    hb = ops.h2mat_bytes if H2 else ops.x3mat_bytes
    rows = []
    u8 = lambda n: torch.empty(n, device=dev, dtype=torch.uint8)
    for B, tag in ((16, 'UNet B=16'), (12, 'ResUnet B=12')):
        for lvl, (h, ci, co) in enumerate([(32, 512, 256), (64, 256, 128), (128, 128, 64), (256, 64, 32)]):
            w = torch.randn(ci, co, 2, 2, device=dev) * 0.02; b = torch.randn(co, device=dev)
            f = u8(hb(ci, 4 * co)); d = u8(hb(4 * co, ci))
            j = ops.PackJobs(); sw = j.add_h2_convt(w, f, d) if H2 else j.add_x3_convt(w, f, d); j.run()
            x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, 2 * h, 2 * h, co, device=dev); dx = torch.empty_like(x)
            fl = 8.0 * B * h * h * ci * co; by = 4.0 * B * h * h * (ci + 4 * co)
            if H2:
                sx = slot(x); ops.convt_h2_fwd(x, sx, f, sw, b, y, co); sy = slot(y)
                rows.append((f'{tag} convT upv{6 + lvl} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.convt_h2_fwd(x, sx, f, sw, b, y, co)),
                             timeit(lambda: ops.convt_h2_bwd_data(y, sy, d, sw, dx))))
            else:
                rows.append((f'{tag} convT upv{6 + lvl} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.convt_x3_fwd(x, f, b, y, co)),
                             timeit(lambda: ops.convt_x3_bwd_data(y, d, dx))))
        if B != 12:
            continue
        for l, (h, ci, co) in enumerate([(512, 32, 64), (256, 64, 128), (128, 128, 256), (64, 256, 512)]):
            w = torch.randn(co, ci, 3, 3, device=dev) * 0.02; b = torch.randn(co, device=dev)
            ok = not H2 or (ops.gemm_h2_supported(ci, co) and ops.gemm_h2_supported(co, ci))
            f = u8(hb(9 * ci, co)); d = u8(9 * hb(co, ci))
            j = ops.PackJobs(); sw = (j.add_h2_s2(w, f, d) if ok else None) if H2 else j.add_x3_s2(w, f, d); j.run()
            x = torch.randn(B, h, h, ci, device=dev); y = torch.empty(B, h // 2, h // 2, co, device=dev); dx = torch.empty_like(x)
            fl = 2.0 * B * (h // 2) ** 2 * ci * co * 9; by = 4.0 * B * (h * h * ci + (h // 2) ** 2 * co)
            if H2 and not ok:
                rows.append((f'{tag} s2 pool{l + 1} {ci}->{co} @{h}', fl, by, nan, nan))
            elif H2:
                sx = slot(x); ops.conv_s2_h2_fwd(x, sx, f, sw, b, y, co, 0); sy = slot(y)
                rows.append((f'{tag} s2 pool{l + 1} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.conv_s2_h2_fwd(x, sx, f, sw, b, y, co, 0)),
                             timeit(lambda: ops.conv_s2_h2_bwd_data(y, sy, d, sw, dx))))
            else:
                rows.append((f'{tag} s2 pool{l + 1} {ci}->{co} @{h}', fl, by, timeit(lambda: ops.conv_s2_x3_fwd(x, f, b, y, co)),
                             timeit(lambda: ops.conv_s2_x3_bwd_data(y, d, dx))))
        for i, (h, c) in enumerate([(64, 256), (128, 128), (256, 64), (512, 32)]):
            w = torch.randn(c, 2 * c, 1, 1, device=dev) * 0.02
            ok = not H2 or (ops.gemm_h2_supported(c, c) and ops.gemm_h2_supported(c, 2 * c))
            f = u8(hb(2 * c, c)); d = u8(hb(c, 2 * c))
            j = ops.PackJobs(); sw = (j.add_h2_1x1(w, f, d) if ok else None) if H2 else j.add_x3_1x1(w, f, d); j.run()
            x1 = torch.randn(B, h, h, c, device=dev); x2 = torch.randn(B, h, h, c, device=dev); y = torch.empty(B, h, h, c, device=dev)
            d1 = torch.empty_like(x1); d2 = torch.empty_like(x2)
            fl = 2.0 * B * h * h * c * 2 * c; by = 4.0 * B * h * h * 3 * c
            if H2 and not ok:
                rows.append((f'{tag} 1x1 sc{6 + i} {2 * c}->{c} @{h}', fl, by, nan, nan))
            elif H2:
                s1, s2 = slot(x1), slot(x2); ops.conv1x1_h2_fwd(x1, s1, x2, s2, f, sw, None, y, c, 0); sy = slot(y)
                rows.append((f'{tag} 1x1 sc{6 + i} {2 * c}->{c} @{h}', fl, by, timeit(lambda: ops.conv1x1_h2_fwd(x1, s1, x2, s2, f, sw, None, y, c, 0)),
                             timeit(lambda: ops.conv1x1_h2_bwd_data(y, sy, d, sw, d1, dx2=d2))))
            else:
                rows.append((f'{tag} 1x1 sc{6 + i} {2 * c}->{c} @{h}', fl, by, timeit(lambda: ops.conv1x1_x3_fwd(x1, x2, f, None, y, c, 0)),
                             timeit(lambda: ops.conv1x1_x3_bwd_data(y, d, d1, dx2=d2))))
    print(f'{"layer":40s} {"fwd ms":>8s} {"TF":>7s} {"dgrad ms":>8s} {"TF":>7s} {"HBM floor ms":>12s}')
    tf = td = 0.0
    for name, fl, by, t1, t2 in rows:
        print(f'{name:40s} {t1:8.3f} {fl / t1 / 1e9:7.1f} {t2:8.3f} {fl / t2 / 1e9:7.1f} {by / 5e9:12.3f}')
    for key in ('UNet B=16 convT', 'ResUnet B=12 convT', 's2', '1x1'):
        s1 = sum(r[3] for r in rows if key in r[0] and r[3] == r[3]); s2 = sum(r[4] for r in rows if key in r[0] and r[4] == r[4])
        print(f'total {key:34s} {s1:8.3f} {"":7s} {s2:8.3f}')


if __name__ == '__main__':
    main()
