#!/usr/bin/env python3
"""Per-layer timing of the conv-stack kernels at the UNet nf=32 shapes (B crops of 4xSxS):
forward, backward-data and backward-weight of every distinct layer shape, HIP events on the
launch stream, median of `--reps` interleaved repetitions.  Prints TFLOP/s per layer and op.

    python tools/layer_bench.py [--batch 16] [--size 512] [--reps 5] [--only fwd,dgrad,wgrad]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnnp_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--size', type=int, default=512)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--only', default='fwd,dgrad,wgrad')
    ap.add_argument('--layers', default='')
    ap.add_argument('--x3', action='store_true', help='bf16x3 kernels (csrc/conv_x3.hip) for fwd / dgrad')
    ap.add_argument('--h2', action='store_true', help='fp16x2 kernels (csrc/conv_h2s.hip) for fwd / dgrad (dgrad masks as sign bits; --h2-fmask: as float32)')
    ap.add_argument('--h2-fmask', action='store_true')
    a = ap.parse_args()
    B, S = a.batch, a.size
    dev = torch.device('cuda')
    nf = 32
    ch = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
    layers = []   # name, H, C1, C2, Cout
    for l in range(5):
        layers.append((f'conv{l + 1}_1', S >> l, 8 if l == 0 else ch[l - 1], 0, ch[l]))
        layers.append((f'conv{l + 1}_2', S >> l, ch[l], 0, ch[l]))
    for i in range(6, 10):
        l = 9 - i
        layers.append((f'conv{i}_1', S >> l, ch[l], ch[l], ch[l]))
    if a.layers:
        layers = [x for x in layers if x[0] in a.layers.split(',')]
    kinds = a.only.split(',')
    res = {}
    for name, H, C1, C2, Co in layers:
        zs = 0.0 if os.environ.get('LB_ZEROS') else 1.0      # LB_ZEROS=1: all-zero operands (same instructions, less switching power: the clock effect)
        x1 = torch.randn(B, H, H, C1, device=dev) * zs
        x2 = torch.randn(B, H, H, C2, device=dev) * zs if C2 else None
        w = torch.randn(Co, C1 + C2, 3, 3, device=dev) * 0.05 * zs
        bias = torch.randn(Co, device=dev)
        f = torch.empty(w.numel(), device=dev); d = torch.empty(w.numel(), device=dev)
        ops.pack_conv_weight(w, f, d)
        y = torch.empty(B, H, H, Co, device=dev)
        g = torch.randn(B, H, H, Co, device=dev) * zs
        dx1 = torch.empty_like(x1); dx2 = torch.empty_like(x2) if C2 else None
        dW = torch.empty_like(w); db = torch.empty(Co, device=dev)
        ws = torch.empty(ops.wgrad_workspace_floats(B, H, H, Co, C1 + C2, 9), device=dev)
        ws3 = torch.empty(max(1, ops.x3_wgrad_workspace_floats(B, H, H, Co, C1 + C2)), device=dev) if (a.x3 and C1 % 32 == 0) else None
        flops = 2.0 * B * H * H * Co * (C1 + C2) * 9
        if a.x3:
            jobs = ops.PackJobs()
            f3 = torch.zeros(ops.x3_weight_bytes(C1 + C2, Co), dtype=torch.uint8, device=dev)
            d3 = torch.zeros(ops.x3_weight_bytes(Co, C1 + C2), dtype=torch.uint8, device=dev)
            jobs.add_x3(w, f3, d3, cin_pad=(C1 + C2 + 15) // 16 * 16); jobs.run()
        if a.h2:
            jobs = ops.PackJobs()
            fh = torch.zeros(ops.h2_weight_bytes(C1 + C2, Co), dtype=torch.uint8, device=dev)
            dh = torch.zeros(ops.h2_weight_bytes(Co, C1 + C2), dtype=torch.uint8, device=dev)
            sw = jobs.add_h2(w, fh, dh, cin_pad=(C1 + C2 + 15) // 16 * 16); jobs.run()
            slot = lambda *ts: [ops.amax(t, s_) for s_ in [torch.zeros(1, dtype=torch.int32, device=dev)] for t in ts][-1]
            s1 = slot(x1); s2 = slot(x2) if C2 else None; sg = slot(g)
            sy = torch.zeros(1, dtype=torch.int32, device=dev); sd1 = torch.zeros(1, dtype=torch.int32, device=dev); sd2 = torch.zeros(1, dtype=torch.int32, device=dev)
            by = torch.zeros(ops.h2_bits_words(B, H, H, Co), dtype=torch.int32, device=dev)
            b1 = torch.randint(-2 ** 31, 2 ** 31 - 1, (ops.h2_bits_words(B, H, H, C1),), dtype=torch.int32, device=dev) if C1 % 32 == 0 else None
            b2 = torch.randint(-2 ** 31, 2 ** 31 - 1, (ops.h2_bits_words(B, H, H, C2),), dtype=torch.int32, device=dev) if C2 else None
        fns = {'fwd': (lambda: ops.conv_x3_fwd(x1, x2, f3, bias, y, Co, 1)) if a.x3 else (lambda: ops.conv_fwd(x1, x2, f, bias, y, Co, 9, 1)),
               'dgrad': (lambda: ops.conv_x3_bwd_data(g, d3, dx1, mask1=x1, mode1=1, dx2=dx2, mask2=x2, mode2=1)) if (a.x3 and C1 % 32 == 0) else
                        (lambda: ops.conv_bwd_data(g, d, dx1, mask1=x1, mode1=1, dx2=dx2, mask2=x2, mode2=1)),
               'wgrad': (lambda: ops.conv_x3_bwd_weight(g, Co, x1, C1, x2, dW, db, ws3)) if (a.x3 and C1 % 32 == 0) else (lambda: ops.conv_bwd_weight(g, Co, x1, C1, x2, dW, db, 9, ws))}
        if a.h2:
            fns['fwd'] = lambda: ops.conv_h2_fwd(x1, x2, fh, sw, bias, y, Co, 1, s1, s2, amax_y=sy, bits_y=by)
            if C1 % 32 == 0:
                ws3h = torch.empty(max(1, ops.x3_wgrad_workspace_floats(B, H, H, Co, C1 + C2)), device=dev)
                fns['wgrad'] = lambda: ops.conv_h2_bwd_weight(g, sg, Co, x1, s1, C1, x2, s2, dW, db, ws3h)
                if a.h2_fmask:
                    fns['dgrad'] = lambda: ops.conv_h2_bwd_data(g, sg, dh, sw, dx1, mask1=x1, mode1=1, amax_dx1=sd1, dx2=dx2, mask2=x2, mode2=1, amax_dx2=sd2)
                else:
                    fns['dgrad'] = lambda: ops.conv_h2_bwd_data(g, sg, dh, sw, dx1, bits1=b1, mode1=1, amax_dx1=sd1, dx2=dx2, bits2=b2, mode2=1, amax_dx2=sd2)
        for k in kinds:
            fns[k](); torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); fns[k](); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            ts.sort()
            res[(name, k)] = (ts[len(ts) // 2], flops)
        del x1, x2, y, g, dx1, dx2, ws
    tot = {k: [0.0, 0.0] for k in kinds}
    print(f'{"layer":10s} ' + ' '.join(f'{k + " ms":>9s} {"TF":>6s}' for k in kinds))
    for name, H, C1, C2, Co in layers:
        line = f'{name:10s} '
        for k in kinds:
            ms, fl = res[(name, k)]
            tot[k][0] += ms; tot[k][1] += fl
            line += f'{ms:9.3f} {fl / ms / 1e9:6.1f} '
        print(line + f' H={H} Cin={C1 + C2} Cout={Co}')
    print('total      ' + ' '.join(f'{tot[k][0]:9.3f} {tot[k][1] / tot[k][0] / 1e9:6.1f}' for k in kinds))


if __name__ == '__main__':
    main()
