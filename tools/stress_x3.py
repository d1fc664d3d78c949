#!/usr/bin/env python3
"""Random-shape stress of the bf16x3 convolution entry points: run with the library under test and (PNNP_LIB=...) with another build of it,
each writing its outputs to a file; `--compare a.pt b.pt` reports the largest relative difference per operation.  The two builds must agree
to float32 rounding (same arithmetic, different accumulation schedule at most)."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(out, n=48, seed=1234):
    from pnnp_amd import ops
    rnd = random.Random(seed)
    res = {}
    dev = 'cuda'
    for it in range(n):
        B = rnd.choice([1, 2, 3]); H = rnd.randint(3, 70); W = rnd.randint(3, 90)
        C1 = rnd.choice([8, 16, 24, 32, 64, 96, 128]); C2 = rnd.choice([0, 0, C1]) if C1 % 16 == 0 else 0
        Co = rnd.choice([32, 64, 96, 128, 256])
        g = torch.Generator(device=dev).manual_seed(seed + it)
        x1 = torch.randn(B, H, W, C1, device=dev, generator=g); x2 = torch.randn(B, H, W, C2, device=dev, generator=g) if C2 else None
        w = torch.randn(Co, C1 + C2, 3, 3, device=dev, generator=g) * 0.1; b = torch.randn(Co, device=dev, generator=g)
        f = torch.zeros(ops.x3_weight_bytes((C1 + C2 + 15) // 16 * 16 if not C2 else C1 + C2, Co), dtype=torch.uint8, device=dev)
        d = torch.zeros(ops.x3_weight_bytes(Co, C1 + C2), dtype=torch.uint8, device=dev) if C1 % 32 == 0 else None      # (two destinations: each a multiple of 32 channels)
        jobs = ops.PackJobs(); jobs.add_x3(w, f, d, cin_pad=(C1 + C2 + 15) // 16 * 16 if not C2 else None); jobs.run()
        act = rnd.choice([0, 1, 2])
        y = torch.full((B, H, W, Co), float('nan'), device=dev)
        ops.conv_x3_fwd(x1, x2, f, b, y, Co, act)
        res[f'{it} fwd {B}x{H}x{W} {C1}+{C2}->{Co} act{act}'] = y.cpu()
        if H % 2 == 0 and W % 2 == 0 and not C2:
            yp = torch.full((B, H, W, Co), float('nan'), device=dev); p = torch.full((B, H // 2, W // 2, Co), float('nan'), device=dev)
            c = torch.full((B, H // 2, W // 2, Co), 255, dtype=torch.uint8, device=dev)
            ops.conv_x3_fwd_pool(x1, None, f, b, yp, p, c, Co, 1)
            res[f'{it} pool'] = torch.cat([p.flatten().cpu(), c.flatten().float().cpu()])
        if d is not None:
            gy = torch.randn(B, H, W, Co, device=dev, generator=g)
            m1 = torch.randn(B, H, W, C1, device=dev, generator=g)
            d1 = torch.full((B, H, W, C1), float('nan'), device=dev)
            d2 = torch.full((B, H, W, C2), float('nan'), device=dev) if C2 else None
            m2 = torch.randn(B, H, W, C2, device=dev, generator=g) if C2 else None
            mode = rnd.choice([0, 1, 2])
            ops.conv_x3_bwd_data(gy, d, d1, mask1=m1 if mode else None, mode1=mode, dx2=d2, mask2=m2 if C2 else None, mode2=1 if C2 else 0)
            res[f'{it} dgrad mode{mode}'] = torch.cat([d1.flatten().cpu()] + ([d2.flatten().cpu()] if C2 else []))
            if C1 % 32 == 0 and Co % 32 == 0:
                ws = torch.empty(ops.x3_wgrad_workspace_floats(B, H, W, Co, C1 + C2), device=dev)
                dW = torch.full((Co, C1 + C2, 3, 3), float('nan'), device=dev); db = torch.full((Co,), float('nan'), device=dev)
                ops.conv_x3_bwd_weight(gy, Co, x1, C1, x2, dW, db, ws)
                res[f'{it} wgrad'] = torch.cat([dW.flatten().cpu(), db.cpu()])
    u8 = lambda n: torch.zeros(n, device=dev, dtype=torch.uint8)
    for it in range(n // 2):                                     # the pointwise GEMMs: ConvTranspose2d, 1x1 (two inputs), stride-2 3x3
        B = rnd.choice([1, 2]); h = rnd.randint(2, 40); w_ = rnd.randint(2, 50)
        ci = rnd.choice([32, 64, 128, 256]); co = rnd.choice([32, 64, 128])
        g = torch.Generator(device=dev).manual_seed(seed + 1000 + it)
        wt = torch.randn(ci, co, 2, 2, device=dev, generator=g) * 0.1; b = torch.randn(co, device=dev, generator=g)
        f = u8(ops.x3mat_bytes(ci, 4 * co)); d = u8(ops.x3mat_bytes(4 * co, ci))
        j = ops.PackJobs(); j.add_x3_convt(wt, f, d); j.run()
        x = torch.randn(B, h, w_, ci, device=dev, generator=g); y = torch.full((B, 2 * h, 2 * w_, co), float('nan'), device=dev)
        ops.convt_x3_fwd(x, f, b, y, co)
        m = torch.randn(B, h, w_, ci, device=dev, generator=g); dx = torch.full((B, h, w_, ci), float('nan'), device=dev)
        ops.convt_x3_bwd_data(y, d, dx, mask=m, mode=1)
        res[f'{it} convt {B}x{h}x{w_} {ci}->{co}'] = torch.cat([y.flatten().cpu(), dx.flatten().cpu()])
        w1 = torch.randn(co, 2 * ci, 1, 1, device=dev, generator=g) * 0.1
        f = u8(ops.x3mat_bytes(2 * ci, co)); d = u8(ops.x3mat_bytes(co, 2 * ci))
        j = ops.PackJobs(); j.add_x3_1x1(w1, f, d); j.run()
        x2 = torch.randn(B, h, w_, ci, device=dev, generator=g); y1 = torch.full((B, h, w_, co), float('nan'), device=dev)
        ops.conv1x1_x3_fwd(x, x2, f, None, y1, co, 0)
        d1 = torch.full((B, h, w_, ci), float('nan'), device=dev); d2 = torch.full((B, h, w_, ci), float('nan'), device=dev)
        ops.conv1x1_x3_bwd_data(y1, d, d1, dx2=d2)
        res[f'{it} pw1x1'] = torch.cat([y1.flatten().cpu(), d1.flatten().cpu(), d2.flatten().cpu()])
        H2, W2 = 2 * h, 2 * w_
        w3 = torch.randn(co, ci, 3, 3, device=dev, generator=g) * 0.1
        f = u8(ops.x3mat_bytes(9 * ci, co)); d = u8(9 * ops.x3mat_bytes(co, ci))
        j = ops.PackJobs(); j.add_x3_s2(w3, f, d); j.run()
        xs = torch.randn(B, H2, W2, ci, device=dev, generator=g); ys = torch.full((B, h, w_, co), float('nan'), device=dev)
        ops.conv_s2_x3_fwd(xs, f, b, ys, co)
        dxs = torch.zeros(B, H2, W2, ci, device=dev)
        ops.conv_s2_x3_bwd_data(ys, d, dxs)
        res[f'{it} s2'] = torch.cat([ys.flatten().cpu(), dxs.flatten().cpu()])
    torch.save(res, out)
    print('wrote', out, len(res), 'results')


def compare(a, b):
    ra, rb = torch.load(a), torch.load(b)
    worst = {}
    for k in ra:
        x, y = ra[k].double(), rb[k].double()
        assert x.shape == y.shape and not x.isnan().any() and not y.isnan().any(), k
        rel = float((x - y).abs().max() / (x.abs().max() + 1e-30))
        op = k.split()[1]
        if rel > worst.get(op, (0, ''))[0]: worst[op] = (rel, k)
        if rel > 2e-5: print('LARGE', k, rel)
    for op, (rel, k) in sorted(worst.items()): print(f'{op:6s} worst max-abs difference / max-abs value = {rel:.3e}   ({k})')


if __name__ == '__main__':
    if sys.argv[1] == '--compare': compare(sys.argv[2], sys.argv[3])
    else: run(sys.argv[1])
