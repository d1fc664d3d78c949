#!/usr/bin/env python3
"""HBM-bound kernels of the path (everything except the conv stack) at BASELINE.json's sizes: time per call, achieved
ALGORITHMIC GB/s (the bytes the operation must read + write once) and its fraction of the 8 TB/s HBM3E
peak.  Inputs are resident in HBM.  Two clocks: (a) HIP events over `reps` back-to-back calls of the Python operator
(includes host launch overhead, which dominates the sub-50 us kernels); (b) the kernels' own durations from
`rocprofv3 --kernel-trace --stats` of this script, joined by tools/aux_join.py (every operator runs WARM + reps times).
    python tools/aux_bench.py [--json rows.json];  rocprofv3 --kernel-trace --stats ... -- python3 tools/aux_bench.py --json rows.json;
    python tools/aux_join.py <kernel_stats.csv> rows.json"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 8000.0      # GB/s, MI355X_MICROARCH.md


WARM = 5


def timeit(fn, reps=20):
    for _ in range(WARM):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser(); ap.add_argument('--json', default=None); a = ap.parse_args()
    from pnnp_amd import augment, isp_ops, metrics, ops, process
    from pnnp_amd.archs import NoiseFlow
    dev = 'cuda'
    rows = []

    def add(name, what, nbytes, fn, reps=20, kern=None):
        s = timeit(fn, reps)
        rows.append(dict(kernel=name, workload=what, alg_MB=nbytes / 1e6, host_us=s * 1e6, host_GBps=nbytes / s / 1e9,
                         match=kern or ['::' + name.split(' ')[0].rstrip('*')], invocations=WARM + reps))
        print(f'{name:34s} {what:46s} {nbytes/1e6:9.1f} MB {s*1e6:9.1f} us {nbytes/s/1e9:8.0f} GB/s  {100*nbytes/s/1e9/PEAK:5.1f}% of 8 TB/s', flush=True)

    g = torch.Generator(device=dev).manual_seed(0)
    # ---- Bayer pack / unpack: 8 full Sony frames per call through the C ABI (2848 x 4256 uint16 <-> 4 x 1424 x 2128 f32); one frame
    # alone is a 12-30 us kernel whose time is launch ramp, and 73 MB stay in the 256 MB Infinity Cache between repeats
    import ctypes as C
    from pnnp_amd import _lib
    L = _lib.lib()
    H, W, NF = 2848, 4256, 8
    raws = torch.randint(512, 16383, (NF, H, W), device=dev, generator=g, dtype=torch.int32).to(torch.uint16)
    raw = raws[0]
    pk = torch.empty(NF, 4, H // 2, W // 2, device=dev); back = torch.empty_like(raws)
    black = (C.c_double * 4)(512., 512., 512., 512.)
    add('pack_bayer_kernel (raw2bayer)', '8 full SID frames 2848x4256 u16 -> f32', NF * H * W * (2 + 4),
        lambda: _lib.check(L.pnnp_pack_bayer_u16(_lib.ptr(raws), NF, H, W, C.c_int64(W), C.c_int64(H * W), _lib.ptr(pk), black, C.c_double(16383.), 1, 1, _lib.stream())))
    add('unpack_bayer_kernel (bayer2raw)', '8 full SID frames f32 -> u16', NF * H * W * (4 + 2),
        lambda: _lib.check(L.pnnp_unpack_bayer_u16(_lib.ptr(pk), NF, H // 2, W // 2, _lib.ptr(back), 16383, 512, _lib.stream())))
    # ---- crop + pack + augment: 16 crops of 1024x1024 Bayer out of the frame (2 B/px in, 4 B/px out)
    ca = augment.CropAugment(dict(H=H, W=W, patch_size=512, crop_per_image=16), ways=8)
    np.random.seed(0); ca.init_random_crop_point(mode='random', raw_crop=False)
    add('crop_pack_bayer (crop+pack+aug)', '16 crops 4x512x512 from one u16 frame', 16 * 1024 * 1024 * (2 + 4), lambda: ca.crop_pack(raw, 16383, 512, True, True), kern=['::crop_aug_kernel'])
    # ---- physics noise sampler, config 3 batch
    hr = torch.rand(16, 4, 512, 512, device=dev, generator=g)
    np.random.seed(1); plist = [process.sample_params_max('SonyA7S2') for _ in range(16)]
    prm = process.pack_params(plist, dev)
    for code in ('pr',):                   # ('prq', 'pgrq' run the same kernel: time them separately, the rocprof join is per kernel name)
        fl = process.noise_flags(code, ori=False, clip=True, torch_mode=True)
        if 'g' in code:
            fl |= process.F_TORCH_TUKEY
        out = torch.empty_like(hr)
        add(f"noise_sample_kernel '{code}'", '16 crops 4x512x512 (read + write f32)', hr.numel() * 8,
            lambda fl=fl, out=out: process.noise_sample(hr, prm, fl, seed=1997, offset=0, out=out))
    # ---- L1 + clamp (loss + gradient), Adam over the UNet's parameters, max-pool fwd/bwd at level 1
    pred = torch.rand(16, 4, 512, 512, device=dev, generator=g)
    g8 = torch.empty(16, 512, 512, 8, device=dev); loss = torch.empty(17, device=dev); lws = torch.empty(128 * 16, device=dev)
    add('l1_clamp_kernel (+finish)', '16 crops: read pred, hr; write grad [..,8]', pred.numel() * 8 + g8.numel() * 4, lambda: ops.l1_clamp_loss(pred, hr, g8, loss, lws), kern=['l1_clamp_kernel', 'l1_finish'])
    n = 7760484
    p = torch.randn(n, device=dev); gr = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    add('adam_kernel', 'UNet 7.76 M parameters (4 reads + 3 writes)', n * 4 * 7, lambda: ops.adam_step(p, gr, m, v, 1e-4, 1))
    x = torch.randn(16, 512, 512, 32, device=dev); y = torch.empty(16, 256, 256, 32, device=dev)
    add('maxpool_fwd_kernel', 'level 1: 16x512x512x32 NHWC', (x.numel() + y.numel()) * 4, lambda: ops.maxpool_fwd(x, y))
    gy = torch.randn_like(y); gx = torch.empty_like(x)
    add('maxpool_bwd_kernel', 'level 1 (reads x, dy; writes dx)', (2 * x.numel() + gy.numel()) * 4, lambda: ops.maxpool_bwd(x, gy, gx, 0, 0))
    # ---- eval metrics on a full packed SID frame
    a_ = torch.rand(1, 4, 1424, 2128, device=dev, generator=g); b_ = (a_ + 0.02 * torch.randn(a_.shape, device=dev, generator=g)).clamp(0, 1)
    ic = metrics.IlluminanceCorrect()
    add('illum_* (IlluminanceCorrect)', 'full SID frame 4x1424x2128 (2 reads + r/w)', a_.numel() * 4 * 4, lambda: ic(a_, b_), kern=['illum'])
    add('psnr_ssim_* (quality_assess)', 'full SID frame, 7x7 SSIM', a_.numel() * 4 * 2, lambda: metrics.quality_assess(a_, b_), kern=['psnr_ssim'])
    # ---- NoiseFlow: sample (config 5 batch), density evaluation, fitting (reference batch)
    np.random.seed(0); torch.manual_seed(0)
    nf = NoiseFlow({'x_shape': (4, 512, 512), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'})
    with torch.no_grad():
        for k, t in nf.state_dict().items():
            if k.endswith('conv2d_3.weight'):
                t.normal_(0, 0.05)
    nf = nf.to(dev).eval()
    clean = torch.rand(12, 4, 512, 512, device=dev, generator=g) * 0.01
    add('nf_step_kernel x8 (NoiseFlow.sample)', '12 crops 4x512x512, 8 pairs (r+w each)', clean.numel() * 4 * (8 * 2 + 1), lambda: nf.sample(clean=clean, iso=6400.0), reps=5, kern=['nf_step_kernel', 'normal_fill'])
    noise = torch.randn_like(clean) * 0.01
    add('nf_fwd_step_kernel x8 (loss, eval)', '12 crops, 8 pairs (r+w each) + clean', clean.numel() * 4 * (8 * 2 + 1), lambda: nf.loss(noise=noise, clean=clean, iso=6400.0), reps=5, kern=['nf_fwd_step_kernel'])
    nft = NoiseFlow({'x_shape': (4, 64, 64), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'}).to(dev).train()
    c2 = torch.rand(256, 4, 64, 64, device=dev, generator=g) * 0.05; n2 = torch.randn_like(c2) * torch.sqrt(c2 * 2e-3 + 1e-5)
    def fit():
        nft.zero_grad(set_to_none=True)
        nll, _ = nft.loss(noise=n2, clean=c2, iso=1600.0); nll.backward()
    add('nf_tr_* (NoiseFlow loss+backward)', '256 crops 4x64x64: 8 pairs x (32 + 56) floats/px', 256 * 64 * 64 * 4 * (8 * 88 + 12), fit, reps=10, kern=['nf_tr_'])
    if a.json:
        json.dump(dict(peak_GBps=PEAK, rows=rows), open(a.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
