"""NoiseFlow fitting step (trainer_NF_SID.py:116-126) at the reference's batch: 256 crops of 4x64x64 (runfiles/SonyA7S2/
NoiseFlow.yml: patch_size 64, crop_per_image 256, batch_size 1), Adam lr 2e-3.  Prints ms/step, the kernels' algorithmic
HBM bytes per step and the achieved rate.  python tools/nf_fit_bench.py [--steps 50] [--B 256] [--P 64]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=50); ap.add_argument('--B', type=int, default=256); ap.add_argument('--P', type=int, default=64)
    a = ap.parse_args()
    from pnnp_amd.archs import NoiseFlow
    np.random.seed(0); torch.manual_seed(0)
    net = NoiseFlow({'x_shape': (4, a.P, a.P), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'}).cuda().train()
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-3)
    clean = torch.rand(a.B, 4, a.P, a.P, device='cuda') * 0.05
    noise = torch.randn_like(clean) * torch.sqrt(clean * 2e-3 + 1e-5)
    def step():
        opt.zero_grad(set_to_none=True)
        nll, sd = net.loss(noise=noise, clean=clean, iso=1600.0)
        nll.backward(); opt.step()
        return nll
    t0 = time.time()
    while time.time() - t0 < 2.0:
        first = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.steps):
        last = step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    npix = a.B * a.P * a.P
    alg = npix * 4 * (8 * (32 + 56) + 4 + 4 + 4)          # floats per pixel: see DESIGN.md (NoiseFlow fitting)
    print({'ms_per_step': ms, 'crops_per_s': a.B / ms * 1e3, 'nll_first': float(first), 'nll_last': float(last),
           'alg_GB_per_step': alg / 1e9, 'alg_GBps_whole_step': alg / ms / 1e6})


if __name__ == '__main__':
    main()
