"""Sampler timing by regime: python tools/noise_bench.py [--lib path/to/other/libpnnp_hip.so]
(the --lib form times an older build on the same box: copy it over the in-tree library of the box's snapshot first)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pnnp_amd import process


def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None      # e.g. pr:dark
    g = torch.Generator(device='cuda').manual_seed(0)
    base = torch.rand(16, 4, 512, 512, device='cuda', generator=g)
    np.random.seed(1); plist = [process.sample_params_max('SonyA7S2') for _ in range(16)]
    prm = process.pack_params(plist, 'cuda')
    out = torch.empty_like(base)
    for code in ('pr', 'prq', 'pgrq'):
        for name, scale in (('bright', 1.0), ('mid', 0.1), ('dark', 0.01)):
            if only and only != f'{code}:{name}':
                continue
            hr = base * scale
            fl = process.noise_flags(code, ori=False, clip=True, torch_mode=True)
            if 'g' in code and 'p' in code: fl |= process.F_TORCH_TUKEY
            us = timeit(lambda: process.noise_sample(hr, prm, fl, seed=1997, offset=0, out=out))
            print(f'{code:5s} {name:7s} {us:8.1f} us  {hr.numel()*8/us/1e3:7.0f} GB/s', flush=True)


main()
