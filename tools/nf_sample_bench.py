#!/usr/bin/env python3
"""NoiseFlow.sample at config 5's shape (12 crops of 4x512x512): ms per sample() in eval and training mode, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd.archs import NoiseFlow
torch.manual_seed(0)
net = NoiseFlow({'x_shape': (4, 512, 512), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'}).cuda()
hr = torch.rand(12, 4, 512, 512, device='cuda') * 0.01
for mode in ('eval', 'train'):
    net.train(mode == 'train')
    for _ in range(3):
        net.sample_mixed(hr, 4.0, 6400, -float('inf'), 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        net.sample_mixed(hr, 4.0, 6400, -float('inf'), 1.0)
    e1.record(); torch.cuda.synchronize()
    print(f'{mode}: {e0.elapsed_time(e1) / 20:.3f} ms per sample_mixed (12 x 4 x 512 x 512)')
