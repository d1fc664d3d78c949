#!/usr/bin/env python3
"""BASELINE config 2: eval-mode UNet forward on SID-sized full frames (4x1424x2128) and 512^2 crops,
1 x MI355X.  Prints ms/frame and TFLOP/s (fwd 96.771 GFLOP per 4x512x512 crop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd.archs import UNetSeeInDark, ResUnet, initialize_weights

def run(cls, gf, B, H, W, reps=10):
    torch.manual_seed(0)
    net = cls(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda().eval()
    x = torch.rand(B, 4, H, W, device='cuda')
    with torch.no_grad():
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.5:      # warm-up long enough for the clocks to ramp after the idle set-up
            net(x); torch.cuda.synchronize()
        net(x); torch.cuda.synchronize(); t0 = time.perf_counter(); net(x); torch.cuda.synchronize()
        reps = max(reps, int(0.5 / (time.perf_counter() - t0)))
        t0 = time.perf_counter()
        for _ in range(reps): y = net(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    fl = gf * 1e9 * B * H * W / (512 * 512)
    print(f'{cls.__name__:14s} B={B} {H}x{W}: {dt*1e3:7.2f} ms/forward  {fl/dt/1e12:6.1f} TFLOP/s  {B*H*W/(512*512)/dt:7.1f} crop-equiv/s')

if __name__ == '__main__':
    from pnnp_amd import ops
    split = int(os.environ.get('PNNP_SPLIT', '0'))          # > 0: force pnnp_set_persistent_split (0 = what the engine chooses)
    if split:
        ops.set_persistent_split(split)
        print(f'persistent split forced to {split}')
    run(UNetSeeInDark, 96.771, 1, 1424, 2128)
    run(UNetSeeInDark, 96.771, 16, 512, 512)
    run(UNetSeeInDark, 96.771, 1, 512, 512)
    run(ResUnet, 125.225, 1, 1744, 2320)     # IMX686 frame reflect-padded to a multiple of 16 (trainer_SID.py:221-226)
