"""Timing of the thin-end kernels (csrc/thin.hip) at config 3's size against the generic kernels they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pnnp_amd import ops


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B, H, W, nf = 16, 512, 512, 32
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(B, H, W, nf, device='cuda', generator=g)
    w = torch.randn(4, nf, 1, 1, device='cuda', generator=g); b = torch.randn(4, device='cuda', generator=g)
    out = torch.empty(B, 4, H, W, device='cuda')
    g8 = torch.randn(B, H, W, 8, device='cuda', generator=g); gx = torch.empty_like(x)
    dW = torch.empty_like(w); db = torch.empty_like(b)
    ws = torch.empty(max(ops.head_bwd_workspace_floats(nf), ops.first_wgrad_workspace_floats(nf), ops.wgrad_workspace_floats(B, H, W, nf, 8, 9)), device='cuda')
    mb = lambda n: n * 4 / 1e6
    t = timeit(lambda: ops.head_fwd(x, w, b, out)); n = mb(x.numel() + out.numel())
    print(f'head_fwd     {t:8.1f} us  {n:8.1f} MB  {n/t*1e3:6.0f} GB/s')
    t = timeit(lambda: ops.head_bwd(g8, x, w, gx, dW, db, ws, mode=1)); n = mb(2 * x.numel() + g8.numel())
    print(f'head_bwd     {t:8.1f} us  {n:8.1f} MB  {n/t*1e3:6.0f} GB/s')
    x8 = torch.zeros(B, H, W, 8, device='cuda'); x8[..., :4] = torch.randn(B, H, W, 4, device='cuda', generator=g)
    dW1 = torch.empty(nf, 4, 3, 3, device='cuda'); db1 = torch.empty(nf, device='cuda')
    t = timeit(lambda: ops.first_bwd_weight(x, nf, x8, 4, dW1, db1, ws)); n = mb(x.numel() + x8.numel())
    print(f'first_wgrad  {t:8.1f} us  {n:8.1f} MB  {n/t*1e3:6.0f} GB/s')
    t = timeit(lambda: ops.conv_bwd_weight(x, nf, x8, 4, None, dW1, db1, 9, ws))
    print(f'  (generic conv_bwd_weight taps=9: {t:8.1f} us)')


main()
