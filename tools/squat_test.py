#!/usr/bin/env python3
"""What happens to a persistent convolution layer when k CUs are occupied by another kernel?

    python tools/squat_test.py [k=32]

Builds tools/ubench/libsquat.so if missing (hipcc), then times conv2_2 forward (B = 16, 256 x 256, 64 -> 64: 2048 tiles, 8 per CU)
alone, and beside a kernel that sits on k CUs, with pnnp_set_persistent_split(1) (one workgroup per CU, static shares: the workgroups
of the occupied CUs wait and then run their whole share: ~2x) and with 4 (four per CU of a quarter share each, dispatched as CUs free up).
What ANY dynamic scheme can reach is set by the tile granularity: the 1024 quarter shares of 2 tiles take ceil(1024 / (256 - k)) rounds,
e.g. 5 rounds x 2 tiles = 10 tile times against 8 alone at k = 32 (x 1.25; 256 / (256 - k) = 1.14 is the continuous limit)."""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from pnnp_amd import _lib, ops  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def squat_lib(build_dir=None, may_build=True):
    """tools/ubench/libsquat.so is built by tools/build.py together with the library (before any GPU call); ``may_build`` = False (the tests): never
    spawn a compiler from this process, raise FileNotFoundError instead."""
    so = os.path.join(build_dir or os.path.join(HERE, 'ubench'), 'libsquat.so')
    src = os.path.join(HERE, 'ubench', 'squat.hip')
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        if not may_build:
            raise FileNotFoundError(so)
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', '-o', so, src])
    L = C.CDLL(so)
    L.squat_launch.argtypes = [C.c_int, C.c_double, C.c_void_p]
    return L


def measure(k=32, S=256, Ci=64, Co=64, B=16, reps=5, build_dir=None, may_build=True):
    L = squat_lib(build_dir, may_build)
    x = torch.randn(B, S, S, Ci, device='cuda'); w = torch.randn(Co, Ci, 3, 3, device='cuda') * 0.05; b = torch.randn(Co, device='cuda')
    wx = torch.empty(ops.x3_weight_bytes(Ci, Co), dtype=torch.uint8, device='cuda')
    jobs = ops.PackJobs(); jobs.add_x3(w, wx, None, cin_pad=Ci); jobs.run()
    y = torch.empty(B, S, S, Co, device='cuda')
    side = torch.cuda.Stream()
    lib = _lib.lib()
    lib.pnnp_set_persistent_split.argtypes = [C.c_int]

    def layer_ms(split, squat):
        lib.pnnp_set_persistent_split(split)
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            if squat:
                assert L.squat_launch(k, 4000.0, C.c_void_p(side.cuda_stream)) == 0          # 4 ms on k CUs
                time.sleep(0.0005)                                                           # (it is resident before the layer is dispatched)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv_x3_fwd(x, None, wx, b, y, Co, 1); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        lib.pnnp_set_persistent_split(1)
        return sorted(ts)[len(ts) // 2]

    for _ in range(3):
        ops.conv_x3_fwd(x, None, wx, b, y, Co, 1)
    ref = y.clone()
    out = dict(k=k, alone_1=layer_ms(1, False), alone_4=layer_ms(4, False), beside_1=layer_ms(1, True), beside_4=layer_ms(4, True))
    lib.pnnp_set_persistent_split(4); ops.conv_x3_fwd(x, None, wx, b, y, Co, 1); lib.pnnp_set_persistent_split(1)
    out['same_result'] = bool(torch.equal(ref, y))
    return out


if __name__ == '__main__':
    r = measure(int(sys.argv[1]) if len(sys.argv) > 1 else 32)
    print(r)
    t0 = r['alone_1']
    tiles = 16 * (256 // 32) * (256 // 16)
    rounds4 = -(-1024 // (256 - r['k'])) * 2 / (tiles / 256)
    print(f"granularity bound for split 4: x{rounds4:.2f}")
    print(f"k = {r['k']} CUs occupied: alone {t0:.3f} ms (split 4: {r['alone_4']:.3f}); beside the squatter: split 1 {r['beside_1']:.3f} ms = x{r['beside_1'] / t0:.2f}, "
          f"split 4 {r['beside_4']:.3f} ms = x{r['beside_4'] / t0:.2f}  (256 / (256 - k) = {256 / (256 - r['k']):.2f})")
