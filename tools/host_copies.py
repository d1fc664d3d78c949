#!/usr/bin/env python3
"""Which Python lines of a train step cause host<->device copies / syncs?  One config-5 (or config-3) step under torch.profiler
with stacks; prints the call sites of aten::_to_copy / aten::copy_ / aten::_local_scalar_dense / aten::item, most frequent first.
    python tools/host_copies.py [--config 3|5]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser(); ap.add_argument('--config', type=int, default=5); a = ap.parse_args()
from pnnp_amd.archs import NoiseFlow, ResUnet, UNetSeeInDark, initialize_weights
from pnnp_amd.trainer import HipTrainStep

torch.manual_seed(0); np.random.seed(0)
dev = torch.device('cuda')
net = (ResUnet if a.config == 5 else UNetSeeInDark)(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)); initialize_weights(net); net = net.to(dev)
proxy = None
if a.config == 5:
    proxy = NoiseFlow({'x_shape': (4, 256, 256), 'arch': 'sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc'}).to(dev).eval()
ts = HipTrainStep(net, lr=1e-4, clip=2)
hr = torch.rand(4, 4, 256, 256, device=dev) * (0.01 if proxy is not None else 1.0)


def step():
    if proxy is not None:
        noisy, _, _ = ts.make_noisy_proxy(hr, proxy, ratio_choices=(1, 2, 4, 8, 16), iso=6400)
        return ts.step(hr, noisy=noisy)
    return ts.step(hr)


for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
sites = collections.Counter()


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*args, **kw):
        st = [fr for fr in traceback.extract_stack()[:-1] if '/pnnp_amd/' in fr.filename or 'host_copies' in fr.filename]
        if st:
            sites[(name, f'{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].line}')] += 1
        return orig(*args, **kw)
    setattr(owner, name, f)


for n in ('cpu', 'item', 'to', 'clone', 'contiguous', '__float__', 'copy_', 'tolist', 'float'):
    wrap(torch.Tensor, n)
for n in ('tensor', 'from_numpy', 'full', 'zeros', 'empty', 'eye', 'cat', 'stack'):
    wrap(torch, n)
step()
torch.cuda.synchronize()
for (name, site), n in sites.most_common(60):
    print(f'{n:4d}  {name:12s} {site[:150]}')
