"""Size-independent properties at BASELINE.json's full sizes (the oracle is too slow there):
* config 3 batch (16 crops of 4x512x512, nf=32): every crop's output equals that crop run alone (no cross-crop leakage
  in tiles, halos or persistent work lists), and the Winograd and direct kernels agree;
* config 2 frame (4x1424x2128: odd 89x133 maps at the bottom level): Winograd and direct forward agree."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(seed=3):
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    torch.manual_seed(seed)
    net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    initialize_weights(net)
    with torch.no_grad():                      # N(0, 0.02) weights give a nearly dead net: scale up so every layer matters
        for p in net.parameters():
            p.mul_(8.0)
    return net.cuda().eval()


def _fwd(net, x, wino):
    old = os.environ.get('PNNP_WINO')
    os.environ['PNNP_WINO'] = '1' if wino else '0'
    try:
        net.engine._pack_key = None            # re-pack for the other kernel family
        with torch.no_grad():
            return net(x).clone()
    finally:
        if old is None:
            os.environ.pop('PNNP_WINO')
        else:
            os.environ['PNNP_WINO'] = old


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def test_batch_of_16_crops_is_16_independent_crops_and_wino_equals_direct():
    net = _net()
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.rand(16, 4, 512, 512, device='cuda', generator=g)
    yw = _fwd(net, x, True)
    yd = _fwd(net, x, False)
    assert torch.isfinite(yw).all() and float(yw.abs().max()) > 1e-3
    assert _rel(yw, yd) < 2e-5                                  # same fp32 arithmetic up to summation order / transforms
    for b in (0, 7, 15):
        alone = _fwd(net, x[b:b + 1].contiguous(), True)
        assert _rel(alone[0], yw[b]) < 1e-6                     # identical tiling per crop: (near) bit-equal


def test_full_sid_frame_wino_equals_direct():
    net = _net()
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.rand(1, 4, 1424, 2128, device='cuda', generator=g)
    yw = _fwd(net, x, True)
    yd = _fwd(net, x, False)
    assert yw.shape == x.shape and torch.isfinite(yw).all()
    assert _rel(yw, yd) < 2e-5
