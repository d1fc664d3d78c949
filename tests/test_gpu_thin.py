"""The thin ends of the networks (csrc/thin.hip) against float64 torch on the same inputs: the 1x1 head conv10_1
(archs/Unet.py:80,93) forward and its one-pass backward, and the first 3x3 convolution's weight gradient (archs/Unet.py:31).
Float32 kernels, fixed summation order: tolerances are those of float32 sums of the given length."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.mark.parametrize('cin,cout,res', [(32, 4, False), (32, 4, True), (32, 3, False), (16, 4, False), (64, 2, True)])
@pytest.mark.parametrize('shape', [(1, 16, 16), (3, 48, 80)])
def test_head_forward(cin, cout, res, shape):
    from pnnp_amd import ops
    B, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin + 8, device='cuda', generator=g)          # channel stride > cin: the tail must be ignored
    w = torch.randn(cout, cin, 1, 1, device='cuda', generator=g) * 0.3
    b = torch.randn(cout, device='cuda', generator=g)
    r = torch.randn(B, cout, H, W, device='cuda', generator=g) if res else None
    assert ops.head_supported(cin, cout, B * H * W)
    out = torch.full((B, cout, H, W), float('nan'), device='cuda')
    ops.head_fwd(x, w, b, out, residual=r)
    ref = F.conv2d(x[..., :cin].permute(0, 3, 1, 2).double(), w.double(), b.double())
    if res:
        ref = ref + r.double()
    assert _rel(out, ref) < 1e-6


@pytest.mark.parametrize('cin,cout,mode', [(32, 4, 1), (32, 4, 0), (32, 3, 2), (16, 4, 1), (64, 4, 1)])
@pytest.mark.parametrize('shape', [(1, 16, 16), (2, 64, 96)])
def test_head_backward_is_dgrad_and_wgrad_in_one_pass(cin, cout, mode, shape):
    from pnnp_amd import ops
    B, H, W = shape
    gen = torch.Generator(device='cuda').manual_seed(7 * cin + mode)
    x = torch.randn(B, H, W, cin, device='cuda', generator=gen)
    g8 = torch.randn(B, H, W, 8, device='cuda', generator=gen)              # channels >= cout hold garbage: must not matter
    w = torch.randn(cout, cin, 1, 1, device='cuda', generator=gen) * 0.3
    gx = torch.full((B, H, W, cin), float('nan'), device='cuda')
    dW = torch.full((cout, cin, 1, 1), float('nan'), device='cuda'); db = torch.full((cout,), float('nan'), device='cuda')
    ws = torch.empty(max(ops.head_bwd_workspace_floats(cin), 1), device='cuda')
    ops.head_bwd(g8, x, w, gx, dW, db, ws, mode=mode)
    gd, xd, wd = g8[..., :cout].double(), x.double(), w.double().reshape(cout, cin)
    slope = {0: 1.0, 1: 0.2, 2: 0.0}[mode]
    ref_gx = (gd @ wd) * torch.where(xd > 0, torch.ones_like(xd), torch.full_like(xd, slope))
    ref_dW = torch.einsum('bhwo,bhwi->oi', gd, xd)
    ref_db = gd.sum((0, 1, 2))
    assert _rel(gx, ref_gx) < 1e-6
    assert _rel(dW.reshape(cout, cin), ref_dW) < 2e-5
    assert _rel(db, ref_db) < 2e-5
    # accumulate: a second call adds the same sums again; and the result is bit-reproducible
    dW2, db2 = dW.clone(), db.clone()
    ops.head_bwd(g8, x, w, gx, dW2, db2, ws, mode=mode, accumulate=1)
    assert torch.equal(dW2, dW + dW) and torch.equal(db2, db + db)


@pytest.mark.parametrize('cin,cout', [(4, 32), (3, 32), (4, 64), (1, 32)])
@pytest.mark.parametrize('shape', [(1, 16, 16), (3, 64, 112), (2, 32, 48)])
def test_first_layer_backward_weight(cin, cout, shape):
    from pnnp_amd import ops
    B, H, W = shape
    gen = torch.Generator(device='cuda').manual_seed(cin * 100 + cout)
    x8 = torch.zeros(B, H, W, 8, device='cuda')
    x8[..., :cin] = torch.randn(B, H, W, cin, device='cuda', generator=gen)
    g = torch.randn(B, H, W, cout, device='cuda', generator=gen)
    assert ops.first_wgrad_supported(cin, cout, H, W)
    dW = torch.full((cout, cin, 3, 3), float('nan'), device='cuda'); db = torch.full((cout,), float('nan'), device='cuda')
    ws = torch.empty(ops.first_wgrad_workspace_floats(cout), device='cuda')
    ops.first_bwd_weight(g, cout, x8, cin, dW, db, ws)
    xd = x8[..., :cin].permute(0, 3, 1, 2).double().requires_grad_(False)
    wd = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device='cuda', requires_grad=True)
    y = F.conv2d(xd, wd, padding=1)
    y.backward(g.permute(0, 3, 1, 2).double())
    assert _rel(dW, wd.grad) < 2e-5
    assert _rel(db, g.double().sum((0, 1, 2))) < 2e-5
    dW2, db2 = dW.clone(), db.clone()
    ops.first_bwd_weight(g, cout, x8, cin, dW2, db2, ws, accumulate=1)
    assert torch.equal(dW2, dW + dW) and torch.equal(db2, db + db)


def test_unsupported_shapes_are_refused_not_miscomputed():
    from pnnp_amd import ops
    from pnnp_amd._lib import PnnpError
    assert not ops.head_supported(24, 4, 256) and not ops.head_supported(32, 5, 256) and not ops.head_supported(32, 4, 100)
    assert not ops.first_wgrad_supported(8, 32, 16, 16) and not ops.first_wgrad_supported(4, 48, 16, 16)
    x = torch.zeros(1, 16, 16, 24, device='cuda'); w = torch.zeros(4, 24, 1, 1, device='cuda')
    with pytest.raises(PnnpError):
        ops.head_fwd(x, w, torch.zeros(4, device='cuda'), torch.zeros(1, 4, 16, 16, device='cuda'))


@pytest.mark.parametrize('cin,cout,act', [(4, 32, 1), (3, 32, 2), (4, 64, 0), (1, 32, 1)])
@pytest.mark.parametrize('shape', [(1, 16, 16), (3, 64, 112), (2, 32, 48)])
def test_first_layer_forward(cin, cout, act, shape):
    from pnnp_amd import ops
    B, H, W = shape
    gen = torch.Generator(device='cuda').manual_seed(cin * 10 + cout + act)
    x8 = torch.zeros(B, H, W, 8, device='cuda')
    x8[..., :cin] = torch.randn(B, H, W, cin, device='cuda', generator=gen)
    w = torch.randn(cout, cin, 3, 3, device='cuda', generator=gen) * 0.3
    b = torch.randn(cout, device='cuda', generator=gen)
    y = torch.full((B, H, W, cout + 4), float('nan'), device='cuda')         # channel stride > cout: the tail stays untouched
    ops.first_fwd(x8, w, b, y, act)
    ref = F.conv2d(x8[..., :cin].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    ref = {0: ref, 1: F.leaky_relu(ref, 0.2), 2: F.relu(ref)}[act].permute(0, 2, 3, 1)
    assert _rel(y[..., :cout], ref) < 2e-6
    assert torch.isnan(y[..., cout:]).all()
