"""Size limits of the convolution kernels (32-bit byte offsets: include/pnnp_hip.h, pnnp_x3_image_fits / pnnp_x3_wgrad_fits) and
what the engines do past them (the policy used to check channel divisibility only and `check()` raised
PNNP_E_UNSUPPORTED in the middle of backward for B = 64 crops of 512 x 512 at nf = 32): the batch-wide limit of the bf16x3
backward-weight kernel falls back to the fp32 kernels; the per-image limit, which every family shares, is refused up front."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fit_queries_mirror_the_launchers():
    from pnnp_amd import ops
    assert ops.x3_image_fits(512, 512, 32) and ops.x3_image_fits(1424, 2128, 64)
    assert not ops.x3_image_fits(4096, 4112, 32)                     # (H + 4) W C 4 >= 2^31: refused by every family (test below)
    assert ops.x3_wgrad_fits(16, 512, 512, 32) and ops.x3_wgrad_fits(16, 512, 512, 64)
    assert not ops.x3_wgrad_fits(64, 512, 512, 32) and not ops.x3_wgrad_fits(16, 1024, 1024, 32)
    # the launcher agrees: one launch past the limit is refused with PNNP_E_UNSUPPORTED, not executed with wrapped offsets
    from pnnp_amd._lib import PnnpError
    B, H, W, C = 64, 512, 512, 32
    g = torch.zeros(B, H, W, C, device='cuda'); x = torch.zeros(B, H, W, C, device='cuda')
    ws = torch.empty(ops.x3_wgrad_workspace_floats(B, H, W, C, C), device='cuda')
    dW = torch.empty(C, C, 3, 3, device='cuda')
    with pytest.raises(PnnpError):
        ops.conv_x3_bwd_weight(g, C, x, C, None, dW, None, ws)


def test_train_step_past_the_wgrad_limit_falls_back_instead_of_failing():
    """B = 64 crops of 4 x 512 x 512 at nf = 32: the top-level backward-weight layers exceed wgrad_x3's whole-tensor offsets and
    run on the Winograd / direct fp32 kernels; everything else stays on bf16x3.  The step completes with a finite loss and the
    per-crop result equals the B = 16 step's on the same crops (the kernels' results do not depend on the batch split)."""
    from pnnp_amd import ops
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    torch.manual_seed(3)
    net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
    initialize_weights(net)
    net = net.cuda()
    ts = HipTrainStep(net, lr=0.0, clip=2)                             # lr 0: the weights stay, two steps are comparable
    g = torch.Generator(device='cuda').manual_seed(1)
    hr = torch.rand(64, 4, 512, 512, device='cuda', generator=g)
    noisy = (hr + 0.05 * torch.randn(hr.shape, device='cuda', generator=g)).clamp_(max=1.0)
    ops.PROFILE, ops.PROFILE_KINDS = [], None
    loss64 = ts.step(hr, noisy=noisy).clone()
    kinds = {r[0] for r in ops.PROFILE}
    ops.PROFILE = None
    assert torch.isfinite(loss64).all()
    # (the fp16x2 family -- the default -- shares wgrad_x3's whole-tensor offsets, limits and fall-back)
    assert ('conv9_wgrad_h2' in kinds or 'conv9_wgrad_x3' in kinds) and (('conv9_wgrad_wino' in kinds) or ('conv9_wgrad' in kinds)), kinds
    g64 = net.engine.params.grad.clone()
    assert torch.isfinite(g64).all()
    sse = []
    gsum = torch.zeros_like(g64)
    for i in range(4):
        lo = ts.step(hr[16 * i:16 * i + 16], noisy=noisy[16 * i:16 * i + 16])
        sse.append(lo[1:].clone()); gsum += net.engine.params.grad
    assert torch.allclose(torch.cat(sse), loss64[1:], rtol=1e-5, atol=0)
    rel = float((gsum / 4 - g64).norm() / g64.norm())
    assert rel < 2e-3, rel                                            # different kernels + split-K partitions: float32 rounding level


def test_single_frame_past_the_image_limit_is_refused_up_front():
    """Every convolution kernel (bf16x3 and fp32-MFMA alike) addresses one image of a map with 32-bit byte offsets: a frame whose
    top-level map exceeds 2 GB per image (4 x 4096 x 4112 at nf = 32) cannot run untiled.  The engine says so BEFORE packing or
    launching anything -- not with PNNP_E_UNSUPPORTED from some layer in the middle of the network -- and stays usable."""
    from pnnp_amd._lib import PnnpError
    from pnnp_amd.archs import ResUnet, UNetSeeInDark, initialize_weights
    torch.manual_seed(4)
    for cls in (UNetSeeInDark, ResUnet):
        net = cls(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4))
        initialize_weights(net)
        net = net.cuda().eval()
        x = torch.zeros(1, 4, 4096, 4112, device='cuda')
        gen0 = net.engine.gen
        with torch.no_grad(), pytest.raises(PnnpError, match='too large'):
            net(x)
        assert net.engine.gen == gen0 and not net.engine.bufs                # nothing was allocated or launched
        with torch.no_grad():
            small = net(torch.rand(1, 4, 256, 256, device='cuda'))           # the engine is still usable
        assert torch.isfinite(small).all()
