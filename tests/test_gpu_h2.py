"""GPU parity of the fp16x2 ("h2") convolution kernels (csrc/conv_h2s.hip: float32 operands scaled by a per-tensor power of two and split
into two fp16 pieces, three fp16 products per multiply on the fp16 matrix cores) -- the SAME cases, references and bars as the bf16x3
family's tests (tests/test_gpu_x3.py: torch-fp32 CPU references at rtol 1e-4 / atol 1e-5 of the largest sum, and the float64 yardsticks
next to the fp32-MFMA kernel: VERDICT round 4, item 1's gate), plus what the family adds: amax slots, sign-bit masks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv import close, nchw, nhwc, _rand
from test_gpu_x3 import CASES

pytestmark = pytest.mark.gpu


def _slot(*tensors):
    """A fresh amax slot holding max |.| over the given CUDA tensors (the stand-alone kernel)."""
    from pnnp_amd import ops
    s = torch.zeros(1, dtype=torch.int32, device='cuda')
    for t in tensors:
        ops.amax(t, s)
    return s


def _slot_value(s):
    return float(s.cpu().view(torch.float32)[0])


def _packs(w, fwd=True, dgrad=True, cin_pad=None):
    from pnnp_amd import ops
    co, ci = w.shape[:2]
    jobs = ops.PackJobs()
    f = torch.zeros(ops.h2_weight_bytes(cin_pad or ci, co), dtype=torch.uint8, device='cuda') if fwd else None
    d = torch.zeros(ops.h2_weight_bytes(co, ci), dtype=torch.uint8, device='cuda') if dgrad else None
    sw = jobs.add_h2(w, f, d, cin_pad=cin_pad)
    jobs.run()
    return f, d, sw


def _decode_bits(bits, B, H, W, C):
    """tile-private sign-bit image (csrc/h2.h) -> bool [B, H, W, C]"""
    ty, tx, nb = (H + 15) // 16, (W + 31) // 32, (C + 31) // 32
    a = bits.cpu().numpy().view(np.uint32).reshape(B, ty, tx, nb, 8, 64)
    out = np.zeros((B, ty * 16, tx * 32, nb * 32), dtype=bool)
    lane = np.arange(64)
    for i in range(2):
        for h in range(2):
            for jj in range(2):
                for c in range(4):
                    bit = 31 - (((i * 2 + h) * 2 + jj) * 4 + c)          # (shifted in in production order: csrc/conv_h2s.hip)
                    v = (a >> bit) & 1                                            # [B, ty, tx, nb, wave, lane]
                    for wv in range(8):
                        rows = np.arange(ty) * 16 + 2 * wv + i
                        for ln in lane:
                            px = np.arange(tx) * 32 + 16 * h + (ln & 15)
                            ch = np.arange(nb) * 32 + 16 * jj + 4 * (ln >> 4) + c
                            out[:, rows[:, None, None], px[None, :, None], ch[None, None, :]] = v[:, :, :, :, wv, ln].astype(bool)
    return out[:, :H, :W, :C]


def test_h2_pack_reconstructs_the_scaled_weights_to_22_bits():
    """hi' + lo' == w 2^se to 2^-22 relative (elements within 2^-18 of the maximum), the scale is the power of two that puts max |w| into
    [2^14, 2^15), and the pack order is the one the kernel streams: [N/32][K16] x (9 hi' taps, 9 lo' taps, 1 tap of zeros) x [octet][32][8]."""
    from pnnp_amd import ops
    w = (_rand(64, 24, 3, 3, seed=3) * torch.logspace(-3, 1, 64 * 24 * 9).reshape(64, 24, 3, 3)).cuda()
    f, d, sw = _packs(w)
    amax = _slot_value(sw)
    assert amax == float(w.abs().max())
    se = 14 - int(np.floor(np.log2(amax)))
    def unpack(buf, K, N):
        K16 = (K + 15) // 16
        a = buf.view(torch.float16).cpu().numpy().astype(np.float64).reshape(N // 32, K16, 19, 2, 32, 8)       # 9 hi' taps, 9 lo' taps, 1 tap of zeros
        assert not a[:, :, 18].any()
        a = a[:, :, :18].reshape(N // 32, K16, 2, 9, 2, 32, 8)
        return a.transpose(2, 1, 4, 6, 0, 5, 3).reshape(2, K16 * 16, N, 9)          # [piece][k][n][tap]
    wn = w.cpu().numpy().astype(np.float64) * 2.0 ** se
    pf = unpack(f, 24, 64)
    want = wn.transpose(1, 0, 2, 3).reshape(24, 64, 9)
    rec = pf[0] + pf[1]
    big = np.abs(want) >= np.abs(want).max() * 2.0 ** -18
    assert np.abs(rec[:24] - want)[big].max() <= 2.0 ** -22 * np.abs(want)[big].max() and not rec[24:].any()
    assert (np.abs(rec[:24] - want) <= 2.0 ** -22 * np.abs(want) + 2.0 ** -25).all()                # below that: half an fp16 subnormal step
    assert 2.0 ** 14 <= np.abs(pf[0]).max() < 2.0 ** 15
    pd = unpack(d, 64, 32)
    want = wn.reshape(64, 24, 9)[:, :, ::-1]
    assert (np.abs(pd[0][:, :24] + pd[1][:, :24] - want) <= 2.0 ** -22 * np.abs(want) + 2.0 ** -25).all() and not pd[:, :, 24:].any()


def test_amax_kernel_and_slot_order():
    from pnnp_amd import ops
    x = torch.randn(3, 5, 7, 9, device='cuda') * 3.0
    x[1, 2, 3, 4] = -123.5
    s = _slot(x)
    assert _slot_value(s) == 123.5
    ops.amax(torch.full((5,), 7.0, device='cuda'), s)                  # a smaller tensor does not lower the slot
    assert _slot_value(s) == 123.5
    assert _slot_value(_slot(torch.zeros(1000, device='cuda'))) == 0.0


@pytest.mark.parametrize('case', CASES)
def test_h2_fwd(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2); b = _rand(Co, seed=4)
    xin = torch.cat([x1, x2], 1) if C2 else x1
    f, _, sw = _packs(w.cuda(), dgrad=False)
    x1c = nhwc(x1).cuda(); x2c = nhwc(x2).cuda() if C2 else None
    s1 = _slot(x1c); s2 = _slot(x2c) if C2 else None
    for act in (0, 1, 2):
        ref = F.conv2d(xin, w, b, padding=1)
        ref = F.leaky_relu(ref, 0.2) if act == 1 else (F.relu(ref) if act == 2 else ref)
        y = torch.full((B, H, W, Co), float('nan'), device='cuda')
        sy = torch.zeros(1, dtype=torch.int32, device='cuda')
        bits = torch.full((ops.h2_bits_words(B, H, W, Co),), -1, dtype=torch.int32, device='cuda')
        ops.conv_h2_fwd(x1c, x2c, f, sw, b.cuda(), y, Co, act, s1, s2, amax_y=sy, bits_y=bits)
        close(nchw(y), ref, what=f'h2 fwd {case} act{act}')
        assert _slot_value(sy) == float(y.abs().max()), 'amax of the stored output'
        assert np.array_equal(_decode_bits(bits, B, H, W, Co), (y > 0).cpu().numpy()), 'sign bits of the stored output'
    r = _rand(B, Co, H, W, seed=9)
    y = torch.empty((B, H, W, Co), device='cuda')
    sy = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd(x1c, x2c, f, sw, b.cuda(), y, Co, 2, s1, s2, amax_y=sy, residual=nhwc(r).cuda())
    close(nchw(y), F.relu(F.conv2d(xin, w, b, padding=1) + r), what='h2 residual')
    assert _slot_value(sy) == float(y.abs().max())


@pytest.mark.parametrize('case', [c for c in CASES if c[3] % 32 == 0])
def test_h2_bwd_data(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2)
    g = _rand(B, Co, H, W, seed=5)
    xin = _rand(B, C1 + C2, H, W, seed=6).requires_grad_(True)
    F.conv2d(xin, w, None, padding=1).backward(g)
    ref = xin.grad
    _, dg, sw = _packs(w.cuda(), fwd=False)
    gc = nhwc(g).cuda(); sg = _slot(gc)
    m1 = _rand(B, C1, H, W, seed=7); m2 = _rand(B, max(C2, 1), H, W, seed=8)
    d1 = torch.full((B, H, W, C1), float('nan'), device='cuda')
    d2 = torch.full((B, H, W, C2), float('nan'), device='cuda') if C2 else None
    a1 = torch.zeros(1, dtype=torch.int32, device='cuda'); a2 = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.conv_h2_bwd_data(gc, sg, dg, sw, d1, amax_dx1=a1, dx2=d2, amax_dx2=a2)
    close(nchw(d1), ref[:, :C1], what=f'h2 dgrad {case}')
    assert _slot_value(a1) == float(d1.abs().max())
    if C2:
        close(nchw(d2), ref[:, C1:], what=f'h2 dgrad2 {case}')
        assert _slot_value(a2) == float(d2.abs().max())
    base2 = _rand(B, max(C2, 1), H, W, seed=10)
    d1 = torch.empty((B, H, W, C1), device='cuda')
    d2 = nhwc(base2).cuda().clone() if C2 else None
    a2.zero_()
    ops.conv_h2_bwd_data(gc, sg, dg, sw, d1, mask1=nhwc(m1).cuda(), mode1=1, dx2=d2,
                         mask2=nhwc(m2).cuda() if C2 else None, mode2=2, accum2=1, amax_dx2=a2)
    close(nchw(d1), ref[:, :C1] * torch.where(m1 > 0, 1.0, 0.2), what='h2 mask1')
    if C2:
        close(nchw(d2), base2 + ref[:, C1:] * (m2 > 0).float(), what='h2 mask2+accum')
        assert _slot_value(a2) == float(d2.abs().max()), 'an accumulating destination reports the SUM it stored'
    if not C2:
        add = _rand(B, C1, H, W, seed=11)
        dx = torch.empty((B, H, W, C1), device='cuda')
        ops.conv_h2_bwd_data_res(gc, sg, dg, sw, dx, addsrc=nhwc(add).cuda(), mask=nhwc(m1).cuda(), mode=2)
        close(nchw(dx), (ref + add) * (m1 > 0).float(), what='h2 dgrad res')


@pytest.mark.parametrize('case', [(2, 8, 32, 32, 0, 32), (1, 12, 40, 32, 0, 64), (2, 8, 32, 32, 32, 64), (1, 9, 33, 64, 64, 32), (1, 16, 64, 128, 128, 64),
                                  (3, 19, 50, 64, 32, 96)])
def test_h2_bwd_data_with_bit_masks_equals_float_masks(case):
    """The act' mask as the sign bits a forward h2 layer wrote (tile-private layout) against the float32 activation itself: the SAME
    kernel arithmetic, so the results must be bit-identical -- single destination, and the decoder's two destinations with the mask on
    the second only (archs/Unet.py:74-93: cat([up, skip]))."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2)
    _, dg, sw = _packs(w.cuda(), fwd=False)
    gc = nhwc(_rand(B, Co, H, W, seed=5)).cuda(); sg = _slot(gc)
    # the tensor whose sign is the mask: output of a forward h2 layer with LeakyReLU (channels = the masked destination's)
    Cm = C2 if C2 else C1
    wm = _rand(Cm, 32, 3, 3, seed=12, scale=0.2).cuda(); bm = _rand(Cm, seed=13).cuda()
    fm, _, swm = _packs(wm, dgrad=False)
    xm = nhwc(_rand(B, 32, H, W, seed=14)).cuda()
    ym = torch.empty((B, H, W, Cm), device='cuda')
    bits = torch.zeros(ops.h2_bits_words(B, H, W, Cm), dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd(xm, None, fm, swm, bm, ym, Cm, 1, _slot(xm), bits_y=bits)
    assert 0.2 < float((ym > 0).float().mean()) < 0.8
    if C2:
        ra = torch.empty((B, H, W, C1), device='cuda'); rb = torch.empty((B, H, W, C2), device='cuda')
        ops.conv_h2_bwd_data(gc, sg, dg, sw, ra, dx2=rb, mask2=ym, mode2=1)
        qa = torch.full_like(ra, float('nan')); qb = torch.full_like(rb, float('nan'))
        ops.conv_h2_bwd_data(gc, sg, dg, sw, qa, dx2=qb, bits2=bits, mode2=1)
        assert torch.equal(ra, qa) and torch.equal(rb, qb)
    else:
        for mode in (1, 2):
            ra = torch.empty((B, H, W, C1), device='cuda'); qa = torch.full_like(ra, float('nan'))
            ops.conv_h2_bwd_data(gc, sg, dg, sw, ra, mask1=ym, mode1=mode)
            ops.conv_h2_bwd_data(gc, sg, dg, sw, qa, bits1=bits, mode1=mode)
            assert torch.equal(ra, qa), mode


def test_h2_padded_network_input():
    """conv1_1: the 4-channel input travels as 8-channel NHWC; the pack pads the reduction to 16 with zeros."""
    from pnnp_amd import ops
    B, H, W, Co = 2, 16, 64, 32
    x = _rand(B, 4, H, W, seed=1); w = _rand(Co, 4, 3, 3, seed=2, scale=0.3); b = _rand(Co, seed=3)
    x8 = torch.empty((B, H, W, 8), device='cuda'); ops.nchw_to_nhwc(x.cuda(), x8, 8)
    f, _, sw = _packs(w.cuda(), dgrad=False, cin_pad=16)
    y = torch.empty((B, H, W, Co), device='cuda')
    ops.conv_h2_fwd(x8, None, f, sw, b.cuda(), y, Co, 1, _slot(x8))
    close(nchw(y), F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2), what='h2 conv1_1')


@pytest.mark.parametrize('cin,cout,shape', [(32, 32, (2, 32, 64)), (32, 64, (1, 48, 96)), (64, 64, (2, 16, 32)), (16, 128, (1, 32, 32))])
def test_h2_fused_maxpool_equals_conv_then_pool_kernel(cin, cout, shape):
    """conv3x3 + LeakyReLU + MaxPool2d(2) in one kernel (archs/Unet.py:33-35): y, the pooled map, the argmax / sign codes, the amax slot and
    the sign bits are bit-identical to the un-fused h2 conv followed by the pool kernel."""
    from pnnp_amd import ops
    B, H, W = shape
    g = torch.Generator(device='cuda').manual_seed(cin + cout)
    x = torch.randn(B, H, W, cin, device='cuda', generator=g)
    x[0, :4, :8] = 0.0
    w = torch.randn(cout, cin, 3, 3, device='cuda', generator=g) * 0.1
    b = torch.randn(cout, device='cuda', generator=g) * 0.1
    b[: cout // 2] = 0.0
    f, _, sw = _packs(w, dgrad=False, cin_pad=(cin + 15) // 16 * 16)
    sx = _slot(x)
    nbits = ops.h2_bits_words(B, H, W, cout)
    y0 = torch.empty(B, H, W, cout, device='cuda'); p0 = torch.empty(B, H // 2, W // 2, cout, device='cuda')
    c0 = torch.empty(B, H // 2, W // 2, cout, dtype=torch.uint8, device='cuda')
    a0 = torch.zeros(1, dtype=torch.int32, device='cuda'); b0 = torch.zeros(nbits, dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd(x, None, f, sw, b, y0, cout, 1, sx, amax_y=a0, bits_y=b0)
    ops.maxpool_fwd(y0, p0, codes=c0)
    y1 = torch.full_like(y0, float('nan')); p1 = torch.full_like(p0, float('nan')); c1 = torch.full_like(c0, 255)
    a1 = torch.zeros(1, dtype=torch.int32, device='cuda'); b1 = torch.full((nbits,), -1, dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd_pool(x, None, f, sw, b, y1, p1, c1, cout, 1, sx, amax_y=a1, bits_y=b1)
    assert torch.equal(y1, y0) and torch.equal(p1, p0) and torch.equal(c1, c0) and torch.equal(a1, a0) and torch.equal(b1, b0)
    assert int((c0 & 3 != 0).sum()) > 0 and int((c0 >> 2 == 0).sum()) > 0


# ---------------------------------------------------------------------------------------------------------------------
# The float64 yardsticks of the bf16x3 family (tests/test_gpu_x3.py:115-452) at the SAME bars, next to the fp32-MFMA kernel.
def _both(x, w, dgrad=False):
    """(h2 result, fp32-MFMA result) of conv2d(x, w, padding=1) (or its backward-data for dgrad: x is the gradient), NCHW on the CPU in double."""
    from pnnp_amd import ops
    B, _, H, W = x.shape
    xc = nhwc(x).cuda()
    if not dgrad:
        Co = w.shape[0]
        f, _, sw = _packs(w.cuda(), dgrad=False)
        f32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), f32, None)
        y2 = torch.empty((B, H, W, Co), device='cuda'); y32 = torch.empty_like(y2)
        ops.conv_h2_fwd(xc, None, f, sw, None, y2, Co, 0, _slot(xc))
        ops.conv_fwd(xc, None, f32, None, y32, Co, 9, 0)
    else:
        Ci = w.shape[1]
        _, d, sw = _packs(w.cuda(), fwd=False)
        d32 = torch.empty(w.numel(), device='cuda'); ops.pack_conv_weight(w.cuda(), None, d32)
        y2 = torch.empty((B, H, W, Ci), device='cuda'); y32 = torch.empty_like(y2)
        ops.conv_h2_bwd_data(xc, _slot(xc), d, sw, y2)
        ops.conv_bwd_data(xc, d32, y32)
    return nchw(y2).cpu().double(), nchw(y32).cpu().double()


def test_h2_is_as_accurate_as_the_fp32_mfma_kernel():
    B, H, W, Ci, Co = 1, 16, 32, 512, 64
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Ci, H, W, generator=g) * torch.logspace(-4, 4, Ci, base=10.0).reshape(1, Ci, 1, 1).roll(1, 1)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * 0.05
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y2, y32 = _both(x, w)
    e2, e32 = float((y2 - ref).norm() / ref.norm()), float((y32 - ref).norm() / ref.norm())
    print(f'relative L2 error vs float64: h2 {e2:.2e}, fp32 MFMA {e32:.2e}')
    assert e2 < 2.0 * e32 + 1e-8 and e2 < 5e-7


def test_h2_dgrad_is_as_accurate_as_the_fp32_mfma_kernel():
    B, H, W, Ci, Co = 1, 16, 32, 64, 512
    gen = torch.Generator().manual_seed(1)
    g = torch.randn(B, Co, H, W, generator=gen) * torch.logspace(-4, 4, Co, base=10.0).reshape(1, Co, 1, 1).roll(3, 1)
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.05
    ref = F.conv_transpose2d(g.double(), w.double(), None, padding=1)
    y2, y32 = _both(g, w, dgrad=True)
    e2, e32 = float((y2 - ref).norm() / ref.norm()), float((y32 - ref).norm() / ref.norm())
    m2, m32 = float((y2 - ref).abs().max() / ref.abs().max()), float((y32 - ref).abs().max() / ref.abs().max())
    print(f'dgrad vs float64: rel L2 h2 {e2:.2e} fp32-MFMA {e32:.2e}; max-element / max|ref| h2 {m2:.2e} fp32-MFMA {m32:.2e}')
    assert e2 < 2.0 * e32 + 1e-8 and e2 < 5e-7
    assert m2 < 2.0 * m32 + 1e-8


def test_h2_max_element_error_under_cancellation():
    B, H, W, Ci, Co = 1, 16, 32, 256, 64
    gen = torch.Generator().manual_seed(4)
    xa = torch.randn(B, Ci // 2, H, W, generator=gen)
    xb = xa * (1 + 2.0 ** -12 * torch.randn(B, Ci // 2, H, W, generator=gen))
    x = torch.stack([xa, xb], 2).reshape(B, Ci, H, W)
    wa = torch.randn(Co, Ci // 2, 3, 3, generator=gen) * 0.1
    w = torch.stack([wa, -wa], 2).reshape(Co, Ci, 3, 3)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    terms = F.conv2d(x.double().abs(), w.double().abs(), None, padding=1)
    assert float(ref.abs().mean() / terms.mean()) < 1e-3
    y2, y32 = _both(x, w)
    r2, r32 = float(((y2 - ref).abs() / terms).max()), float(((y32 - ref).abs() / terms).max())
    print(f'cancellation: max |err| / sum|terms|: h2 {r2:.2e}, fp32-MFMA {r32:.2e} (2^-24 = {2.0 ** -24:.2e})')
    assert r2 < 2.0 * r32 + 2.0 ** -26 and r2 < 8 * 2.0 ** -24


@pytest.mark.parametrize('xs,wsc', [(1e-30, 1.0), (1e+30, 1e-3), (1e-15, 1e-15), (1.0, 1e-30)])
def test_h2_dynamic_range(xs, wsc):
    """Per-tensor scales: operands at 1e-30 / 1e+30 / both 1e-15 give results as accurate (vs float64) as at scale 1."""
    B, H, W, Ci, Co = 1, 8, 32, 128, 32
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(B, Ci, H, W, generator=gen) * xs
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.05 * wsc
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y2, y32 = _both(x, w)
    e2, e32 = float((y2 - ref).norm() / ref.norm()), float((y32 - ref).norm() / ref.norm())
    print(f'scale x {xs:g} w {wsc:g}: rel L2 vs float64 h2 {e2:.2e}, fp32-MFMA {e32:.2e}')
    assert torch.isfinite(y2).all()
    assert e2 < 2.0 * e32 + 1e-8 and e2 < 1e-6


def test_h2_tiny_operands_keep_float32_accuracy():
    """x at 1e-36 (where the bf16x3 family degrades to one piece): the per-tensor scale brings it back into fp16's range."""
    B, H, W, Ci, Co = 1, 8, 32, 64, 32
    gen = torch.Generator().manual_seed(6)
    x = torch.randn(B, Ci, H, W, generator=gen) * 1e-36
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.5
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y2, _ = _both(x, w)
    e2 = float((y2 - ref).norm() / ref.norm())
    print(f'x at 1e-36: rel L2 vs float64 {e2:.2e}')
    assert torch.isfinite(y2).all() and e2 < 1e-6


def test_h2_wide_range_inside_one_tensor_degrades_gracefully():
    """The documented limit (include/pnnp_hip.h): elements below 2^-18 of the tensor's maximum carry an ABSOLUTE error of 2^-40 of that maximum.
    One channel at 1e+6, the rest at 1: the output of filters that IGNORE the large channel is still accurate to 1e-5 relative (2^-40 x 1e6 x
    sqrt(K) / typical sum), finite, and the filters that see it are at float32 level."""
    B, H, W, Ci, Co = 1, 8, 32, 64, 32
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, Ci, H, W, generator=gen); x[:, 0] *= 1e6
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.1; w[:16, 0] = 0.0
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y2, _ = _both(x, w)
    e_blind = float((y2[:, :16] - ref[:, :16]).norm() / ref[:, :16].norm())
    e_see = float((y2[:, 16:] - ref[:, 16:]).norm() / ref[:, 16:].norm())
    print(f'one channel 1e6 x the others: rel L2 of outputs that ignore it {e_blind:.2e}, that see it {e_see:.2e}')
    assert torch.isfinite(y2).all() and e_blind < 1e-5 and e_see < 5e-7


@pytest.mark.parametrize('positive', [False, True])
def test_h2_forward_signed_mean_error_is_bounded(positive):
    B, H, W, Ci, Co = 1, 16, 32, 512, 64
    g = torch.Generator().manual_seed(0)
    x = torch.rand(B, Ci, H, W, generator=g) + 0.5
    w = (torch.rand(Co, Ci, 3, 3, generator=g) + 0.5) * 0.05
    if not positive:
        x = x * (torch.randint(0, 2, x.shape, generator=g) * 2 - 1)
        w = w * (torch.randint(0, 2, w.shape, generator=g) * 2 - 1)
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    y2, y32 = _both(x, w)

    def stats(y):
        d = y - ref
        return float(d.norm() / ref.norm()), float(d.mean() / ref.abs().mean())
    (l2, b2), (l32, b32) = stats(y2), stats(y32)
    print(f'K=4608 {"positive" if positive else "rnd-sign"}: h2 L2 {l2:.2e} signed mean {b2:+.2e} | fp32-MFMA L2 {l32:.2e} signed mean {b32:+.2e}')
    assert l2 < 2.0 * l32 + 1e-8
    assert abs(b2) < (5e-7 if positive else 4e-6)


def test_h2_nan_and_inf_propagate():
    """A diverged tensor (amax = inf / NaN) takes scale 1: the fp16 conversion carries inf / NaN into the products, the output is not finite."""
    from pnnp_amd import ops
    B, H, W, Ci, Co = 1, 16, 32, 32, 32
    x = nhwc(_rand(B, Ci, H, W, seed=1)).cuda(); w = _rand(Co, Ci, 3, 3, seed=2, scale=0.2).cuda()
    f, _, sw = _packs(w, dgrad=False)
    for bad in (float('nan'), float('inf')):
        xb = x.clone(); xb[0, 5, 7, 3] = bad
        y = torch.zeros((B, H, W, Co), device='cuda')
        ops.conv_h2_fwd(xb, None, f, sw, None, y, Co, 0, _slot(xb))
        assert not torch.isfinite(y[0, 4:7, 6:9]).all()


# ---- backward-weight on the fp16 matrix cores (csrc/wgrad_h2s.hip): the cases and bars of tests/test_gpu_x3.py::test_x3_bwd_weight / the float64 yardstick
@pytest.mark.parametrize('case', [(2, 16, 48, 32, 0, 32), (1, 12, 40, 32, 0, 64), (1, 6, 70, 64, 0, 128), (2, 8, 32, 32, 32, 32),
                                  (1, 8, 36, 64, 64, 64), (1, 5, 17, 128, 128, 128), (1, 4, 4, 256, 0, 256), (1, 16, 32, 32, 0, 256),
                                  (3, 32, 64, 32, 0, 32), (2, 16, 32, 64, 64, 64), (3, 19, 50, 64, 32, 96), (1, 9, 33, 96, 0, 32)])
def test_h2_bwd_weight(case):
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.2).requires_grad_(True)
    b = _rand(Co, seed=4).requires_grad_(True)
    g = _rand(B, Co, H, W, seed=5)
    x1 = _rand(B, C1, H, W, seed=1); x2 = _rand(B, C2, H, W, seed=2) if C2 else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    F.conv2d(xin, w, b, padding=1).backward(g)
    assert ops.x3_wgrad_supported(H, W, Co, C1, C2)
    ws = torch.empty(ops.x3_wgrad_workspace_floats(B, H, W, Co, C1 + C2), device='cuda')
    gc, x1c, x2c = nhwc(g).cuda(), nhwc(x1).cuda(), nhwc(x2).cuda() if C2 else None
    sg, s1, s2 = _slot(gc), _slot(x1c), _slot(x2c) if C2 else None
    dW = torch.full(w.shape, float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_h2_bwd_weight(gc, sg, Co, x1c, s1, C1, x2c, s2, dW, db, ws)
    close(dW, w.grad, rtol=2e-4, atol=2e-5, what=f'h2 wgrad {case}')
    close(db, b.grad, rtol=2e-4, atol=2e-5, what=f'h2 bgrad {case}')
    ops.conv_h2_bwd_weight(gc, sg, Co, x1c, s1, C1, x2c, s2, dW, db, ws, accumulate=1)
    close(dW, 2 * w.grad, rtol=2e-4, atol=2e-5, what='h2 wgrad accumulate')
    close(db, 2 * b.grad, rtol=2e-4, atol=4e-5, what='h2 bgrad accumulate')
    dW2 = torch.full(w.shape, float('nan'), device='cuda')
    ops.conv_h2_bwd_weight(gc, sg, Co, x1c, s1, C1, x2c, s2, dW2, None, ws)
    close(dW2, w.grad, rtol=2e-4, atol=2e-5, what='h2 wgrad without dbias')


@pytest.mark.parametrize('shape', [(16, 512, 512, 32, 32), (4, 128, 128, 64, 128)])
def test_h2_wgrad_is_as_accurate_as_the_fp32_mfma_kernel(shape):
    """K = 16 x 512 x 512 = 4.2 M terms per weight at the benchmark's top level, operands spanning 4 decades across channels: relative L2 AND
    max-element error (per (co, ci) scale) within 2x the fp32-MFMA backward-weight kernel's, against float64 sums."""
    from pnnp_amd import ops
    from test_gpu_x3 import _f64_wgrad
    B, H, W, Ci, Co = shape
    gen = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen) * torch.logspace(-2, 2, Ci, device='cuda').roll(5)
    g = torch.randn(B, H, W, Co, device='cuda', generator=gen) * torch.logspace(-3, 1, Co, device='cuda')
    ref = _f64_wgrad(g, x)
    ws = torch.empty(max(ops.x3_wgrad_workspace_floats(B, H, W, Co, Ci), ops.wgrad_workspace_floats(B, H, W, Co, Ci, 9)), device='cuda')
    d2 = torch.empty(Co, Ci, 3, 3, device='cuda'); d32 = torch.empty_like(d2)
    ops.conv_h2_bwd_weight(g, _slot(g), Co, x, _slot(x), Ci, None, None, d2, None, ws)
    ops.conv_bwd_weight(g, Co, x, Ci, None, d32, None, 9, ws)
    rn = ref.norm()
    e2, e32 = float((d2.double() - ref).norm() / rn), float((d32.double() - ref).norm() / rn)
    scale = ref.abs().amax(dim=(2, 3), keepdim=True).clamp_min(1e-30)
    m2, m32 = float(((d2.double() - ref).abs() / scale).max()), float(((d32.double() - ref).abs() / scale).max())
    print(f'wgrad {shape} vs float64: rel L2 h2 {e2:.2e} fp32-MFMA {e32:.2e}; max-element (per (co,ci) scale) h2 {m2:.2e} fp32-MFMA {m32:.2e}')
    assert e2 < 2.0 * e32 + 1e-8, (e2, e32)
    assert m2 < 2.0 * m32 + 1e-7, (m2, m32)


# ---------------------------------------------------------------- the family's one caveat, in all three directions
def test_h2_wide_range_inside_one_tensor_backward_data():
    """Backward-data with one GRADIENT channel at 1e+6 x the others: dx channels whose filters ignore that channel keep 1e-5 relative (the absolute
    error 2^-40 x max of the small elements), the channels that see it are at float32 level -- the same contract as the forward test above."""
    B, H, W, Ci, Co = 1, 8, 32, 32, 64
    gen = torch.Generator().manual_seed(8)
    g = torch.randn(B, Co, H, W, generator=gen); g[:, 0] *= 1e6
    w = torch.randn(Co, Ci, 3, 3, generator=gen) * 0.1; w[0, :16] = 0.0
    ref = F.conv_transpose2d(g.double(), w.double(), None, padding=1)
    y2, y32 = _both(g, w, dgrad=True)
    e_blind = float((y2[:, :16] - ref[:, :16]).norm() / ref[:, :16].norm())
    e_see = float((y2[:, 16:] - ref[:, 16:]).norm() / ref[:, 16:].norm())
    e32 = float((y32[:, :16] - ref[:, :16]).norm() / ref[:, :16].norm())
    print(f'dgrad, one channel 1e6 x the others: rel L2 of dx channels that ignore it {e_blind:.2e} (fp32-MFMA {e32:.2e}), that see it {e_see:.2e}')
    assert torch.isfinite(y2).all() and e_blind < 1e-5 and e_see < 5e-7


def _wgrad_both(g, x):
    """(h2, fp32-MFMA, float64) weight gradient of a 3x3 layer from NHWC CUDA tensors."""
    from pnnp_amd import ops
    from test_gpu_x3 import _f64_wgrad
    B, H, W, Co = g.shape
    Ci = x.shape[3]
    ws = torch.empty(max(ops.x3_wgrad_workspace_floats(B, H, W, Co, Ci), ops.wgrad_workspace_floats(B, H, W, Co, Ci, 9)), device='cuda')
    d2 = torch.empty(Co, Ci, 3, 3, device='cuda'); d32 = torch.empty_like(d2)
    ops.conv_h2_bwd_weight(g, _slot(g), Co, x, _slot(x), Ci, None, None, d2, None, ws)
    ops.conv_bwd_weight(g, Co, x, Ci, None, d32, None, 9, ws)
    return d2.double(), d32.double(), _f64_wgrad(g, x)


def test_h2_wide_range_inside_one_tensor_backward_weight():
    """Backward-weight with one activation channel AND one gradient channel at 1e+6 x the others.  A weight's gradient is a sum of products g x: where
    at least one factor is a SMALL element of its tensor (absolute accuracy 2^-40 x max = relative 1e-6 of itself) the product inherits that -- dW[1:, 1:],
    dW[1:, 0] and dW[0, 1:] keep 1e-5 relative against float64 sums (measured 2-5e-6); only dW[0, 0], large x large, is at float32 level."""
    B, H, W, Ci, Co = 2, 32, 64, 32, 32
    gen = torch.Generator(device='cuda').manual_seed(9)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen); x[..., 0] *= 1e6
    g = torch.randn(B, H, W, Co, device='cuda', generator=gen); g[..., 0] *= 1e6
    d2, d32, ref = _wgrad_both(g, x)
    rel = lambda d, sl: float((d[sl] - ref[sl]).norm() / ref[sl].norm())
    blind, col, row, big = (slice(1, None), slice(1, None)), (slice(1, None), slice(0, 1)), (slice(0, 1), slice(1, None)), (slice(0, 1), slice(0, 1))
    print(f'wgrad, channel 0 of x and of g 1e6 x the others: rel L2 vs float64 of dW[1:,1:] {rel(d2, blind):.2e} (fp32-MFMA {rel(d32, blind):.2e}), '
          f'dW[1:,0] {rel(d2, col):.2e}, dW[0,1:] {rel(d2, row):.2e}, dW[0,0] {rel(d2, big):.2e}')
    assert torch.isfinite(d2).all() and rel(d2, blind) < 1e-5 and rel(d2, col) < 1e-5 and rel(d2, row) < 1e-5 and rel(d2, big) < 5e-7


@pytest.mark.parametrize('direction', ['fwd', 'dgrad', 'wgrad'])
def test_h2_single_outlier_bounds_the_damage(direction):
    """ONE element at 1e+8 x the rest of its tensor (a hot pixel times ratio, a gradient spike: ADVICE round 5).  The other elements sit 2^-26.6 below the
    maximum: their split keeps an ABSOLUTE accuracy of 2^-40 x 1e8 = 9e-5 (csrc/h2.h), i.e. ~13 bits -- the documented price of ONE scale per tensor.  The test
    pins that bound (outputs that do not touch the outlier: relative L2 below 2^-40 x 1e8 = 9.1e-5, finite, no saturation) and that the outputs that DO see it are at
    float32 level relative to their own size; it also prints what the exact bf16x3 / fp32 kernels give, which is what `set_policy(h2=False)` buys."""
    gen = torch.Generator().manual_seed(10)
    bound = 2.0 ** -40 * 1e8
    if direction in ('fwd', 'dgrad'):
        B, H, W, Ca, Cb = 1, 16, 32, 64, 32
        a = torch.randn(B, Ca, H, W, generator=gen); a[0, 3, 8, 16] = 1e8
        w = torch.randn(Cb, Ca, 3, 3, generator=gen) * 0.1 if direction == 'fwd' else torch.randn(Ca, Cb, 3, 3, generator=gen) * 0.1
        ref = F.conv2d(a.double(), w.double(), None, padding=1) if direction == 'fwd' else F.conv_transpose2d(a.double(), w.double(), None, padding=1)
        y2, y32 = _both(a, w, dgrad=direction == 'dgrad')
        near = torch.zeros(H, W, dtype=torch.bool); near[7:10, 15:18] = True          # the 3 x 3 pixels the outlier reaches
        far = ~near
        e_far, e32_far = float((y2[..., far] - ref[..., far]).norm() / ref[..., far].norm()), float((y32[..., far] - ref[..., far]).norm() / ref[..., far].norm())
        e_near = float((y2[..., near] - ref[..., near]).norm() / ref[..., near].norm())
        print(f'{direction}, one element 1e8: rel L2 away from it {e_far:.2e} (fp32-MFMA {e32_far:.2e}; bound {bound:.1e}), at the pixels it reaches {e_near:.2e}')
        assert torch.isfinite(y2).all() and e_far < bound and e_near < 5e-7
    else:
        B, H, W, Ci, Co = 2, 32, 64, 32, 32
        gc = torch.Generator(device='cuda').manual_seed(11)
        x = torch.randn(B, H, W, Ci, device='cuda', generator=gc); x[1, 5, 9, 2] = 1e8
        g = torch.randn(B, H, W, Co, device='cuda', generator=gc)
        d2, d32, ref = _wgrad_both(g, x)
        oth = [c for c in range(Ci) if c != 2]
        e_oth, e32_oth = float((d2[:, oth] - ref[:, oth]).norm() / ref[:, oth].norm()), float((d32[:, oth] - ref[:, oth]).norm() / ref[:, oth].norm())
        e_hit = float((d2[:, 2] - ref[:, 2]).norm() / ref[:, 2].norm())
        print(f'wgrad, one element of x 1e8: rel L2 of the other input channels: weights {e_oth:.2e} (fp32-MFMA {e32_oth:.2e}; bound {bound:.1e}), of channel 2 {e_hit:.2e}')
        assert torch.isfinite(d2).all() and e_oth < bound and e_hit < 5e-7


# ---------------------------------------------------------------- split-K for small grids (round 6; VERDICT rounds 3-5)
@pytest.mark.parametrize('case', [(1, 32, 32, 512, 0, 512), (1, 64, 64, 128, 128, 256), (2, 24, 40, 256, 0, 64), (1, 16, 32, 64, 0, 32)])
def test_h2_splitk_forward_equals_the_unsplit_launch(case):
    """K cut into slices (pnnp_h2_splitk), raw partial sums into a slab tensor, one fixed-order reduce with bias, LeakyReLU, amax and the sign bits: the
    result equals the ordinary launch up to float32 rounding of another partition of the K sum (4e-6 of the largest output), is deterministic (two runs
    bit-identical), the amax slot holds max |y| of what was stored, and the decoded sign bits are (y > 0)."""
    from pnnp_amd import ops
    B, H, W, C1, C2, Co = case
    x1 = nhwc(_rand(B, C1, H, W, seed=1)).cuda(); x2 = nhwc(_rand(B, C2, H, W, seed=2)).cuda() if C2 else None
    w = _rand(Co, C1 + C2, 3, 3, seed=3, scale=0.05).cuda(); bias = _rand(Co, seed=4).cuda()
    f, _, sw = _packs(w, dgrad=False)
    s1, s2 = _slot(x1), (_slot(x2) if C2 else None)
    chunks = (2 if C2 else 1) * ((C1 + 15) // 16)
    ks = ops.h2_splitk(B, H, W, chunks, Co)
    assert ks > 1 and chunks % ks == 0, (ks, chunks)
    y0 = torch.empty(B, H, W, Co, device='cuda'); a0 = torch.zeros(1, dtype=torch.int32, device='cuda')
    ops.conv_h2_fwd(x1, x2, f, sw, bias, y0, Co, 1, s1, s2, amax_y=a0)

    def split():
        y = torch.full((B, H, W, Co), float('nan'), device='cuda'); am = torch.zeros(1, dtype=torch.int32, device='cuda')
        bits = torch.zeros(ops.h2_bits_words(B, H, W, Co), dtype=torch.int32, device='cuda')
        ws = torch.empty(ks * y.numel(), device='cuda')
        ops.conv_h2_fwd_splitk(x1, x2, f, sw, bias, y, Co, 1, s1, ks, ws, amax_x2=s2, amax_y=am, bits_y=bits)
        return y, am, bits
    y, am, bits = split()
    y_b, am_b, bits_b = split()
    assert torch.equal(y, y_b) and torch.equal(bits, bits_b) and torch.equal(am, am_b)
    d = float((y - y0).abs().max() / y0.abs().max())
    print(f'split-K x {ks} {case}: max |diff| / max |y| vs the unsplit launch {d:.2e}')
    assert d < 4e-6                                              # (K = 4608 terms summed in 8 slices instead of one chain: float32 rounding of another partition)
    assert _slot_value(am) == float(y.abs().max())
    assert np.array_equal(_decode_bits(bits, B, H, W, Co), (y > 0).cpu().numpy())


def test_splitk_engine_forward_and_step_on_one_crop():
    """One 64 x 64 ... 512 x 512 crop through the engine: with split-K (default) and without (`set_policy(splitk=False)`) the eval forward agrees to 1e-5 of
    the largest output and a training step's loss to 1e-6; the split path runs the un-fused pool kernel behind conv{2,3,4}_2 and the same backward."""
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    from pnnp_amd.trainer import HipTrainStep
    torch.manual_seed(5)
    g = torch.Generator(device='cuda').manual_seed(6)
    x = torch.rand(1, 4, 256, 256, device='cuda', generator=g); t = torch.rand(1, 4, 256, 256, device='cuda', generator=g)
    outs, losses = [], []
    for sk in (True, False):
        torch.manual_seed(7)
        net = UNetSeeInDark(dict(nframes=1, res=False, nf=32, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda()
        net.engine.set_policy(splitk=sk)
        with torch.no_grad():
            outs.append(net(x).clone())
        ts = HipTrainStep(net, lr=1e-4, clip=0)
        losses.append([float(ts.step(t, noisy=x)[0]) for _ in range(3)])
    d = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
    print(f'engine, one 256 x 256 crop: eval forward split-K vs not: {d:.2e}; losses {losses}')
    assert d < 1e-5
    assert all(abs(p - q) < 1e-6 * max(1.0, abs(q)) for p, q in zip(*losses))


@pytest.mark.parametrize('B', [1, 2, 4, 5, 6, 11])
@pytest.mark.parametrize('chan', [(64, 64), (32, 32), (32, 64)])
def test_h2_bwd_weight_tiles_per_workgroup(B, chan):
    """Round 6: the producers of wgrad_h2s_kernel re-request a staging slot for the tile AFTER NEXT as soon as it is staged (rolling refill), with the round-5
    order left for a workgroup's last two tiles.  One output tile of 64 x 64 / 32 x 32 / 32 x 64 channels on a 64 x 128 map = one slab per CU, so the batch
    sets the pixel tiles per workgroup: 1 (no loop), 2 (no rolling iteration), 3-4 (one or two), uneven shares, many -- against float64 sums."""
    from pnnp_amd import ops
    from test_gpu_x3 import _f64_wgrad
    Co, Ci = chan
    H, W = 64, 128
    gen = torch.Generator(device='cuda').manual_seed(B * 7 + Co)
    x = torch.randn(B, H, W, Ci, device='cuda', generator=gen); g = torch.randn(B, H, W, Co, device='cuda', generator=gen)
    ref = _f64_wgrad(g, x)
    ws = torch.empty(ops.x3_wgrad_workspace_floats(B, H, W, Co, Ci), device='cuda')
    dW = torch.full((Co, Ci, 3, 3), float('nan'), device='cuda'); db = torch.full((Co,), float('nan'), device='cuda')
    ops.conv_h2_bwd_weight(g, _slot(g), Co, x, _slot(x), Ci, None, None, dW, db, ws)
    err = (dW.double() - ref).norm() / ref.norm()
    assert err < 2e-6, (B, chan, float(err))
    bref = g.double().sum((0, 1, 2))
    assert (db.double() - bref).abs().max() < 1e-5 * g.double().abs().sum((0, 1, 2)).max()


@pytest.mark.parametrize('shape', [(32, 32), (512, 512), (96, 48)])
def test_weight_amax_job_one_atomic_per_block(shape):
    """Round 6: the weight tensors' amax job (pack_jobs kind 5) reduces a block's four waves through LDS and issues ONE atomic (every wave queued its own: up to
    4 096 on one slot).  The slot is max |w| as a bit pattern; a NaN anywhere flags the tensor (its pattern orders above inf), whichever block holds it."""
    from pnnp_amd import ops
    Co, Ci = shape
    w = (_rand(Co, Ci, 3, 3, seed=11, scale=0.3)).cuda()
    w[Co // 2, Ci // 3, 1, 2] = -7.25                                   # the maximum, negative, somewhere in the middle
    jobs = ops.PackJobs()
    f = torch.zeros(ops.h2_weight_bytes((Ci + 15) // 16 * 16, Co), dtype=torch.uint8, device='cuda')
    slot = jobs.add_h2(w, f, None)
    jobs.run()
    assert slot.view(torch.float32).item() == 7.25
    jobs.run()                                                          # (the slots are zeroed per run: the same answer, not an accumulation)
    assert slot.view(torch.float32).item() == 7.25
    w[-1, -1, 2, 2] = float('nan')                                      # the last element: the last block's last wave
    jobs.run()
    assert slot.view(torch.float32).isnan().item()
    w[-1, -1, 2, 2] = float('inf')
    jobs.run()
    assert slot.view(torch.float32).isinf().item()
