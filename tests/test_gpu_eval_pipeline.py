"""The eval loop of trainer_SID.py:208-246 end to end on the device against the same steps on the CPU oracle, same weights and
same noisy input: net -> clamp -> IlluminanceCorrect -> tensor2im -> raw-domain PSNR / SSIM.  north_star asks for eval PSNR
within +-0.02 dB; with identical weights the two paths agree to < 1e-3 dB, SSIM to 2e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('arch', ['unet', 'resunet'])
def test_eval_psnr_ssim_gpu_equals_cpu_path(arch):
    from oracle import metrics_np as M, net_torch as O
    from pnnp_amd import process
    from pnnp_amd.archs import ResUnet, UNetSeeInDark
    from pnnp_amd.metrics import IlluminanceCorrect, quality_assess
    torch.manual_seed(3); np.random.seed(3)
    nthr = torch.get_num_threads(); torch.set_num_threads(min(nthr, 8))      # the CPU leg is small convs: more threads only add overhead
    shapes = O.unet_param_shapes(nf=32) if arch == 'unet' else O.resunet_param_shapes(nf=32)
    sd = O.init_state(shapes, seed=5, std=0.02)
    sd = {k: v * 6.0 for k, v in sd.items()}                     # N(0, 0.02) weights give a dead network: make every layer matter
    net = (UNetSeeInDark if arch == 'unet' else ResUnet)(dict(nframes=1, res=True, nf=32, in_nc=4, out_nc=4))
    net.load_state_dict(sd); net = net.cuda().eval()
    g = torch.Generator().manual_seed(7)
    hr = (torch.rand(1, 4, 192, 192, generator=g) ** 2.2) * 0.1 * 100.0 / 100.0          # dark-ish linear raw
    hr = (hr * 8).clamp(0, 1)
    p = process.sample_params_max('SonyA7S2', ratio=100, iso=1600)
    lr = process.generate_noisy_torch(hr[0].cuda(), param=p, noise_code='pr', ori=False, clip=False).unsqueeze(0).clamp(0, 1)
    with torch.no_grad():
        dn_gpu = net(lr).clamp(0, 1)
        dn_gpu = IlluminanceCorrect()(dn_gpu, hr.cuda())
        res = quality_assess(dn_gpu, hr.cuda()).cpu().numpy()
        fwd = O.unet_forward if arch == 'unet' else O.resunet_forward
        dn_cpu = fwd(sd, lr.cpu(), res=True).clamp(0, 1)
        dn_cpu = O.illuminance_correct(dn_cpu, hr)
    psnr_cpu = M.psnr(M.tensor2im(hr.numpy()), M.tensor2im(dn_cpu.numpy()))
    ssim_cpu = M.ssim(M.tensor2im(hr.numpy()), M.tensor2im(dn_cpu.numpy()))
    assert np.isfinite(res).all() and 5.0 < psnr_cpu < 60.0
    assert abs(float(res[0]) - psnr_cpu) < 1e-3, (float(res[0]), psnr_cpu)            # dB
    assert abs(float(res[1]) - ssim_cpu) < 2e-5, (float(res[1]), ssim_cpu)
    torch.set_num_threads(nthr)


def test_evaluate_product_function_on_the_imx686_frame_vs_cpu_oracle():
    """`pnnp_amd.evaluate` (row f1 as a product function) on the LRID frame 4x1736x2312 -- width % 16 = 8, so the
    reflect-pad-4 / crop branch of trainer_SID.py:221-228 runs (4x1744x2320 through the net) -- with `ori` brightening,
    IlluminanceCorrect, and PSNR/SSIM of the denoised AND the noisy frame, against the same steps on the CPU oracle.
    Bars: PSNR 1e-3 dB (north_star: 0.02 dB), SSIM 2e-5; then the log line / metrics dict layout of :309-315."""
    import torch.nn.functional as F
    from oracle import metrics_np as M, net_torch as O
    from pnnp_amd.archs import UNetSeeInDark
    from pnnp_amd.evaluate import EvalLoop, evaluate
    sd = O.init_state_he(O.unet_param_shapes(nf=32), seed=21)
    net = UNetSeeInDark(dict(nframes=1, res=True, nf=32, in_nc=4, out_nc=4))
    net.load_state_dict({k: v.clone() for k, v in sd.items()}); net = net.cuda().eval()
    g = torch.Generator().manual_seed(13)
    hr = (torch.rand(1, 4, 1736, 2312, generator=g) ** 2.2)
    ratio = 8.0
    lr = (hr / ratio + torch.randn(1, 4, 1736, 2312, generator=g) * 0.004)          # dark, un-brightened input (dst.ori = True)
    out = evaluate(net, lr.cuda(), hr.cuda(), ratio=ratio, ori=True, brightness_correct=True, epoch=-1)
    got = out['metrics'].cpu().numpy()
    assert out['dn'].shape == hr.shape
    nthr = torch.get_num_threads()
    with torch.no_grad():
        p = F.pad(lr, (4, 4, 4, 4), mode='reflect')
        dn = O.unet_forward(sd, p, res=True)[..., 4:-4, 4:-4]
        lr_c = (lr * ratio).clamp(0, 1); dn = (dn * ratio).clamp(0, 1)
        dn = O.illuminance_correct(dn, hr)
    tgt = M.tensor2im(hr.numpy())
    ref = [M.psnr(tgt, M.tensor2im(dn.numpy())), M.ssim(tgt, M.tensor2im(dn.numpy())),
           M.psnr(tgt, M.tensor2im(lr_c.numpy())), M.ssim(tgt, M.tensor2im(lr_c.numpy()))]
    torch.set_num_threads(nthr)
    assert abs(got[0] - ref[0]) < 1e-3 and abs(got[2] - ref[2]) < 1e-3, (got, ref)
    assert abs(got[1] - ref[1]) < 2e-5 and abs(got[3] - ref[3]) < 2e-5, (got, ref)
    loop = EvalLoop(net, ori=True, brightness_correct=True)
    loop.step('frame_a', lr.cuda(), hr.cuda(), ratio=ratio)
    loop.step('frame_b', lr.cuda(), hr.cuda(), ratio=ratio)
    metrics, text = loop.finish(epoch=-1)
    assert list(metrics) == ['frame_a', 'frame_b'] and abs(metrics['frame_a'][0] - ref[0]) < 1e-3
    import re
    lines = text.splitlines()
    assert re.fullmatch(r'Epoch -1: PSNR=\d+\.\d\d', lines[0])
    assert re.fullmatch(r'psnrs_lr=\d+\.\d\d, psnrs_dn=\d+\.\d\d', lines[1])
    assert re.fullmatch(r'ssims_lr=\d\.\d{4}, ssims_dn=\d\.\d{4}', lines[2])
    nums = [float(v) for v in re.findall(r'=(-?\d+\.\d+)', text)]
    assert np.allclose(nums, [ref[0], ref[2], ref[0], ref[3], ref[1]], atol=6e-3)


@pytest.mark.parametrize('res', [False, True])
def test_eval_tail_kernel_equals_the_tensor_ops_bit_for_bit(res):
    """`evaluate` crops the reflect-padded output, adds the input residual of a `res` network, multiplies by the ratio (`ori`) and
    clamps both frames in ONE kernel (pnnp_eval_post_f32): against the reference's sequence of tensor ops (trainer_SID.py:226-235)
    on the same network output the result must be identical, with a host ratio and with a device ratio (no synchronisation)."""
    import torch.nn.functional as F
    from pnnp_amd.archs import UNetSeeInDark, initialize_weights
    from pnnp_amd.evaluate import evaluate
    torch.manual_seed(4)
    net = UNetSeeInDark(dict(nframes=1, res=res, nf=8, in_nc=4, out_nc=4)); initialize_weights(net); net = net.cuda().eval()
    g = torch.Generator(device='cuda').manual_seed(5)
    hr = torch.rand(1, 4, 56, 72, device='cuda', generator=g)                      # 72 % 16 = 8: the padded branch (64 x 80 through the net)
    lr = hr / 5.0 + torch.randn(1, 4, 56, 72, device='cuda', generator=g) * 0.02
    with torch.no_grad():
        ref_dn = net(F.pad(lr, (4, 4, 4, 4), mode='reflect'))[..., 4:-4, 4:-4]          # the module's own forward (residual inside)
        ref_dn = (ref_dn * 5.0).clamp(0, 1); ref_lr = (lr * 5.0).clamp(0, 1)
    for ratio in (5.0, torch.tensor([5.0], device='cuda')):
        out = evaluate(net, lr, hr, ratio=ratio, ori=True, brightness_correct=False)
        # (a `res` network: f(pad x)[crop] + x here, (f(pad x) + pad x)[crop] in the module -- the same float32 sum)
        assert torch.equal(out['lr'], ref_lr)
        assert torch.equal(out['dn'], ref_dn), float((out['dn'] - ref_dn).abs().max())


def test_eval_tail_kernel_propagates_nan_like_torch_clamp():
    """A diverged network (NaN / +-inf in its output) must report NaN metrics, not finite ones: `(x * ratio).clamp(0, 1)` keeps a NaN
    (trainer_SID.py:234-235), and so does pnnp_eval_post_f32 (fminf / fmaxf would have turned it into 0)."""
    import ctypes as C
    from pnnp_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(6)
    out = torch.randn(4, 24, 40, device='cuda', generator=g)
    lr = torch.randn(4, 16, 32, device='cuda', generator=g)
    out[0, 5, 6] = float('nan'); out[1, 7, 8] = float('inf'); out[2, 9, 10] = -float('inf'); lr[3, 1, 2] = float('nan')
    dn = torch.empty_like(lr); lo = torch.empty_like(lr)
    for add_res in (0, 1):
        _lib.check(_lib.lib().pnnp_eval_post_f32(_lib.ptr(out), _lib.ptr(lr), _lib.ptr(dn), _lib.ptr(lo), 4, 16, 32, 24, 40, 4,
                                                 C.c_float(3.0), None, add_res, _lib.stream()), 'eval_post')
        ref = out[:, 4:-4, 4:-4] + (lr if add_res else 0.0)
        ref_dn, ref_lr = (ref * 3.0).clamp(0, 1), (lr * 3.0).clamp(0, 1)
        assert torch.isnan(dn[0, 1, 2]) and float(dn[1, 3, 4]) == 1.0 and float(dn[2, 5, 6]) == 0.0 and torch.isnan(lo[3, 1, 2])
        assert torch.equal(torch.nan_to_num(dn, nan=-7.0), torch.nan_to_num(ref_dn, nan=-7.0))
        assert torch.equal(torch.nan_to_num(lo, nan=-7.0), torch.nan_to_num(ref_lr, nan=-7.0))
