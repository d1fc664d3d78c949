"""GPU parity of the eval epilogue (row f1): IlluminanceCorrect vs the reference's golden output,
PSNR/SSIM vs the oracle restatement of the scikit-image definition (parity unpinned: skimage absent)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_illuminance_correct_golden(golden_dir):
    from pnnp_amd.metrics import IlluminanceCorrect
    g = np.load(os.path.join(golden_dir, 'misc.npz'))
    a = torch.from_numpy(g['psnr_a'])
    pred = (a[:1] * 1.3 - 0.1).cuda(); src = torch.from_numpy(g['ic_src']).cuda()
    out = IlluminanceCorrect()(pred, src)
    np.testing.assert_allclose(out.cpu().numpy(), g['ic_out'], rtol=2e-6, atol=1e-7)
    # batch of predictions against one source, like the reference's forward()
    outb = IlluminanceCorrect()(pred.repeat(2, 1, 1, 1), src)
    np.testing.assert_allclose(outb[1].cpu().numpy(), g['ic_out'][0], rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('shape', [(4, 64, 64), (4, 70, 101), (4, 1424, 2128)])
def test_psnr_ssim_vs_oracle(shape):
    from oracle import metrics_np as M
    from pnnp_amd.metrics import quality_assess
    g = torch.Generator().manual_seed(shape[1])
    t = torch.rand(1, *shape, generator=g) ** 2
    o = (t + 0.03 * torch.randn(1, *shape, generator=g)).clamp(-0.1, 1.1)
    res = quality_assess(o.cuda(), t.cuda()).cpu().numpy()
    X, Y = M.tensor2im(o.numpy()), M.tensor2im(t.numpy())
    assert abs(res[0] - M.psnr(Y, X)) < 1e-3
    assert abs(res[1] - M.ssim(Y, X)) < 2e-5
