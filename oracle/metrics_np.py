"""ORACLE (test infrastructure, not product): the raw-domain eval metrics of the reference.

* ``tensor2im``            utils/visualization.py:9-24 (x255, clip to [0,255], HWC float)
* ``psnr``                 skimage.metrics.peak_signal_noise_ratio(data_range=255) (utils/visualization.py:29)
* ``ssim``                 skimage.metrics.structural_similarity(data_range=255, channel_axis=-1) (:30)
* ``illuminance_correct``  in oracle/net_torch.py (pinned by tests/golden/misc.npz)

scikit-image is NOT installed in the build image and the reference does not pin its version:
**parity unpinned** for PSNR/SSIM -- restated from the published definition (Wang et al. 2004 as
implemented by scikit-image: 7x7 uniform window, K1=0.01, K2=0.03, sample covariance NP/(NP-1), mean of
the SSIM map cropped by (win-1)//2, per channel, averaged over channels; float32 inputs stay float32).
"""
import numpy as np
from scipy.ndimage import uniform_filter


def tensor2im(t):
    a = np.asarray(t, np.float32)[0]
    return np.clip(np.transpose(a, (1, 2, 0)) * 255.0, 0, 255)


def psnr(true, test, data_range=255):
    err = np.mean((true.astype(np.float64) - test.astype(np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / err)


def ssim(im1, im2, data_range=255, win=7, k1=0.01, k2=0.03):
    vals = []
    for c in range(im1.shape[-1]):
        x = im1[..., c].astype(np.float32); y = im2[..., c].astype(np.float32)
        npx = win * win
        cov = npx / (npx - 1)
        ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
        uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
        vx, vy, vxy = cov * (uxx - ux * ux), cov * (uyy - uy * uy), cov * (uxy - ux * uy)
        c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        p = (win - 1) // 2
        vals.append(s[p:-p, p:-p].mean(dtype=np.float64))
    return float(np.mean(vals))
