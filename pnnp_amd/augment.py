"""Dataset-side crop / augmentation / white-balance gains on the device.

Mirrors the crop logic the reference's dataset base classes run on DataLoader workers:

* ``SynBase_Dataset.init_random_crop_point / data_aug / random_crop``  data_process/syn_datasets.py:69-107,162-173
  (8-way: rot90 k = mode % 4, flip = mode // 4 -- "row noise has a direction")
* ``RealBase_Dataset`` same names, 4-way (rot 180 = mode % 2, flip = mode // 2)  data_process/real_datasets.py:98-137
* ``random_gains``  data_process/unprocess.py:60-77 and its use syn_datasets.py:313-322
* the linear dark-shading subtraction in front of raw2bayer  real_datasets.py:360-372

The crop points and augmentation modes are drawn on the host with the reference's numpy call order (so
``np.random.seed`` reproduces them); the pixels never leave the GPU: ``crop_pack`` takes the uint16 Bayer
frame (2 B/px over PCIe) and emits normalised, cropped, rotated, flipped and gained float32 crops in one
kernel (csrc/cropaug.hip).  No CPU path.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributions as tdist

from . import _lib
from .isp_ops import _to_dev, _ret


def random_gains(camera_type='SonyA7S2'):
    """unprocess.py:60-77: brightening gain 1/N(0.8, 0.1) (torch RNG), red gain U(a,b) (numpy RNG), blue gain a
    quadratic fit of the red gain.  Returns three float32 tensors of shape [1] like the reference."""
    n = tdist.Normal(loc=torch.tensor([0.8]), scale=torch.tensor([0.1]))
    rgb_gain = 1.0 / n.sample()
    if camera_type == 'SonyA7S2':
        red_gain = np.random.uniform(1.75, 2.65)
        fit = [14.65, -9.63942308, 1.80288462]
    elif camera_type == 'IMX686':
        red_gain = np.random.uniform(1.4, 2.3)
        fit = [6.14381188, -3.65620261, 0.70205967]
    else:
        raise NotImplementedError
    blue_gain = fit[0] + fit[1] * red_gain + fit[2] * red_gain ** 2
    return rgb_gain, torch.FloatTensor(np.array([red_gain])).view(1), torch.FloatTensor(np.array([blue_gain])).view(1)


class CropAugment:
    """The crop state of a dataset object: ``args`` needs H, W (raw frame), patch_size, crop_per_image.
    ``ways=8`` is SynBase_Dataset's augmentation, ``ways=4`` RealBase_Dataset's."""

    def __init__(self, args, ways=8):
        self.args = dict(args)
        self.ways = ways
        self.get_shape()
        self.h_start, self.w_start, self.h_end, self.w_end = [], [], [], []
        self.aug = np.zeros(self.args['crop_per_image'], np.int64)

    def get_shape(self):
        self.H, self.W = self.args['H'], self.args['W']
        self.h, self.w, self.c = self.H // 2, self.W // 2, 4

    def init_random_crop_point(self, mode='non-overlapped', raw_crop=False):
        """Same numpy draws, in the same order, as syn_datasets.py:69-98 / real_datasets.py:98-127."""
        self.h_start, self.w_start, self.h_end, self.w_end = [], [], [], []
        ps = self.args['patch_size']
        self.aug = np.random.randint(self.ways, size=self.args['crop_per_image'])
        h, w = (self.H, self.W) if raw_crop else (self.h, self.w)
        if mode == 'non-overlapped':
            nh, nw = h // ps, w // ps
            h0 = np.random.randint(0, h - nh * ps + 1)
            w0 = np.random.randint(0, w - nw * ps + 1)
            for i in range(nh):
                for j in range(nw):
                    self.h_start.append(h0 + i * ps); self.w_start.append(w0 + j * ps)
                    self.h_end.append(h0 + (i + 1) * ps); self.w_end.append(w0 + (j + 1) * ps)
        else:
            for _ in range(self.args['crop_per_image']):
                h0 = np.random.randint(0, h - ps + 1)
                w0 = np.random.randint(0, w - ps + 1)
                self.h_start.append(h0); self.w_start.append(w0)
                self.h_end.append(h0 + ps); self.w_end.append(w0 + ps)

    def _rot_flip(self, mode):
        mode = int(mode)
        if self.ways == 8:
            return mode % 4, mode // 4
        return 2 * (mode % 2), mode // 2

    def data_aug(self, data, mode=0):
        """One crop [..., ps, ps] through the device kernel (square crops, like the datasets use)."""
        x, host = _to_dev(data)
        x = x.float()
        lead = x.shape[:-2]
        ps = x.shape[-1]
        if x.shape[-2] != ps:
            raise _lib.PnnpError('data_aug: square crops only')
        img = x.reshape(-1, ps, ps)
        rot, flip = self._rot_flip(mode)
        out = _crop_aug(img, [(0, 0, rot, flip)], ps, None, False)[0]
        return _ret(out.reshape(*lead, ps, ps), host)

    def _desc(self, n):
        return [(int(self.h_start[i]), int(self.w_start[i])) + self._rot_flip(self.aug[i]) for i in range(n)]

    def random_crop(self, img, gains=None, clip=False):
        """syn_datasets.py:162-173: packed image [c,h,w] -> [crop_per_image, c, ps, ps] float32 (device kernel)."""
        x, host = _to_dev(img)
        n, ps = self.args['crop_per_image'], self.args['patch_size']
        return _ret(_crop_aug(x.float(), self._desc(n), ps, gains, clip), host)

    def crop_pack(self, raw, wp=16383, bl=512, norm=True, clip=True, bias=np.array([0, 0, 0, 0]), gains=None,
                  post_clip=False, darkshading=None, dark_add=0.0):
        """Fused ``random_crop(raw2bayer(raw - darkshading + dark_add, wp, bl, norm, clip, bias))`` [* gains]
        on a uint16 Bayer frame [H,W]: -> [crop_per_image,4,ps,ps] float32 on the device.
        ``gains`` = (rgb, red, blue) scalars applied as syn_datasets.py:317-319 does; ``post_clip`` = the final
        ``hr_crops.clip(0,1)`` (:341)."""
        x, host = _to_dev(raw)
        if x.dtype != torch.uint16 or x.dim() != 2:
            raise _lib.PnnpError('crop_pack wants a uint16 [H,W] Bayer frame')
        H, W = x.shape
        n, ps = self.args['crop_per_image'], self.args['patch_size']
        desc = torch.tensor(self._desc(n), dtype=torch.int32).reshape(n, 4).to(x.device)
        out = torch.empty((n, 4, ps, ps), dtype=torch.float32, device=x.device)
        black = (C.c_double * 4)(*[float(b) + float(bl) for b in np.broadcast_to(np.asarray(bias, np.float64).reshape(-1), (4,))])
        g = _gains_tensor(gains, n, x.device)
        dk = None
        if darkshading is not None:
            dk, _ = _to_dev(darkshading)
            if dk.dtype not in (torch.float32, torch.float64) or tuple(dk.shape) != (H, W):
                raise _lib.PnnpError('darkshading must be a float32/float64 [H,W] map')
        _lib.check(_lib.lib().pnnp_crop_pack_bayer_u16(
            _lib.ptr(x), H, W, _lib.ptr(dk), int(dk is not None and dk.dtype == torch.float64), C.c_double(float(dark_add)),
            _lib.ptr(out), n, ps, _lib.ptr(desc), _lib.ptr(g), black, C.c_double(float(wp)), int(bool(norm)), int(bool(clip)),
            int(bool(post_clip)), _lib.stream()), 'crop_pack_bayer')
        return _ret(out, host)


def _gains_tensor(gains, n, device):
    if gains is None:
        return None
    g = np.asarray([[float(np.asarray(v).reshape(-1)[0]) for v in gains]] * n, np.float64)
    return torch.from_numpy(g).to(device)


def _crop_aug(img, desc, ps, gains, clip):
    _lib.require_cuda(img)
    img = img.contiguous()
    Cc, h, w = img.shape
    n = len(desc)
    d = torch.tensor(desc, dtype=torch.int32).reshape(n, 4).to(img.device)
    out = torch.empty((n, Cc, ps, ps), dtype=torch.float32, device=img.device)
    g = _gains_tensor(gains, n, img.device)
    _lib.check(_lib.lib().pnnp_crop_aug_f32(_lib.ptr(img), Cc, h, w, _lib.ptr(out), n, ps, _lib.ptr(d), _lib.ptr(g),
                                            int(bool(clip)), _lib.stream()), 'crop_aug')
    return out
