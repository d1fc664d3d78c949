"""The eval step of the reference trainers as a product function, all on the device.

One iteration of `SID_Trainer.eval` (trainer_SID.py:208-248; trainer_LRID.py has the same body):

    if W % 16 != 0:  reflect-pad 4 px on every side, run the net, crop both back        (:221-228)
    if dst.ori:      imgs_lr *= ratio;  imgs_dn *= ratio                                 (:231-233)
    clamp both to [0, 1]                                                                 (:234-235)
    if brightness_correct and epoch < 0:  imgs_dn = IlluminanceCorrect(imgs_dn, imgs_hr) (:238-239)
    PSNR / SSIM of tensor2im(imgs_dn) and of tensor2im(imgs_lr) against tensor2im(imgs_hr), data_range 255  (:242-255)

and the bookkeeping around it: the `metrics` dict that goes to `metrics.pkl` (:247,314-315: name -> [PSNR, SSIM] of the
denoised frame, or (psnr_dn, ssim_dn) when plots are on) and the log line (:309-312).  Rendering (rawpy), plotting and
checkpoint selection stay outside (out of scope).  The IMX686 frame 4x1736x2312 takes the padded branch
(2312 % 16 = 8 -> 4x1744x2320).
"""
import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import PnnpError
from .metrics import IlluminanceCorrect, quality_assess


def evaluate(net, imgs_lr, imgs_hr, ratio=None, ori=False, brightness_correct=True, epoch=-1, corrector=None, with_lr=True):
    """-> dict(dn=<clamped (and corrected) prediction>, lr=<clamped input>, metrics=<device tensor [PSNR_dn, SSIM_dn, PSNR_lr, SSIM_lr]>).

    ``imgs_lr``, ``imgs_hr``: CUDA [1,C,H,W] (the eval loaders use batch_size 1); ``ratio``: scalar / [1] tensor, needed when
    ``ori``.  Nothing synchronises: read ``metrics`` (``.tolist()``) when the numbers are needed."""
    if not imgs_lr.is_cuda or not imgs_hr.is_cuda:
        raise PnnpError('evaluate needs CUDA tensors (pnnp_amd has no CPU path)')
    if imgs_lr.dim() != 4 or imgs_lr.shape[0] != 1:
        raise PnnpError('evaluate expects one frame per call: [1,C,H,W] (DataLoader batch_size 1, trainer_SID.py:52)')
    with torch.no_grad():
        lr_in = imgs_lr.contiguous().float()
        _, Cc, H, W = lr_in.shape
        pad = 4 if W % 16 != 0 else 0                         # :221 -- the reference tests the width only
        eng = getattr(net, 'engine', None)
        res = bool(getattr(net, 'res', False))
        if eng is not None and hasattr(eng, 'forward'):
            # the reflection is folded into the layout pass the input goes through anyway (no F.pad kernel, no padded copy); a `res` network
            # runs without its input residual, which the tail kernel adds after the crop
            out = eng.forward(lr_in, False, reflect_pad=pad, add_residual=not res)
            add_res = res
        else:
            out = net(F.pad(lr_in, (4, 4, 4, 4), mode='reflect')) if pad else net(lr_in)
            add_res = False
        if ori and ratio is None:
            raise PnnpError('ori=True needs the ratio (trainer_SID.py:231-233)')
        # (one frame per call: one scalar; a device tensor stays on the device -- nothing here synchronises)
        r_dev = ratio.reshape(-1)[:1].float().contiguous() if (ori and torch.is_tensor(ratio) and ratio.is_cuda) else None
        r = 1.0 if (not ori or r_dev is not None) else float(torch.as_tensor(ratio).reshape(-1)[0])
        # crop, residual, x ratio and both clamps (:226-235) in one pass
        out = out.contiguous()
        dn = torch.empty_like(lr_in); lr = torch.empty_like(lr_in)
        _lib.check(_lib.lib().pnnp_eval_post_f32(_lib.ptr(out), _lib.ptr(lr_in), _lib.ptr(dn), _lib.ptr(lr), Cc, H, W, out.shape[-2], out.shape[-1], pad,
                                                 C.c_float(r), _lib.ptr(r_dev), int(add_res), _lib.stream()), 'eval_post')
        if brightness_correct and epoch < 0:
            dn = (corrector or IlluminanceCorrect())(dn.contiguous(), imgs_hr)
        m_dn = quality_assess(dn, imgs_hr)
        m_lr = quality_assess(lr, imgs_hr) if with_lr else torch.full((2,), float('nan'), device=dn.device)
    return dict(dn=dn, lr=lr, metrics=torch.cat([m_dn, m_lr]))


class AverageMeter:
    """utils/utils.py AverageMeter as far as the eval log needs it."""

    def __init__(self):
        self.sum, self.count = 0.0, 0

    def update(self, v):
        self.sum += float(v); self.count += 1

    @property
    def avg(self):
        return self.sum / self.count if self.count else 0.0


class EvalLoop:
    """Accumulates what `SID_Trainer.eval` reports over a dataset: `metrics` (name -> [PSNR, SSIM], the content of
    metrics.pkl) and the log line of trainer_SID.py:309-312."""

    def __init__(self, net, ori=False, brightness_correct=True):
        self.net, self.ori, self.brightness_correct = net, ori, brightness_correct
        self.corrector = IlluminanceCorrect()
        self.reset()

    def reset(self):
        self.metrics = {}
        self.psnr, self.ssim = AverageMeter(), AverageMeter()
        self.psnr_lr, self.ssim_lr = AverageMeter(), AverageMeter()
        self._pending = []

    def step(self, name, imgs_lr, imgs_hr, ratio=None, epoch=-1):
        out = evaluate(self.net, imgs_lr, imgs_hr, ratio=ratio, ori=self.ori, brightness_correct=self.brightness_correct,
                       epoch=epoch, corrector=self.corrector)
        self._pending.append((name, out['metrics']))          # device tensors: one host sync per dataset, in finish()
        return out

    def finish(self, epoch=-1):
        """-> (metrics dict, log text).  Reads the accumulated device metrics once."""
        if self._pending:
            vals = torch.stack([m for _, m in self._pending]).cpu().tolist()
            for (name, _), (p_dn, s_dn, p_lr, s_lr) in zip(self._pending, vals):
                self.metrics[name] = [p_dn, s_dn]
                self.psnr.update(p_dn); self.ssim.update(s_dn)
                self.psnr_lr.update(p_lr); self.ssim_lr.update(s_lr)
            self._pending = []
        text = (f"Epoch {epoch}: PSNR={self.psnr.avg:.2f}\n"
                f"psnrs_lr={self.psnr_lr.avg:.2f}, psnrs_dn={self.psnr.avg:.2f}"
                f"\nssims_lr={self.ssim_lr.avg:.4f}, ssims_dn={self.ssim.avg:.4f}")
        return self.metrics, text
