"""Eval-side epilogue on the device (reference: trainer_SID.py:230-248): IlluminanceCorrect and the
raw-domain PSNR / SSIM, without leaving the GPU."""
import ctypes as C

import torch

from . import _lib


class IlluminanceCorrect(torch.nn.Module):
    """data_process/__init__.py:144-175 (ELD's brightness alignment): per image,
    out = <p,s>/<p,p> * p with p = clamp(predict,0,1), dots over source != 1."""

    def forward(self, predict, source):
        _lib.require_cuda(predict, source)
        predict = predict.contiguous().float(); source = source.contiguous().float()
        out = torch.empty_like(predict)
        ws = torch.empty(512, dtype=torch.float64, device=predict.device)
        n_img = predict.shape[0]
        for i in range(n_img):
            s = source[i] if source.shape[0] != 1 else source[0]
            _lib.check(_lib.lib().pnnp_illuminance_correct_f32(_lib.ptr(predict[i]), _lib.ptr(s), _lib.ptr(out[i]),
                                                               C.c_int64(predict[i].numel()), _lib.ptr(ws), _lib.stream()),
                       'illuminance_correct')
        return out

    def correct(self, predict, source):
        assert predict.shape[0] == 1
        return self.forward(predict, source)


def quality_assess(output, target):
    """PSNR / SSIM of two [1,C,H,W] (or [C,H,W]) tensors in [0,1], as
    quality_assess(tensor2im(output), tensor2im(target), data_range=255) (utils/visualization.py:9-31).
    Returns a device tensor [psnr, ssim] (no sync)."""
    _lib.require_cuda(output, target)
    a = output[0] if output.dim() == 4 else output
    b = target[0] if target.dim() == 4 else target
    a = a.contiguous().float(); b = b.contiguous().float()
    Cc, H, W = a.shape
    ws = torch.empty(2 * Cc * ((H + 31) // 32) * ((W + 31) // 32), dtype=torch.float64, device=a.device)
    out = torch.empty(2, dtype=torch.float32, device=a.device)
    # the reference passes (Y=target, X=output): image_true = target
    _lib.check(_lib.lib().pnnp_psnr_ssim_f32(_lib.ptr(b), _lib.ptr(a), _lib.ptr(out), Cc, H, W, _lib.ptr(ws), _lib.stream()),
               'psnr_ssim')
    return out
