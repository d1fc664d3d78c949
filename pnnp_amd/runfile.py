"""The reference's run files (`runfiles/<camera>/*.yml`) as the configuration surface of the hot path.

The trainers read them with `yaml.load` and pull the hot-path settings out of three sections
(trainer_SID.py:13-75, base_trainer.py:16-75):

    arch:       name, in_nc, out_nc, nf, nframes, res            -> the denoiser class in `archs`
    dst_train:  camera_type, noise_code, ori, clip, patch_size, crop_per_image, wp, bl, ratio_list
    hyper:      learning_rate, batch_size, last_epoch, stop_epoch, step_size, T      -> Adam + cosine restarts

`build(cfg)` turns an (unmodified) run file into the device pipeline: the network by name, the fused
`HipTrainStep` and the per-epoch learning rate.  Datasets, checkpoints and logging are outside the hot path
(SURVEY section 8): `python -m pnnp_amd.runfile RUNFILE --synthetic` drives the step on synthetic crops of the
configured shape, which is what bench.py measures.
"""
import argparse
import os
import sys

import numpy as np
import torch
import yaml

from . import archs, process
from .trainer import HipTrainStep, NoiseFlowFitStep, get_cos_lr
from .utils import load_weights

PROXY_DATASETS = ('NF_Syn_Dataset', 'IMX686_NF_Syn_Dataset')      # trainer_SID.py:463, trainer_LRID.py:419


def build_proxy(cfg, device='cuda', allow_uninitialised_proxy=False):
    """`arch_proxy` of a run file -> the NoiseFlow proxy the train step samples from (trainer_SID.py:33-42,
    trainer_LRID.py:33-39): built by name, weights from `<fast_ckpt>/<camera>_NoiseFlow_last_model.pth`.  A missing file raises
    FileNotFoundError, as the reference's unconditional ``torch.load`` does (trainer_SID.py:39-40, trainer_LRID.py:36-37): a wrong
    `fast_ckpt` must not train the denoiser on the noise of a random flow.  ``allow_uninitialised_proxy=True`` (tests, benchmarks:
    the checkpoints are not distributed with the reference) keeps the initial weights and says so.
    Deviation: trainer_SID.py:39 hard-codes 'SonyA7S2_NoiseFlow_last_model.pth' whatever the camera; here the file name follows
    dst.camera_type, which is what trainer_LRID.py:36 does and equals the reference for the SonyA7S2 run files.
    The two trainers differ and both are kept: trainer_SID loads by_name=True and calls ``.eval()`` (:41-42); trainer_LRID loads
    by_name=False and never calls ``.eval()`` (:37-39), so its proxy samples with BatchNorm on BATCH statistics (NoiseFlow.sample
    in training mode).  Returns None when the run file has no proxy."""
    proxy = cfg.get('arch_proxy')
    if not proxy or cfg.get('mode', 'train') != 'train':
        return None
    cls = getattr(archs, proxy['name'], None)
    if cls is None:
        raise KeyError(proxy['name'])
    net = cls(proxy)
    dst = cfg.get('dst_train', cfg.get('dst'))
    path = os.path.join(str(cfg.get('fast_ckpt', '')), f"{dst['camera_type']}_NoiseFlow_last_model.pth")
    lrid = dst.get('dataset') == 'IMX686_NF_Syn_Dataset'
    if os.path.exists(path):
        net = load_weights(net, torch.load(path, map_location='cpu'), by_name=not lrid)
    elif allow_uninitialised_proxy:
        print(f'No checkpoint file!!!  ({path}: the {proxy["name"]} proxy keeps its initial weights)', flush=True)
    else:
        raise FileNotFoundError(f'{path} (the NoiseFlow proxy checkpoint of arch_proxy; pass allow_uninitialised_proxy=True to '
                                f'train on a random-initialised flow)')
    net = net.to(device)
    return net.train() if lrid else net.eval()


def load(path):
    with open(path) as f:
        return yaml.safe_load(f)          # resolves the `<<: *base_dst` merges the run files use


def lr_schedule(hyper):
    """base_trainer.py:131-149: cosine with warm restarts, evaluated per epoch."""
    T = int(hyper.get('T', 1))
    period = (int(hyper['stop_epoch']) - int(hyper['last_epoch'])) // T
    peak, lr0 = int(hyper['step_size']), float(hyper['learning_rate'])
    return lambda epoch: get_cos_lr(epoch - int(hyper['last_epoch']), period=period, peak=peak, lr=lr0)


def build(cfg, device='cuda', rank=0, world=1, group=None, allow_uninitialised_proxy=False):
    """-> (net, train_step, lr_of_epoch, shapes).  Unknown architectures raise KeyError like `globals()[name]`."""
    arch = cfg['arch']
    cls = getattr(archs, arch['name'], None)
    if cls is None:
        raise KeyError(arch['name'])
    net = cls(arch)
    dst = cfg.get('dst_train', cfg.get('dst'))
    hyper = cfg['hyper']
    if arch['name'] == 'NoiseFlow':                                 # trainer_NF_SID.py: the proxy itself is fitted (NLL)
        net = net.to(device)
        step = NoiseFlowFitStep(net, lr=float(hyper['learning_rate']), camera_type=dst['camera_type'], noise_code=dst['noise_code'],
                                ori=bool(dst.get('ori', False)), clip=dst.get('clip', False), rank=rank, world=world, group=group)
        shapes = dict(batch=int(hyper.get('batch_size', 1)) * int(dst.get('crop_per_image', 1)), patch=int(dst['patch_size']),
                      channels=int(arch['x_shape'][0]))
        return net, step, lr_schedule(hyper), shapes
    archs.initialize_weights(net)                                   # trainer_SID.py:31
    net = net.to(device)
    proxy_kw = {}
    if dst.get('dataset') in PROXY_DATASETS:                        # the preprocess branch is chosen by dst_train.dataset
        proxy_net = build_proxy(cfg, device, allow_uninitialised_proxy)
        if proxy_net is None:
            raise KeyError('arch_proxy')                            # the reference would fail on self.proxy_net
        proxy_kw = dict(proxy_net=proxy_net)
        if dst['dataset'] == 'IMX686_NF_Syn_Dataset':               # trainer_LRID.py:33: legal_ratio = [1, 2, 4, 8, 16]
            proxy_kw['proxy_ratio_choices'] = (1, 2, 4, 8, 16)
    step = HipTrainStep(net, lr=float(hyper['learning_rate']), camera_type=dst['camera_type'], noise_code=dst['noise_code'],
                        ori=bool(dst.get('ori', False)), clip=dst.get('clip', False), rank=rank, world=world, group=group, **proxy_kw)
    shapes = dict(batch=int(hyper.get('batch_size', 1)) * int(dst.get('crop_per_image', 1)), patch=int(dst['patch_size']),
                  channels=int(arch['in_nc']) * int(arch.get('nframes', 1)))
    return net, step, lr_schedule(hyper), shapes


def main(argv=None):
    ap = argparse.ArgumentParser(description='run the PNNP hot path from a reference run file on synthetic crops')
    ap.add_argument('runfile')
    ap.add_argument('--synthetic', action='store_true', help='required: datasets are outside the hot path')
    ap.add_argument('--epochs', type=int, default=2)
    ap.add_argument('--steps', type=int, default=4, help='steps per epoch')
    ap.add_argument('--patch', type=int, default=0, help='override dst.patch_size')
    ap.add_argument('--allow-uninitialised-proxy', action='store_true',
                    help='arch_proxy run files: train on a random-initialised NoiseFlow when its checkpoint is missing (default: fail like the reference)')
    a = ap.parse_args(argv)
    if not a.synthetic:
        sys.exit('only --synthetic data is available here (datasets / rawpy are out of scope)')
    cfg = load(a.runfile)
    net, step, lr_of, sh = build(cfg, allow_uninitialised_proxy=a.allow_uninitialised_proxy)
    S = a.patch or sh['patch']
    hyper = cfg['hyper']
    e0 = int(hyper['last_epoch'])
    g = torch.Generator(device='cuda').manual_seed(0)
    for epoch in range(e0 + 1, e0 + 1 + a.epochs):
        lr = lr_of(epoch)
        losses, psnrs = [], []
        for k in range(a.steps):
            np.random.seed(1997 + epoch * 1000 + k)
            hr = torch.rand(sh['batch'], sh['channels'], S, S, device='cuda', generator=g)
            if isinstance(step, NoiseFlowFitStep):                  # log line of trainer_NF_SID.py:136: nll, std
                nll, sd = step.step(hr * 0.1, iso=(800, 1600, 3200)[k % 3], lr=lr)
                losses.append(float(nll)); psnrs.append(float(sd))
                continue
            if step.proxy_net is not None and step.proxy_ratio_choices is not None:
                out = step.step(hr * 0.05, lr=lr, iso=6400)         # LRID: the ISO comes with the data (trainer_LRID.py:423); 6400 = the calibrated one
            else:
                out = step.step(hr, lr=lr)
            losses.append(float(out[0])); psnrs.append(step.psnr_from(out, sh['channels'] * S * S))
        if isinstance(step, HipTrainStep):
            step.check_proxy()                                      # signal_dependant.py:50, deferred: at every epoch end, before the log line / a save
        # base_trainer / trainer_SID log line format: epoch, lr, loss, psnr
        if isinstance(step, NoiseFlowFitStep):
            print(f"Epoch {epoch:04d} | lr {lr:.3e} | nll {np.mean(losses):.5f} | std {np.mean(psnrs):.4f}", flush=True)
            continue
        print(f"Epoch {epoch:04d} | lr {lr:.3e} | loss {np.mean(losses):.5f} | psnr {np.mean(psnrs):.2f}", flush=True)
    return 0


if __name__ == '__main__':
    sys.exit(main())
