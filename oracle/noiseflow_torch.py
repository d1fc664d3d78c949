"""ORACLE (test infrastructure, not product): plain torch-fp32 restatement of
``NoiseFlow.sample`` for the published arch string ``sdn|unc|unc|unc|unc|giso|unc|unc|unc|unc``.

Functional on the reference's state_dict (222 keys): ``model.0`` SignalDependantISO,
``model.{1,3,5,7,10,12,14,16}`` Conv2d1x1 (LU-parametrised), ``model.{2,4,6,8,11,13,15,17}``
AffineCoupling, ``model.9`` GainISO.

* ``sample``            archs/noise_flow.py:173-188 (reversed chain of ``_inverse``)
* ``conv1x1_inverse``   archs/flow_layers/conv2d1x1.py:47-92 (inverse of P·L·U in float64)
* ``coupling_inverse``  archs/flow_layers/affine_coupling.py:27-34,245-295 (BatchNorm in eval mode,
                        the SID trainer calls ``proxy_net.eval()``, trainer_SID.py:42)
* ``sdn_scale``         archs/flow_layers/signal_dependant.py:37-51
* ``gain_scale``        archs/flow_layers/gain.py:79-86

Pinned by tests/golden/noiseflow.npz (outputs of the imported reference with an injected z; ``tr_*``: the NLL, its
gradients and the updated BatchNorm buffers of one train-mode ``loss().backward()`` of the reference).
"""
import numpy as np
import torch
import torch.nn.functional as F

LEGAL_ISO = [50, 64, 80, 100, 125, 160, 200, 250, 320, 400, 500, 640, 800, 1000, 1250, 1600,
             2000, 2500, 3200, 4000, 5000, 6400, 8000, 10000, 12800, 16000, 20000, 25600, 32000, 40000, 51200]
CONV_IDX = (1, 3, 5, 7, 10, 12, 14, 16)
COUPLING_IDX = (2, 4, 6, 8, 11, 13, 15, 17)
BN_EPS = 1e-5


def _interp(table, iso):
    """searchsorted left/right + linear interpolation of exp(table) (signal_dependant.py:39-43)."""
    legal = torch.tensor(LEGAL_ISO, dtype=torch.float32)
    iso = torch.as_tensor(iso, dtype=torch.float32)
    l = int(torch.searchsorted(legal, iso, right=False))
    r = int(torch.searchsorted(legal, iso, right=True))
    iso_l, iso_r = legal[l], legal[r]
    pl, pr = torch.exp(table[l]), torch.exp(table[r])
    if float(iso_r - iso_l) != 0:
        return ((iso - iso_l) * pr + (iso_r - iso) * pl) / (iso_r - iso_l)
    return pl


def sdn_scale(sd, clean, iso, k=0):
    cam = _interp(sd[f'model.{k}.cam_param'], iso)
    beta1 = torch.exp(sd[f'model.{k}.beta1'] * cam[0])
    beta2 = torch.exp(sd[f'model.{k}.beta2'] * cam[1])
    gain = torch.exp(sd[f'model.{k}.gain'] * cam[2]) * iso
    scale = beta1 * clean / gain + beta2
    assert float(scale.detach().min()) >= 0            # signal_dependant.py:50
    return torch.sqrt(scale)


def gain_scale(sd, iso, k=9):
    cam = _interp(sd[f'model.{k}.cam_param'], iso)
    return torch.exp(cam * sd[f'model.{k}.gain_params']) * iso


def conv1x1_inverse_matrix(sd, k):
    """W^-1 = U^-1 L^-1 P^-1 with L, U assembled as in get_weight(); float64 inverses, float32 result."""
    l_mask = torch.tril(torch.ones(4, 4), -1)
    eye = torch.eye(4)
    l = sd[f'model.{k}.l'] * l_mask + eye
    u = sd[f'model.{k}.u'] * l_mask.t() + torch.diag(sd[f'model.{k}.sign_s'] * torch.exp(sd[f'model.{k}.log_s']))
    li = torch.inverse(l.double()).float()
    ui = torch.inverse(u.double()).float()
    return torch.matmul(ui, torch.matmul(li, sd[f'model.{k}.p'].inverse()))


def _bn_eval(x, sd, pre, training=False):
    """nn.BatchNorm2d: running statistics in eval mode; in training mode the batch statistics normalise and the running
    ones are updated in place (momentum 0.1, unbiased variance), as the module does."""
    if training:
        y = F.batch_norm(x, sd[pre + '.running_mean'], sd[pre + '.running_var'], sd[pre + '.weight'], sd[pre + '.bias'],
                         training=True, momentum=0.1, eps=BN_EPS)
        if pre + '.num_batches_tracked' in sd:
            sd[pre + '.num_batches_tracked'] += 1
        return y
    return F.batch_norm(x, sd[pre + '.running_mean'], sd[pre + '.running_var'], sd[pre + '.weight'], sd[pre + '.bias'],
                        training=False, eps=BN_EPS)


def shift_and_log_scale(sd, k, z0, training=False):
    p = f'model.{k}._shift_and_log_scale'
    h = F.relu(_bn_eval(F.conv2d(z0, sd[p + '.conv2d_1.weight'], sd[p + '.conv2d_1.bias'], padding=1), sd, p + '.net.1', training))
    h = F.relu(_bn_eval(F.conv2d(h, sd[p + '.conv2d_2.weight'], sd[p + '.conv2d_2.bias']), sd, p + '.net.4', training))
    h = F.pad(h, (1, 1, 1, 1, 0, 1), value=0.)          # ConstantPad3d((1,1,1,1,0,1)): +1 channel, +1 px border
    ring = torch.ones(h.shape[-2:]); ring[1:-1, 1:-1] = 0.0
    h = torch.cat([h[:, :4], (h[:, 4] + ring).unsqueeze(1)], dim=1)        # channel 4: ones on the pad ring (:268-271, out of place)
    h = F.conv2d(h, sd[p + '.conv2d_3.weight'], sd[p + '.conv2d_3.bias'])
    h = h * torch.exp(sd[p + '.logs'] * 3)
    shift, log_scale = torch.split(h, 2, dim=1)
    return shift, sd[p + '.scale'] * torch.tanh(log_scale)


def coupling_inverse(sd, k, z):
    z0, z1 = z[:, :2], z[:, 2:]
    shift, log_scale = shift_and_log_scale(sd, k, z0)
    return torch.cat([z0, (z1 - shift) * torch.exp(-log_scale)], dim=1)


def sample(sd, clean, iso, z):
    """noise_flow.py:173-188 with the prior draw ``z`` given explicitly."""
    x = z
    for k in range(17, -1, -1):
        if k in COUPLING_IDX:
            x = coupling_inverse(sd, k, x)
        elif k in CONV_IDX:
            x = F.conv2d(x, conv1x1_inverse_matrix(sd, k).view(4, 4, 1, 1))
        elif k == 9:
            x = x * gain_scale(sd, iso)
        else:
            x = x * sdn_scale(sd, clean, iso)
    return x


# ---------------------------------------------------------------- density direction (row f4)
def conv1x1_matrix(sd, k):
    """W = P L U (conv2d1x1.py:58-65)."""
    l_mask = torch.tril(torch.ones(4, 4), -1)
    l = sd[f'model.{k}.l'] * l_mask + torch.eye(4)
    u = sd[f'model.{k}.u'] * l_mask.t() + torch.diag(sd[f'model.{k}.sign_s'] * torch.exp(sd[f'model.{k}.log_s']))
    return torch.matmul(sd[f'model.{k}.p'], torch.matmul(l, u))


def forward(sd, noise, clean, iso, training=False):
    """noise_flow.py:113-130: x -> z and the summed log|det J| of the chain (eval-mode BatchNorm unless ``training``).
    Quirk kept: Conv2d1x1's log-det is sum(log_s) * W * W (`pixels*pixels`, conv2d1x1.py:49,65: square inputs assumed)."""
    z = noise
    obj = torch.zeros(noise.shape[0], dtype=torch.float32)
    for k in range(18):
        if k in CONV_IDX:
            z = F.conv2d(z, conv1x1_matrix(sd, k).view(4, 4, 1, 1))
            obj = obj + sd[f'model.{k}.log_s'].sum() * noise.shape[-1] * noise.shape[-1]
        elif k in COUPLING_IDX:
            z0, z1 = z[:, :2], z[:, 2:]
            shift, log_scale = shift_and_log_scale(sd, k, z0, training)
            z = torch.cat([z0, z1 * torch.exp(log_scale) + shift], dim=1)
            obj = obj + log_scale.sum(dim=[1, 2, 3])
        elif k == 9:
            scale = gain_scale(sd, iso) + z * 0.0
            z = z / scale
            obj = obj - torch.log(scale).sum(dim=[1, 2, 3])
        else:
            scale = sdn_scale(sd, clean, iso)
            z = z / scale
            obj = obj - torch.log(scale).sum(dim=[1, 2, 3])
    return z, obj


def loss(sd, noise, clean, iso, training=False):
    """noise_flow.py:132-165: (mean NLL per dimension, mean std of the noise)."""
    z, obj = forward(sd, noise, clean, iso, training)
    log_z = (-0.5 * (np.log(2 * np.pi) + z ** 2)).sum(dim=[1, 2, 3])
    nll = -(obj + log_z)
    sd_z = torch.sqrt(torch.var(noise, dim=[1, 2, 3])).mean()
    return nll.mean() / np.prod(noise.shape[1:]), sd_z


# ---------------------------------------------------------------- fitting (row f4): one NLL step of trainer_NF_SID.py:116-126
TRAINABLE_SUFFIX = ('.l', '.log_s', '.u', '.scale', '.logs', 'conv2d_1.weight', 'conv2d_1.bias', 'conv2d_2.weight', 'conv2d_2.bias',
                    'conv2d_3.weight', 'conv2d_3.bias', 'net.1.weight', 'net.1.bias', 'net.4.weight', 'net.4.bias',
                    'model.0.gain', 'model.0.beta1', 'model.0.beta2', 'model.9.cam_param', 'model.9.gain_params')


def trainable(key):
    """nn.Parameters with requires_grad (model.0.cam_param is frozen, signal_dependant.py:25; p, sign_s and the BatchNorm
    running statistics are buffers); the aliases net.0 / net.3 of conv2d_1 / conv2d_2 are the same tensors."""
    return key.endswith(TRAINABLE_SUFFIX) and '.net.0.' not in key and '.net.3.' not in key


def loss_and_grads(sd, noise, clean, iso):
    """net.train(); nll, sd_z = net.loss(...); nll.backward() (trainer_NF_SID.py:102,122-125) by autograd on this
    restatement.  Returns (nll, sd_z, {key: grad}, {key: updated BatchNorm buffer})."""
    sd = {k: v.clone() for k, v in sd.items()}
    leaves = {}
    for k in list(sd.keys()):
        if trainable(k):
            sd[k] = sd[k].detach().requires_grad_(True); leaves[k] = sd[k]
    nll, sd_z = loss(sd, noise, clean, iso, training=True)
    grads = torch.autograd.grad(nll, list(leaves.values()), allow_unused=True)
    g = {k: (torch.zeros_like(leaves[k]) if v is None else v) for k, v in zip(leaves.keys(), grads)}
    buffers = {k: v.detach() for k, v in sd.items() if 'running_' in k or 'num_batches' in k}
    return nll.detach(), sd_z, g, buffers
