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

Pinned by tests/golden/noiseflow.npz (outputs of the imported reference with an injected z).
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
    assert float(scale.min()) >= 0            # signal_dependant.py:50
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


def _bn_eval(x, sd, pre):
    return F.batch_norm(x, sd[pre + '.running_mean'], sd[pre + '.running_var'], sd[pre + '.weight'], sd[pre + '.bias'],
                        training=False, eps=BN_EPS)


def shift_and_log_scale(sd, k, z0):
    p = f'model.{k}._shift_and_log_scale'
    h = F.relu(_bn_eval(F.conv2d(z0, sd[p + '.conv2d_1.weight'], sd[p + '.conv2d_1.bias'], padding=1), sd, p + '.net.1'))
    h = F.relu(_bn_eval(F.conv2d(h, sd[p + '.conv2d_2.weight'], sd[p + '.conv2d_2.bias']), sd, p + '.net.4'))
    h = F.pad(h, (1, 1, 1, 1, 0, 1), value=0.)          # ConstantPad3d((1,1,1,1,0,1)): +1 channel, +1 px border
    h[:, 4, :1, :] = 1.0; h[:, 4, -1:, :] = 1.0; h[:, 4, :, :1] = 1.0; h[:, 4, :, -1:] = 1.0
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


def forward(sd, noise, clean, iso):
    """noise_flow.py:113-130: x -> z and the summed log|det J| of the chain (eval-mode BatchNorm).
    Quirk kept: Conv2d1x1's log-det is sum(log_s) * W * W (`pixels*pixels`, conv2d1x1.py:49,65: square inputs assumed)."""
    z = noise
    obj = torch.zeros(noise.shape[0], dtype=torch.float32)
    for k in range(18):
        if k in CONV_IDX:
            z = F.conv2d(z, conv1x1_matrix(sd, k).view(4, 4, 1, 1))
            obj = obj + sd[f'model.{k}.log_s'].sum() * noise.shape[-1] * noise.shape[-1]
        elif k in COUPLING_IDX:
            z0, z1 = z[:, :2], z[:, 2:]
            shift, log_scale = shift_and_log_scale(sd, k, z0)
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


def loss(sd, noise, clean, iso):
    """noise_flow.py:132-165: (mean NLL per dimension, mean std of the noise)."""
    z, obj = forward(sd, noise, clean, iso)
    log_z = (-0.5 * (np.log(2 * np.pi) + z ** 2)).sum(dim=[1, 2, 3])
    nll = -(obj + log_z)
    sd_z = torch.sqrt(torch.var(noise, dim=[1, 2, 3])).mean()
    return nll.mean() / np.prod(noise.shape[1:]), sd_z
