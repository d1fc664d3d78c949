"""ORACLE (test infrastructure, not product): plain torch-fp32 restatement of the
denoiser stack the HIP kernels implement.

Functional style on a ``{name: tensor}`` state dict with the reference's key
names, so the same weights drive the oracle and the HIP path.

* ``unet_forward``     archs/Unet.py:54-99 (UNetSeeInDark)
* ``resunet_forward``  archs/ResUnet.py:46-88 + archs/modules.py:130-153,176-197
* ``l1_clamp_loss``    losses/base_loss.py:92-107 at call site trainer_SID.py:99
* ``psnr_loss``        losses/__init__.py:4-15
* ``get_cos_lr``       base_trainer.py:140-149
* ``adam_step``        torch.optim.Adam defaults used at trainer_SID.py:44
* ``init_state``       archs/__init__.py:12-19 (N(0,0.02) init) + layer shapes
* ``illuminance_correct`` data_process/__init__.py:165-175

Pinned against outputs of the imported reference modules (torch 2.10.0 CPU):
tests/golden/unet_*.npz, resunet_*.npz, trainstep_*.npz, misc_*.npz.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- shapes
def unet_param_shapes(nf=32, in_nc=4, out_nc=4, nframes=1):
    """Ordered {name: shape} of UNetSeeInDark.state_dict()  (archs/Unet.py:16-51)."""
    s = {}
    def conv(name, ci, co, k=3):
        s[name + '.weight'] = (co, ci, k, k)
        s[name + '.bias'] = (co,)
    def convt(name, ci, co):
        s[name + '.weight'] = (ci, co, 2, 2)
        s[name + '.bias'] = (co,)
    c = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
    conv('conv1_1', in_nc * nframes, c[0]); conv('conv1_2', c[0], c[0])
    conv('conv2_1', c[0], c[1]); conv('conv2_2', c[1], c[1])
    conv('conv3_1', c[1], c[2]); conv('conv3_2', c[2], c[2])
    conv('conv4_1', c[2], c[3]); conv('conv4_2', c[3], c[3])
    conv('conv5_1', c[3], c[4]); conv('conv5_2', c[4], c[4])
    convt('upv6', c[4], c[3]); conv('conv6_1', c[4], c[3]); conv('conv6_2', c[3], c[3])
    convt('upv7', c[3], c[2]); conv('conv7_1', c[3], c[2]); conv('conv7_2', c[2], c[2])
    convt('upv8', c[2], c[1]); conv('conv8_1', c[2], c[1]); conv('conv8_2', c[1], c[1])
    convt('upv9', c[1], c[0]); conv('conv9_1', c[1], c[0]); conv('conv9_2', c[0], c[0])
    conv('conv10_1', c[0], out_nc, k=1)
    return s


def resunet_param_shapes(nf=32, in_nc=4, out_nc=4, nframes=1):
    """Ordered {name: shape} of ResUnet.state_dict()  (archs/ResUnet.py:15-44)."""
    s = {}
    c = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
    s['conv_in.weight'] = (c[0], in_nc * nframes, 3, 3); s['conv_in.bias'] = (c[0],)
    def block(name, ci, co):
        s[f'{name}.block.0.conv.conv.weight'] = (co, ci, 3, 3)
        s[f'{name}.block.1.conv.conv.weight'] = (co, co, 3, 3)
        if ci != co:
            s[f'{name}.short_cut.0.conv.conv.weight'] = (co, ci, 1, 1)
    def down(name, ci, co):
        s[f'{name}.conv.weight'] = (co, ci, 3, 3); s[f'{name}.conv.bias'] = (co,)
    def convt(name, ci, co):
        s[name + '.weight'] = (ci, co, 2, 2); s[name + '.bias'] = (co,)
    block('conv1', c[0], c[0]); down('pool1', c[0], c[1])
    block('conv2', c[1], c[1]); down('pool2', c[1], c[2])
    block('conv3', c[2], c[2]); down('pool3', c[2], c[3])
    block('conv4', c[3], c[3]); down('pool4', c[3], c[4])
    block('conv5', c[4], c[4])
    convt('upv6', c[4], c[3]); block('conv6', c[4], c[3])
    convt('upv7', c[3], c[2]); block('conv7', c[3], c[2])
    convt('upv8', c[2], c[1]); block('conv8', c[2], c[1])
    convt('upv9', c[1], c[0]); block('conv9', c[1], c[0])
    s['conv10.weight'] = (out_nc, c[0], 1, 1); s['conv10.bias'] = (out_nc,)
    return s


def init_state(shapes, seed=0, std=0.02):
    """archs/__init__.py:12-19: Conv2d weight+bias ~ N(0,0.02); ConvTranspose2d
    weight ~ N(0,0.02) and its bias left at the framework default.  For oracle use
    a single generator draws everything (including convT biases) ~ N(0,std)."""
    g = torch.Generator().manual_seed(seed)
    return {k: torch.randn(v, generator=g) * std for k, v in shapes.items()}


def init_state_he(shapes, seed=0, bias_std=0.02, res_scale=1.0, head_scale=1.0, head_bias=0.0):
    """Variance-preserving weights (std = sqrt(1.92 / fan_in), the LeakyReLU(0.2) gain) so that a 10-level nf=32 network
    keeps O(1) activations and live gradients in every layer -- N(0, 0.02) gives a nearly dead net at full depth.  Used by
    the 512x512 backward golden (res_scale = 0.25, head_scale = 0.004, head_bias = 0.5) (tests/golden/make_golden.py `nets512`) and the test that replays it."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in shapes.items():
        if k.endswith('.weight'):
            if k.startswith('upv'):                 # ConvTranspose2d [Cin][Cout][2][2] k2 s2: one tap per output pixel
                fan_in = v[0]
            else:
                fan_in = v[1] * v[2] * v[3]
            out[k] = torch.randn(v, generator=g) * (1.92 / fan_in) ** 0.5
            if '.block.1.' in k:                    # second conv of a ResidualBlock: keep the residual sum from growing
                out[k] = out[k] * res_scale
        else:
            out[k] = torch.randn(v, generator=g) * bias_std
    # the 1x1 output layer (conv10_1 / conv10): head_scale < 1 with head_bias = 0.5 keeps every prediction well inside (0, 1), so
    # the clamp of the loss never bites and its gradient is dense (what a trained denoiser looks like)
    for k in out:
        if k.startswith('conv10'):
            out[k] = out[k] * head_scale + (head_bias if k.endswith('.bias') else 0.0)
    return out


# --------------------------------------------------------------------------- UNet
def _lrelu(x):
    return F.leaky_relu(x, 0.2)


def unet_forward(sd, x, res=False):
    """archs/Unet.py:54-99."""
    def c3(name, t):
        return _lrelu(F.conv2d(t, sd[name + '.weight'], sd[name + '.bias'], padding=1))
    def up(name, t):
        return F.conv_transpose2d(t, sd[name + '.weight'], sd[name + '.bias'], stride=2)
    c1 = c3('conv1_2', c3('conv1_1', x));                   p1 = F.max_pool2d(c1, 2)
    c2 = c3('conv2_2', c3('conv2_1', p1));                  p2 = F.max_pool2d(c2, 2)
    c3_ = c3('conv3_2', c3('conv3_1', p2));                 p3 = F.max_pool2d(c3_, 2)
    c4 = c3('conv4_2', c3('conv4_1', p3));                  p4 = F.max_pool2d(c4, 2)
    c5 = c3('conv5_2', c3('conv5_1', p4))
    c6 = c3('conv6_2', c3('conv6_1', torch.cat([up('upv6', c5), c4], 1)))
    c7 = c3('conv7_2', c3('conv7_1', torch.cat([up('upv7', c6), c3_], 1)))
    c8 = c3('conv8_2', c3('conv8_1', torch.cat([up('upv8', c7), c2], 1)))
    c9 = c3('conv9_2', c3('conv9_1', torch.cat([up('upv9', c8), c1], 1)))
    out = F.conv2d(c9, sd['conv10_1.weight'], sd['conv10_1.bias'])
    return out + x if res else out


# --------------------------------------------------------------------------- ResUnet
def _resblock(sd, name, x):
    """modules.py:176-197 with is_activate=False (ResUnet.py:17-41): conv(no bias)
    -> ReLU -> conv(no bias) -> identity 'activation' -> + shortcut (1x1 conv, no
    bias, when channel counts differ)."""
    t = F.relu(F.conv2d(x, sd[f'{name}.block.0.conv.conv.weight'], None, padding=1))
    t = F.conv2d(t, sd[f'{name}.block.1.conv.conv.weight'], None, padding=1)
    key = f'{name}.short_cut.0.conv.conv.weight'
    sc = F.conv2d(x, sd[key], None) if key in sd else x
    return t + sc


def resunet_forward(sd, x, res=False):
    """archs/ResUnet.py:46-88.  ``conv3x3`` (modules.py:130-138) attaches its ReLU as
    a child of nn.Conv2d, which Conv2d.forward never calls: the down-sampling convs
    are stride-2 conv + bias with NO activation."""
    def down(name, t):
        return F.conv2d(t, sd[f'{name}.conv.weight'], sd[f'{name}.conv.bias'], stride=2, padding=1)
    def up(name, t):
        return F.conv_transpose2d(t, sd[name + '.weight'], sd[name + '.bias'], stride=2)
    t0 = F.relu(F.conv2d(x, sd['conv_in.weight'], sd['conv_in.bias'], padding=1))
    c1 = _resblock(sd, 'conv1', t0)
    c2 = _resblock(sd, 'conv2', down('pool1', c1))
    c3 = _resblock(sd, 'conv3', down('pool2', c2))
    c4 = _resblock(sd, 'conv4', down('pool3', c3))
    c5 = _resblock(sd, 'conv5', down('pool4', c4))
    c6 = _resblock(sd, 'conv6', torch.cat([up('upv6', c5), c4], 1))
    c7 = _resblock(sd, 'conv7', torch.cat([up('upv7', c6), c3], 1))
    c8 = _resblock(sd, 'conv8', torch.cat([up('upv8', c7), c2], 1))
    c9 = _resblock(sd, 'conv9', torch.cat([up('upv9', c8), c1], 1))
    out = F.conv2d(c9, sd['conv10.weight'], sd['conv10.bias'])
    return out + x if res else out


# --------------------------------------------------------------------------- loss etc.
def l1_clamp_loss(pred, hr):
    """trainer_SID.py:99: Unet_Loss()(pred.clamp(0,1), imgs_hr) == mean |.|."""
    return F.l1_loss(pred.clamp(0, 1), hr)


def psnr_loss(low, high):
    """losses/__init__.py:4-15: mean over the batch of -10*log10(MSE_b)."""
    if low.dim() <= 3:
        return -10.0 * torch.log(torch.mean((high - low) ** 2)) / math.log(10.0)
    per = [-10.0 * torch.log(torch.mean((high[i] - low[i]) ** 2)) / math.log(10.0)
           for i in range(low.shape[0])]
    return torch.stack(per).mean()


def get_cos_lr(step, period=1000, peak=20, lr=1e-4, ratio=0.2):
    """base_trainer.py:140-149 (SGDR with warm-up on restarts; returns absolute lr)."""
    T = step // period
    s = step % period
    if s <= peak and T > 0:
        mul = s / peak
    else:
        mul = (1 - ratio) * (np.cos((s - peak) / (period - peak) * math.pi) * 0.5 + 0.5) + ratio
    return lr * mul / (2 ** T)


def adam_step(params, grads, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam step (defaults of trainer_SID.py:44), in place.
    ``step`` is 1-based.  Same operation order as torch's single-tensor path."""
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k in params:
        g = grads[k]
        m[k].lerp_(g, 1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        params[k].addcdiv_(m[k], denom, value=-lr / bc1)


def illuminance_correct(pred, source):
    """data_process/__init__.py:165-175 for N == 1."""
    pred = pred.clamp(0, 1)
    keep = source != 1
    num = torch.dot(pred[keep], source[keep])
    den = torch.dot(pred[keep], pred[keep])
    return num / den * pred


def train_step(sd, m, v, step, lr_in, hr, lr=1e-4, arch='unet', res=False):
    """Fwd + L1(clamp) + bwd + Adam on a fixed (noisy, clean) pair
    (trainer_SID.py:93-101).  Returns (loss, psnr, grads)."""
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in sd.items()}
    fwd = unet_forward if arch == 'unet' else resunet_forward
    pred = fwd(leaves, lr_in, res=res)
    loss = l1_clamp_loss(pred, hr)
    loss.backward()
    grads = {k: t.grad for k, t in leaves.items()}
    with torch.no_grad():
        adam_step(sd, grads, m, v, step, lr=lr)
        psnr = psnr_loss(pred.detach().clamp(0, 1), hr.clamp(0, 1))
    return float(loss.detach()), float(psnr), grads
