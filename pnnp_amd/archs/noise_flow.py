"""NoiseFlow proxy with the reference's module structure and state_dict (222 keys), sampling on
HIP kernels (reference: archs/noise_flow.py:24-221, archs/flow_layers/{conv2d1x1,affine_coupling,
signal_dependant,gain}.py).

``sample(clean=, iso=)`` is what the denoiser-training hot path uses (trainer_SID.py:464-472,
trainer_LRID.py:420-427); ``forward``/``loss``/``inverse`` are the density direction.  In eval mode BatchNorm inside the
coupling networks uses its running statistics, as in the SID trainer which calls ``proxy_net.eval()`` (trainer_SID.py:42);
trainer_LRID.py:34-39 forgets that call and samples with batch statistics -- a reference quirk that is NOT reproduced.

Fitting the flow (trainer_NF_SID.py:102,116-126: ``net.train(); nll, _ = net.loss(...); nll.backward(); optimizer.step()``)
works unchanged: in training mode ``loss()`` runs the chain with batch statistics on the kernels of csrc/nf_train.hip and
returns an autograd-connected scalar whose ``backward()`` runs the hand-written backward kernels and fills ``.grad`` of every
trainable parameter.  Only the 4x4 / scalar parameter algebra (W = P L U, the ISO tables) is left to torch autograd.
"""
import ctypes as C

import numpy as np
import scipy.linalg
import torch
import torch.nn as nn

from .. import _lib
from .._lib import PnnpError

LEGAL_ISO = [50, 64, 80, 100, 125, 160, 200, 250, 320, 400, 500, 640, 800, 1000, 1250, 1600,
             2000, 2500, 3200, 4000, 5000, 6400, 8000, 10000, 12800, 16000, 20000, 25600, 32000, 40000, 51200]
BN_EPS = 1e-5


class Conv2d1x1(nn.Module):                       # flow_layers/conv2d1x1.py:19-45 (LU-parametrised)
    def __init__(self, num_channels=4, name='Conv2d1x1'):
        super().__init__()
        self.name = name
        w = np.linalg.qr(np.random.randn(num_channels, num_channels))[0].astype(np.float32)
        p, l, u = scipy.linalg.lu(w)
        s = np.diag(u)
        self.register_buffer('p', torch.tensor(p.astype(np.float32)))
        self.register_buffer('sign_s', torch.tensor(np.sign(s).astype(np.float32)))
        self.l = nn.Parameter(torch.tensor(l.astype(np.float32)))
        self.log_s = nn.Parameter(torch.tensor(np.log(np.abs(s)).astype(np.float32)))
        self.u = nn.Parameter(torch.tensor(np.triu(u, k=1).astype(np.float32)))

    def matrix(self):
        """conv2d1x1.py:58-65: W = P L U (forward direction)."""
        mask = torch.tril(torch.ones(4, 4), -1)
        l = self.l.detach().cpu() * mask + torch.eye(4)
        u = self.u.detach().cpu() * mask.t() + torch.diag(self.sign_s.cpu() * torch.exp(self.log_s.detach().cpu()))
        return torch.matmul(self.p.cpu(), torch.matmul(l, u))

    def inverse_matrix(self):
        """conv2d1x1.py:66-74: U^-1 L^-1 P^-1 with float64 inverses (host side, 4x4)."""
        mask = torch.tril(torch.ones(4, 4), -1)
        l = self.l.detach().cpu() * mask + torch.eye(4)
        u = self.u.detach().cpu() * mask.t() + torch.diag(self.sign_s.cpu() * torch.exp(self.log_s.detach().cpu()))
        li = torch.inverse(l.double()).float(); ui = torch.inverse(u.double()).float()
        return torch.matmul(ui, torch.matmul(li, self.p.cpu().inverse()))


class ShiftAndLogScale(nn.Module):                # flow_layers/affine_coupling.py:245-277
    def __init__(self, num_in=2, num_out=4, width=4):
        super().__init__()
        self.scale = nn.Parameter(torch.full((1,), 1e-4))
        self.conv2d_1 = nn.Conv2d(num_in, width, kernel_size=3, padding=1)
        nn.init.normal_(self.conv2d_1.weight, mean=0.0, std=width / 512 * 0.05); self.conv2d_1.bias.data.fill_(0.0)
        self.conv2d_2 = nn.Conv2d(width, width, kernel_size=1, padding=0)
        nn.init.normal_(self.conv2d_2.weight, mean=0.0, std=width / 512 * 0.05); self.conv2d_2.bias.data.fill_(0.0)
        self.net = nn.Sequential(self.conv2d_1, nn.BatchNorm2d(width), nn.ReLU(), self.conv2d_2, nn.BatchNorm2d(width), nn.ReLU())
        self.conv2d_3 = nn.Conv2d(width + 1, num_out, kernel_size=3, padding=0)
        self.conv2d_3.weight.data.fill_(0.0); self.conv2d_3.bias.data.fill_(0.0)
        self.logs = nn.Parameter(torch.zeros([1, num_out, 1, 1]))


class AffineCoupling(nn.Module):                  # flow_layers/affine_coupling.py:19-34
    def __init__(self, x_shape, name='real_nvp'):
        super().__init__()
        self.name = name
        self._shift_and_log_scale = ShiftAndLogScale(num_in=x_shape[0] // 2, num_out=2 * (x_shape[0] - x_shape[0] // 2))


class SignalDependantISO(nn.Module):              # flow_layers/signal_dependant.py:19-29
    def __init__(self, name='sdn'):
        super().__init__()
        self.name = name
        self.cam_param = nn.Parameter(torch.zeros(len(LEGAL_ISO), 3), requires_grad=False)
        self.gain = nn.Parameter(torch.tensor(-6.0))
        self.beta1 = nn.Parameter(torch.tensor(-5.0))
        self.beta2 = nn.Parameter(torch.tensor(-4.0))


class GainISO(nn.Module):                         # flow_layers/gain.py:65-72
    def __init__(self, name='giso'):
        super().__init__()
        self.name = name
        self.cam_param = nn.Parameter(torch.zeros(len(LEGAL_ISO)))
        self.gain_params = nn.Parameter(torch.tensor(-5.0))


def _interp(table, iso):
    """searchsorted(left/right) + linear interpolation of exp(table) (signal_dependant.py:39-43)."""
    legal = np.asarray(LEGAL_ISO, np.float32)
    iso = np.float32(iso)
    l = int(np.searchsorted(legal, iso, side='left')); r = int(np.searchsorted(legal, iso, side='right'))
    if l >= len(legal) or r >= len(legal):
        raise IndexError('iso beyond the calibrated table (the reference indexes out of range too)')
    pl, pr = np.exp(table[l].astype(np.float32)), np.exp(table[r].astype(np.float32))
    if legal[r] - legal[l] != 0:
        return ((iso - legal[l]) * pr + (legal[r] - iso) * pl) / (legal[r] - legal[l])
    return pl


# ---------------------------------------------------------------------------------------------- fitting (training mode)
_CPARAM_SIZES = (72, 4, 4, 4, 16, 4, 4, 4, 180, 4, 4, 1)        # the prm block of csrc/nf_train.hip, in order


def _coupling_params(ac):
    s = ac._shift_and_log_scale
    return [s.conv2d_1.weight, s.conv2d_1.bias, s.net[1].weight, s.net[1].bias, s.conv2d_2.weight, s.conv2d_2.bias,
            s.net[4].weight, s.net[4].bias, s.conv2d_3.weight, s.conv2d_3.bias, s.logs, s.scale]


def _interp_t(table, iso):
    """signal_dependant.py:39-43 / gain.py:74-78 as torch ops on the parameter's device (keeps the autograd graph)."""
    legal = np.asarray(LEGAL_ISO, np.float32)
    iso = np.float32(iso)
    l = int(np.searchsorted(legal, iso, side='left')); r = int(np.searchsorted(legal, iso, side='right'))
    if l >= len(legal) or r >= len(legal):
        raise IndexError('iso beyond the calibrated table (the reference indexes out of range too)')
    pl, pr = torch.exp(table[l]), torch.exp(table[r])
    if legal[r] - legal[l] != 0:
        return (float(iso - legal[l]) * pr + float(legal[r] - iso) * pl) / float(legal[r] - legal[l])
    return pl


class _PairChainNLL(torch.autograd.Function):
    """F = mean_b( sum of the pixel log-det terms of crop b - 0.5 * sum z_b^2 ) through the training-mode chain.

    Inputs that carry gradients: wstack [P,4,4] (Conv2d1x1 matrices, GainISO folded), ab [2] (signal-dependent scale), and the
    12 parameters of each coupling.  Everything heavy runs in libpnnp_hip.so (pnnp_nf_train_{fwd,bwd}_pair_f32)."""

    @staticmethod
    def forward(ctx, net, noise, clean, wstack, ab, *cparams):
        L = _lib.lib()
        B, Cc, H, W = noise.shape
        P = wstack.shape[0]
        dev = noise.device
        tiles, pb = L.pnnp_nf_train_tiles(B, H, W), L.pnnp_nf_train_pblocks(B, H, W)
        f32 = dict(dtype=torch.float32, device=dev)
        prm = torch.cat([t.detach().reshape(-1) for t in cparams]).view(P, 301)
        wm = wstack.detach().contiguous().view(P, 16)
        abd = ab.detach().contiguous()
        xs = torch.empty((P + 1, B, 4, H, W), **f32)
        xs[0].copy_(noise)
        h1 = torch.empty((P, B, 4, H, W), **f32); h2 = torch.empty_like(h1); out3 = torch.empty_like(h1)
        bn = torch.empty((P, 24), **f32)
        ldpart = torch.empty((P, tiles, 2), **f32)
        part = torch.empty(max(tiles * 197, pb * 28), **f32)
        st = _lib.stream()
        for k in range(P):
            cl = clean if k == net._sdn_pair else None
            _lib.check(L.pnnp_nf_train_fwd_pair_f32(_lib.ptr(xs[k]), _lib.ptr(cl), _lib.ptr(abd), _lib.ptr(wm[k]), _lib.ptr(prm[k]),
                                                    _lib.ptr(bn[k]), _lib.ptr(h1[k]), _lib.ptr(h2[k]), _lib.ptr(out3[k]),
                                                    _lib.ptr(xs[k + 1]), _lib.ptr(ldpart[k]), _lib.ptr(part), B, H, W, st),
                       'nf_train_fwd_pair')
        per = ldpart.view(P, B, tiles // B, 2).double()
        F = (per[..., 0].sum(dim=(0, 2)) - 0.5 * per[P - 1, :, :, 1].sum(dim=1)).mean().float()
        ctx.net, ctx.shape, ctx.clean = net, (B, H, W, P, tiles, pb), clean
        ctx.saved = (xs, h1, h2, out3, bn, prm, wm, abd, part)
        net._last_bn = bn                       # BatchNorm buffers are updated by the caller (needs N and the momentum)
        return F

    @staticmethod
    def backward(ctx, gF):
        L = _lib.lib()
        net, clean = ctx.net, ctx.clean
        B, H, W, P, tiles, pb = ctx.shape
        xs, h1, h2, out3, bn, prm, wm, abd, part = ctx.saved
        g = float(gF)                            # one host read per step (the reference's loop reads nll.item() as well)
        cobj = g / B
        f32 = dict(dtype=torch.float32, device=xs.device)
        sums = torch.empty((P, 319), **f32)
        dy2 = torch.empty((B, 4, H, W), **f32); dy1 = torch.empty_like(dy2)
        dv23 = torch.empty((B, 2, H, W), **f32)
        dbuf = [torch.empty((B, 4, H, W), **f32), torch.empty((B, 4, H, W), **f32)]
        st = _lib.stream()
        dz, dzmul = xs[P], -cobj                 # d(-0.5 z^2)/dz = -z
        for k in range(P - 1, -1, -1):
            cl = clean if k == net._sdn_pair else None
            dx = dbuf[k & 1]
            _lib.check(L.pnnp_nf_train_bwd_pair_f32(_lib.ptr(xs[k]), _lib.ptr(cl), _lib.ptr(abd), _lib.ptr(wm[k]), _lib.ptr(prm[k]),
                                                    _lib.ptr(bn[k]), _lib.ptr(h1[k]), _lib.ptr(h2[k]), _lib.ptr(out3[k]), _lib.ptr(dz),
                                                    C.c_float(dzmul), C.c_float(cobj), _lib.ptr(dx), _lib.ptr(sums[k]), _lib.ptr(dy2),
                                                    _lib.ptr(dy1), _lib.ptr(dv23), _lib.ptr(part), B, H, W, st), 'nf_train_bwd_pair')
            dz, dzmul = dx, 1.0
        # sums -> gradients in the order of the inputs
        o2, o3 = 197, 225
        cg = []
        for k in range(P):
            r = sums[k]
            cg += [r[o3:o3 + 72].view(4, 2, 3, 3), r[o3 + 72:o3 + 76], r[o2 + 24:o2 + 28], r[o2 + 20:o2 + 24],
                   r[o2:o2 + 16].view(4, 4, 1, 1), r[o2 + 16:o2 + 20], r[193:197], r[189:193],
                   r[0:180].view(4, 5, 3, 3), r[180:184], r[184:188].view(1, 4, 1, 1), r[188:189]]
        dw = sums[:, o3 + 76:o3 + 92].reshape(P, 4, 4)
        dab = sums[net._sdn_pair, o3 + 92:o3 + 94].clone() if net._sdn_pair is not None else torch.zeros(2, **f32)
        dnoise = dz if ctx.needs_input_grad[1] else None
        return (None, dnoise, None, dw, dab) + tuple(cg)


class NoiseFlow(nn.Module):
    """Drop-in for archs/noise_flow.py:24 (args keys: x_shape, arch, flow_permutation, param_inits, lu_decomp)."""

    def __init__(self, args=None):
        super().__init__()
        self.args = {'x_shape': (4, 256, 256), 'arch': 'sdn|unc|unc|unc|unc|gain|unc|unc|unc|unc',
                     'flow_permutation': 1, 'param_inits': None, 'lu_decomp': True}
        if args is not None:
            self.args.update(args)
        self.x_shape = self.args['x_shape']
        self.arch = self.args['arch']
        layers = []
        for i, lyr in enumerate(self.arch.split('|')):          # noise_flow.py:46-111
            if lyr == 'unc':
                if self.args['flow_permutation'] == 1:
                    layers.append(Conv2d1x1(self.x_shape[0], name=f'Conv2d_1x1_{i}'))
                layers.append(AffineCoupling(self.x_shape, name=f'unc_{i}'))
            elif lyr == 'sdn':
                layers.append(SignalDependantISO(name=f'sdn_{i}'))
            elif lyr == 'giso':
                layers.append(GainISO(name=f'giso_{i}'))
        self.model = nn.ModuleList(layers)
        self.seed, self.offset = 1997, 0
        self._sdn_pair, self._last_bn = None, None

    # ------------------------------------------------------------------ host-side step tables
    def _plan(self):
        """Reversed chain -> list of (AffineCoupling, Conv2d1x1, gain_after: bool, sdn_after: module|None)."""
        rev = list(self.model)[::-1]
        plan, i = [], 0
        while i < len(rev):
            m = rev[i]
            if isinstance(m, AffineCoupling):
                if i + 1 >= len(rev) or not isinstance(rev[i + 1], Conv2d1x1):
                    raise PnnpError('HIP NoiseFlow.sample supports unc = [Conv2d1x1, AffineCoupling] pairs (flow_permutation=1)')
                plan.append([m, rev[i + 1], None, None]); i += 2
            elif isinstance(m, GainISO):
                plan[-1][2] = m; i += 1
            elif isinstance(m, SignalDependantISO):
                plan[-1][3] = m; i += 1
            else:
                raise PnnpError(f'unsupported flow layer {type(m).__name__}')
        return plan

    def _tables(self, train=False):
        """Host-side step tables.  ``train``: the BatchNorm slots hold weight / bias (the kernel forms scale / offset from the batch
        statistics it finds on the device) and the cache does not depend on the running buffers -- which training-mode sampling moves
        every step; eval: the running statistics are folded in here."""
        key = tuple(p._version for p in self.parameters()) + (() if train else tuple(b._version for b in self.buffers()))
        cache = self.__dict__.setdefault('_tcache', {})
        hit = cache.get(bool(train))
        if hit is not None and hit[0] == key:
            return hit[1]
        steps = []
        for ac, cv, g_after, s_after in self._plan():
            s = ac._shift_and_log_scale
            f = lambda t: t.detach().cpu().float().numpy().reshape(-1)
            bn1, bn2 = s.net[1], s.net[4]
            def fold(bn):
                if train:
                    return f(bn.weight), f(bn.bias)
                sc = bn.weight.detach().cpu() / torch.sqrt(bn.running_var.cpu() + BN_EPS)
                return f(sc), f(bn.bias.detach().cpu() - bn.running_mean.cpu() * sc)
            s1, o1 = fold(bn1); s2, o2 = fold(bn2)
            vec = np.concatenate([f(s.conv2d_1.weight), f(s.conv2d_1.bias), s1, o1,
                                  f(s.conv2d_2.weight), f(s.conv2d_2.bias), s2, o2,
                                  f(s.conv2d_3.weight), f(s.conv2d_3.bias), np.exp(3.0 * f(s.logs)),
                                  f(s.scale), np.zeros(16, np.float32)]).astype(np.float32)
            assert vec.size == 317
            host = {'w': cv.matrix().numpy().astype(np.float32), 'log_s_sum': float(cv.log_s.detach().sum())}
            if g_after is not None:
                host['g_cam'] = g_after.cam_param.detach().cpu().numpy(); host['g_gain'] = np.float32(g_after.gain_params.item())
            if s_after is not None:
                host['s_cam'] = s_after.cam_param.detach().cpu().numpy()
                host['s_beta1'], host['s_beta2'], host['s_gain'] = (np.float32(s_after.beta1.item()), np.float32(s_after.beta2.item()),
                                                                   np.float32(s_after.gain.item()))
            steps.append((vec, cv.inverse_matrix().numpy().astype(np.float32), g_after, s_after, host))
        cache[bool(train)] = (key, steps)
        return steps

    # ------------------------------------------------------------------ API
    def forward(self, **kwargs):
        """noise_flow.py:113-130.  mode 'sample' / 'loss' / 'inverse' dispatch like the reference; the default is the
        density direction: kwargs noise [B,4,H,W] (CUDA), clean, iso -> (z, sum of log|det J| per crop).
        BatchNorm runs with its running statistics (eval mode); gradients are not produced (NLL *evaluation*)."""
        mode = kwargs.get('mode', 'forward')
        if mode == 'sample':
            return self.sample(**kwargs)
        if mode == 'loss':
            return self.loss(**kwargs)
        if mode == 'inverse':
            return self.inverse(**kwargs)
        x = kwargs['noise']
        _lib.require_cuda(x)
        x = x.contiguous().float()
        clean = kwargs['clean'].contiguous().float() if kwargs.get('clean') is not None else None
        iso = float(kwargs['iso'])
        B, Cc, H, W = x.shape
        L = _lib.lib()
        nblk = ((H + 31) // 32) * ((W + 31) // 32)
        partial = torch.zeros((8, B, nblk), dtype=torch.float32, device=x.device)
        scalar = 0.0                                   # log-det terms that do not depend on the pixel
        cur, nxt = x.clone(), torch.empty_like(x)
        tables = self._tables()[::-1]                  # forward order: pairs from the data side to the prior side
        for k, (vec, _winv, g_after, s_after, host) in enumerate(tables):    # parameters come from the cached host copies: no D2H here
            w = host['w']
            scalar += host['log_s_sum'] * W * W                       # conv2d1x1.py:49,65 `pixels*pixels` (square inputs assumed)
            # in the forward chain the layer that FOLLOWS this pair in the reversed plan precedes it here:
            # SignalDependantISO before the first pair, GainISO before the fifth (both stored on the previous reversed entry)
            a = b = np.float32(0.0); cl = None
            if s_after is not None:
                cam = _interp(host['s_cam'], iso)
                beta1 = np.exp(host['s_beta1'] * cam[0]); beta2 = np.exp(host['s_beta2'] * cam[1])
                gain = np.exp(host['s_gain'] * cam[2]) * np.float32(iso)
                a, b, cl = np.float32(beta1 / gain), np.float32(beta2), clean
                if cl is None:
                    raise PnnpError("NoiseFlow.forward needs 'clean' for the signal-dependent layer")
                if float(a) * float(clean.min()) + float(b) < 0:
                    raise AssertionError('scale must be non-negative')      # signal_dependant.py:50
            if g_after is not None:
                gs = np.float32(np.exp(_interp(host['g_cam'], iso) * host['g_gain']) * np.float32(iso))
                w = w / gs                                 # z = x / scale, then W: folded
                scalar -= float(np.log(gs)) * Cc * H * W   # gain.py:101-108
            vec = vec.copy(); vec[301:317] = w.reshape(-1)
            buf = (C.c_float * 317)(*vec.tolist())
            _lib.check(L.pnnp_nf_fwd_step_f32(_lib.ptr(cur), _lib.ptr(nxt), _lib.ptr(partial[k]), B, H, W, buf, _lib.ptr(cl),
                                              C.c_float(a), C.c_float(b), _lib.stream()), 'nf_fwd_step')
            cur, nxt = nxt, cur
        objective = partial.double().sum(dim=(0, 2)).float() + scalar
        return cur, objective

    def loss(self, **kwargs):
        """noise_flow.py:132-165: (mean NLL per dimension, mean per-crop std of the noise).  Eval mode: running BatchNorm
        statistics, no gradients.  Training mode: batch statistics, autograd-connected (see the module docstring)."""
        x = kwargs['noise']
        if self.training:
            return self._loss_train(**kwargs)
        kw = dict(kwargs); kw['mode'] = 'forward'
        z, objective = self.forward(**kw)
        log_z = (-0.5 * (np.log(2 * np.pi) + z.double() ** 2)).sum(dim=[1, 2, 3])        # prior N(0, I), noise_flow.py:190-219
        nll = -(objective.double() + log_z)
        sd_z = torch.sqrt(torch.var(x.float(), dim=[1, 2, 3])).mean()
        return (nll.mean() / float(np.prod(x.shape[1:]))).float(), sd_z

    def _forward_plan(self):
        """Forward-order pairs [(coupling, conv1x1, gain_before | None, sdn_before | None)] and the index of the sdn pair."""
        plan = self._plan()[::-1]
        self._sdn_pair = next((k for k, e in enumerate(plan) if e[3] is not None), None)
        return plan

    def _loss_train(self, **kwargs):
        """One training-mode evaluation of loss() (trainer_NF_SID.py:116-123).  The per-pixel work is in HIP; the 4x4
        matrices W = P L U (conv2d1x1.py:58-65), the ISO-table scalars and the pixel-independent log-det terms are torch
        expressions of the parameters so that autograd finishes the chain rule for l, u, log_s, gain, beta1/2, cam_param."""
        x = kwargs['noise']
        _lib.require_cuda(x)
        x = x.contiguous().float()
        clean = kwargs['clean'].contiguous().float() if kwargs.get('clean') is not None else None
        iso = float(kwargs['iso'])
        B, Cc, H, W = x.shape
        dev = x.device
        plan = self._forward_plan()
        P = len(plan)
        mask = torch.tril(torch.ones(4, 4, device=dev), -1); eye = torch.eye(4, device=dev)
        # the eight W = P L U at once (conv2d1x1.py:58-65), batched so that the step launches a handful of tiny kernels
        cvs = [e[1] for e in plan]
        ls = torch.stack([cv.l for cv in cvs]) * mask + eye
        log_s = torch.stack([cv.log_s for cv in cvs])
        us = torch.stack([cv.u for cv in cvs]) * mask.t() + torch.diag_embed(torch.stack([cv.sign_s for cv in cvs]) * torch.exp(log_s))
        wstack = torch.matmul(torch.stack([cv.p for cv in cvs]), torch.matmul(ls, us))
        scalar = log_s.sum() * float(W * W)                              # conv2d1x1.py:49,65 `pixels*pixels`
        ab = torch.zeros(2, device=dev)
        for k, (ac, cv, g_before, s_before) in enumerate(plan):
            if g_before is not None:                                     # gain.py:79-110: z = x / scale, log-det -log(scale) per element
                gs = torch.exp(_interp_t(g_before.cam_param, iso) * g_before.gain_params) * iso
                div = torch.ones(P, 1, 1, device=dev).index_put((torch.tensor([k], device=dev),), gs.reshape(1, 1, 1))
                wstack = wstack / div
                scalar = scalar - torch.log(gs) * float(Cc * H * W)
            if s_before is not None:                                     # signal_dependant.py:37-51
                if clean is None:
                    raise PnnpError("NoiseFlow.loss needs 'clean' for the signal-dependent layer")
                cam = _interp_t(s_before.cam_param, iso)
                beta1 = torch.exp(s_before.beta1 * cam[0]); beta2 = torch.exp(s_before.beta2 * cam[1])
                gain = torch.exp(s_before.gain * cam[2]) * iso
                ab = torch.stack([beta1 / gain, beta2])
                if float((ab[0].detach() * clean.min() + ab[1].detach())) < 0:
                    raise AssertionError('scale must be non-negative')  # signal_dependant.py:50
        cparams = [t for ac, _cv, _g, _s in plan for t in _coupling_params(ac)]
        F = _PairChainNLL.apply(self, x, clean, wstack, ab, *cparams)
        D = float(Cc * H * W)
        nll = -(F + scalar - 0.5 * D * float(np.log(2 * np.pi))) / D
        # BatchNorm buffers (nn.BatchNorm2d, momentum 0.1, unbiased variance); the kernels' means are bias-free
        n = float(B * H * W)
        with torch.no_grad():
            bn = self._last_bn.view(P, 2, 3, 4)                          # [pair][layer][mean, rstd, var][channel]
            sls = [e[0]._shift_and_log_scale for e in plan]
            bias = torch.stack([torch.stack([sl.conv2d_1.bias, sl.conv2d_2.bias]) for sl in sls])
            mean = ((bn[:, :, 0] + bias) * 0.1).reshape(2 * P, 4).unbind(0)
            var = (bn[:, :, 2] * (0.1 * n / max(n - 1.0, 1.0))).reshape(2 * P, 4).unbind(0)
            bns = [m for sl in sls for m in (sl.net[1], sl.net[4])]
            rm, rv = [m.running_mean for m in bns], [m.running_var for m in bns]
            torch._foreach_mul_(rm, 0.9); torch._foreach_add_(rm, list(mean))
            torch._foreach_mul_(rv, 0.9); torch._foreach_add_(rv, list(var))
            torch._foreach_add_([m.num_batches_tracked for m in bns], 1)
        sd_z = torch.sqrt(torch.var(x, dim=[1, 2, 3])).mean()
        return nll, sd_z

    def inverse(self, **kwargs):
        """noise_flow.py:167-171: the reversed chain applied to ``noise`` (= sample() with that tensor as the draw)."""
        kw = dict(kwargs); kw['z'] = kwargs['noise']
        if 'clean' not in kw:
            raise PnnpError("NoiseFlow.inverse needs 'clean' for the signal-dependent layer")
        return self.sample(**kw)

    def sample(self, **kwargs):
        """noise_flow.py:173-188.  kwargs: clean [B,4,H,W] (CUDA), iso (scalar / 0-dim tensor);
        optional ``z`` injects the prior draw (else N(0,1) from the counter-based generator).
        In TRAINING mode (trainer_LRID.py:34-39 never calls .eval() on the proxy it samples from, :420-427) the BatchNorm layers
        of every coupling use the statistics of the batch being sampled and move their running buffers, as nn.BatchNorm2d does."""
        clean = kwargs['clean'] if 'clean' in kwargs else kwargs['noise']
        _lib.require_cuda(clean)
        clean = clean.contiguous().float()
        iso = float(kwargs['iso'])
        B, Cc, H, W = clean.shape
        L = _lib.lib()
        z = kwargs.get('z')
        if z is None:
            z = torch.empty_like(clean)
            _lib.check(L.pnnp_normal_fill_f32(_lib.ptr(z), C.c_int64(z.numel()), C.c_uint64(self.seed), C.c_uint64(self.offset),
                                              _lib.stream()), 'normal_fill')
            self.offset += 1
        else:
            z = z.contiguous().float().clone()       # the ping-pong below overwrites its buffers
        cur, nxt = z, torch.empty_like(clean)
        train = self.training
        if train:
            tiles, pb = L.pnnp_nf_train_tiles(B, H, W), L.pnnp_nf_train_pblocks(B, H, W)
            f32 = dict(dtype=torch.float32, device=clean.device)
            h1 = torch.empty((B, 4, H, W), **f32); h2 = torch.empty_like(h1)
            part = torch.empty(max(tiles, pb) * 8, **f32)
            bn_all = torch.empty((len(self._plan()), 24), **f32)          # one statistics record per coupling: nothing waits for the host
            n = float(B * H * W)
            # the running buffers are moved through raw pointers below (no tensor version bump): the eval-mode tables are stale after this
            self.__dict__.setdefault('_tcache', {}).pop(False, None)
            pkey = (clean.device,) + tuple(p._version for p in self.parameters())
            pc = self.__dict__.get('_prm_cache')
            if pc is None or pc[0] != pkey:                             # the couplings' parameters as flat device rows, rebuilt only when they change
                pc = (pkey, [torch.cat([t.detach().reshape(-1) for t in _coupling_params(ac)]).contiguous() for ac, _c, _g, _s in self._plan()],
                      torch.eye(4, **f32).reshape(-1).contiguous())
                self.__dict__['_prm_cache'] = pc
            prm_rows, ident = pc[1], pc[2]
        _clean_div, _mix, _defer_check = kwargs.get('_clean_div'), kwargs.get('_mix'), bool(kwargs.get('_defer_check', False))
        tables, plan = self._tables(train), self._plan()
        n_pairs = len(plan)
        for pair_index, ((vec, winv, g_after, s_after, host), (ac, _cv, _g, _s)) in enumerate(zip(tables, plan)):
            if train:
                # batch statistics of this coupling's hidden maps for the tensor it is about to transform (its first two planes are
                # the coupling network's input in both directions), folded into the BatchNorm scale / offset slots of the step table
                sl = ac._shift_and_log_scale
                prm = prm_rows[pair_index]
                bn = bn_all[pair_index]
                _lib.check(L.pnnp_nf_train_stats_f32(_lib.ptr(cur), _lib.ptr(ident), _lib.ptr(prm), _lib.ptr(bn), _lib.ptr(h1), _lib.ptr(h2),
                                                     _lib.ptr(part), B, H, W, _lib.stream()), 'nf_train_stats')
                # the step kernel forms BatchNorm's scale / offset from `bn` itself (its table holds weight / bias in training mode) and
                # the running buffers move on the device: nn.BatchNorm2d semantics (momentum 0.1, unbiased variance) without .cpu()
                b1m, b2m = sl.net[1], sl.net[4]
                _lib.check(L.pnnp_nf_bn_update_f32(_lib.ptr(bn), _lib.ptr(sl.conv2d_1.bias.detach()), _lib.ptr(sl.conv2d_2.bias.detach()),
                                                   _lib.ptr(b1m.running_mean), _lib.ptr(b1m.running_var), _lib.ptr(b2m.running_mean),
                                                   _lib.ptr(b2m.running_var), C.c_void_p(b1m.num_batches_tracked.data_ptr()),
                                                   C.c_void_p(b2m.num_batches_tracked.data_ptr()), C.c_double(n), _lib.stream()), 'nf_bn_update')
                bn_dev = bn
            else:
                bn_dev = None
            w = winv.copy()
            if g_after is not None:        # gain.py:79-86: x * exp(cam*gain_params) * iso  (scalar: folded into W^-1)
                w *= np.float32(np.exp(_interp(host['g_cam'], iso) * host['g_gain']) * np.float32(iso))
            vec = vec.copy(); vec[301:317] = w.reshape(-1)
            a = b = np.float32(0.0); cl = None
            if s_after is not None:        # signal_dependant.py:37-51: sqrt(beta1*clean/gain + beta2)
                cam = _interp(host['s_cam'], iso)
                beta1 = np.exp(host['s_beta1'] * cam[0]); beta2 = np.exp(host['s_beta2'] * cam[1])
                gain = np.exp(host['s_gain'] * cam[2]) * np.float32(iso)
                a, b, cl = np.float32(beta1 / gain), np.float32(beta2), clean
            last = pair_index == n_pairs - 1
            mix = _mix if (last and _mix is not None) else None
            cdiv_t, cdiv_s = (None, 1.0)
            if cl is not None and _clean_div is not None:
                cdiv_t, cdiv_s = (_clean_div, 1.0) if torch.is_tensor(_clean_div) else (None, float(_clean_div))
            flag = None
            if cl is not None and _defer_check:
                flag = self._flag(clean.device)                             # the device records a negative scale; check_scale_flag() reads it
            elif cl is not None:
                if float(a) * float(clean.min()) + float(b) < 0:
                    raise AssertionError('scale must be non-negative')      # signal_dependant.py:50
            buf = vec.ctypes.data_as(C.POINTER(C.c_float))
            if mix is None and cdiv_t is None and cdiv_s == 1.0 and flag is None and bn_dev is None:
                _lib.check(L.pnnp_nf_step_f32(_lib.ptr(cur), _lib.ptr(nxt), B, H, W, buf, _lib.ptr(cl), C.c_float(a), C.c_float(b),
                                              C.c_float(1.0), _lib.stream()), 'nf_step')
            else:
                base, mul, lo, hi = mix if mix is not None else (None, 1.0, 0.0, 0.0)
                mul_t, mul_s = (mul, 1.0) if torch.is_tensor(mul) else (None, float(mul))
                _lib.check(L.pnnp_nf_step_mix_f32(_lib.ptr(cur), _lib.ptr(nxt), B, H, W, buf, _lib.ptr(cl), C.c_float(a), C.c_float(b),
                                                  C.c_float(1.0), _lib.ptr(cdiv_t), C.c_float(cdiv_s), _lib.ptr(base), _lib.ptr(mul_t),
                                                  C.c_float(mul_s), C.c_float(lo), C.c_float(hi),
                                                  C.c_void_p(flag.data_ptr()) if flag is not None else None, _lib.ptr(bn_dev), _lib.stream()), 'nf_step_mix')
            cur, nxt = nxt, cur
        return cur

    # ------------------------------------------------------------------ the trainer's preprocess around sample(), fused
    def _flag(self, device):
        f = getattr(self, '_scale_flag', None)
        if f is None or f.device != device:
            f = self._scale_flag = torch.zeros(1, dtype=torch.int32, device=device)
        return f

    def check_scale_flag(self):
        """The reference asserts ``scale >= 0`` inside every sample() (signal_dependant.py:50), which costs a device-to-host round
        trip per step.  ``sample_mixed`` records the condition in a device flag instead; this reads it (one sync) and raises the same
        AssertionError if any sample since the last check saw a negative scale."""
        f = getattr(self, '_scale_flag', None)
        if f is not None and int(f.item()) != 0:
            f.zero_()
            raise AssertionError('scale must be non-negative')

    def sample_mixed(self, hr, ratio, iso, clamp_lo=-float('inf'), clamp_hi=float('inf'), z=None):
        """preprocess() of the proxy branches as ONE chain of kernels (trainer_SID.py:463-472,481-485; trainer_LRID.py:419-427):

            clean = hr / ratio;  noise = sample(clean=clean, iso=iso) * ratio;  lr = (hr + noise).clamp(clamp_lo, clamp_hi)

        ``ratio``: a host scalar (LRID: one per batch) or a tensor [B] / [B,1,1,1] (SID: one per crop).  The division happens where
        the clean crop enters the signal-dependent scale, the multiply / add / clamp in the last step's store, and the scale >= 0
        assertion is deferred to ``check_scale_flag``: no elementwise torch kernels, no host synchronisation."""
        _lib.require_cuda(hr)
        r = ratio.reshape(-1).contiguous().float() if torch.is_tensor(ratio) else float(ratio)
        kw = {} if z is None else {'z': z}
        return self.sample(clean=hr, iso=iso, _clean_div=r, _mix=(hr.contiguous().float(), r, float(clamp_lo), float(clamp_hi)), _defer_check=True, **kw)
