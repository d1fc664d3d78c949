"""ResUnet on hand-written HIP kernels (reference: archs/ResUnet.py:3-88, building blocks
archs/modules.py:130-153,176-197).

Same contract as UNetSeeInDark: the reference's constructor, attribute names and state_dict
keys (``conv_in``, ``conv{1..9}.block.{0,1}.conv.conv.weight``, ``conv{6..9}.short_cut.0.conv.conv
.weight``, ``pool{1..4}.conv.{weight,bias}``, ``upv{6..9}``, ``conv10``); children only own
parameters, forward/backward run through libpnnp_hip.so.

Reference quirks kept: ``conv3x3`` attaches its ReLU as a child of nn.Conv2d, which never runs
(the down-sampling convs are stride-2 conv + bias, NO activation); ResidualBlock is built with
``is_activate=False`` so the block output has no activation: out = conv(relu(conv(x))) + shortcut(x).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import ops
from .._lib import PnnpError
from .unet import FlatParams, _Bufs, _EngineBase, RELU


class _ConvHolder(nn.Module):      # modules.py:140-153 convWithBN(is_bn=False): .conv = Sequential(conv=Conv2d(bias=False))
    def __init__(self, ci, co, k):
        super().__init__()
        self.conv = nn.Sequential(OrderedDict([('conv', nn.Conv2d(ci, co, kernel_size=k, padding=k // 2, stride=1, bias=False))]))


class _ResBlockHolder(nn.Module):  # modules.py:176-197
    def __init__(self, ci, co):
        super().__init__()
        self.block = nn.Sequential(_ConvHolder(ci, co, 3), _ConvHolder(co, co, 3))
        self.short_cut = nn.Sequential(_ConvHolder(ci, co, 1)) if ci != co else nn.Sequential(OrderedDict([]))


class _DownHolder(nn.Module):      # modules.py:130-138 conv3x3(stride=2)
    def __init__(self, ci, co):
        super().__init__()
        self.conv = nn.Conv2d(ci, co, kernel_size=3, padding=1, stride=2)


class ResUnetEngine(_EngineBase):
    def __init__(self, module):
        self._init_base()
        self.m = module
        self.params = FlatParams(module)
        self.bufs = {}
        self.packed = {}
        nf = module.nf
        if nf % 8:
            raise PnnpError('ResUnet on HIP needs nf % 8 == 0')
        self.ch = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
        self.cin = module.in_nc * module.nframes
        self.cin_pad = (self.cin + 7) // 8 * 8
        self.cout = module.out_nc
        self.cout_pad = (self.cout + 7) // 8 * 8

    # ---------------------------------------------------------------- weights
    def _buf(self, key, n, dev, dtype=torch.float32):
        k = (key, dev)
        if k not in self.packed or self.packed[k].numel() != n:
            self.packed[k] = torch.empty(n, dtype=dtype, device=dev)
        return self.packed[k]

    def grad_out_channels(self, B, H, W):
        """channels of the NHWC loss gradient backward() wants (UNetEngine.grad_out_channels)"""
        return 4 if (self.cout == 4 and self._pol.use_thin_head(self.ch[0], self.cout, B * H * W)) else self.cout_pad

    def pack_weights(self, train):
        """Re-pack every layer's weights for the kernels; the job table is built once per device / mode / parameter storage
        and runs in a few launches per step (ops.PackJobs)."""
        dev = self.params.flat.device
        P = dict(self.m.named_parameters())
        key = (dev, train, self._pol.key(), tuple(p.data_ptr() for p in P.values()))
        if self._jobs_key != key:
            self._jobs, self._jobs_key = self._build_pack_jobs(train, dev, P), key
        self._jobs.run()

    def _build_pack_jobs(self, train, dev, P):
        jobs = ops.PackJobs(cap=512)
        W = {}
        self.WU, self.WX = {}, {}
        self.WH, self.WS = {}, {}              # fp16x2 packs (forward, backward-data) per layer; the weight tensor's amax slot (csrc/h2.h)
        self.WM = {}                           # pointwise / stride-2 / ConvTranspose2d layers on the fp16x2 GEMM kernel (csrc/gemm_h2s.hip): kind-6 packs
        h2pw = self._pol.h2 and self._pol.h2_pointwise
        def conv(name, pname, cin_pad=None, cout_pad=None, dgrad=True, c1=None):
            w = P[pname]
            co, ci, kh, kw = w.shape
            t = kh * kw
            cip = cin_pad or ci
            bwd = train and dgrad
            xf, xd = self._pol.use_x3(co, cip, t, c1)
            xd = xd and bwd
            hf, hd = self._pol.use_h2(co, cip, t, c1)
            hf, hd = hf and xf, hd and xd
            if hf or hd:                                               # the fp16x2 kernel takes what bf16x3 would have taken
                self.WH[name] = (self._buf(name + ':h2f', ops.h2_weight_bytes(cip, co), dev, torch.uint8) if hf else None,
                                 self._buf(name + ':h2d', ops.h2_weight_bytes(co, ci), dev, torch.uint8) if hd else None)
                self.WS[name] = jobs.add_h2(w, self.WH[name][0], self.WH[name][1], cin_pad=(cip + 15) // 16 * 16)
                xf, xd = xf and not hf, xd and not hd
            p1 = t == 1 and self._pol.use_x3_pointwise(ci, co) and self._pol.use_x3_pointwise(co, ci) and (c1 is None or c1 % 32 == 0)
            if p1 and h2pw and ops.gemm_h2_supported(c1 if c1 else ci, co) and ops.gemm_h2_supported(co, ci):
                mf = self._buf(name + ':h2mf', ops.h2mat_bytes(ci, co), dev, torch.uint8)
                md = self._buf(name + ':h2md', ops.h2mat_bytes(co, ci), dev, torch.uint8) if bwd else None
                self.WS[name] = jobs.add_h2_1x1(w, mf, md)
                W[name] = (None, None); self.WU[name] = (None, None); self.WM[name] = (mf, md if bwd else mf)
                return
            if p1:                                                     # 1x1 (ResidualBlock shortcut) on the pointwise bf16x3 kernel
                x3f = self._buf(name + ':x3f', ops.x3mat_bytes(ci, co), dev, torch.uint8)
                x3d = self._buf(name + ':x3d', ops.x3mat_bytes(co, ci), dev, torch.uint8) if bwd else None
                jobs.add_x3_1x1(w, x3f, x3d)
                W[name] = (None, None); self.WU[name] = (None, None); self.WX[name] = (x3f, x3d if bwd else x3f)
                return
            wf, wd = self._wino(co, ci, t)
            wf, wd = wf and not (xf or hf), wd and bwd and not (xd or hd)
            df, dd = not (xf or wf or hf), bwd and not (xd or wd or hd)        # what is left for the direct fp32 kernels
            f = self._buf(name + ':f', t * cip * co, dev) if df else None
            d = self._buf(name + ':d', t * (cout_pad or co) * ci, dev) if dd else None
            if df or dd:
                jobs.add_conv(w, f, d, cin_pad=cin_pad, cout_pad=cout_pad)
            uf = self._buf(name + ':uf', 16 * co * ci, dev) if wf else None
            ud = self._buf(name + ':ud', 16 * co * ci, dev) if wd else None
            if wf or wd:
                jobs.add_wino(w, uf, ud)
            x3f = self._buf(name + ':x3f', ops.x3_weight_bytes(cip, co), dev, torch.uint8) if xf else None
            x3d = self._buf(name + ':x3d', ops.x3_weight_bytes(co, ci), dev, torch.uint8) if xd else None
            if xf or xd:
                jobs.add_x3(w, x3f, x3d, cin_pad=(cip + 15) // 16 * 16)
            W[name] = (f, d)
            self.WU[name] = (uf, ud)
            self.WX[name] = (x3f, x3d)
        conv('conv_in', 'conv_in.weight', cin_pad=self.cin_pad, dgrad=False)
        for i in range(1, 10):
            conv(f'b{i}_0', f'conv{i}.block.0.conv.conv.weight', c1=self.ch[9 - i] if i >= 6 else None)     # decoder: cat([up, skip])
            conv(f'b{i}_1', f'conv{i}.block.1.conv.conv.weight')
            if i >= 6:
                conv(f'sc{i}', f'conv{i}.short_cut.0.conv.conv.weight', c1=self.ch[9 - i])
        for l in range(1, 5):
            w = P[f'pool{l}.conv.weight']
            co, ci = w.shape[0], w.shape[1]
            if h2pw and self._pol.use_x3_pointwise(ci, co) and ops.gemm_h2_supported(ci, co) and ops.gemm_h2_supported(co, ci):
                f = self._buf(f'pool{l}:h2mf', ops.h2mat_bytes(9 * ci, co), dev, torch.uint8)
                d = self._buf(f'pool{l}:h2md', 9 * ops.h2mat_bytes(co, ci), dev, torch.uint8) if train else None
                self.WS[f'pool{l}'] = jobs.add_h2_s2(w, f, d)
                self.WM[f'pool{l}'] = (f, d if train else f)
                continue
            if self._pol.use_x3_pointwise(ci, co) and self._pol.use_x3_pointwise(co, ci):      # stride-2 conv on the pointwise bf16x3 kernel
                f = self._buf(f'pool{l}:x3f', ops.x3mat_bytes(9 * ci, co), dev, torch.uint8)
                d = self._buf(f'pool{l}:x3d', 9 * ops.x3mat_bytes(co, ci), dev, torch.uint8) if train else None
                jobs.add_x3_s2(w, f, d)
                self.WX[f'pool{l}'] = (f, d if train else f)
                continue
            f = self._buf(f'pool{l}:f', w.numel(), dev)
            jobs.add_conv(w, f, None)
            d = None
            if train:
                d = self._buf(f'pool{l}:d', w.numel(), dev)
                jobs.add_s2_dgrad(w, d)
            W[f'pool{l}'] = (f, d)
        for i in range(6, 10):
            w = P[f'upv{i}.weight']
            ci, co = w.shape[0], w.shape[1]
            # (only beside an fp16x2 shortcut: its backward-data leaves the amax slot of the summed gradient this layer's backward splits)
            if h2pw and f'sc{i}' in self.WM and self._pol.use_x3_pointwise(ci, 4 * co) and ops.gemm_h2_supported(ci, 4 * co) and ops.gemm_h2_supported(co, ci):
                f = self._buf(f'upv{i}:h2mf', ops.h2mat_bytes(ci, 4 * co), dev, torch.uint8)
                d = self._buf(f'upv{i}:h2md', ops.h2mat_bytes(4 * co, ci), dev, torch.uint8) if train else None
                self.WS[f'upv{i}'] = jobs.add_h2_convt(w, f, d)
                self.WM[f'upv{i}'] = (f, d if train else f)
                continue
            if self._pol.use_x3_pointwise(ci, 4 * co) and self._pol.use_x3_pointwise(co, ci):
                f = self._buf(f'upv{i}:x3f', ops.x3mat_bytes(ci, 4 * co), dev, torch.uint8)
                d = self._buf(f'upv{i}:x3d', ops.x3mat_bytes(4 * co, ci), dev, torch.uint8) if train else None
                jobs.add_x3_convt(w, f, d)
                self.WX[f'upv{i}'] = (f, d if train else f)
                continue
            f = self._buf(f'upv{i}:f', w.numel(), dev)
            d = self._buf(f'upv{i}:d', w.numel(), dev) if train else None
            jobs.add_convt(w, f, d)
            W[f'upv{i}'] = (f, d)
        conv('conv10', 'conv10.weight', cout_pad=self.cout_pad)
        self.W = W
        return jobs

    def _wino(self, co, ci, taps=9):
        """(forward, backward-data) through the Winograd F(2x2,3x3) kernel?  Same rule as the UNet engine (self.policy)."""
        return self._pol.use_wino(co, ci, taps)

    def _cf(self, name, src, src2, bias, out, cout, act, residual=None):
        """3x3 forward: bf16x3 / Winograd kernel where packed for it, else the direct implicit GEMM."""
        x3 = self.WX.get(name, (None, None))[0]
        if x3 is not None:
            return ops.conv_x3_fwd(src, src2, x3, bias, out, cout, act, residual=residual)
        u = self.WU.get(name, (None, None))[0]
        if u is not None:
            return ops.conv_wino_fwd(src, src2, u, bias, out, cout, act, residual=residual)
        return ops.conv_fwd(src, src2, self.W[name][0], bias, out, cout, 9, act, residual=residual)

    def _dg(self, name, gsrc, dx1, **kw):
        x3 = self.WX.get(name, (None, None))[1]
        if x3 is not None:
            return ops.conv_x3_bwd_data(gsrc, x3, dx1, **kw)
        u = self.WU.get(name, (None, None))[1]
        if u is not None:
            return ops.conv_wino_bwd_data(gsrc, u, dx1, **kw)
        return ops.conv_bwd_data(gsrc, self.W[name][1], dx1, **kw)

    # ---------------------------------------------------------------- forward
    def forward(self, x, train, reflect_pad=0, add_residual=True):
        """``reflect_pad`` > 0 (eval loop, trainer_SID.py:221-226): the network runs on the frame reflect-padded by that many pixels on
        every side -- the padding happens inside the NCHW -> NHWC layout pass, the result has the PADDED size (the caller crops).
        ``add_residual=False``: a `res` network returns f(x) without `+ x` (the caller adds the un-padded input after cropping:
        (f(pad x) + pad x)[crop] = f(pad x)[crop] + x; pnnp_eval_post_f32)."""
        if not x.is_cuda:
            raise PnnpError('ResUnet.forward: input must be a CUDA tensor (pnnp_amd has no CPU path)')
        x = x.contiguous().float()
        B, Cin, H, Wd = x.shape
        if reflect_pad:
            if train or (self.m.res and add_residual):
                raise PnnpError('reflect_pad is an eval-mode option; a `res` network needs add_residual=False (the caller adds the input after cropping)')
            H, Wd = H + 2 * reflect_pad, Wd + 2 * reflect_pad
        if Cin != self.cin or H % 16 or Wd % 16:
            raise PnnpError(f'input must be [B,{self.cin},H,W] with H,W multiples of 16, got {tuple(x.shape)}')
        dev = x.device
        self.params.ensure(dev)
        # packed weights are re-used while no parameter changed (eval loops); in-place torch updates bump
        # tensor._version, the fused Adam kernel goes through mark_dirty()
        self._pol = self.effective_policy(H, Wd, max(self.ch[0], self.cin_pad, self.cout_pad))
        self._packs_ready(train, dev)
        gen = self._begin_forward((B, H, Wd, dev), train)
        bufs = self.bufs.setdefault((B, H, Wd, dev), _Bufs())
        P = dict(self.m.named_parameters())
        ch, W = self.ch, self.W
        g = lambda n, s: bufs.get(n, s, dev)
        hs = [H >> i for i in range(5)]; ws = [Wd >> i for i in range(5)]
        a = {}
        a['x8'] = ops.nchw_to_nhwc(x, g('x8', (B, H, Wd, self.cin_pad)), self.cin_pad, reflect_pad=reflect_pad)
        # fp16x2 family (csrc/h2.h): amax slots of the activations, keyed by the layer that wrote the tensor; sign bits of the ReLU outputs
        # that backward-data will need as masks
        h2_on = bool(self.WH) or bool(self.WM)       # amax-slot upkeep whenever ANY layer runs on an fp16x2 kernel
        if h2_on:
            bufs.slots('f', dev).zero_()
        sl = lambda n: bufs.slot('f', n, dev)
        src_name = {}

        def produced(t, name, fused):
            src_name[id(t)] = name
            if h2_on and not fused:
                ops.amax(t, sl(name))
            return t

        def fslot(t, name):                        # the amax slot of an activation an fp16x2 kernel is about to split (filled here if nobody did)
            if id(t) not in src_name:
                produced(t, name, fused=False)
            return sl(src_name[id(t)])

        def cf(name, src, src2, bias, out, cout, act, residual=None):
            hp = self.WH.get(name, (None, None))[0]
            if hp is None:
                return produced(self._cf(name, src, src2, bias, out, cout, act, residual=residual), name, fused=False)
            bits = None
            if train and act != 0 and residual is None:
                bits = a['bits:' + name] = bufs.bits(name, B, out.shape[1], out.shape[2], cout, dev)
            if id(src) not in src_name:                                # (the zero-padded network input: a kernel of its own fills its slot)
                produced(src, 'in:' + name, fused=False)
            ops.conv_h2_fwd(src, src2, hp, self.WS[name], bias, out, cout, act, sl(src_name[id(src)]),
                            sl(src_name[id(src2)]) if src2 is not None else None, amax_y=sl(name), bits_y=bits, residual=residual)
            return produced(out, name, fused=True)

        if self._pol.use_thin_first(self.cin, ch[0], H, Wd, a['x8'].shape[3]):
            a['t0'] = produced(ops.first_fwd(a['x8'], P['conv_in.weight'], P['conv_in.bias'], g('t0', (B, H, Wd, ch[0])), RELU,
                                             amax_y=sl('conv_in') if h2_on else None), 'conv_in', fused=True)
        else:
            a['t0'] = cf('conv_in', a['x8'], None, P['conv_in.bias'], g('t0', (B, H, Wd, ch[0])), ch[0], RELU)
        xin = a['t0']
        for l in range(1, 6):
            lv = l - 1
            shp = (B, hs[lv], ws[lv], ch[lv])
            a[f't{l}'] = cf(f'b{l}_0', xin, None, None, g(f't{l}', shp), ch[lv], RELU)
            a[f'c{l}'] = cf(f'b{l}_1', a[f't{l}'], None, None, g(f'c{l}', shp), ch[lv], 0, residual=xin)
            if l < 5:
                if f'pool{l}' in self.WM:
                    a[f'd{l}'] = produced(ops.conv_s2_h2_fwd(a[f'c{l}'], fslot(a[f'c{l}'], f'c{l}'), self.WM[f'pool{l}'][0], self.WS[f'pool{l}'], P[f'pool{l}.conv.bias'],
                                                             g(f'd{l}', (B, hs[l], ws[l], ch[l])), ch[l], 0, amax_y=sl(f'pool{l}')), f'pool{l}', fused=True)
                elif f'pool{l}' in self.WX:
                    a[f'd{l}'] = produced(ops.conv_s2_x3_fwd(a[f'c{l}'], self.WX[f'pool{l}'][0], P[f'pool{l}.conv.bias'],
                                                             g(f'd{l}', (B, hs[l], ws[l], ch[l])), ch[l], amax_y=sl(f'pool{l}') if h2_on else None), f'pool{l}', fused=True)
                else:
                    a[f'd{l}'] = produced(ops.conv_s2_fwd(a[f'c{l}'], W[f'pool{l}'][0], P[f'pool{l}.conv.bias'],
                                                          g(f'd{l}', (B, hs[l], ws[l], ch[l])), ch[l]), f'pool{l}', fused=False)
                xin = a[f'd{l}']
        cur = a['c5']
        for i in range(6, 10):
            lv = 9 - i
            shp = (B, hs[lv], ws[lv], ch[lv])
            if f'upv{i}' in self.WM:
                u = produced(ops.convt_h2_fwd(cur, fslot(cur, f'in_upv{i}'), self.WM[f'upv{i}'][0], self.WS[f'upv{i}'], P[f'upv{i}.bias'], g(f'u{i}', shp), ch[lv],
                                              amax_y=sl(f'upv{i}')), f'upv{i}', fused=True)
            elif f'upv{i}' in self.WX:
                u = produced(ops.convt_x3_fwd(cur, self.WX[f'upv{i}'][0], P[f'upv{i}.bias'], g(f'u{i}', shp), ch[lv],
                                              amax_y=sl(f'upv{i}') if h2_on else None), f'upv{i}', fused=True)
            else:
                u = produced(ops.convt_fwd(cur, W[f'upv{i}'][0], P[f'upv{i}.bias'], g(f'u{i}', shp), ch[lv]), f'upv{i}', fused=False)
            skip = a[f'c{lv + 1}']
            a[f'u{i}'] = u
            a[f't{i}'] = cf(f'b{i}_0', u, skip, None, g(f't{i}', shp), ch[lv], RELU)
            if f'sc{i}' in self.WM:
                sc = ops.conv1x1_h2_fwd(u, fslot(u, f'upv{i}'), skip, fslot(skip, f'c{lv + 1}'), self.WM[f'sc{i}'][0], self.WS[f'sc{i}'], None, g(f'sc{i}', shp), ch[lv], 0)
            elif self.WX.get(f'sc{i}', (None, None))[0] is not None:
                sc = ops.conv1x1_x3_fwd(u, skip, self.WX[f'sc{i}'][0], None, g(f'sc{i}', shp), ch[lv], 0)
            else:
                sc = ops.conv_fwd(u, skip, W[f'sc{i}'][0], None, g(f'sc{i}', shp), ch[lv], 1, 0)
            a[f'c{i}'] = cf(f'b{i}_1', a[f't{i}'], None, None, g(f'c{i}', shp), ch[lv], 0, residual=sc)
            cur = a[f'c{i}']
        out = torch.empty((B, self.cout, H, Wd), dtype=torch.float32, device=dev)
        if self._pol.use_thin_head(ch[0], self.cout, B * H * Wd):
            ops.head_fwd(a['c9'], P['conv10.weight'], P['conv10.bias'], out, residual=x if (self.m.res and add_residual) else None)
        else:
            o = ops.conv_fwd(a['c9'], None, W['conv10'][0], P['conv10.bias'], g('o', (B, H, Wd, self.cout)), self.cout, 1, 0)
            ops.nhwc_to_nchw(o, out, residual=x if (self.m.res and add_residual) else None)
        if train:
            a['_pol'] = self._pol
            a['_src_name'] = src_name
            self.saved = (a, (B, H, Wd, dev), gen)
        return out

    # ---------------------------------------------------------------- backward
    def backward(self, g_out8, need_dx=False, accumulate=False, on_ready=None):
        a, (B, H, Wd, dev), _ = self.saved
        self._pol = a['_pol']            # the kernel families this forward ran on (effective_policy)
        bufs = self.bufs[(B, H, Wd, dev)]
        ch, W = self.ch, self.W
        gb = lambda n, like: bufs.get('g_' + n, like.shape, dev)
        P = dict(self.m.named_parameters())
        G = lambda name: self.params.grad_view(name, P[name].shape)
        acc = 1 if accumulate else 0
        wsf = bufs.get('wgrad_ws', (self._ws_floats(B, H, Wd),), dev)

        def done(pname):
            if on_ready is not None:
                on_ready(self.params.slices[pname][0])

        # fp16x2 family: amax slots of the gradients (zeroed per backward), the activations' slots are the forward's
        h2_on = bool(self.WH) or bool(self.WM)       # amax-slot upkeep whenever ANY layer runs on an fp16x2 kernel
        if h2_on:
            bufs.slots('b', dev).zero_()
        src_name = a.get('_src_name', {})
        slf = lambda t: bufs.slot('f', src_name[id(t)], dev)
        gname = {}
        gslot = lambda t: bufs.slot('b', gname[id(t)], dev)
        bslot = lambda n: bufs.slot('b', n, dev) if h2_on else None

        def gproduced(t, name, fused):
            gname[id(t)] = name
            if h2_on and not fused:
                ops.amax(t, bufs.slot('b', name, dev))
            return t

        def bneed(t, name):                        # the amax slot of a gradient an fp16x2 kernel is about to split (filled here if it is stale / missing)
            if id(t) not in gname:
                gproduced(t, name, fused=False)
            return gslot(t)

        def dg(name, gsrc, dx1, **kw):
            hp = self.WH.get(name, (None, None))[1]
            dx2 = kw.get('dx2')
            if hp is None:
                self._dg(name, gsrc, dx1, **kw)
                gproduced(dx1, 'd1:' + name, fused=False)
                if dx2 is not None:
                    gproduced(dx2, 'd2:' + name, fused=False)
                return
            for k_mask, k_bits in (('mask1', 'bits1'), ('mask2', 'bits2')):      # act' masks as the forward kernels' sign bits
                m = kw.get(k_mask)
                if m is not None and ('bits:' + src_name.get(id(m), '?')) in a:
                    kw[k_bits] = a['bits:' + src_name[id(m)]]
                    kw[k_mask] = None
            gname[id(dx1)] = 'd1:' + name
            if dx2 is not None:
                gname[id(dx2)] = 'd2:' + name
            ops.conv_h2_bwd_data(gsrc, gslot(gsrc), hp, self.WS[name], dx1, amax_dx1=gslot(dx1), amax_dx2=gslot(dx2) if dx2 is not None else None, **kw)

        def wgrad(pname, gpre, cout, x1, c1, x2=None, taps=9, bias=None):
            c2 = x2.shape[3] if x2 is not None else 0
            if (taps == 9 and h2_on and self._pol.h2_wgrad and id(gpre) in gname and id(x1) in src_name and (x2 is None or id(x2) in src_name) and
                    self._pol.use_x3_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, batch=gpre.shape[0],
                                           cs=max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0))):
                ops.conv_h2_bwd_weight(gpre, gslot(gpre), cout, x1, slf(x1), c1, x2, slf(x2) if x2 is not None else None,
                                       G(pname), G(bias) if bias else None, wsf, accumulate=acc)
                return
            if taps == 9 and self._pol.use_x3_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, batch=gpre.shape[0],
                                                    cs=max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0)):
                ops.conv_x3_bwd_weight(gpre, cout, x1, c1, x2, G(pname), G(bias) if bias else None, wsf, accumulate=acc)
            elif taps == 9 and self._pol.use_wino_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, gpre.shape[3], x1.shape[3]):
                ops.conv_wino_bwd_weight(gpre, cout, x1, c1, x2, G(pname), G(bias) if bias else None, wsf, accumulate=acc)
            elif (taps == 1 and h2_on and self._pol.h2_pointwise and self._pol.x3 and not self._pol.use_x3g_wgrad(ops.X3G_PW, cout, c1 + c2, gpre.shape[0], gpre.shape[1],
                                                                                                                    gpre.shape[2], gpre.shape[1], gpre.shape[2], max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0))
                  and ops.h2g_wgrad_supported(ops.X3G_PW, cout, c1 + c2) and id(gpre) in gname and id(x1) in src_name and (x2 is None or id(x2) in src_name)
                  and ops.x3_wgrad_fits(gpre.shape[0], gpre.shape[1], gpre.shape[2], max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0))
                  and wsf.numel() >= ops.h2g_wgrad_workspace_floats(ops.X3G_PW, gpre.shape[0], gpre.shape[1], gpre.shape[2], cout, c1 + c2)):
                # a shape only the fp16x2 kernel has a tile for (sc9: 32 x 64, round 6; it ran on the fp32-MFMA kernel)
                ops.conv1x1_h2_bwd_weight(gpre, gslot(gpre), cout, x1, slf(x1), c1, x2, slf(x2) if x2 is not None else None, G(pname), G(bias) if bias else None,
                                          wsf, accumulate=acc)
            elif taps == 1 and self._pol.use_x3g_wgrad(ops.X3G_PW, cout, c1 + c2, gpre.shape[0], gpre.shape[1], gpre.shape[2], gpre.shape[1], gpre.shape[2],
                                                       max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0)):
                if h2_on and self._pol.h2_pointwise and id(gpre) in gname and id(x1) in src_name and (x2 is None or id(x2) in src_name):
                    ops.conv1x1_h2_bwd_weight(gpre, gslot(gpre), cout, x1, slf(x1), c1, x2, slf(x2) if x2 is not None else None, G(pname), G(bias) if bias else None,
                                              wsf, accumulate=acc)
                else:
                    ops.conv1x1_x3_bwd_weight(gpre, cout, x1, c1, x2, G(pname), G(bias) if bias else None, wsf, accumulate=acc)
            else:
                ops.conv_bwd_weight(gpre, cout, x1, c1, x2, G(pname), G(bias) if bias else None, taps, wsf, accumulate=acc)

        # head
        g = gb('c9', a['c9'])
        if self._pol.use_thin_head(ch[0], self.cout, B * H * Wd):
            ops.head_bwd(g_out8, a['c9'], P['conv10.weight'], g, G('conv10.weight'), G('conv10.bias'), wsf, mode=0, accumulate=acc, amax_gx=bslot('head'))
            gproduced(g, 'head', fused=True)
        else:
            wgrad('conv10.weight', g_out8, self.cout, a['c9'], ch[0], taps=1, bias='conv10.bias')
            ops.conv_bwd_data(g_out8, W['conv10'][1], g, taps=1)
            gproduced(g, 'head', fused=False)
        done('conv10.weight')
        for i in range(9, 5, -1):                    # decoder blocks, top-down
            lv = 9 - i
            u, skip, t = a[f'u{i}'], a[f'c{lv + 1}'], a[f't{i}']
            wgrad(f'conv{i}.short_cut.0.conv.conv.weight', g, ch[lv], u, ch[lv], x2=skip, taps=1)
            wgrad(f'conv{i}.block.1.conv.conv.weight', g, ch[lv], t, ch[lv])
            g_t = gb(f't{i}', t)
            dg(f'b{i}_1', g, g_t, mask1=t, mode1=RELU)
            wgrad(f'conv{i}.block.0.conv.conv.weight', g_t, ch[lv], u, ch[lv], x2=skip)
            done(f'conv{i}.block.0.conv.conv.weight')
            g_u, g_skip = gb(f'u{i}', u), gb(f'c{lv + 1}', skip)
            dg(f'b{i}_0', g_t, g_u, dx2=g_skip)
            # (the shortcut's gradient is ACCUMULATED into g_u and g_skip next: their slots are stale from here on -- no fp16x2 kernel reads
            #  them before ConvTranspose2d's backward / the stride-2 backward rewrite or finish them)
            gname.pop(id(g_u), None); gname.pop(id(g_skip), None)
            if f'sc{i}' in self.WM:
                # (g_u now holds block gradient + shortcut gradient: the kernel reports max |sum| -- its slot is valid again; g_skip's stays stale)
                ops.conv1x1_h2_bwd_data(g, bneed(g, f'gc{i}'), self.WM[f'sc{i}'][1], self.WS[f'sc{i}'], g_u, accum1=1, amax_dx1=bslot(f'gu{i}'), dx2=g_skip, accum2=1)
                gname[id(g_u)] = f'gu{i}'
            elif self.WX.get(f'sc{i}', (None, None))[0] is not None:
                ops.conv1x1_x3_bwd_data(g, self.WX[f'sc{i}'][1], g_u, accum1=1, dx2=g_skip, accum2=1)
            else:
                ops.conv_bwd_data(g, W[f'sc{i}'][1], g_u, accum1=1, dx2=g_skip, accum2=1, taps=1)
            below = a['c5'] if i == 6 else a[f'c{i - 1}']
            ct_wgrad = ops.convt_x3_bwd_weight if self._pol.use_x3g_wgrad(ops.X3G_CT, below.shape[3], g_u.shape[3], B, below.shape[1], below.shape[2],
                                                                           g_u.shape[1], g_u.shape[2], max(below.shape[3], g_u.shape[3])) else ops.convt_bwd_weight
            if ct_wgrad is ops.convt_x3_bwd_weight and h2_on and self._pol.h2_pointwise and id(below) in src_name and id(g_u) in gname:
                ops.convt_h2_bwd_weight(below, slf(below), g_u, gslot(g_u), G(f'upv{i}.weight'), wsf, accumulate=acc, dbias=G(f'upv{i}.bias'))
            else:
                ct_wgrad(below, g_u, G(f'upv{i}.weight'), wsf, accumulate=acc, dbias=G(f'upv{i}.bias'))
            done(f'upv{i}.weight')
            g = gb('c5' if i == 6 else f'c{i - 1}', below)
            if f'upv{i}' in self.WM:
                ops.convt_h2_bwd_data(g_u, bneed(g_u, f'gu{i}'), self.WM[f'upv{i}'][1], self.WS[f'upv{i}'], g, amax_dx=bslot(f'upv{i}'))
                gproduced(g, f'upv{i}', fused=True)
            elif f'upv{i}' in self.WX:
                ops.convt_x3_bwd_data(g_u, self.WX[f'upv{i}'][1], g, amax_dx=bslot(f'upv{i}'))
                gproduced(g, f'upv{i}', fused=True)
            else:
                ops.convt_bwd_data(g_u, W[f'upv{i}'][1], g)
                gproduced(g, f'upv{i}', fused=False)
        for l in range(5, 0, -1):                    # encoder blocks, bottom-up; g = dL/d c_l
            lv = l - 1
            t = a[f't{l}']
            xin = a['t0'] if l == 1 else a[f'd{l - 1}']
            wgrad(f'conv{l}.block.1.conv.conv.weight', g, ch[lv], t, ch[lv])
            g_t = gb(f't{l}', t)
            dg(f'b{l}_1', g, g_t, mask1=t, mode1=RELU)
            wgrad(f'conv{l}.block.0.conv.conv.weight', g_t, ch[lv], xin, ch[lv])
            done(f'conv{l}.block.0.conv.conv.weight')
            g_x = gb('t0' if l == 1 else f'd{l - 1}', xin)
            # identity shortcut: d/d(xin) = dgrad(block) + g ; xin = t0 is a ReLU output (mask), d_l is not
            ud = self.WU.get(f'b{l}_0', (None, None))[1]
            x3d = self.WX.get(f'b{l}_0', (None, None))[1]
            h2d = self.WH.get(f'b{l}_0', (None, None))[1]
            if h2d is not None:
                ops.conv_h2_bwd_data_res(g_t, gslot(g_t), h2d, self.WS[f'b{l}_0'], g_x, addsrc=g, mask=xin if l == 1 else None, mode=RELU, amax_dx=bslot(f'gx{l}'))
                gname[id(g_x)] = f'gx{l}'                          # (the stride-2 layer's fp16x2 backward kernels split it next)
            elif x3d is not None:
                ops.conv_x3_bwd_data_res(g_t, x3d, g_x, addsrc=g, mask=xin if l == 1 else None, mode=RELU)
            elif ud is not None:
                ops.conv_wino_bwd_data_res(g_t, ud, g_x, addsrc=g, mask=xin if l == 1 else None, mode=RELU)
            else:
                ops.conv_bwd_data_res(g_t, W[f'b{l}_0'][1], g_x, addsrc=g, mask=xin if l == 1 else None, mode=RELU)
            if l > 1:
                c_prev = a[f'c{l - 1}']
                s2_wgrad = ops.conv_s2_x3_bwd_weight if self._pol.use_x3g_wgrad(ops.X3G_S2, g_x.shape[3], c_prev.shape[3], B, g_x.shape[1], g_x.shape[2],
                                                                                c_prev.shape[1], c_prev.shape[2], max(g_x.shape[3], c_prev.shape[3])) else ops.conv_s2_bwd_weight
                h2g_ok = (h2_on and self._pol.h2_pointwise and self._pol.x3 and id(c_prev) in src_name and ops.h2g_wgrad_supported(ops.X3G_S2, g_x.shape[3], c_prev.shape[3])
                          and ops.x3_wgrad_fits(B, g_x.shape[1], g_x.shape[2], max(g_x.shape[3], c_prev.shape[3])) and ops.x3_wgrad_fits(B, c_prev.shape[1], c_prev.shape[2], max(g_x.shape[3], c_prev.shape[3])))
                if h2g_ok:                                           # (the fp16x2 kernel also has a tile for Cout = 64: pool1, which bf16x3 left to the fp32-MFMA kernel)
                    ops.conv_s2_h2_bwd_weight(g_x, bneed(g_x, f'gx{l}'), c_prev, slf(c_prev), G(f'pool{l - 1}.conv.weight'), G(f'pool{l - 1}.conv.bias'), wsf, accumulate=acc)
                else:
                    s2_wgrad(g_x, c_prev, G(f'pool{l - 1}.conv.weight'), G(f'pool{l - 1}.conv.bias'), wsf, accumulate=acc)
                done(f'pool{l - 1}.conv.weight')
                g = gb(f'c{l - 1}', c_prev)                          # already holds the skip gradient
                if f'pool{l - 1}' in self.WM:
                    ops.conv_s2_h2_bwd_data(g_x, bneed(g_x, f'gx{l}'), self.WM[f'pool{l - 1}'][1], self.WS[f'pool{l - 1}'], g, accum=1, amax_dx=bslot(f'pool{l - 1}'))
                    gproduced(g, f'pool{l - 1}', fused=True)
                elif f'pool{l - 1}' in self.WX:
                    ops.conv_s2_x3_bwd_data(g_x, self.WX[f'pool{l - 1}'][1], g, accum=1, amax_dx=bslot(f'pool{l - 1}'))
                    gproduced(g, f'pool{l - 1}', fused=True)           # (the sums it stored: skip gradient + this layer's)
                else:
                    ops.conv_s2_bwd_data(g_x, W[f'pool{l - 1}'][1], g, accum=1)
                    gproduced(g, f'pool{l - 1}', fused=False)
            else:
                if self._pol.use_thin_first(self.cin, ch[0], H, Wd, a['x8'].shape[3]):
                    ops.first_bwd_weight(g_x, ch[0], a['x8'], self.cin, G('conv_in.weight'), G('conv_in.bias'), wsf, accumulate=acc)
                else:
                    wgrad('conv_in.weight', g_x, ch[0], a['x8'], self.cin, bias='conv_in.bias')
                done('conv_in.weight')
        if need_dx:
            raise PnnpError('gradient w.r.t. the network input is not implemented on the HIP path')
        return None

    def _ws_floats(self, B, H, W):
        ch = self.ch
        need = 1024 * max(ch)
        for lv in range(5):
            h, w, c = H >> lv, W >> lv, ch[lv]
            need = max(need, ops.wino_wgrad_workspace_floats(B, h, w, c, c), ops.wino_wgrad_workspace_floats(B, h, w, c, 2 * c),
                       ops.x3_wgrad_workspace_floats(B, h, w, c, c), ops.x3_wgrad_workspace_floats(B, h, w, c, 2 * c))
            need = max(need, ops.wgrad_workspace_floats(B, h, w, c, c, 9), ops.wgrad_workspace_floats(B, h, w, c, 2 * c, 9),
                       ops.wgrad_workspace_floats(B, h, w, c, 2 * c, 1), ops.wgrad_workspace_floats(B, h, w, c, self.cin, 9))
            need = max(need, ops.x3g_wgrad_workspace_floats(ops.X3G_PW, B, h, w, c, 2 * c))
            if lv < 4:
                need = max(need, ops.wgrad_workspace_floats(B, h >> 1, w >> 1, ch[lv + 1], c, 4),
                           ops.wgrad_workspace_floats(B, h >> 1, w >> 1, ch[lv + 1], c, 18),
                           ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, h >> 1, w >> 1, ch[lv + 1], c),
                           ops.x3g_wgrad_workspace_floats(ops.X3G_S2, B, h >> 1, w >> 1, ch[lv + 1], c),
                           ops.h2g_wgrad_workspace_floats(ops.X3G_S2, B, h >> 1, w >> 1, ch[lv + 1], c))
        need = max(need, ops.head_bwd_workspace_floats(ch[0]), ops.first_wgrad_workspace_floats(ch[0]))
        return max(need, ops.wgrad_workspace_floats(B, H, W, self.cout, ch[0], 1))


class _ResUnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, train, *params):
        ctx.engine = engine
        ctx.x_needs = x.requires_grad
        out = engine.forward(x, train)
        ctx.gen = engine.gen
        return out

    @staticmethod
    def backward(ctx, grad_out):
        e = ctx.engine
        e.check_saved(ctx.gen)
        B, _, H, W = grad_out.shape
        bufs = e.bufs[(B, H, W, grad_out.device)]
        g8 = ops.nchw_to_nhwc(grad_out.contiguous().float(), bufs.get('g_out8', (B, H, W, e.cout_pad), grad_out.device), e.cout_pad)
        e.backward(g8, need_dx=ctx.x_needs)
        grads = [e.params.grad_view(n, p.shape).clone() if p.requires_grad else None for n, p in e.m.named_parameters()]
        return (None, None, None) + tuple(grads)


class ResUnet(nn.Module):
    """Drop-in for archs/ResUnet.py:3-88 (``args`` keys: nframes, res, nf, in_nc, out_nc)."""

    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.nframes = args['nframes']
        self.cf = args['nframes'] // 2
        self.res = args['res']
        nf = self.nf = args['nf']
        self.in_nc = args['in_nc']
        self.out_nc = args['out_nc']
        c = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
        self.conv_in = nn.Conv2d(self.in_nc * self.nframes, nf, kernel_size=3, stride=1, padding=1)
        for l in range(1, 6):
            setattr(self, f'conv{l}', _ResBlockHolder(c[l - 1], c[l - 1]))
            if l < 5:
                setattr(self, f'pool{l}', _DownHolder(c[l - 1], c[l]))
        for i in range(6, 10):
            lv = 9 - i
            setattr(self, f'upv{i}', nn.ConvTranspose2d(c[lv + 1], c[lv], 2, stride=2))
            setattr(self, f'conv{i}', _ResBlockHolder(c[lv + 1], c[lv]))
        self.conv10 = nn.Conv2d(nf, self.out_nc, kernel_size=1, stride=1)
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            object.__setattr__(self, '_engine', ResUnetEngine(self))
        return self._engine

    def forward(self, x, noise_map=None):
        params = list(self.parameters())
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _ResUnetFn.apply(x, self.engine, train, *params)
