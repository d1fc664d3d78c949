"""UNetSeeInDark on hand-written HIP kernels (reference: archs/Unet.py:4-99).

The module keeps the reference's constructor (``args`` dict), attribute names and
``state_dict`` layout (46 tensors, ``conv{1..9}_{1,2}``, ``upv{6..9}``, ``conv10_1``) so
released checkpoints load unchanged and ``initialize_weights`` / ``load_weights`` work on
it.  The nn.Conv2d / nn.ConvTranspose2d children only own parameters: ``forward`` runs the
whole network through libpnnp_hip.so (NHWC fp32 activations, fp32 MFMA) and backward is a
hand-sequenced pass over the same kernels, exposed to autograd as one Function so
``loss.backward(); optimizer.step()`` of the reference trainer works as is.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from .._lib import PnnpError

LRELU, RELU = 1, 2


class ConvPolicy:
    """Which kernel family a 3x3 layer runs on.
    ``x3``: forward / backward-data on the bf16 matrix cores with float32 operands split into three bf16 pieces
    (csrc/conv_x3.hip: float32-accurate, 6/16 of the fp32-MFMA time) wherever the layer qualifies (reduction % 8 == 0,
    channels written % 32 == 0) -- the default;
    ``wino``: Winograd F(2x2,3x3) on the fp32 matrix cores for forward / backward-data where x3 is off and the layer
    qualifies (channels written % 64 == 0, reduction >= ``wino_mink`` channels), ``wino_wgrad``: the Winograd
    backward-weight kernel likewise; everything else (and everything when all are off) uses the direct fp32 implicit-GEMM
    kernels.  ``thin``: the 4-channel ends -- the 1x1 head and the first layer's backward-weight -- on the streaming vector-ALU
    kernels of csrc/thin.hip instead of the channel-padded GEMM kernels.  An engine takes DEFAULT_POLICY at construction; ``engine.set_policy(...)`` switches it (tests compare the
    families against each other at full size)."""

    def __init__(self, wino=True, wino_wgrad=True, wino_mink=32, x3=True, thin=True, pool_fused=True, h2=True):
        self.wino, self.wino_wgrad, self.wino_mink, self.x3, self.thin = bool(wino), bool(wino_wgrad), int(wino_mink), bool(x3), bool(thin)
        self.pool_fused = bool(pool_fused)         # training forward: MaxPool2d(2) in the epilogue of the bf16x3 / fp16x2 conv in front of it
        # ``h2``: the 3x3 layers that qualify for x3 run on the fp16 matrix cores instead, float32 operands split into TWO scaled fp16 pieces
        # (csrc/conv_h2s.hip, csrc/h2.h: half the matrix instructions of bf16x3; amax slots travel beside the tensors, the act' masks of the
        # backward pass are the forward kernels' sign bits).  The default since round 5: every float64 yardstick and reference-golden test of the
        # bf16x3 family passes at the same bars (tests/test_gpu_h2.py, tests/test_gpu_fullsize.py); ``set_policy(h2=False)`` = the bf16x3 family.
        self.h2 = bool(h2)
        self.h2_wgrad = bool(h2) and os.environ.get('PNNP_H2_WGRAD', '1') != '0'      # (host-side A/B switch: backward-weight stays on bf16x3 with 0)
        self.h2_pointwise = bool(h2) and os.environ.get('PNNP_H2_POINTWISE', '1') != '0'      # (A/B switch: ConvTranspose2d stays on bf16x3 with 0)
        self.head_fused = bool(h2) and os.environ.get('PNNP_HEAD_FUSED', '1') != '0'          # (A/B switch) conv10_1 inside conv9_2's epilogue (round 6)
        self.splitk = bool(h2) and os.environ.get('PNNP_SPLITK', '1') != '0'                  # (A/B switch) split-K forward launches for small grids (round 6)
        self.convt_bits = bool(h2) and os.environ.get('PNNP_CONVT_BITS', '1') != '0'          # (A/B switch) ConvTranspose2d backward-data masks with sign bits (round 6)

    def key(self):
        return (self.wino, self.wino_wgrad, self.wino_mink, self.x3, self.thin, self.pool_fused, self.h2, self.h2_wgrad, self.h2_pointwise, self.head_fused, self.splitk, self.convt_bits)

    def use_thin_head(self, cin, cout, npix):
        return self.thin and ops.head_supported(cin, cout, npix)

    def use_thin_first(self, cin, cout, h, w, x_cs):
        return self.thin and x_cs >= 4 and ops.first_wgrad_supported(cin, cout, h, w)

    def use_x3(self, co, ci, taps=9, c1=None):
        """(forward, backward-data) of a 3x3 Conv2d(ci -> co) on the bf16x3 kernel?  ``c1``: channels of the first of two
        concatenated inputs (its gradient is a separate destination: the split must fall on a 32-column block)."""
        if taps != 9 or not self.x3:
            return False, False
        return (ops.x3_supported(ci, co) and (c1 is None or c1 % 16 == 0),
                ops.x3_supported(co, ci) and (c1 is None or c1 % 32 == 0))

    def use_h2(self, co, ci, taps=9, c1=None):
        """(forward, backward-data) of a 3x3 Conv2d(ci -> co) on the fp16x2 kernel?  Same shape rules as use_x3 (+ at most 1024 channels written)."""
        if taps != 9 or not (self.h2 and self.x3):
            return False, False
        return (ops.h2_supported(ci, co) and (c1 is None or c1 % 16 == 0),
                ops.h2_supported(co, ci) and (c1 is None or c1 % 32 == 0))

    def use_wino(self, co, ci, taps=9):
        """(forward, backward-data) of a Conv2d(ci -> co, taps) on the Winograd kernel?"""
        if taps != 9 or not self.wino:
            return False, False
        return (ops.wino_supported(ci, co) and ci >= self.wino_mink, ops.wino_supported(co, ci) and co >= self.wino_mink)

    def use_x3_pointwise(self, K, N):
        """a one-tap-per-segment layer (ConvTranspose2d, 1x1, stride-2 3x3) on the pointwise bf16x3 GEMM kernel (csrc/gemm_x3.hip)?"""
        return self.x3 and ops.gemm_x3_supported(K, N)

    def use_x3_wgrad(self, h, w, cout, c1, c2, batch=None, cs=None):
        """backward-weight of a 3x3 layer on the bf16x3 kernel (csrc/wgrad_x3.hip)?  ``batch`` / ``cs`` (largest channel stride of
        the tensors involved): the kernel addresses a whole [B][H][W][cs] map with 32-bit byte offsets; past that the layer falls
        back to the Winograd / direct fp32 kernels instead of failing inside backward."""
        if not (self.x3 and ops.x3_wgrad_supported(h, w, cout, c1, c2)):
            return False
        return batch is None or ops.x3_wgrad_fits(batch, h, w, cs if cs is not None else max(cout, c1, c2))

    def use_x3g_wgrad(self, kind, M, N, batch, uh, uw, sh, sw, cs):
        """backward-weight of a ConvTranspose2d / stride-2 3x3 / 1x1 layer on the bf16x3 kernel of csrc/wgrad_x3g.hip?  (M, N) must have a
        tile configuration and both whole maps must fit 32-bit byte offsets; otherwise the fp32-MFMA kernel of csrc/wgrad.hip takes it."""
        if not (self.x3 and ops.x3g_wgrad_supported(kind, M, N)):
            return False
        return ops.x3_wgrad_fits(batch, uh, uw, cs) and ops.x3_wgrad_fits(batch, sh, sw, cs)

    def use_wino_wgrad(self, h, w, cout, c1, c2, g_cs, x_cs):
        return self.wino and self.wino_wgrad and g_cs == cout and x_cs == c1 and ops.wino_wgrad_supported(h, w, cout, c1, c2)


DEFAULT_POLICY = ConvPolicy(wino=os.environ.get('PNNP_WINO', '1') != '0', x3=os.environ.get('PNNP_X3', '1') != '0', h2=os.environ.get('PNNP_H2', '1') != '0')      # host-side defaults only; the library reads no environment


class _EngineBase:
    """What the two network engines share: the kernel-family policy and the bookkeeping that ties an autograd backward to
    the training forward whose activations it needs."""

    def _init_base(self):
        self.policy = ConvPolicy(*DEFAULT_POLICY.key()[:7])
        self._h2_sub = {}            # explicit set_policy(h2_wgrad= / h2_pointwise=) overrides: they survive later set_policy calls
        self._pol = self.policy      # the effective policy of the current forward (effective_policy)
        self.saved = None
        self.gen = 0                 # bumped by every forward that (re)writes an activation buffer set
        self._pack_key = None
        self._jobs_key = None
        self._dirty_epoch = 0

    def _pack_state_key(self, train, dev):
        return (train, dev, self._dirty_epoch, self._pol.key()) + tuple(p._version for p in self.m.parameters())

    def _packs_ready(self, train, dev):
        """Start of a forward: (re-)pack the weights if the packs are not the ones this forward needs (mode, policy, parameter versions)."""
        key = self._pack_state_key(train, dev)
        if key != self._pack_key:
            self.pack_weights(train)
            self._pack_key = key

    def set_policy(self, policy=None, **kw):
        """``set_policy(x3=False)`` etc.: fields not named keep their current value."""
        if policy is None:
            cur = dict(wino=self.policy.wino, wino_wgrad=self.policy.wino_wgrad, wino_mink=self.policy.wino_mink, x3=self.policy.x3,
                       thin=self.policy.thin, pool_fused=self.policy.pool_fused, h2=self.policy.h2)
            for k in ('h2_wgrad', 'h2_pointwise', 'head_fused', 'splitk', 'convt_bits'):   # sub-switches of h2 (host-side A/B): not constructor arguments, kept as overrides
                if k in kw:
                    self._h2_sub[k] = bool(kw.pop(k))
            cur.update(kw)
            policy = ConvPolicy(**cur)                             # (defaults of the sub-switches: environment, as at construction)
            for k, v in self._h2_sub.items():
                setattr(policy, k, v and policy.h2)
        self.policy = self._pol = policy
        self._pack_key = None        # re-pack for the other kernel family
        self._jobs_key = None

    def effective_policy(self, H, W, cs_max):
        """The policy a forward on [.,.,H,W] inputs runs with (also what its backward uses, whatever set_policy does in between).
        Every convolution kernel of the library -- bf16x3 and fp32-MFMA families alike -- addresses ONE image of a map through a
        buffer resource with 32-bit byte offsets, so the largest map of the network ([H][W][cs_max] floats) must stay below 2 GB
        per image (pnnp_x3_image_fits; ~16.7 M pixels at nf = 32).  A larger frame is refused HERE, before anything is packed or
        launched, instead of failing with PNNP_E_UNSUPPORTED somewhere inside the network: tile the frame.  (The batch-wide limit of
        the bf16x3 backward-weight kernel is different: past it the layer falls back to the fp32 kernels, ConvPolicy.use_x3_wgrad.)"""
        if not ops.x3_image_fits(H, W, cs_max):
            raise PnnpError(f'frame {H} x {W} is too large for the HIP convolution kernels: one image of a {cs_max}-channel map must stay '
                            f'below 2 GB ((H + 4) * W * {cs_max} * 4 bytes); run the frame in tiles')
        return self.policy

    def h2_range_report(self, sample=1 << 22):
        """Debug aid for the fp16x2 family's ONE precision caveat (csrc/h2.h): the scale is per TENSOR, so elements below
        2^-18 of a tensor's largest magnitude keep only absolute accuracy (2^-40 of that maximum).  After a training forward + backward this walks the
        tensors the kernels split on the fly -- the saved activations and the gradient buffers of that step -- and returns, per tensor,
        ``dict(name, kind, amax, median, log2_ratio = log2(amax / median |x|), frac_small, l2_small)``: ``median`` over the non-zero elements
        of a strided sample, ``frac_small`` the share of non-zero elements below 2^-18 amax, ``l2_small`` the share of the tensor's sum of squares they
        carry (what an output that depends on them alone would lose).  Plain torch ops: for tools and tests (tools/soak.py), never on the hot path."""
        if self.saved is None:
            raise PnnpError('h2_range_report: no saved training forward')
        a, key, _ = self.saved
        rows = []

        def add(name, kind, t):
            if not torch.is_tensor(t) or t.dtype != torch.float32 or t.dim() != 4:
                return
            v = t.reshape(-1)
            amax = float(v.abs().max())
            step = max(1, v.numel() // sample)
            sv = v[::step].abs()
            nz = sv[sv > 0]
            if amax == 0.0 or nz.numel() == 0 or not torch.isfinite(sv).all():
                rows.append(dict(name=name, kind=kind, amax=amax, median=0.0, log2_ratio=float('nan'), frac_small=0.0, l2_small=0.0))
                return
            med = float(nz.median())
            small = nz < amax * 2.0 ** -18
            rows.append(dict(name=name, kind=kind, amax=amax, median=med, log2_ratio=float(torch.log2(torch.tensor(amax / med))),
                             frac_small=float(small.float().mean()), l2_small=float((nz[small] ** 2).sum() / (nz ** 2).sum())))

        for k, t in a.items():
            if isinstance(k, str) and not k.startswith(('bits:', 'pc', '_')):
                add(k, 'act', t)
        for k, t in self.bufs[key].t.items():
            if isinstance(k, str) and k.startswith('g_') and k not in ('g_out8', 'g_out4'):
                add(k, 'grad', t)
        return rows

    def mark_dirty(self):
        """Parameters were modified behind torch's back (fused Adam on the flat buffer)."""
        self._dirty_epoch += 1

    def _begin_forward(self, key, train):
        """Activation buffers are per input shape and reused: a later forward on the same shape overwrites what an earlier
        training forward saved for its backward.  Every forward gets a generation number; `saved` remembers the one it
        belongs to and is dropped when its buffers are about to be overwritten by a forward that does not replace it."""
        self.gen += 1
        if self.saved is not None and self.saved[1] == key and not train:
            self.saved = None
        return self.gen

    def check_saved(self, gen):
        if self.saved is None or self.saved[2] != gen:
            raise PnnpError('backward: the activations of this forward were overwritten by a later forward on the same input shape '
                            '(the HIP engine keeps ONE saved forward per shape: run backward before the next forward, '
                            'or accumulate gradients step by step)')


class FlatParams:
    """All parameters (and their gradients) of a module as views of two flat fp32 buffers,
    16-byte aligned per tensor: one fused Adam launch and one (bucketed) all-reduce."""

    def __init__(self, module):
        self.module = module
        self.flat = None
        self.grad = None
        self.slices = {}

    def ensure(self, device):
        ps = list(self.module.named_parameters())
        ok = self.flat is not None and self.flat.device == device and all(
            p.data_ptr() == self.flat.data_ptr() + 4 * self.slices[n][0] for n, p in ps)
        if ok:
            return
        off = 0
        self.slices = {}
        for n, p in ps:
            self.slices[n] = (off, p.numel())
            off += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(off, dtype=torch.float32, device=device)
        grad = torch.zeros(off, dtype=torch.float32, device=device)
        for n, p in ps:
            o, k = self.slices[n]
            flat[o:o + k].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[o:o + k].view(p.shape)
            p.grad = None
        self.flat, self.grad = flat, grad

    def grad_view(self, name, shape):
        o, k = self.slices[name]
        return self.grad[o:o + k].view(shape)


class _Bufs:
    """Activation / gradient buffers for one input shape, allocated once and reused."""

    def __init__(self):
        self.t = {}

    def get(self, name, shape, device):
        b = self.t.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.device != device:
            b = torch.empty(shape, dtype=torch.float32, device=device)
            self.t[name] = b
        return b

    def scratch(self, name, n, device):
        """a 1-D float32 scratch buffer of at least n elements (grows, never shrinks)"""
        b = self.t.get(name)
        if b is None or b.numel() < n or b.device != device:
            b = self.t[name] = torch.empty(int(n), dtype=torch.float32, device=device)
        return b

    # ---- fp16x2 family (csrc/h2.h): one 4-byte amax slot per tensor the kernels split, in two tables -- 'f' (activations, zeroed when a
    # forward starts) and 'b' (gradients, zeroed when a backward starts) -- and the sign-bit images of the activations that serve as act' masks
    NSLOTS = 64

    def slots(self, which, device):
        key = 'slots_' + which
        t = self.t.get(key)
        if t is None or t.device != device:
            t = self.t[key] = torch.zeros(self.NSLOTS, dtype=torch.int32, device=device)
            self.t[key + '_idx'] = {}
        return t

    def slot(self, which, name, device):
        t = self.slots(which, device)
        idx = self.t['slots_' + which + '_idx']
        if name not in idx:
            if len(idx) >= self.NSLOTS:
                raise PnnpError('amax slot table full')
            idx[name] = len(idx)
        i = idx[name]
        return t[i:i + 1]

    def bits(self, name, B, H, W, C_, device):
        n = ops.h2_bits_words(B, H, W, C_)
        key = 'bits_' + name
        b = self.t.get(key)
        if b is None or b.numel() != n or b.device != device:
            b = self.t[key] = torch.empty(n, dtype=torch.int32, device=device)
        return b


class UNetEngine(_EngineBase):
    """Forward / backward schedule of UNetSeeInDark over the C-ABI layer kernels."""

    def __init__(self, module):
        self._init_base()
        self.m = module
        self.params = FlatParams(module)
        self.bufs = {}
        self.packed = {}
        nf = module.nf
        self.ch = [nf, nf * 2, nf * 4, nf * 8, nf * 16]
        if nf % 8:
            raise PnnpError('UNetSeeInDark on HIP needs nf % 8 == 0')
        self.cin = module.in_nc * module.nframes
        self.cin_pad = (self.cin + 7) // 8 * 8
        self.cout = module.out_nc
        self.cout_pad = (self.cout + 7) // 8 * 8

    # ------------------------------------------------------------------ weights
    def _conv_names(self):
        return ['conv%d_%d' % (i, j) for i in range(1, 10) for j in (1, 2)] + ['conv10_1']

    def pack_weights(self, need_dgrad):
        """Re-pack every layer's weights into the kernels' K-major order; called once per forward because the optimiser has
        moved them.  The table of pack jobs is built once (per device / mode / parameter storage) and runs in two or three
        launches (ops.PackJobs) instead of ~70."""
        dev = self.params.flat.device
        P = dict(self.m.named_parameters())
        key = (dev, need_dgrad, self._pol.key(), tuple(p.data_ptr() for p in P.values()))
        if self._jobs_key != key:
            self._jobs, self._jobs_key = self._build_pack_jobs(need_dgrad, dev, P), key
        self._jobs.run()

    def _build_pack_jobs(self, need_dgrad, dev, P):
        jobs = ops.PackJobs()
        self._x3, self._wn = {}, {}            # per layer: (forward, backward-data) on the bf16x3 / Winograd kernel
        self._h2, self._wslot = {}, {}         # per layer: (forward, backward-data) fp16x2 packs; the weight tensor's amax slot
        self._h2m = {}                         # ConvTranspose2d layers on the fp16x2 GEMM kernel: (forward, backward-data) kind-6 packs
        def buf(key, n, dt=torch.float32):
            if key not in self.packed:
                self.packed[key] = torch.empty(n, dtype=dt, device=dev)
            return self.packed[key]
        for name in self._conv_names():
            w = P[name + '.weight']
            co, ci, kh, kw = w.shape
            taps = kh * kw
            cip = self.cin_pad if name == 'conv1_1' else ci
            cop = self.cout_pad if name == 'conv10_1' else co
            bwd = need_dgrad and name != 'conv1_1'                 # no gradient w.r.t. the network input
            c1 = ci // 2 if (name.endswith('_1') and name[4] in '6789') else None       # decoder conv{6..9}_1 read cat([up, skip])
            xf, xd = self._pol.use_x3(co, cip, taps, c1)
            xd = xd and bwd
            hf, hd = self._pol.use_h2(co, cip, taps, c1)
            hf, hd = hf and xf, hd and xd
            if hf or hd:                                           # the fp16x2 kernel takes what bf16x3 would have taken
                self._h2[name] = (buf((name, dev, 'h2f'), ops.h2_weight_bytes(cip, co), torch.uint8) if hf else None,
                                  buf((name, dev, 'h2d'), ops.h2_weight_bytes(co, ci), torch.uint8) if hd else None)
                self._wslot[name] = jobs.add_h2(w, self._h2[name][0], self._h2[name][1], cin_pad=(cip + 15) // 16 * 16)
                xf, xd = xf and not hf, xd and not hd
            wf, wd = self._wino(name, co, ci, taps)
            wf, wd = wf and not xf, wd and bwd and not xd
            df, dd = not (xf or wf), bwd and not (xd or wd)        # what is left for the direct fp32 kernels
            if name in self._h2:
                df, dd, wf, wd = df and not hf, dd and not hd, wf and not hf, wd and not hd
            self._x3[name], self._wn[name] = (xf, xd), (wf, wd)
            if df or dd:
                jobs.add_conv(w, buf((name, dev, 'f'), taps * cip * co) if df else None, buf((name, dev, 'd'), taps * cop * ci) if dd else None,
                              cin_pad=cip, cout_pad=cop)
            if xf or xd:
                jobs.add_x3(w, buf((name, dev, 'x3f'), ops.x3_weight_bytes(cip, co), torch.uint8) if xf else None,
                            buf((name, dev, 'x3d'), ops.x3_weight_bytes(co, ci), torch.uint8) if xd else None, cin_pad=(cip + 15) // 16 * 16)
            if wf or wd:
                jobs.add_wino(w, buf((name, dev, 'uf'), 16 * co * ci) if wf else None, buf((name, dev, 'ud'), 16 * co * ci) if wd else None)
        for name in ('upv6', 'upv7', 'upv8', 'upv9'):
            w = P[name + '.weight']
            ci, co = w.shape[0], w.shape[1]
            if (self._pol.h2 and self._pol.h2_pointwise and self._pol.use_x3_pointwise(ci, 4 * co) and self._pol.use_x3_pointwise(co, ci)
                    and ops.gemm_h2_supported(ci, 4 * co) and ops.gemm_h2_supported(co, ci)):
                self._x3[name] = (False, False)                    # ConvTranspose2d on the pointwise fp16x2 GEMM kernel (csrc/gemm_h2s.hip)
                self._h2m[name] = (buf((name, dev, 'h2mf'), ops.h2mat_bytes(ci, 4 * co), torch.uint8),
                                   buf((name, dev, 'h2md'), ops.h2mat_bytes(4 * co, ci), torch.uint8) if need_dgrad else None)
                self._wslot[name] = jobs.add_h2_convt(w, self._h2m[name][0], self._h2m[name][1])
                continue
            if self._pol.use_x3_pointwise(ci, 4 * co) and self._pol.use_x3_pointwise(co, ci):
                self._x3[name] = (True, True)                      # ConvTranspose2d on the pointwise bf16x3 GEMM kernel
                jobs.add_x3_convt(w, buf((name, dev, 'x3f'), ops.x3mat_bytes(ci, 4 * co), torch.uint8),
                                  buf((name, dev, 'x3d'), ops.x3mat_bytes(4 * co, ci), torch.uint8) if need_dgrad else None)
                continue
            self._x3[name] = (False, False)
            key = (name, dev)
            if key not in self.packed:
                self.packed[key] = (torch.empty(w.numel(), dtype=torch.float32, device=dev),
                                    torch.empty(w.numel(), dtype=torch.float32, device=dev))
            f, d = self.packed[key]
            jobs.add_convt(w, f, d if need_dgrad else None)
        return jobs

    def _head_fusable(self):
        """conv10_1 inside conv9_2's epilogue (ops.conv_h2_fwd_head)?  nf = 32 (one 32-column tile holds a pixel's channels), 4 output planes, conv9_2 on
        the fp16x2 kernel; reflect-padded eval frames included (the head writes the PADDED NCHW planes like head_fwd did)."""
        return (self._pol.head_fused and self.ch[0] == 32 and self.cout == 4 and self._h2.get('conv9_2', (None, None))[0] is not None
                and self.m.conv10_1.weight.shape[1] == 32)

    def grad_out_channels(self, B, H, W):
        """channels of the NHWC loss gradient backward() wants: the streaming head kernel takes the 4 real ones (half the bytes of the zero-padded copy)"""
        return 4 if (self.cout == 4 and self._pol.use_thin_head(self.ch[0], self.cout, B * H * W)) else self.cout_pad

    def _w(self, name):
        """(forward, backward-data) direct packs of a layer (ConvTranspose2d: the pair built by add_convt)."""
        dev = self.params.flat.device
        if (name, dev) in self.packed:
            return self.packed[(name, dev)]
        return self.packed.get((name, dev, 'f')), self.packed.get((name, dev, 'd'))

    def _wino(self, name, co, ci, taps=9):
        """(forward, backward-data) through the Winograd F(2x2,3x3) kernel?  (self.policy)"""
        return self._pol.use_wino(co, ci, taps)

    def _wino_wgrad(self, h, w, cout, c1, c2, g_cs, x_cs):
        """Backward-weight through the Winograd kernel?  (self.policy)"""
        return self._pol.use_wino_wgrad(h, w, cout, c1, c2, g_cs, x_cs)

    def _wu(self, name):
        dev = self.params.flat.device
        return self.packed.get((name, dev, 'uf')), self.packed.get((name, dev, 'ud'))

    def _wx(self, name):
        dev = self.params.flat.device
        return self.packed.get((name, dev, 'x3f')), self.packed.get((name, dev, 'x3d'))

    # ------------------------------------------------------------------ forward
    def forward(self, x, train, reflect_pad=0, add_residual=True):
        """``reflect_pad`` > 0 (eval loop, trainer_SID.py:221-226): the network runs on the frame reflect-padded by that many pixels on
        every side -- the padding happens inside the NCHW -> NHWC layout pass, the result has the PADDED size (the caller crops).
        ``add_residual=False``: a `res` network returns f(x) without `+ x` (the caller adds the un-padded input after cropping:
        (f(pad x) + pad x)[crop] = f(pad x)[crop] + x; pnnp_eval_post_f32)."""
        if not x.is_cuda:
            raise PnnpError('UNetSeeInDark.forward: input must be a CUDA tensor (pnnp_amd has no CPU path)')
        x = x.contiguous().float()
        B, Cin, H, W = x.shape
        if reflect_pad:
            if train or (self.m.res and add_residual):
                raise PnnpError('reflect_pad is an eval-mode option; a `res` network needs add_residual=False (the caller adds the input after cropping)')
            H, W = H + 2 * reflect_pad, W + 2 * reflect_pad
        if Cin != self.cin or H % 16 or W % 16:
            raise PnnpError(f'input must be [B,{self.cin},H,W] with H,W multiples of 16, got {tuple(x.shape)}')
        dev = x.device
        self.params.ensure(dev)
        # packed weights are re-used while no parameter changed (eval loops); in-place torch updates bump
        # tensor._version, the fused Adam kernel goes through mark_dirty()
        self._pol = self.effective_policy(H, W, max(self.ch[0], self.cin_pad, self.cout_pad))
        self._packs_ready(train, dev)
        gen = self._begin_forward((B, H, W, dev), train)
        bufs = self.bufs.setdefault((B, H, W, dev), _Bufs())
        P = dict(self.m.named_parameters())
        ch = self.ch
        g = lambda n, s: bufs.get(n, s, dev)
        a = {}
        # fp16x2 family: amax slots of the activations (keyed by the name of the layer that wrote the tensor; a pooled map shares its
        # full-resolution map's slot) and, in a training forward, the sign bits of every LeakyReLU output that backward-data will need
        h2_on = bool(self._h2) or bool(self._h2m)     # amax-slot upkeep whenever ANY layer runs on an fp16x2 kernel (3x3 or pointwise)
        if h2_on:
            bufs.slots('f', dev).zero_()
        sl = lambda n: bufs.slot('f', n, dev)
        src_name = {}                          # id(tensor) -> the layer name its amax slot is keyed by

        # the zero-padded NHWC copy of the network input; its amax rides on the layout pass when conv1_1 runs on the fp16x2 kernel
        first_h2 = self._h2.get('conv1_1', (None, None))[0] is not None
        a['x8'] = ops.nchw_to_nhwc(x, g('x8', (B, H, W, self.cin_pad)), self.cin_pad, reflect_pad=reflect_pad, amax=sl('x8') if first_h2 else None)
        src_name[id(a['x8'])] = 'x8'

        def produced(t, name, fused):
            """`t` was just written by layer `name`; without a fused amax (and if an fp16x2 kernel will read it) a kernel of its own fills the slot."""
            src_name[id(t)] = name
            if h2_on and not fused:
                ops.amax(t, sl(name))
            return t

        def splitk(src, src2, h, w, cout):
            """K slices of a 3x3 fp16x2 forward launch on this map (1: none).  Eval forwards only: in a training forward the split launch gives up the fused
            max-pool and pays for the sign bits in its reduce -- measured neutral at B = 1 and 5 % slower at B = 2 (profiles/r6/small_batch.txt), where the step
            is bound by the host's ~110 launches anyway."""
            if not self._pol.splitk or train:
                return 1
            return ops.h2_splitk(B, h, w, (2 if src2 is not None else 1) * ((src.shape[3] + 15) // 16), cout)

        def conv(name, src, src2, h, w, cout, act=LRELU, taps=9, out=None):
            y = out if out is not None else g(name, (B, h, w, cout))
            hp = self._h2.get(name, (None, None))[0]
            if hp is not None:
                bits = bufs.bits(name, B, h, w, cout, dev) if (train and act == LRELU) else None
                if bits is not None:
                    a['bits:' + name] = bits
                ks = splitk(src, src2, h, w, cout)
                if ks > 1:                                           # a small grid: K in slices, one reduce (bias, activation, amax, sign bits)
                    ops.conv_h2_fwd_splitk(src, src2, hp, self._wslot[name], P[name + '.bias'], y, cout, act, sl(src_name[id(src)]), ks,
                                           bufs.scratch('splitk_ws', ks * y.numel(), dev), amax_x2=sl(src_name[id(src2)]) if src2 is not None else None,
                                           amax_y=sl(name), bits_y=bits)
                else:
                    ops.conv_h2_fwd(src, src2, hp, self._wslot[name], P[name + '.bias'], y, cout, act, sl(src_name[id(src)]),
                                    sl(src_name[id(src2)]) if src2 is not None else None, amax_y=sl(name), bits_y=bits)
                src_name[id(y)] = name
                return y
            if h2_on:
                src_name[id(y)] = name
            if self._x3.get(name, (False, False))[0]:
                ops.conv_x3_fwd(src, src2, self._wx(name)[0], P[name + '.bias'], y, cout, act)
            elif self._wn.get(name, (False, False))[0]:
                ops.conv_wino_fwd(src, src2, self._wu(name)[0], P[name + '.bias'], y, cout, act)
            else:
                ops.conv_fwd(src, src2, self._w(name)[0], P[name + '.bias'], y, cout, taps, act)
            return produced(y, name, fused=False) if (h2_on and taps == 9) else y

        hs = [H >> i for i in range(5)]
        ws = [W >> i for i in range(5)]
        cur = a['x8']
        for lvl in range(5):               # encoder: conv{l}_1, conv{l}_2, pool
            i = lvl + 1
            if lvl == 0 and not first_h2 and self._pol.use_thin_first(self.cin, ch[0], H, W, cur.shape[3]):
                # conv1_1 on the streaming kernel: its 4 input channels are not worth a (padded) GEMM chunk.  (With the fp16x2 family the
                # matrix-core kernel is as fast -- a padded chunk is 14 instructions, not 27 -- and writes the sign bits that make conv1_2's
                # backward-data read 1/32 of the bytes: the streaming kernel keeps conv1_1's weight gradient only.)
                a['c1a'] = produced(ops.first_fwd(cur, P['conv1_1.weight'], P['conv1_1.bias'], g('conv1_1', (B, H, W, ch[0])), LRELU,
                                                  amax_y=sl('conv1_1') if h2_on else None), 'conv1_1', fused=True)
            else:
                a[f'c{i}a'] = conv(f'conv{i}_1', cur, None, hs[lvl], ws[lvl], ch[lvl])
            if lvl == 4:
                a[f'c{i}'] = conv(f'conv{i}_2', a[f'c{i}a'], None, hs[lvl], ws[lvl], ch[lvl])
                continue
            codes = None
            fused = self._pol.pool_fused and (self._x3.get(f'conv{i}_2', (False, False))[0] or self._h2.get(f'conv{i}_2', (None, None))[0] is not None)
            if train or fused:                     # argmax + sign codes: the backward pass then does not re-read the full-resolution map
                codes = bufs.t.get(f'pc{i}')
                shp = (B, hs[lvl + 1], ws[lvl + 1], ch[lvl])
                if codes is None or tuple(codes.shape) != shp or codes.device != dev:
                    codes = bufs.t[f'pc{i}'] = torch.empty(shp, dtype=torch.uint8, device=dev)
                a[f'pc{i}'] = codes
            pooled = g(f'p{i}', (B, hs[lvl + 1], ws[lvl + 1], ch[lvl]))
            hp = self._h2.get(f'conv{i}_2', (None, None))[0]
            if hp is not None and self._pol.pool_fused and splitk(a[f'c{i}a'], None, hs[lvl], ws[lvl], ch[lvl]) > 1:
                # a small grid: the split-K launch has no fused pool -- conv (slices + reduce), then the pool kernel
                a[f'c{i}'] = conv(f'conv{i}_2', a[f'c{i}a'], None, hs[lvl], ws[lvl], ch[lvl])
                a[f'p{i}'] = ops.maxpool_fwd(a[f'c{i}'], pooled, codes=codes)
            elif hp is not None and self._pol.pool_fused:
                # conv{i}_2 writes the pooled map, the codes, its amax (the pooled map is a subset) and the sign bits from its own epilogue
                name = f'conv{i}_2'
                bits = bufs.bits(name, B, hs[lvl], ws[lvl], ch[lvl], dev) if train else None
                if bits is not None:
                    a['bits:' + name] = bits
                a[f'c{i}'] = ops.conv_h2_fwd_pool(a[f'c{i}a'], None, hp, self._wslot[name], P[name + '.bias'], g(name, (B, hs[lvl], ws[lvl], ch[lvl])),
                                                  pooled, codes, ch[lvl], LRELU, sl(src_name[id(a[f'c{i}a'])]), amax_y=sl(name), bits_y=bits)
                src_name[id(a[f'c{i}'])] = name
                a[f'p{i}'] = pooled
            elif fused:
                # conv{i}_2 writes the pooled map and the codes from its own epilogue (csrc/conv_x3.hip)
                a[f'c{i}'] = produced(ops.conv_x3_fwd_pool(a[f'c{i}a'], None, self._wx(f'conv{i}_2')[0], P[f'conv{i}_2.bias'],
                                                           g(f'conv{i}_2', (B, hs[lvl], ws[lvl], ch[lvl])), pooled, codes, ch[lvl], LRELU), f'conv{i}_2', fused=False)
                a[f'p{i}'] = pooled
            else:
                a[f'c{i}'] = conv(f'conv{i}_2', a[f'c{i}a'], None, hs[lvl], ws[lvl], ch[lvl])
                a[f'p{i}'] = ops.maxpool_fwd(a[f'c{i}'], pooled, codes=codes)
            src_name[id(a[f'p{i}'])] = f'conv{i}_2'                 # max |pooled| <= max |full-resolution map|
            cur = a[f'p{i}']
        cur = a['c5']
        for i in range(6, 10):             # decoder: upv{i}, conv{i}_1 on [up, skip], conv{i}_2
            lvl = 9 - i
            if f'upv{i}' in self._h2m:
                if id(cur) not in src_name:
                    produced(cur, f'in_upv{i}', fused=False)
                u = ops.convt_h2_fwd(cur, sl(src_name[id(cur)]), self._h2m[f'upv{i}'][0], self._wslot[f'upv{i}'], P[f'upv{i}.bias'],
                                     g(f'u{i}', (B, hs[lvl], ws[lvl], ch[lvl])), ch[lvl], amax_y=sl(f'upv{i}'))
                a[f'u{i}'] = produced(u, f'upv{i}', fused=True)
            elif self._x3.get(f'upv{i}', (False, False))[0]:
                u = ops.convt_x3_fwd(cur, self._wx(f'upv{i}')[0], P[f'upv{i}.bias'], g(f'u{i}', (B, hs[lvl], ws[lvl], ch[lvl])), ch[lvl],
                                     amax_y=sl(f'upv{i}') if h2_on else None)
                a[f'u{i}'] = produced(u, f'upv{i}', fused=True)
            else:
                u = ops.convt_fwd(cur, self._w(f'upv{i}')[0], P[f'upv{i}.bias'], g(f'u{i}', (B, hs[lvl], ws[lvl], ch[lvl])), ch[lvl])
                a[f'u{i}'] = produced(u, f'upv{i}', fused=False)
            a[f'c{i}a'] = conv(f'conv{i}_1', u, a[f'c{lvl + 1}'], hs[lvl], ws[lvl], ch[lvl])
            if i == 9 and self._head_fusable():
                break
            a[f'c{i}'] = conv(f'conv{i}_2', a[f'c{i}a'], None, hs[lvl], ws[lvl], ch[lvl])
            cur = a[f'c{i}']
        out = torch.empty((B, self.cout, H, W), dtype=torch.float32, device=dev)
        if 'c9' not in a:
            # conv9_2 + LeakyReLU + conv10_1 in one kernel (csrc/conv_h2s.hip EK_HEAD): the 4 output planes come straight from the accumulators; the
            # 32-channel map c9 is stored (with its sign bits and amax) only for a backward pass -- an eval forward neither writes nor re-reads it
            name = 'conv9_2'
            y = g(name, (B, H, W, ch[0])) if train else None
            bits = bufs.bits(name, B, H, W, ch[0], dev) if train else None
            if bits is not None:
                a['bits:' + name] = bits
            ops.conv_h2_fwd_head(a['c9a'], None, self._h2[name][0], self._wslot[name], P[name + '.bias'], y, ch[0], LRELU, sl(src_name[id(a['c9a'])]),
                                 P['conv10_1.weight'], P['conv10_1.bias'], out, amax_y=sl(name) if train else None, bits_y=bits,
                                 residual=x if (self.m.res and add_residual) else None)
            if train:
                a['c9'] = y
                src_name[id(y)] = name
        elif self._pol.use_thin_head(ch[0], self.cout, B * H * W):
            ops.head_fwd(a['c9'], P['conv10_1.weight'], P['conv10_1.bias'], out, residual=x if (self.m.res and add_residual) else None)
        else:
            o = conv('conv10_1', a['c9'], None, H, W, self.cout, act=0, taps=1, out=g('o', (B, H, W, self.cout)))
            ops.nhwc_to_nchw(o, out, residual=x if (self.m.res and add_residual) else None)
        if train:
            a['_pol'] = self._pol
            a['_src_name'] = src_name
            self.saved = (a, (B, H, W, dev), gen)
        return out

    # ------------------------------------------------------------------ backward
    def backward(self, g_out8, need_dx=False, accumulate=False, on_ready=None):
        """g_out8: dL/d(out) as NHWC [B,H,W,cout_pad] (zero padded).  Fills the flat gradient
        buffer.  ``on_ready(offset)`` is called as soon as flat_grad[offset:] is final (layers
        finish in exactly the reverse of the flat parameter order) so a data-parallel reducer
        can start all-reducing the tail while the rest of the backward pass still runs."""
        a, (B, H, W, dev), _ = self.saved
        self._pol = a['_pol']            # the kernel families this forward ran on (effective_policy)
        bufs = self.bufs[(B, H, W, dev)]
        ch = self.ch
        gb = lambda n, s: bufs.get('g_' + n, s, dev)
        G = self.params.grad_view
        P = dict(self.m.named_parameters())
        acc = 1 if accumulate else 0
        hs = [H >> i for i in range(5)]
        ws = [W >> i for i in range(5)]
        wsf = bufs.get('wgrad_ws', (self._ws_floats(B, H, W),), dev)

        # fp16x2 family: amax slots of the gradients (keyed by the buffer name), zeroed per backward; the activations' slots are the forward's
        h2_on = bool(self._h2) or bool(self._h2m)
        if h2_on:
            bufs.slots('b', dev).zero_()
        src_name = a.get('_src_name', {})
        slf = lambda t: bufs.slot('f', src_name[id(t)], dev)
        gname = {}                             # id(gradient tensor) -> slot name

        def gslot(t):
            return bufs.slot('b', gname[id(t)], dev)

        def gproduced(t, name, fused):
            gname[id(t)] = name
            if h2_on and not fused:
                ops.amax(t, bufs.slot('b', name, dev))
            return t

        def dgrad(name, gsrc, dx1, **kw):
            hp = self._h2.get(name, (None, None))[1]
            dx2 = kw.get('dx2')
            if hp is not None:
                # the act' masks as the forward kernels' sign bits where the masking activation came out of an fp16x2 layer
                for k_mask, k_bits in (('mask1', 'bits1'), ('mask2', 'bits2')):
                    m = kw.get(k_mask)
                    if m is not None and ('bits:' + src_name.get(id(m), '?')) in a:
                        kw[k_bits] = a['bits:' + src_name[id(m)]]
                        kw[k_mask] = None
                gname[id(dx1)] = 'd1:' + name
                if dx2 is not None:
                    gname[id(dx2)] = 'd2:' + name
                ops.conv_h2_bwd_data(gsrc, gslot(gsrc), hp, self._wslot[name], dx1, amax_dx1=gslot(dx1),
                                     amax_dx2=gslot(dx2) if dx2 is not None else None, **kw)
                return
            if self._x3.get(name, (False, False))[1]:
                ops.conv_x3_bwd_data(gsrc, self._wx(name)[1], dx1, **kw)
            elif self._wn.get(name, (False, False))[1]:
                ops.conv_wino_bwd_data(gsrc, self._wu(name)[1], dx1, **kw)
            else:
                ops.conv_bwd_data(gsrc, self._w(name)[1], dx1, **kw)
            if h2_on:
                gproduced(dx1, 'd1:' + name, fused=False)
                if dx2 is not None:
                    gproduced(dx2, 'd2:' + name, fused=False)

        def done(name):
            if on_ready is not None:
                on_ready(self.params.slices[name + '.weight'][0])

        def wgrad(name, gpre, cout, x1, c1, x2=None, taps=9):
            c2 = x2.shape[3] if x2 is not None else 0
            if (taps == 9 and h2_on and self._pol.h2_wgrad and id(gpre) in gname and id(x1) in src_name and (x2 is None or id(x2) in src_name) and
                    self._pol.use_x3_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, batch=B, cs=max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0))):
                ops.conv_h2_bwd_weight(gpre, gslot(gpre), cout, x1, slf(x1), c1, x2, slf(x2) if x2 is not None else None,
                                       G(name + '.weight', P[name + '.weight'].shape), G(name + '.bias', (cout,)), wsf, accumulate=acc)
                done(name)
                return
            if taps == 9 and self._pol.use_x3_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, batch=B,
                                                    cs=max(gpre.shape[3], x1.shape[3], x2.shape[3] if x2 is not None else 0)):
                ops.conv_x3_bwd_weight(gpre, cout, x1, c1, x2, G(name + '.weight', P[name + '.weight'].shape),
                                       G(name + '.bias', (cout,)), wsf, accumulate=acc)
            elif taps == 9 and self._wino_wgrad(gpre.shape[1], gpre.shape[2], cout, c1, c2, gpre.shape[3], x1.shape[3]):
                ops.conv_wino_bwd_weight(gpre, cout, x1, c1, x2, G(name + '.weight', P[name + '.weight'].shape),
                                         G(name + '.bias', (cout,)), wsf, accumulate=acc)
            else:
                ops.conv_bwd_weight(gpre, cout, x1, c1, x2, G(name + '.weight', P[name + '.weight'].shape),
                                    G(name + '.bias', (cout,)), taps, wsf, accumulate=acc)
            done(name)

        # conv10_1 (1x1, no activation); its input c9 is a LeakyReLU output
        g_cur = gb('c9', a['c9'].shape)
        if self._pol.use_thin_head(ch[0], self.cout, B * H * W):
            ops.head_bwd(g_out8, a['c9'], P['conv10_1.weight'], g_cur, G('conv10_1.weight', P['conv10_1.weight'].shape),
                         G('conv10_1.bias', (self.cout,)), wsf, mode=LRELU, accumulate=acc, amax_gx=bufs.slot('b', 'head', dev) if h2_on else None)
            done('conv10_1')
            gproduced(g_cur, 'head', fused=True)
        else:
            wgrad('conv10_1', g_out8, self.cout, a['c9'], ch[0], taps=1)
            ops.conv_bwd_data(g_out8, self._w('conv10_1')[1], g_cur, mask1=a['c9'], mode1=LRELU, taps=1)
            gproduced(g_cur, 'head', fused=False)
        for i in range(9, 5, -1):          # decoder, top-down
            lvl = 9 - i
            wgrad(f'conv{i}_2', g_cur, ch[lvl], a[f'c{i}a'], ch[lvl])
            g_a = gb(f'c{i}a', a[f'c{i}a'].shape)
            dgrad(f'conv{i}_2', g_cur, g_a, mask1=a[f'c{i}a'], mode1=LRELU)
            skip = a[f'c{lvl + 1}']
            wgrad(f'conv{i}_1', g_a, ch[lvl], a[f'u{i}'], ch[lvl], x2=skip)
            g_u = gb(f'u{i}', a[f'u{i}'].shape)
            g_skip = gb(f'c{lvl + 1}', skip.shape)
            dgrad(f'conv{i}_1', g_a, g_u, dx2=g_skip, mask2=skip, mode2=LRELU)
            below = a['c5'] if i == 6 else a[f'c{i - 1}']
            ct_wgrad = ops.convt_x3_bwd_weight if self._pol.use_x3g_wgrad(ops.X3G_CT, below.shape[3], ch[lvl], B, below.shape[1], below.shape[2],
                                                                           g_u.shape[1], g_u.shape[2], max(below.shape[3], g_u.shape[3])) else ops.convt_bwd_weight
            if ct_wgrad is ops.convt_x3_bwd_weight and h2_on and self._pol.h2_pointwise and id(below) in src_name and id(g_u) in gname:
                ops.convt_h2_bwd_weight(below, slf(below), g_u, gslot(g_u), G(f'upv{i}.weight', P[f'upv{i}.weight'].shape), wsf, accumulate=acc,
                                        dbias=G(f'upv{i}.bias', (ch[lvl],)))
            else:
                ct_wgrad(below, g_u, G(f'upv{i}.weight', P[f'upv{i}.weight'].shape), wsf, accumulate=acc, dbias=G(f'upv{i}.bias', (ch[lvl],)))
            done(f'upv{i}')
            g_cur = gb('c5' if i == 6 else f'c{i - 1}', below.shape)
            if f'upv{i}' in self._h2m:
                if id(g_u) not in gname:
                    gproduced(g_u, f'gu{i}', fused=False)
                # the act' mask as the sign bits conv{i-1}_2's forward kernel stored (the float32 activation is not read: 503 MB per step over the four layers)
                bits_below = a.get('bits:' + src_name.get(id(below), '?')) if self._pol.convt_bits else None
                ops.convt_h2_bwd_data(g_u, gslot(g_u), self._h2m[f'upv{i}'][1], self._wslot[f'upv{i}'], g_cur, mask=below, mode=LRELU,
                                      amax_dx=bufs.slot('b', f'upv{i}', dev), bits=bits_below if (bits_below is not None and below.shape[3] % 32 == 0) else None)
                gproduced(g_cur, f'upv{i}', fused=True)
            elif self._x3.get(f'upv{i}', (False, False))[1]:
                ops.convt_x3_bwd_data(g_u, self._wx(f'upv{i}')[1], g_cur, mask=below, mode=LRELU, amax_dx=bufs.slot('b', f'upv{i}', dev) if h2_on else None)
                gproduced(g_cur, f'upv{i}', fused=True)
            else:
                ops.convt_bwd_data(g_u, self._w(f'upv{i}')[1], g_cur, mask=below, mode=LRELU)
                gproduced(g_cur, f'upv{i}', fused=False)
        dx = None
        for i in range(5, 0, -1):          # encoder, bottom-up
            lvl = i - 1
            wgrad(f'conv{i}_2', g_cur, ch[lvl], a[f'c{i}a'], ch[lvl])
            g_a = gb(f'c{i}a', a[f'c{i}a'].shape)
            dgrad(f'conv{i}_2', g_cur, g_a, mask1=a[f'c{i}a'], mode1=LRELU)
            if i > 1:
                src = a[f'p{i - 1}']
                wgrad(f'conv{i}_1', g_a, ch[lvl], src, ch[lvl - 1])
                g_p = gb(f'p{i - 1}', src.shape)
                dgrad(f'conv{i}_1', g_a, g_p)
                g_cur = gb(f'c{i - 1}', a[f'c{i - 1}'].shape)      # already holds the skip gradient
                # (the skip gradient + the scattered pooled gradient: a new tensor, a new amax slot)
                ops.maxpool_bwd(a[f'c{i - 1}'], g_p, g_cur, LRELU, 1, codes=a.get(f'pc{i - 1}'), amax_gx=bufs.slot('b', f'pool{i - 1}', dev) if h2_on else None)
                gproduced(g_cur, f'pool{i - 1}', fused=True)
            else:
                if self._pol.use_thin_first(self.cin, ch[0], H, W, a['x8'].shape[3]):
                    ops.first_bwd_weight(g_a, ch[0], a['x8'], self.cin, G('conv1_1.weight', P['conv1_1.weight'].shape),
                                         G('conv1_1.bias', (ch[0],)), wsf, accumulate=acc)
                    done('conv1_1')
                else:
                    wgrad('conv1_1', g_a, ch[0], a['x8'], self.cin)
                if need_dx:
                    raise PnnpError('gradient w.r.t. the network input is not implemented on the HIP path')
        return dx

    def _ws_floats(self, B, H, W):
        ch = self.ch
        need = 1024 * max(ch)
        for lvl in range(5):
            h, w = H >> lvl, W >> lvl
            c = ch[lvl]
            cin = self.cin if lvl == 0 else ch[lvl - 1]
            need = max(need, ops.x3_wgrad_workspace_floats(B, h, w, c, c), ops.x3_wgrad_workspace_floats(B, h, w, c, 2 * c),
                       ops.x3_wgrad_workspace_floats(B, h, w, c, cin) if cin % 32 == 0 else 0)
            need = max(need, ops.wgrad_workspace_floats(B, h, w, c, c, 9), ops.wgrad_workspace_floats(B, h, w, c, cin, 9),
                       ops.wgrad_workspace_floats(B, h, w, c, 2 * c, 9), ops.wino_wgrad_workspace_floats(B, h, w, c, c),
                       ops.wino_wgrad_workspace_floats(B, h, w, c, cin), ops.wino_wgrad_workspace_floats(B, h, w, c, 2 * c))
            if lvl < 4:
                need = max(need, ops.wgrad_workspace_floats(B, h >> 1, w >> 1, ch[lvl + 1], c, 4),
                           ops.x3g_wgrad_workspace_floats(ops.X3G_CT, B, h >> 1, w >> 1, ch[lvl + 1], c))
        need = max(need, ops.wgrad_workspace_floats(B, H, W, self.cout, ch[0], 1))
        need = max(need, ops.head_bwd_workspace_floats(ch[0]), ops.first_wgrad_workspace_floats(ch[0]))
        return need


class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, engine, train, *params):
        # autograd.Function.forward runs with grad mode off: `train` is decided by the caller
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
        # autograd may keep (or accumulate in place into) what we return, and the flat buffer
        # is overwritten by the next backward: hand out copies on this compatibility path
        grads = []
        for n, p in e.m.named_parameters():
            grads.append(e.params.grad_view(n, p.shape).clone() if p.requires_grad else None)
        return (None, None, None) + tuple(grads)


class UNetSeeInDark(nn.Module):
    """Drop-in for archs/Unet.py:4-99 (same ``args`` keys: nframes, res, nf, in_nc, out_nc)."""

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
        prev = self.in_nc * self.nframes
        for lvl in range(5):
            setattr(self, f'conv{lvl + 1}_1', nn.Conv2d(prev, c[lvl], kernel_size=3, stride=1, padding=1))
            setattr(self, f'conv{lvl + 1}_2', nn.Conv2d(c[lvl], c[lvl], kernel_size=3, stride=1, padding=1))
            prev = c[lvl]
        for i in range(6, 10):
            lvl = 9 - i
            setattr(self, f'upv{i}', nn.ConvTranspose2d(c[lvl + 1], c[lvl], 2, stride=2))
            setattr(self, f'conv{i}_1', nn.Conv2d(c[lvl + 1], c[lvl], kernel_size=3, stride=1, padding=1))
            setattr(self, f'conv{i}_2', nn.Conv2d(c[lvl], c[lvl], kernel_size=3, stride=1, padding=1))
        self.conv10_1 = nn.Conv2d(nf, self.out_nc, kernel_size=1, stride=1)
        self._engine = None

    @property
    def engine(self):
        if self._engine is None:
            object.__setattr__(self, '_engine', UNetEngine(self))
        return self._engine

    def forward(self, x):
        params = list(self.parameters())
        train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        return _UNetFn.apply(x, self.engine, train, *params)
