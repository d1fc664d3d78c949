"""Thin ctypes wrappers of the layer-level C ABI (include/pnnp_hip.h).  Tensors are CUDA
fp32; activations NHWC.  No CPU path: every function raises on CPU tensors."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, require_cuda, stream

_i64 = C.c_int64

# Optional per-launch timing with HIP events on the launching stream (bench.py turns it on):
# PROFILE = [] collects (kind, algorithmic_flops, algorithmic_bytes, start_event, end_event, sub); `sub` names the kernel instantiation
# a launch class resolves to where that matters for its roofline ('bn32' / 'bn64': pnnp_h2_tile_columns), else ''.
PROFILE = None
PROFILE_KINDS = None        # None: every launch class; a set: only those (bench.py times just the dominant kernel in its timed region)


class _Timed:
    def __init__(self, kind, flops=0.0, nbytes=0.0, sub=None):
        self.rec = None
        if PROFILE is not None and (PROFILE_KINDS is None or kind in PROFILE_KINDS):
            self.rec = (kind, float(flops), float(nbytes), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                        sub() if callable(sub) else (sub or ''))

    def __enter__(self):
        if self.rec is not None:
            self.rec[3].record(torch.cuda.current_stream())

    def __exit__(self, *a):
        if self.rec is not None:
            self.rec[4].record(torch.cuda.current_stream())
            PROFILE.append(self.rec)


ABI_VERSION = 6                 # the PNNP_ABI_VERSION of include/pnnp_hip.h these wrappers (and the PackJob mirror below) were written against


def _prep():
    L = lib()
    if not getattr(L, '_pnnp_sigs', False):
        got = L.pnnp_abi_version() if hasattr(L, 'pnnp_abi_version') else None
        if got != ABI_VERSION or L.pnnp_pack_job_bytes() != C.sizeof(PackJob):
            raise _lib.PnnpError(f'{_lib.LIB_PATH}: ABI version {got} / PnnpPackJob of {L.pnnp_pack_job_bytes() if got else "?"} bytes, these bindings '
                                 f'were written for version {ABI_VERSION} / {C.sizeof(PackJob)} bytes: rebuild the library (tools/build.py)')
        L.pnnp_wgrad_workspace_floats.restype = C.c_int64
        L.pnnp_wino_weight_floats.restype = C.c_int64
        L.pnnp_wino_wgrad_workspace_floats.restype = C.c_int64
        L.pnnp_x3_weight_bytes.restype = C.c_int64
        L.pnnp_x3_wgrad_workspace_floats.restype = C.c_int64
        L.pnnp_x3g_wgrad_workspace_floats.restype = C.c_int64
        L.pnnp_x3mat_bytes.restype = C.c_int64
        L.pnnp_h2_weight_bytes.restype = C.c_int64
        L.pnnp_h2_bits_words.restype = C.c_int64
        L.pnnp_h2mat_bytes.restype = C.c_int64
        L.pnnp_h2g_wgrad_workspace_floats.restype = C.c_int64
        L.pnnp_head_bwd_workspace_floats.restype = C.c_int64
        L.pnnp_first_wgrad_workspace_floats.restype = C.c_int64
        L._pnnp_sigs = True
    return L


def pack_conv_weight(w, fwd, dgrad, cin_pad=None, cout_pad=None):
    co, ci, kh, kw = w.shape
    check(_prep().pnnp_pack_conv_weight_f32(ptr(w), ptr(fwd), ptr(dgrad), co, ci, kh * kw,
                                            cin_pad or ci, cout_pad or co, stream()), 'pack_conv_weight')


class PackJob(C.Structure):                      # PnnpPackJob of include/pnnp_hip.h
    _fields_ = [('src', C.c_void_p), ('dst', C.c_void_p), ('kind', C.c_int), ('T', C.c_int), ('K', C.c_int), ('N', C.c_int),
                ('sk', C.c_int64), ('sn', C.c_int64), ('st', C.c_int64), ('off', C.c_int64),
                ('flip', C.c_int), ('Kvalid', C.c_int), ('Ndst', C.c_int), ('n_off', C.c_int), ('amax', C.c_void_p)]


class PackJobs:
    """A table of weight re-pack jobs (csrc/pack_jobs.hip): filled once per model / device with the add_* builders, run once
    per optimiser step in ceil(n / 32) launches instead of one or two launches per layer.  The tensors whose pointers are
    recorded must stay alive and in place (parameters as views of the flat buffer, the persistent packed buffers)."""

    def __init__(self, cap=256):
        self.cap = cap
        self.jobs = (PackJob * cap)()
        self.n = C.c_int(0)
        self.keep = []                           # the tensors behind the recorded pointers
        # fp16x2 packs (add_h2) scale every weight tensor by its own maximum: the amax jobs form a table of their own that run() launches
        # FIRST (a pack job reads the slot its amax job filled); `wslots` holds one 4-byte slot per weight tensor
        self.amax_jobs = None
        self.n_amax = C.c_int(0)
        self.wslots = None
        self._wslot_of = {}

    def add_conv(self, w, fwd, dgrad, cin_pad=None, cout_pad=None):
        co, ci, kh, kw = w.shape
        check(_prep().pnnp_pack_jobs_add_conv(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci, kh * kw,
                                              cin_pad or ci, cout_pad or co), 'pack_jobs_add_conv')
        self.keep += [w, fwd, dgrad]

    def add_convt(self, w, fwd, dgrad):
        ci, co = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_convt(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), ci, co), 'pack_jobs_add_convt')
        self.keep += [w, fwd, dgrad]

    def add_wino(self, w, fwd, dgrad):
        co, ci = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_wino(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci), 'pack_jobs_add_wino')
        self.keep += [w, fwd, dgrad]

    def add_x3(self, w, fwd, dgrad, cin_pad=None):
        """bf16x3 packs (csrc/conv_x3.hip) of a 3x3 Conv2d weight; fwd / dgrad: uint8 buffers of x3_weight_bytes, or None."""
        co, ci = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_x3(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci,
                                            cin_pad or (ci + 15) // 16 * 16), 'pack_jobs_add_x3')
        self.keep += [w, fwd, dgrad]

    def weight_slot(self, w):
        """The amax slot (a 1-element int32 view) of weight tensor ``w``; its amax job is added on first use."""
        key = (w.data_ptr(), w.numel())
        if key not in self._wslot_of:
            if self.wslots is None:
                self.wslots = torch.zeros(self.cap, dtype=torch.int32, device=w.device)
                self.amax_jobs = (PackJob * self.cap)()
            i = len(self._wslot_of)
            slot = self.wslots[i:i + 1]
            check(_prep().pnnp_pack_jobs_add_amax(self.amax_jobs, C.byref(self.n_amax), self.cap, ptr(w), C.c_int64(w.numel()), ptr(slot)), 'pack_jobs_add_amax')
            self._wslot_of[key] = slot
            self.keep += [w]
        return self._wslot_of[key]

    def add_h2(self, w, fwd, dgrad, cin_pad=None):
        """fp16x2 packs (csrc/conv_h2s.hip) of a 3x3 Conv2d weight; fwd / dgrad: uint8 buffers of h2_weight_bytes, or None.  Returns the
        weight tensor's amax slot (what the kernels take as ``amax_w``)."""
        co, ci = w.shape[0], w.shape[1]
        slot = self.weight_slot(w)
        check(_prep().pnnp_pack_jobs_add_h2(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci,
                                            cin_pad or (ci + 15) // 16 * 16, ptr(slot)), 'pack_jobs_add_h2')
        self.keep += [w, fwd, dgrad]
        return slot

    def add_h2_convt(self, w, fwd, dgrad):
        """fp16x2 packs (csrc/gemm_h2s.hip) of a ConvTranspose2d(2, 2) weight; fwd / dgrad: uint8 buffers of h2mat_bytes (zero-filled), or None.
        Returns the weight tensor's amax slot."""
        ci, co = w.shape[0], w.shape[1]
        slot = self.weight_slot(w)
        check(_prep().pnnp_pack_jobs_add_h2_convt(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), ci, co, ptr(slot)), 'pack_jobs_add_h2_convt')
        self.keep += [w, fwd, dgrad]
        return slot

    def add_h2_1x1(self, w, fwd, dgrad):
        co, ci = w.shape[0], w.shape[1]
        slot = self.weight_slot(w)
        check(_prep().pnnp_pack_jobs_add_h2_1x1(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci, ptr(slot)), 'pack_jobs_add_h2_1x1')
        self.keep += [w, fwd, dgrad]
        return slot

    def add_h2_s2(self, w, fwd, dgrad):
        co, ci = w.shape[0], w.shape[1]
        slot = self.weight_slot(w)
        check(_prep().pnnp_pack_jobs_add_h2_s2(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci, ptr(slot)), 'pack_jobs_add_h2_s2')
        self.keep += [w, fwd, dgrad]
        return slot

    def add_x3_convt(self, w, fwd, dgrad):
        ci, co = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_x3_convt(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), ci, co), 'pack_jobs_add_x3_convt')
        self.keep += [w, fwd, dgrad]

    def add_x3_1x1(self, w, fwd, dgrad):
        co, ci = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_x3_1x1(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci), 'pack_jobs_add_x3_1x1')
        self.keep += [w, fwd, dgrad]

    def add_x3_s2(self, w, fwd, dgrad):
        co, ci = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_x3_s2(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(fwd), ptr(dgrad), co, ci), 'pack_jobs_add_x3_s2')
        self.keep += [w, fwd, dgrad]

    def add_s2_dgrad(self, w, dst):
        co, ci = w.shape[0], w.shape[1]
        check(_prep().pnnp_pack_jobs_add_conv3x3s2_dgrad(self.jobs, C.byref(self.n), self.cap, ptr(w), ptr(dst), co, ci), 'pack_jobs_add_s2_dgrad')
        self.keep += [w, dst]

    def run(self):
        if self.n_amax.value:
            self.wslots.zero_()
            check(_prep().pnnp_pack_jobs_f32(self.amax_jobs, self.n_amax.value, stream()), 'pack_jobs (amax)')
        check(_prep().pnnp_pack_jobs_f32(self.jobs, self.n.value, stream()), 'pack_jobs')


def pack_convt_weight(w, fwd, dgrad):
    ci, co = w.shape[:2]
    check(_prep().pnnp_pack_convt_weight_f32(ptr(w), ptr(fwd), ptr(dgrad), ci, co, stream()), 'pack_convt_weight')


def conv_fwd(x1, x2, w_packed, bias, y, cout, taps, act, residual=None):
    require_cuda(x1, x2, w_packed, y)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv%d_fwd' % taps, 2.0 * B * H * W * cout * (C1 + C2) * taps, 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv_fwd_f32(ptr(x1), C1, ptr(x2), C2, ptr(w_packed), ptr(bias), ptr(residual), ptr(y),
                                        B, H, W, cout, taps, act, stream()), 'conv_fwd')
    return y


def conv_bwd_data(g, w_dgrad, dx1, mask1=None, mode1=0, accum1=0, dx2=None, mask2=None, mode2=0, accum2=0, taps=9):
    require_cuda(g, w_dgrad, dx1)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]
    C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv%d_dgrad' % taps, 2.0 * B * H * W * Cout * (C1 + C2) * taps, 4.0 * B * H * W * (C1 + C2 + Cout)):
        check(_prep().pnnp_conv_bwd_data_f32(ptr(g), Cout, ptr(w_dgrad), ptr(dx1), C1, ptr(mask1), mode1, accum1,
                                             ptr(dx2), C2, ptr(mask2), mode2, accum2, B, H, W, taps, stream()), 'conv_bwd_data')


def x3_supported(K, N):
    return bool(_prep().pnnp_x3_supported(int(K), int(N)))


def x3_image_fits(H, W, cstride):
    """one [H][W][cstride] image within the 32-bit offsets of the bf16x3 forward / backward-data / pointwise kernels?"""
    return bool(_prep().pnnp_x3_image_fits(int(H), int(W), int(cstride)))


def x3_wgrad_fits(B, H, W, cstride):
    return bool(_prep().pnnp_x3_wgrad_fits(int(B), int(H), int(W), int(cstride)))


def x3_weight_bytes(K, N):
    return int(_prep().pnnp_x3_weight_bytes(int(K), int(N)))


def conv_x3_fwd(x1, x2, w_x3, bias, y, cout, act, residual=None):
    """3x3 / stride 1 / pad 1 forward on the bf16 matrix cores, fp32 operands split in three (same contract as conv_fwd, taps=9)."""
    require_cuda(x1, x2, w_x3, y)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_x3', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv3x3_x3_fwd_f32(ptr(x1), C1, ptr(x2), C2, ptr(w_x3), ptr(bias), ptr(residual), ptr(y), B, H, W, cout, act,
                                              stream()), 'conv_x3_fwd')
    return y


def conv_x3_fwd_pool(x1, x2, w_x3, bias, y, pooled, codes, cout, act):
    """conv_x3_fwd + MaxPool2d(2) of the activated output in the same kernel (pooled [B,H/2,W/2,cout], codes as maxpool_fwd)."""
    require_cuda(x1, x2, w_x3, y, pooled, codes)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_x3', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv3x3_x3_fwd_pool_f32(ptr(x1), C1, ptr(x2), C2, ptr(w_x3), ptr(bias), ptr(y), ptr(pooled), ptr(codes),
                                                   B, H, W, cout, act, stream()), 'conv_x3_fwd_pool')
    return y


def conv_x3_bwd_data(g, w_x3_dgrad, dx1, mask1=None, mode1=0, accum1=0, dx2=None, mask2=None, mode2=0, accum2=0):
    require_cuda(g, w_x3_dgrad, dx1)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]
    C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv9_dgrad_x3', 2.0 * B * H * W * Cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + Cout)):
        check(_prep().pnnp_conv3x3_x3_bwd_data_f32(ptr(g), Cout, ptr(w_x3_dgrad), ptr(dx1), C1, ptr(mask1), mode1, accum1,
                                                   ptr(dx2), C2, ptr(mask2), mode2, accum2, B, H, W, stream()), 'conv_x3_bwd_data')


def conv_x3_bwd_data_res(g, w_x3_dgrad, dx, addsrc, mask=None, mode=0):
    require_cuda(g, w_x3_dgrad, dx, addsrc)
    B, H, W, Cout = g.shape
    C1 = dx.shape[3]
    with _Timed('conv9_dgrad_x3', 2.0 * B * H * W * Cout * C1 * 9, 4.0 * B * H * W * (2 * C1 + Cout)):
        check(_prep().pnnp_conv3x3_x3_bwd_data_res_f32(ptr(g), Cout, ptr(w_x3_dgrad), ptr(dx), C1, ptr(addsrc), ptr(mask), mode,
                                                       B, H, W, stream()), 'conv_x3_bwd_data_res')


# ---- the fp16x2 ("h2") family: csrc/conv_h2s.hip, csrc/h2.h.  amax_* are 1-element int32 CUDA tensors (the slots), bits_* int32 tensors of
# h2_bits_words elements.
def h2_supported(K, N):
    return bool(_prep().pnnp_h2_supported(int(K), int(N)))


def h2_weight_bytes(K, N):
    return int(_prep().pnnp_h2_weight_bytes(int(K), int(N)))


def h2_bits_words(B, H, W, C_):
    return int(_prep().pnnp_h2_bits_words(int(B), int(H), int(W), int(C_)))


def amax(x, slot):
    """slot = max(slot, max |x|) as a kernel of its own (tensors whose producer has no fused amax)."""
    require_cuda(x, slot)
    with _Timed('amax', 0.0, 4.0 * x.numel()):
        check(_prep().pnnp_amax_f32(ptr(x), C.c_int64(x.numel()), ptr(slot), stream()), 'amax')
    return slot


def h2_tile_columns(B, H, W, N, pool=False):
    """GEMM columns per workgroup tile the fp16x2 3x3 kernel picks for this map (32: the HBM-bound instantiations; pnnp_h2_tile_columns)."""
    return int(_prep().pnnp_h2_tile_columns(int(B), int(H), int(W), int(N), int(bool(pool))))


def conv_h2_fwd(x1, x2, w_h2, amax_w, bias, y, cout, act, amax_x1, amax_x2=None, amax_y=None, bits_y=None, residual=None):
    """3x3 / stride 1 / pad 1 forward on the fp16 matrix cores, fp32 operands split in two scaled pieces (contract of conv_fwd, taps=9)."""
    require_cuda(x1, x2, w_h2, y, amax_w, amax_x1)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_h2', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + cout), sub=lambda: 'bn%d' % h2_tile_columns(B, H, W, cout)):
        check(_prep().pnnp_conv3x3_h2_fwd_f32(ptr(x1), C1, ptr(amax_x1), ptr(x2), C2, ptr(amax_x2), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(residual),
                                              ptr(y), ptr(amax_y), ptr(bits_y), B, H, W, cout, act, stream()), 'conv_h2_fwd')
    return y


def h2_splitk(B, H, W, chunks, N):
    """K slices for a small grid (pnnp_h2_splitk): 1 = launch as usual."""
    return int(_prep().pnnp_h2_splitk(int(B), int(H), int(W), int(chunks), int(N)))


def conv_h2_fwd_splitk(x1, x2, w_h2, amax_w, bias, y, cout, act, amax_x1, ksplit, ws, amax_x2=None, amax_y=None, bits_y=None):
    """conv_h2_fwd with K cut into ``ksplit`` slices (small grids): partial sums into ``ws`` (>= ksplit * y.numel() floats), one fixed-order reduce."""
    require_cuda(x1, x2, w_h2, y, amax_w, amax_x1, ws)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_h2', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + cout), sub='splitk'):
        check(_prep().pnnp_conv3x3_h2_fwd_splitk_f32(ptr(x1), C1, ptr(amax_x1), ptr(x2), C2, ptr(amax_x2), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(y), ptr(amax_y),
                                                     ptr(bits_y), B, H, W, cout, act, int(ksplit), ptr(ws), _i64(ws.numel()), stream()), 'conv_h2_fwd_splitk')
    return y


def conv_h2_fwd_head(x1, x2, w_h2, amax_w, bias, y, cout, act, amax_x1, head_w, head_b, out, amax_x2=None, amax_y=None, bits_y=None, residual=None):
    """The last 3x3 layer + activation + the 1x1 head in one kernel (archs/Unet.py:93-94): ``out`` NCHW [B,4,H,W] (+ ``residual``, NCHW);
    ``y`` None: the 32-channel map is not stored (eval forward)."""
    require_cuda(x1, x2, w_h2, y, amax_w, amax_x1, head_w, out, residual)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_h2', 2.0 * B * H * W * cout * (C1 + C2) * 9 + 2.0 * B * H * W * cout * 4, 4.0 * B * H * W * (C1 + C2 + (cout if y is not None else 0) + 4), sub='bn32'):
        check(_prep().pnnp_conv3x3_h2_fwd_head_f32(ptr(x1), C1, ptr(amax_x1), ptr(x2), C2, ptr(amax_x2), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(y), ptr(amax_y),
                                                   ptr(bits_y), ptr(head_w), ptr(head_b), ptr(residual), ptr(out), B, H, W, cout, act, stream()), 'conv_h2_fwd_head')
    return out


def conv_h2_fwd_pool(x1, x2, w_h2, amax_w, bias, y, pooled, codes, cout, act, amax_x1, amax_x2=None, amax_y=None, bits_y=None):
    require_cuda(x1, x2, w_h2, y, pooled, codes, amax_w, amax_x1)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_h2', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + 1.25 * cout) + B * H * W * cout / 4,
                sub=lambda: 'bn%d' % h2_tile_columns(B, H, W, cout, True)):      # (+ the pooled map and its one-byte codes)
        check(_prep().pnnp_conv3x3_h2_fwd_pool_f32(ptr(x1), C1, ptr(amax_x1), ptr(x2), C2, ptr(amax_x2), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(y),
                                                   ptr(pooled), ptr(codes), ptr(amax_y), ptr(bits_y), B, H, W, cout, act, stream()), 'conv_h2_fwd_pool')
    return y


def conv_h2_bwd_data(g, amax_g, w_h2_dgrad, amax_w, dx1, mask1=None, bits1=None, mode1=0, accum1=0, amax_dx1=None,
                     dx2=None, mask2=None, bits2=None, mode2=0, accum2=0, amax_dx2=None):
    require_cuda(g, w_h2_dgrad, dx1, amax_g, amax_w)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]
    C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv9_dgrad_h2', 2.0 * B * H * W * Cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + Cout), sub=lambda: 'bn%d' % h2_tile_columns(B, H, W, C1 + C2)):
        check(_prep().pnnp_conv3x3_h2_bwd_data_f32(ptr(g), Cout, ptr(amax_g), ptr(w_h2_dgrad), ptr(amax_w),
                                                   ptr(dx1), C1, ptr(mask1), ptr(bits1), mode1, accum1, ptr(amax_dx1),
                                                   ptr(dx2), C2, ptr(mask2), ptr(bits2), mode2, accum2, ptr(amax_dx2), B, H, W, stream()), 'conv_h2_bwd_data')


def conv_h2_bwd_data_res(g, amax_g, w_h2_dgrad, amax_w, dx, addsrc, mask=None, mode=0, amax_dx=None):
    require_cuda(g, w_h2_dgrad, dx, addsrc, amax_g, amax_w)
    B, H, W, Cout = g.shape
    C1 = dx.shape[3]
    with _Timed('conv9_dgrad_h2', 2.0 * B * H * W * Cout * C1 * 9, 4.0 * B * H * W * (2 * C1 + Cout), sub=lambda: 'bn%d' % h2_tile_columns(B, H, W, C1)):
        check(_prep().pnnp_conv3x3_h2_bwd_data_res_f32(ptr(g), Cout, ptr(amax_g), ptr(w_h2_dgrad), ptr(amax_w), ptr(dx), C1, ptr(addsrc), ptr(mask), mode,
                                                       ptr(amax_dx), B, H, W, stream()), 'conv_h2_bwd_data_res')


def gemm_x3_supported(K, N):
    return bool(_prep().pnnp_gemm_x3_supported(int(K), int(N)))


def x3mat_bytes(K, N):
    return int(_prep().pnnp_x3mat_bytes(int(K), int(N)))


def convt_x3_fwd(x, w_x3, bias, y, cout, amax_y=None):
    require_cuda(x, w_x3, y)
    B, H, W, Cin = x.shape
    with _Timed('convt_fwd_x3', 8.0 * B * H * W * Cin * cout, 4.0 * B * H * W * (Cin + 4 * cout)):
        check(_prep().pnnp_convt2x2_x3_fwd_amax_f32(ptr(x), Cin, ptr(w_x3), ptr(bias), ptr(y), ptr(amax_y), B, H, W, cout, stream()), 'convt_x3_fwd')
    return y


def convt_x3_bwd_data(g, w_x3_dgrad, dx, mask=None, mode=0, amax_dx=None):
    require_cuda(g, w_x3_dgrad, dx)
    B, H, W, Cin = dx.shape
    with _Timed('convt_dgrad_x3', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        check(_prep().pnnp_convt2x2_x3_bwd_data_amax_f32(ptr(g), g.shape[3], ptr(w_x3_dgrad), ptr(dx), Cin, ptr(mask), mode, ptr(amax_dx), B, H, W, stream()),
              'convt_x3_bwd_data')


def gemm_h2_supported(K, N):
    return bool(_prep().pnnp_gemm_h2_supported(int(K), int(N)))


def h2mat_bytes(K, N):
    return int(_prep().pnnp_h2mat_bytes(int(K), int(N)))


def convt_h2_fwd(x, amax_x, w_h2, amax_w, bias, y, cout, amax_y=None):
    """ConvTranspose2d(2, 2) forward on the fp16 matrix cores (csrc/gemm_h2s.hip): contract of convt_x3_fwd + the amax slots."""
    require_cuda(x, w_h2, y, amax_x, amax_w)
    B, H, W, Cin = x.shape
    with _Timed('convt_fwd_h2', 8.0 * B * H * W * Cin * cout, 4.0 * B * H * W * (Cin + 4 * cout)):
        check(_prep().pnnp_convt2x2_h2_fwd_f32(ptr(x), Cin, ptr(amax_x), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(y), ptr(amax_y), B, H, W, cout, stream()), 'convt_h2_fwd')
    return y


def convt_h2_bwd_data(g, amax_g, w_h2_dgrad, amax_w, dx, mask=None, mode=0, amax_dx=None, bits=None):
    """``bits``: the act' mask as the sign bits a 3x3 fp16x2 forward kernel stored for the layer's input map (then ``mask`` is not read)."""
    require_cuda(g, w_h2_dgrad, dx, amax_g, amax_w, bits)
    B, H, W, Cin = dx.shape
    with _Timed('convt_dgrad_h2', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        if bits is not None and mode:
            check(_prep().pnnp_convt2x2_h2_bwd_data_bits_f32(ptr(g), g.shape[3], ptr(amax_g), ptr(w_h2_dgrad), ptr(amax_w), ptr(dx), Cin, ptr(bits), mode, ptr(amax_dx),
                                                             B, H, W, stream()), 'convt_h2_bwd_data (bit masks)')
            return
        check(_prep().pnnp_convt2x2_h2_bwd_data_f32(ptr(g), g.shape[3], ptr(amax_g), ptr(w_h2_dgrad), ptr(amax_w), ptr(dx), Cin, ptr(mask), mode, ptr(amax_dx),
                                                    B, H, W, stream()), 'convt_h2_bwd_data')


def conv1x1_h2_fwd(x1, amax_x1, x2, amax_x2, w_h2, amax_w, bias, y, cout, act, residual=None, amax_y=None):
    require_cuda(x1, x2, w_h2, y, amax_x1, amax_w)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv1_fwd_h2', 2.0 * B * H * W * cout * (C1 + C2), 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv1x1_h2_fwd_f32(ptr(x1), C1, ptr(amax_x1), ptr(x2), C2, ptr(amax_x2), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(residual), ptr(y), ptr(amax_y),
                                              B, H, W, cout, act, stream()), 'conv1x1_h2_fwd')
    return y


def conv1x1_h2_bwd_data(g, amax_g, w_h2_dgrad, amax_w, dx1, mask1=None, mode1=0, accum1=0, amax_dx1=None, dx2=None, mask2=None, mode2=0, accum2=0):
    require_cuda(g, w_h2_dgrad, dx1, dx2, amax_g, amax_w)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]; C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv1_dgrad_h2', 2.0 * B * H * W * Cout * (C1 + C2), 4.0 * B * H * W * (C1 + C2 + Cout)):
        check(_prep().pnnp_conv1x1_h2_bwd_data_f32(ptr(g), Cout, ptr(amax_g), ptr(w_h2_dgrad), ptr(amax_w), ptr(dx1), C1, ptr(mask1), mode1, int(accum1), ptr(amax_dx1),
                                                   ptr(dx2), C2, ptr(mask2), mode2, int(accum2), B, H, W, stream()), 'conv1x1_h2_bwd_data')


def conv_s2_h2_fwd(x, amax_x, w_h2, amax_w, bias, y, cout, act, amax_y=None):
    require_cuda(x, w_h2, y, amax_x, amax_w)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_fwd_h2', 2.0 * B * (H // 2) * (W // 2) * cout * Cin * 9):
        check(_prep().pnnp_conv3x3s2_h2_fwd_f32(ptr(x), Cin, ptr(amax_x), ptr(w_h2), ptr(amax_w), ptr(bias), ptr(y), ptr(amax_y), B, H, W, cout, act, stream()), 'conv_s2_h2_fwd')
    return y


def conv_s2_h2_bwd_data(g, amax_g, w_h2_s2dgrad, amax_w, dx, mask=None, mode=0, accum=0, amax_dx=None):
    require_cuda(g, w_h2_s2dgrad, dx, amax_g, amax_w)
    B, H, W, Cin = dx.shape
    with _Timed('conv9s2_dgrad_h2', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_h2_bwd_data_f32(ptr(g), g.shape[3], ptr(amax_g), ptr(w_h2_s2dgrad), ptr(amax_w), ptr(dx), Cin, ptr(mask), mode, int(accum), ptr(amax_dx),
                                                     B, H, W, stream()), 'conv_s2_h2_bwd_data')


def conv1x1_x3_fwd(x1, x2, w_x3, bias, y, cout, act, residual=None):
    require_cuda(x1, x2, w_x3, y)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv1_fwd_x3', 2.0 * B * H * W * cout * (C1 + C2), 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv1x1_x3_fwd_f32(ptr(x1), C1, ptr(x2), C2, ptr(w_x3), ptr(bias), ptr(residual), ptr(y), B, H, W, cout, act, stream()),
              'conv1x1_x3_fwd')
    return y


def conv1x1_x3_bwd_data(g, w_x3_dgrad, dx1, mask1=None, mode1=0, accum1=0, dx2=None, mask2=None, mode2=0, accum2=0):
    require_cuda(g, w_x3_dgrad, dx1)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]
    C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv1_dgrad_x3', 2.0 * B * H * W * Cout * (C1 + C2), 4.0 * B * H * W * (C1 + C2 + Cout)):
        check(_prep().pnnp_conv1x1_x3_bwd_data_f32(ptr(g), Cout, ptr(w_x3_dgrad), ptr(dx1), C1, ptr(mask1), mode1, accum1,
                                                   ptr(dx2), C2, ptr(mask2), mode2, accum2, B, H, W, stream()), 'conv1x1_x3_bwd_data')


def conv_s2_x3_fwd(x, w_x3, bias, y, cout, act=0, amax_y=None):
    require_cuda(x, w_x3, y)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_fwd_x3', 2.0 * B * (H // 2) * (W // 2) * cout * Cin * 9):
        check(_prep().pnnp_conv3x3s2_x3_fwd_amax_f32(ptr(x), Cin, ptr(w_x3), ptr(bias), ptr(y), ptr(amax_y), B, H, W, cout, act, stream()), 'conv3x3s2_x3_fwd')
    return y


def conv_s2_x3_bwd_data(g, w_x3_s2dgrad, dx, mask=None, mode=0, accum=0, amax_dx=None):
    require_cuda(g, w_x3_s2dgrad, dx)
    B, H, W, Cin = dx.shape
    with _Timed('conv9s2_dgrad_x3', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_x3_bwd_data_amax_f32(ptr(g), g.shape[3], ptr(w_x3_s2dgrad), ptr(dx), Cin, ptr(mask), mode, accum, ptr(amax_dx),
                                                          B, H, W, stream()), 'conv3x3s2_x3_bwd_data')


def x3_wgrad_supported(H, W, cout, c1, c2=0):
    return bool(_prep().pnnp_x3_wgrad_supported(int(H), int(W), int(cout), int(c1), int(c2)))


def x3_wgrad_workspace_floats(B, H, W, cout, cin):
    return int(_prep().pnnp_x3_wgrad_workspace_floats(B, H, W, cout, cin))


def conv_x3_bwd_weight(g, cout, x1, c1, x2, dW, dbias, workspace, accumulate=0):
    """dW [cout][c1+c2][3][3] (+ dbias) on the bf16 matrix cores, fp32 operands split in three (same contract as conv_bwd_weight, taps=9)."""
    require_cuda(g, x1, dW, workspace)
    B, H, W, gcs = g.shape
    c2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_wgrad_x3', 2.0 * B * H * W * cout * (c1 + c2) * 9, 4.0 * B * H * W * (gcs + x1.shape[3] + c2)):
        check(_prep().pnnp_conv3x3_x3_bwd_weight_f32(ptr(g), gcs, cout, ptr(x1), x1.shape[3], c1, ptr(x2), c2, c2, ptr(dW), ptr(dbias),
                                                     B, H, W, int(accumulate), ptr(workspace), C.c_int64(workspace.numel()), stream()),
              'conv_x3_bwd_weight')


def conv_h2_bwd_weight(g, amax_g, cout, x1, amax_x1, c1, x2, amax_x2, dW, dbias, workspace, accumulate=0):
    """dW [cout][c1+c2][3][3] (+ dbias) on the fp16 matrix cores, both operands split in two scaled pieces (contract of conv_x3_bwd_weight + slots)."""
    require_cuda(g, x1, dW, workspace, amax_g, amax_x1)
    B, H, W, gcs = g.shape
    c2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_wgrad_h2', 2.0 * B * H * W * cout * (c1 + c2) * 9, 4.0 * B * H * W * (gcs + x1.shape[3] + c2)):
        check(_prep().pnnp_conv3x3_h2_bwd_weight_f32(ptr(g), gcs, cout, ptr(amax_g), ptr(x1), x1.shape[3], c1, ptr(amax_x1), ptr(x2), c2, c2, ptr(amax_x2),
                                                     ptr(dW), ptr(dbias), B, H, W, int(accumulate), ptr(workspace), C.c_int64(workspace.numel()), stream()),
              'conv_h2_bwd_weight')


def wino_supported(K, N):
    return bool(_prep().pnnp_wino_supported(int(K), int(N)))


def pack_conv_weight_wino(w, fwd, dgrad):
    co, ci = w.shape[:2]
    check(_prep().pnnp_pack_conv_weight_wino_f32(ptr(w), ptr(fwd), ptr(dgrad), co, ci, stream()), 'pack_conv_weight_wino')


def conv_wino_fwd(x1, x2, u_fwd, bias, y, cout, act, residual=None):
    """3x3 / stride 1 / pad 1 forward through the Winograd F(2x2,3x3) kernel (same contract as conv_fwd, taps=9)."""
    require_cuda(x1, x2, u_fwd, y)
    B, H, W, C1 = x1.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_fwd_wino', 2.0 * B * H * W * cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + cout)):
        check(_prep().pnnp_conv3x3_wino_fwd_f32(ptr(x1), C1, ptr(x2), C2, ptr(u_fwd), ptr(bias), ptr(residual), ptr(y), B, H, W, cout, act,
                                                stream()), 'conv_wino_fwd')
    return y


def conv_wino_bwd_data(g, u_dgrad, dx1, mask1=None, mode1=0, accum1=0, dx2=None, mask2=None, mode2=0, accum2=0):
    require_cuda(g, u_dgrad, dx1)
    B, H, W, Cout = g.shape
    C1 = dx1.shape[3]
    C2 = dx2.shape[3] if dx2 is not None else 0
    with _Timed('conv9_dgrad_wino', 2.0 * B * H * W * Cout * (C1 + C2) * 9, 4.0 * B * H * W * (C1 + C2 + Cout)):
        check(_prep().pnnp_conv3x3_wino_bwd_data_f32(ptr(g), Cout, ptr(u_dgrad), ptr(dx1), C1, ptr(mask1), mode1, accum1,
                                                     ptr(dx2), C2, ptr(mask2), mode2, accum2, B, H, W, stream()), 'conv_wino_bwd_data')


def wino_wgrad_supported(H, W, cout, c1, c2=0):
    return bool(_prep().pnnp_wino_wgrad_supported(int(H), int(W), int(cout), int(c1), int(c2)))


def wino_wgrad_workspace_floats(B, H, W, cout, cin):
    return int(_prep().pnnp_wino_wgrad_workspace_floats(B, H, W, cout, cin))


def conv_wino_bwd_weight(g, cout, x1, c1, x2, dW, dbias, workspace, accumulate=0):
    """dW [cout][c1+c2][3][3] (+ dbias) through the Winograd backward-weight kernel (same contract as conv_bwd_weight, taps=9)."""
    require_cuda(g, x1, dW, workspace)
    B, H, W, gcs = g.shape
    c2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv9_wgrad_wino', 2.0 * B * H * W * cout * (c1 + c2) * 9, 4.0 * B * H * W * (gcs + x1.shape[3] + c2)):
        check(_prep().pnnp_conv3x3_wino_bwd_weight_f32(ptr(g), gcs, cout, ptr(x1), x1.shape[3], c1, ptr(x2), c2, c2, ptr(dW), ptr(dbias),
                                                       B, H, W, int(accumulate), ptr(workspace), C.c_int64(workspace.numel()), stream()),
              'conv_wino_bwd_weight')


def conv_wino_bwd_data_res(g, u_dgrad, dx, addsrc, mask=None, mode=0):
    """dx = (conv_bwd_data(g) + addsrc) * act'(mask) through the Winograd kernel."""
    require_cuda(g, u_dgrad, dx, addsrc)
    B, H, W, Cout = g.shape
    C1 = dx.shape[3]
    with _Timed('conv9_dgrad_wino', 2.0 * B * H * W * Cout * C1 * 9, 4.0 * B * H * W * (2 * C1 + Cout)):
        check(_prep().pnnp_conv3x3_wino_bwd_data_res_f32(ptr(g), Cout, ptr(u_dgrad), ptr(dx), C1, ptr(addsrc), ptr(mask), mode,
                                                         B, H, W, stream()), 'conv_wino_bwd_data_res')


def conv_bwd_data_res(g, w_dgrad, dx, addsrc, mask=None, mode=0, taps=9):
    """dx = (conv_bwd_data(g) + addsrc) * act'(mask): backward through an identity shortcut."""
    require_cuda(g, w_dgrad, dx, addsrc)
    B, H, W, Cout = g.shape
    C1 = dx.shape[3]
    with _Timed('conv%d_dgrad' % taps, 2.0 * B * H * W * Cout * C1 * taps, 4.0 * B * H * W * (2 * C1 + Cout)):
        check(_prep().pnnp_conv_bwd_data_res_f32(ptr(g), Cout, ptr(w_dgrad), ptr(dx), C1, ptr(addsrc), ptr(mask), mode,
                                                 B, H, W, taps, stream()), 'conv_bwd_data_res')


def wgrad_workspace_floats(B, H, W, M, N, taps):
    return int(_prep().pnnp_wgrad_workspace_floats(B, H, W, M, N, taps))


def conv_bwd_weight(g, cout, x1, c1, x2, dW, dbias, taps, ws, accumulate=0):
    """g [B,H,W,>=cout] (first ``cout`` channels used), x1 [B,H,W,>=c1], x2 [B,H,W,C2] or None."""
    require_cuda(g, x1, dW, ws)
    B, H, W, gcs = g.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv%d_wgrad' % taps, 2.0 * B * H * W * cout * (c1 + C2) * taps, 4.0 * B * H * W * (c1 + C2 + cout)):
        check(_prep().pnnp_conv_bwd_weight_f32(ptr(g), gcs, cout, ptr(x1), x1.shape[3], c1, ptr(x2), C2, C2, ptr(dW), ptr(dbias),
                                               B, H, W, taps, accumulate, ptr(ws), _i64(ws.numel()), stream()), 'conv_bwd_weight')


def convt_fwd(x, w_packed, bias, y, cout):
    require_cuda(x, w_packed, y)
    B, H, W, Cin = x.shape
    with _Timed('convt_fwd', 8.0 * B * H * W * Cin * cout, 4.0 * B * H * W * (Cin + 4 * cout)):
        check(_prep().pnnp_convt2x2_fwd_f32(ptr(x), Cin, ptr(w_packed), ptr(bias), ptr(y), B, H, W, cout, stream()), 'convt_fwd')
    return y


def convt_bwd_data(g, w_dgrad, dx, mask=None, mode=0):
    require_cuda(g, w_dgrad, dx)
    B, H, W, Cin = dx.shape
    with _Timed('convt_dgrad', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        check(_prep().pnnp_convt2x2_bwd_data_f32(ptr(g), g.shape[3], ptr(w_dgrad), ptr(dx), Cin, ptr(mask), mode, B, H, W,
                                                 stream()), 'convt_bwd_data')


def set_persistent_split(n):
    """Workgroups per CU of the persistent forward / backward-data convolution kernels (include/pnnp_hip.h: pnnp_set_persistent_split):
    1 = one per CU with a static share (default, fastest alone), 4 = quarter shares handed out by the dispatcher (safe beside a
    collective kernel that occupies CUs: tests/test_gpu_overlap.py)."""
    L = _prep()
    L.pnnp_set_persistent_split.argtypes = [C.c_int]
    L.pnnp_set_persistent_split.restype = None
    L.pnnp_set_persistent_split(int(n))


def get_persistent_split():
    return int(_prep().pnnp_get_persistent_split())


X3G_PW, X3G_CT, X3G_S2 = 0, 1, 2        # geometry kinds of csrc/wgrad_x3g.hip: Conv2d 1x1, ConvTranspose2d 2x2 s2, Conv2d 3x3 s2


def x3g_wgrad_supported(kind, M, N):
    """Does the bf16x3 backward-weight kernel of the pointwise / strided layers have a tile configuration for (M, N)?"""
    return bool(_prep().pnnp_x3g_wgrad_supported(int(kind), int(M), int(N)))


def x3g_wgrad_workspace_floats(kind, B, UH, UW, M, N):
    return int(_prep().pnnp_x3g_wgrad_workspace_floats(int(kind), int(B), int(UH), int(UW), int(M), int(N)))


def convt_x3_bwd_weight(x, g, dW, ws, accumulate=0, dbias=None):
    """convt_bwd_weight on the bf16 matrix cores (fp32 operands split in three, csrc/wgrad_x3g.hip): Cin = M, Cout = N."""
    require_cuda(x, g, dW, ws, dbias)
    B, H, W, Cin = x.shape
    with _Timed('convt_wgrad_x3', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        check(_prep().pnnp_convt2x2_x3_bwd_weight_f32(ptr(x), Cin, ptr(g), g.shape[3], ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                      ptr(ws), _i64(ws.numel()), stream()), 'convt_x3_bwd_weight')


def conv_s2_x3_bwd_weight(g, x, dW, dbias, ws, accumulate=0):
    require_cuda(g, x, dW, ws)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_wgrad_x3', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_x3_bwd_weight_f32(ptr(g), g.shape[3], ptr(x), Cin, ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                       ptr(ws), _i64(ws.numel()), stream()), 'conv3x3s2_x3_bwd_weight')


def conv1x1_x3_bwd_weight(g, cout, x1, c1, x2, dW, dbias, ws, accumulate=0):
    """conv_bwd_weight(taps=1) on the bf16 matrix cores."""
    require_cuda(g, x1, dW, ws)
    B, H, W, gcs = g.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv1_wgrad_x3', 2.0 * B * H * W * cout * (c1 + C2), 4.0 * B * H * W * (c1 + C2 + cout)):
        check(_prep().pnnp_conv1x1_x3_bwd_weight_f32(ptr(g), gcs, cout, ptr(x1), x1.shape[3], c1, ptr(x2), C2, C2, ptr(dW), ptr(dbias),
                                                     B, H, W, accumulate, ptr(ws), _i64(ws.numel()), stream()), 'conv1x1_x3_bwd_weight')


def h2g_wgrad_supported(kind, M, N):
    return bool(_prep().pnnp_h2g_wgrad_supported(int(kind), int(M), int(N)))


def h2g_wgrad_workspace_floats(kind, B, UH, UW, M, N):
    return int(_prep().pnnp_h2g_wgrad_workspace_floats(int(kind), int(B), int(UH), int(UW), int(M), int(N)))


def convt_h2_bwd_weight(x, amax_x, g, amax_g, dW, ws, accumulate=0, dbias=None):
    """convt_x3_bwd_weight on the fp16x2 scheme (csrc/wgrad_h2g.hip): same shapes and workspace, + the operands' amax slots."""
    require_cuda(x, g, dW, ws, dbias, amax_x, amax_g)
    B, H, W, Cin = x.shape
    with _Timed('convt_wgrad_h2', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        check(_prep().pnnp_convt2x2_h2_bwd_weight_f32(ptr(x), Cin, ptr(amax_x), ptr(g), g.shape[3], ptr(amax_g), ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                      ptr(ws), _i64(ws.numel()), stream()), 'convt_h2_bwd_weight')


def conv_s2_h2_bwd_weight(g, amax_g, x, amax_x, dW, dbias, ws, accumulate=0):
    require_cuda(g, x, dW, ws, amax_g, amax_x)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_wgrad_h2', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_h2_bwd_weight_f32(ptr(g), g.shape[3], ptr(amax_g), ptr(x), Cin, ptr(amax_x), ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                       ptr(ws), _i64(ws.numel()), stream()), 'conv3x3s2_h2_bwd_weight')


def conv1x1_h2_bwd_weight(g, amax_g, cout, x1, amax_x1, c1, x2, amax_x2, dW, dbias, ws, accumulate=0):
    require_cuda(g, x1, dW, ws, amax_g, amax_x1)
    B, H, W, gcs = g.shape
    C2 = x2.shape[3] if x2 is not None else 0
    with _Timed('conv1_wgrad_h2', 2.0 * B * H * W * cout * (c1 + C2), 4.0 * B * H * W * (c1 + C2 + cout)):
        check(_prep().pnnp_conv1x1_h2_bwd_weight_f32(ptr(g), gcs, cout, ptr(amax_g), ptr(x1), x1.shape[3], c1, ptr(amax_x1), ptr(x2), C2, C2, ptr(amax_x2), ptr(dW), ptr(dbias),
                                                     B, H, W, accumulate, ptr(ws), _i64(ws.numel()), stream()), 'conv1x1_h2_bwd_weight')


def convt_bwd_weight(x, g, dW, ws, accumulate=0, dbias=None):
    """dW (and dbias = the channel sums of g, gathered by the same kernel while g is staged) of ConvTranspose2d(k2, s2)."""
    require_cuda(x, g, dW, ws, dbias)
    B, H, W, Cin = x.shape
    with _Timed('convt_wgrad', 8.0 * B * H * W * Cin * g.shape[3], 4.0 * B * H * W * (Cin + 4 * g.shape[3])):
        check(_prep().pnnp_convt2x2_bwd_weight_f32(ptr(x), Cin, ptr(g), g.shape[3], ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                   ptr(ws), _i64(ws.numel()), stream()), 'convt_bwd_weight')


def conv_s2_fwd(x, w_packed, bias, y, cout, act=0):
    """Conv2d 3x3 stride 2 pad 1: x [B,H,W,Cin] -> y [B,H/2,W/2,cout]."""
    require_cuda(x, w_packed, y)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_fwd', 2.0 * B * (H // 2) * (W // 2) * cout * Cin * 9):
        check(_prep().pnnp_conv3x3s2_fwd_f32(ptr(x), Cin, ptr(w_packed), ptr(bias), ptr(y), B, H, W, cout, act, stream()),
              'conv3x3s2_fwd')
    return y


def pack_conv_s2_dgrad(w, dst):
    co, ci = w.shape[:2]
    check(_prep().pnnp_pack_conv3x3s2_dgrad_f32(ptr(w), ptr(dst), co, ci, stream()), 'pack_conv3x3s2_dgrad')


def conv_s2_bwd_data(g, w_s2dgrad, dx, mask=None, mode=0, accum=0):
    require_cuda(g, w_s2dgrad, dx)
    B, H, W, Cin = dx.shape
    with _Timed('conv9s2_dgrad', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_bwd_data_f32(ptr(g), g.shape[3], ptr(w_s2dgrad), ptr(dx), Cin, ptr(mask), mode, accum,
                                                  B, H, W, stream()), 'conv3x3s2_bwd_data')


def conv_s2_bwd_weight(g, x, dW, dbias, ws, accumulate=0):
    require_cuda(g, x, dW, ws)
    B, H, W, Cin = x.shape
    with _Timed('conv9s2_wgrad', 2.0 * B * (H // 2) * (W // 2) * g.shape[3] * Cin * 9):
        check(_prep().pnnp_conv3x3s2_bwd_weight_f32(ptr(g), g.shape[3], ptr(x), Cin, ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                    ptr(ws), _i64(ws.numel()), stream()), 'conv3x3s2_bwd_weight')


# ---- the thin ends of the networks (csrc/thin.hip): the 1x1 head and the first layer's backward-weight
def head_supported(cin, cout, npix):
    return bool(_prep().pnnp_head_supported(int(cin), int(cout), _i64(npix)))


def head_bwd_workspace_floats(cin):
    return int(_prep().pnnp_head_bwd_workspace_floats(int(cin)))


def head_fwd(x, weight, bias, out, residual=None):
    """out NCHW [B,cout,H,W] = conv1x1(x NHWC [B,H,W,>=cin]; weight [cout,cin,1,1], bias) (+ residual NCHW)."""
    require_cuda(x, weight, bias, out)
    B, H, W, xcs = x.shape
    cout, cin = weight.shape[0], weight.shape[1]
    with _Timed('head_fwd', 2.0 * B * H * W * cin * cout, 4.0 * B * H * W * (cin + cout)):
        check(_prep().pnnp_head_fwd_f32(ptr(x), xcs, cin, ptr(weight), ptr(bias), ptr(residual), ptr(out), B, H, W, cout, stream()), 'head_fwd')
    return out


def head_bwd(g, x, weight, gx, dW, dbias, ws, mode=0, accumulate=0, amax_gx=None):
    """Backward of the 1x1 head in one pass: gx [B,H,W,>=cin] = (g W) * act'(x), dW [cout,cin,1,1] and dbias (+)=.
    g [B,H,W,gcs>=4] (first cout channels), x [B,H,W,>=cin] the head's input (an activation output when mode != 0)."""
    require_cuda(g, x, weight, gx, dW, ws)
    B, H, W, gcs = g.shape
    cout, cin = weight.shape[0], weight.shape[1]
    with _Timed('head_bwd', 4.0 * B * H * W * cin * cout, 4.0 * B * H * W * (2 * cin + cout)):
        check(_prep().pnnp_head_bwd_amax_f32(ptr(g), gcs, ptr(x), x.shape[3], cin, ptr(weight), ptr(gx), gx.shape[3], mode, ptr(dW), ptr(dbias),
                                             B, H, W, cout, accumulate, ptr(ws), _i64(ws.numel()), ptr(amax_gx), stream()), 'head_bwd')


def first_wgrad_supported(cin, cout, H, W):
    return bool(_prep().pnnp_first_wgrad_supported(int(cin), int(cout), int(H), int(W)))


def first_wgrad_workspace_floats(cout):
    return int(_prep().pnnp_first_wgrad_workspace_floats(int(cout)))


def first_fwd(x, weight, bias, y, act, amax_y=None):
    """y [B,H,W,>=cout] = act(conv3x3(x [B,H,W,xcs>=4], zero in channels cin..3; weight [cout,cin,3,3]) + bias) on the streaming kernel.
    ``amax_y``: an amax slot of the fp16x2 family (max |y| is raised into it)."""
    require_cuda(x, weight, y)
    B, H, W, xcs = x.shape
    cout, cin = weight.shape[0], weight.shape[1]
    with _Timed('first_fwd', 2.0 * B * H * W * cout * cin * 9, 4.0 * B * H * W * (cin + cout)):
        check(_prep().pnnp_first_fwd_amax_f32(ptr(x), xcs, cin, ptr(weight), ptr(bias), ptr(y), y.shape[3], B, H, W, cout, act, ptr(amax_y), stream()), 'first_fwd')
    return y


def first_bwd_weight(g, cout, x, cin, dW, dbias, ws, accumulate=0):
    """dW [cout,cin,3,3], dbias (+)= for the first 3x3 convolution: x [B,H,W,xcs>=4] zero in channels cin..3, g [B,H,W,>=cout]."""
    require_cuda(g, x, dW, ws)
    B, H, W, gcs = g.shape
    with _Timed('first_wgrad', 2.0 * B * H * W * cout * cin * 9, 4.0 * B * H * W * (cin + cout)):
        check(_prep().pnnp_first_bwd_weight_f32(ptr(g), gcs, cout, ptr(x), x.shape[3], cin, ptr(dW), ptr(dbias), B, H, W, accumulate,
                                                ptr(ws), _i64(ws.numel()), stream()), 'first_bwd_weight')


def maxpool_fwd(x, y, codes=None):
    """MaxPool2d(2); with ``codes`` (uint8 [B,H/2,W/2,C]) it also records argmax + signs for maxpool_bwd(codes=...)."""
    require_cuda(x, y, codes)
    B, H, W, Cc = x.shape
    if codes is not None:
        check(_prep().pnnp_maxpool2_fwd_codes_f32(ptr(x), ptr(y), ptr(codes), B, H, W, Cc, stream()), 'maxpool_fwd_codes')
    else:
        check(_prep().pnnp_maxpool2_fwd_f32(ptr(x), ptr(y), B, H, W, Cc, stream()), 'maxpool_fwd')
    return y


def maxpool_bwd(x, gy, gx, act_mode, accumulate, codes=None, amax_gx=None):
    """gx (+)= routed(gy) * act'(x); with the forward pass's ``codes`` the full-resolution x is not read again.  ``amax_gx`` (an amax slot of
    the fp16x2 family): fused into the codes kernel, a kernel of its own behind the other."""
    require_cuda(x, gy, gx, codes)
    B, H, W, Cc = x.shape
    if codes is not None:
        check(_prep().pnnp_maxpool2_bwd_codes_amax_f32(ptr(codes), ptr(gy), ptr(gx), B, H, W, Cc, act_mode, accumulate, ptr(amax_gx), stream()), 'maxpool_bwd_codes')
    else:
        check(_prep().pnnp_maxpool2_bwd_f32(ptr(x), ptr(gy), ptr(gx), B, H, W, Cc, act_mode, accumulate, stream()), 'maxpool_bwd')
        if amax_gx is not None:
            amax(gx, amax_gx)


def nchw_to_nhwc(src, dst, cp, reflect_pad=0, amax=None):
    """NCHW -> zero-padded-channel NHWC; ``reflect_pad`` > 0: dst is the frame reflect-padded by that many pixels on every side.
    ``amax``: an amax slot of the fp16x2 family (max |element| is raised into it)."""
    require_cuda(src, dst)
    B, Cc, H, W = src.shape
    check(_prep().pnnp_nchw_to_nhwc_reflect_amax_f32(ptr(src), ptr(dst), B, Cc, H, W, cp, int(reflect_pad), ptr(amax), stream()), 'nchw_to_nhwc')
    return dst


def nhwc_to_nchw(src, dst, residual=None):
    require_cuda(src, dst, residual)
    B, Cc, H, W = dst.shape
    check(_prep().pnnp_nhwc_to_nchw_f32(ptr(src), ptr(residual), ptr(dst), B, Cc, H, W, src.shape[3], stream()), 'nhwc_to_nchw')
    return dst


def channel_sum(x, out, ws, accumulate=0):
    require_cuda(x, out, ws)
    Cc = x.shape[-1]
    check(_prep().pnnp_channel_sum_f32(ptr(x), ptr(out), _i64(x.numel() // Cc), Cc, accumulate, ptr(ws), stream()), 'channel_sum')


def l1_clamp_loss(pred, hr, grad_nhwc, loss_out, ws, scale=None, clamp_target=False, grad_weight=1.0):
    """``scale`` [B] (or None): the `ori` branch of the train loop, pred * ratio before the loss (trainer_SID.py:97-99).
    ``clamp_target``: clamp ``hr`` to [0,1] inside the kernel (preprocess's imgs_hr.clamp(0,1) under dst.clip, trainer_SID.py:485).
    ``grad_weight`` multiplies dL/dpred only (uneven data-parallel shards of a global batch)."""
    require_cuda(pred, hr, loss_out, ws, scale)
    B, Cc, H, W = pred.shape
    cp = grad_nhwc.shape[3] if grad_nhwc is not None else 0
    check(_prep().pnnp_l1_clamp_loss_w_f32(ptr(pred), ptr(hr), ptr(scale), ptr(grad_nhwc), ptr(loss_out), B, Cc, H, W, cp, ptr(ws),
                                           int(bool(clamp_target)), C.c_float(grad_weight), stream()), 'l1_clamp_loss')


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    require_cuda(p, g, m, v)
    check(_prep().pnnp_adam_step_f32(ptr(p), ptr(g), ptr(m), ptr(v), _i64(p.numel()), C.c_float(lr), C.c_float(beta1),
                                     C.c_float(beta2), C.c_float(eps), int(step), C.c_float(grad_scale), stream()), 'adam_step')
