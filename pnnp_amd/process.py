"""Noise-parameter samplers and the physics noise sampler, with the reference's names
and signatures (data_process/process.py:215-412, 591-673).

Host side (scalar numpy RNG, same draw order as the reference so seeded runs agree):
``get_camera_noisy_params``, ``get_specific_noise_params``, ``sample_params_max``,
``sample_params``.  Device side: ``generate_noisy_obs`` / ``generate_noisy_torch`` run
the fused HIP sampler (csrc/noise.hip); there is no CPU implementation here.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

Dual_ISO_Cameras = ['SonyA7S2']
HALF_CLIP = 2            # process.py:19 -- `clip: 2` in the YAMLs means clamp(-inf, 1)
NPARAM = 16

# ---------------------------------------------------------------------------- tables
# Regression tables (process.py:215-255), one row per camera:
# Kmin Kmax lam q_bits wp bl | sigTL k b sig | sigR k b sig | sigGs k b sig | [sigRead k b sig | uRead k b sig]
_REGRESSION = """
NikonD850        1.2      2.4828  -0.26  14 16383 512  0.906   -0.6754  0.035165  0.8322  -2.3326  0.301333  0.8322  -0.1754 0.035165
IMX686           -0.19118 2.16820 0.102  10 1023  64   0.85187 0.07991  0.02921   0.87611 -2.11455 0.03274   0.85187 0.67991 0.02921
SonyA7S2_lowISO  -1.67214 0.42228 -0.026 14 16383 512  0.74043 0.86182  0.00712   0.78782 -0.34227 0.02832   0.82966 1.49343 0.00359  0.82879 1.50601 0.00362 0.01472 0.01129 0.00034
SonyA7S2_highISO 0.64567  2.51606 -0.025 14 16383 512  0.74901 -0.12348 0.00638   0.62945 -1.51040 0.02609   0.82878 0.44162 0.00153  0.82645 0.45061 0.00156 0.00385 0.00674 0.00039
CRVD             1.31339  3.95448 0.015  12 4095  240  0.95495 0.01618  0.00790   0.93368 -2.19692 0.02473   0.95387 0.01552 0.00855
"""

# Per-ISO calibration of the SonyA7S2 (process.py:257-289), columns:
# iso Kmax lam sigGs sigGssig sigTL sigTLsig sigR sigRsig biassig      (bias 0, q 2^-14, wp 16383, bl 512)
_SONY_ISO = """
50 0.047815 0.1474653 1.0164667 0.005272454 0.70727646 0.004360543 0.13997398 0.0064381803 0.010093017
64 0.0612032 0.13243394 1.0509665 0.008081373 0.71535635 0.0056863446 0.14346549 0.006400559 0.008690166
80 0.076504 0.1121489 1.180899 0.011333668 0.7799473 0.009347968 0.19540153 0.008197397 0.0107246125
100 0.09563 0.14875287 1.0067395 0.0033682834 0.70181876 0.0037532174 0.1391465 0.006530218 0.007235429
125 0.1195375 0.12904578 1.0279676 0.007364685 0.6961967 0.0048687346 0.14485553 0.006731584 0.008026363
160 0.153008 0.094135 1.1293099 0.008340453 0.7258587 0.008032158 0.19755602 0.0082754735 0.0101351
200 0.19126 0.07902429 1.2926387 0.012171176 0.8117464 0.010250768 0.22815849 0.010726711 0.011413908
250 0.239075 0.051688068 1.4345995 0.01606571 0.8630922 0.013844714 0.26271912 0.0130637 0.013569083
320 0.306016 0.040700804 1.7481371 0.019626873 1.0334468 0.017629284 0.3097104 0.016202712 0.017825918
400 0.38252 0.0222538 2.0595572 0.024872316 1.1816813 0.02505812 0.36209714 0.01994737 0.021005306
500 0.47815 -0.0031342343 2.3956928 0.030144656 1.31772 0.028629242 0.42528257 0.025104137 0.02981831
640 0.612032 0.002566592 2.9662898 0.045661453 1.6474211 0.04671843 0.48839623 0.031589635 0.10000693
800 0.76504 -0.008199721 3.5475867 0.052318197 1.9346539 0.046128694 0.5723769 0.037824076 0.025339302
1000 0.9563 -0.021061005 4.2727833 0.06972333 2.2795107 0.059203167 0.6845563 0.04879781 0.027911892
1250 1.195375 -0.032423194 5.177596 0.092677385 2.708437 0.07622563 0.8177013 0.06162229 0.03293372
1600 1.53008 -0.0441045 6.29925 0.1153261 3.2283993 0.09118158 0.988786 0.078567736 0.03877672
2000 1.9126 -0.012963797 2.653871 0.015890995 1.4356787 0.02178686 0.33124214 0.018801652 0.01570677
2500 2.39075 -0.027097283 3.200225 0.019307792 1.6897862 0.025873765 0.38264316 0.023769397 0.018728448
3200 3.06016 -0.034863412 3.9193838 0.02649232 2.0417721 0.032873377 0.44543457 0.030114045 0.021355819
4000 3.8252 -0.043700505 4.8015847 0.03781628 2.4629273 0.042401053 0.52347374 0.03929801 0.026152484
5000 4.7815 -0.053150143 5.8995814 0.0625814 2.9761007 0.061326735 0.6190265 0.05335372 0.058574405
6400 6.12032 -0.07517104 7.1163535 0.08435366 3.4502964 0.08226275 0.7218788 0.0642334 0.059074216
8000 7.6504 -0.08208357 8.916516 0.12763213 4.269624 0.13381928 0.87760293 0.07389065 0.084842026
10000 9.563 -0.073289566 11.291476 0.1639773 5.495318 0.16279395 1.0522343 0.094359785 0.107438326
12800 12.24064 -0.06495205 14.245901 0.17283991 7.038261 0.18822834 1.2749791 0.120479785 0.0944684
16000 15.3008 -0.060692135 17.833515 0.19809262 8.877547 0.23338738 1.5559287 0.15791349 0.09725099
20000 19.126 -0.060213074 22.084776 0.21820943 11.002351 0.28806436 1.8810822 0.18937257 0.4984733
25600 24.48128 -0.09089118 25.853043 0.35371417 12.175712 0.4215717 2.2760193 0.2609267 0.37568903
"""


def _build_regression():
    tab = {}
    for line in _REGRESSION.strip().splitlines():
        f = line.split()
        v = [float(x) for x in f[1:]]
        d = {'Kmin': v[0], 'Kmax': v[1], 'lam': v[2], 'q': 1 / (2 ** int(v[3])), 'wp': int(v[4]), 'bl': int(v[5]),
             'sigTLk': v[6], 'sigTLb': v[7], 'sigTLsig': v[8], 'sigRk': v[9], 'sigRb': v[10], 'sigRsig': v[11],
             'sigGsk': v[12], 'sigGsb': v[13], 'sigGssig': v[14]}
        if len(v) > 15:
            d.update({'sigReadk': v[15], 'sigReadb': v[16], 'sigReadsig': v[17],
                      'uReadk': v[18], 'uReadb': v[19], 'uReadsig': v[20]})
        tab[f[0]] = d
    return tab


def _build_specific():
    sony = {}
    for line in _SONY_ISO.strip().splitlines():
        f = line.split()
        v = [float(x) for x in f[1:]]
        sony[f[0]] = {'Kmax': v[0], 'lam': v[1], 'sigGs': v[2], 'sigGssig': v[3], 'sigTL': v[4], 'sigTLsig': v[5],
                      'sigR': v[6], 'sigRsig': v[7], 'bias': 0, 'biassig': v[8], 'q': 6.103515625e-05,
                      'wp': 16383, 'bl': 512}
    imx = {   # process.py:290-303
        '100': {'Kmax': 0.083805, 'sigGs': 0.6926457, 'sigGssig': 0.002096, 'sigTL': 0.67998, 'lam': 0.015,
                'sigR': 0.23668, 'q': 1 / (2 ** 10), 'wp': 1023, 'bl': 64, 'bias': np.array([0, 0, 0, 0])},
        '6400': {'Kmax': 8.74253, 'sigGs': 14.30362, 'sigGssig': 0.06967, 'sigTL': 12.8901, 'lam': 0.015,
                 'sigR': 0, 'q': 1 / (2 ** 10), 'wp': 1023, 'bl': 64,
                 'bias': np.array([-0.08113494, -0.04906388, -0.9408157, -1.2048522])},
    }
    return {'SonyA7S2': sony, 'IMX686': imx}


_REG = _build_regression()
_SPEC = _build_specific()


def get_camera_noisy_params(camera_type=None):
    """process.py:215-255.  Unknown cameras fall back to NikonD850 like the reference."""
    return dict(_REG[camera_type] if camera_type in _REG else _REG['NikonD850'])


def get_specific_noise_params(camera_type=None, iso='100'):
    """process.py:257-308.  None for cameras without a per-ISO table; KeyError for an
    ISO that is not calibrated (same as the reference's dict lookup)."""
    if camera_type not in _SPEC:
        return None
    return dict(_SPEC[camera_type][str(iso)])


def sample_params_max(camera_type='NikonD850', ratio=None, iso=None):
    """process.py:311-351 -- the sampler the GPU training path uses
    (trainer_SID.py:453).  NOTE the regression tables' ``Kmax`` is already log K."""
    params = None
    if iso is not None:
        params = get_specific_noise_params(camera_type=camera_type, iso=iso)
    if params is None:
        if camera_type in Dual_ISO_Cameras:
            camera_type += '_lowISO' if np.random.randint(2) < 1 else '_highISO'
        params = get_camera_noisy_params(camera_type=camera_type)
        bias = 0
        log_K = params['Kmax'] + np.random.uniform(low=-0.01, high=+0.01)
        K = np.exp(log_K)
        sigTL = np.exp(params['sigTLk'] * log_K + params['sigTLb'])
        sigR = np.exp(params['sigRk'] * log_K + params['sigRb'])
        mu_Gs = params['sigGsk'] * log_K + params['sigGsb']
        sigGs = np.exp(np.random.normal(loc=mu_Gs, scale=params['sigGssig']))
    else:
        K = params['Kmax'] * (1 + np.random.uniform(low=-0.01, high=+0.01))
        sigGs = np.random.normal(loc=params['sigGs'], scale=params['sigGssig']) if 'sigGssig' in params else params['sigGs']
        sigTL = np.random.normal(loc=params['sigTL'], scale=params['sigTLsig']) if 'sigTLsig' in params else params['sigTL']
        sigR = np.random.normal(loc=params['sigR'], scale=params['sigRsig']) if 'sigRsig' in params else params['sigR']
        bias = params['bias']
    if ratio is None:
        if 'SonyA7S2' in camera_type:
            ratio = np.random.uniform(low=100, high=300)
        else:
            ratio = np.exp(np.random.uniform(low=0, high=2.08))
    return {'K': K, 'sigTL': sigTL, 'sigR': sigR, 'sigGs': sigGs, 'bias': bias,
            'lam': params['lam'], 'q': params['q'], 'ratio': ratio, 'wp': params['wp'], 'bl': params['bl']}


_CRVD_A = np.array([3.513262, 6.955588, 13.486051, 26.585953, 52.032536])          # process.py:371-373
_CRVD_B = np.array([11.917691, 38.117816, 130.818508, 484.539790, 1819.818657])
_CRVD_BIAS = np.array([-1.12660, -1.69546, -3.25935, -6.68111, -12.66876])


def sample_params(camera_type='NikonD850', ln_ratio=False):
    """process.py:354-412 (the CPU-dataset variant).  Quirk kept: cameras without the
    ``uRead*`` regression (IMX686, NikonD850) raise KeyError, as the reference does."""
    if camera_type in ['SonyA7S2']:
        camera_type += '_lowISO' if np.random.randint(2) < 1 else '_highISO'
    params = get_camera_noisy_params(camera_type=camera_type)
    q = params['q']
    if camera_type in ['CRVD', 'BM3D']:
        choice = np.random.randint(5)
        log_K = np.log(_CRVD_A)[choice]
        K = _CRVD_A[choice]
        mu_TL = params['sigTLk'] * log_K + params['sigTLb']
        mu_R = params['sigRk'] * log_K + params['sigRb']
        mu_Gs = np.log(np.sqrt(_CRVD_B))[choice]
    else:
        log_K = np.random.uniform(low=params['Kmin'], high=params['Kmax'])
        K = np.exp(log_K)
        mu_TL = params['sigTLk'] * log_K + params['sigTLb']
        mu_R = params['sigRk'] * log_K + params['sigRb']
        mu_Gs = params['sigGsk'] * log_K + params['sigGsb']
        mu_bias = params['uReadk'] * log_K + params['uReadb']      # KeyError for tables without uRead*
    log_sigTL = np.random.normal(loc=mu_TL, scale=params['sigTLsig'])
    log_sigR = np.random.normal(loc=mu_R, scale=params['sigRsig'])
    log_sigGs = np.random.normal(loc=mu_Gs, scale=params['sigGssig'])
    log_bias = np.random.normal(loc=mu_bias, scale=params['uReadsig']) if 'uReadk' in params else 0
    if ln_ratio:
        ratio = np.exp(np.random.uniform(low=-0.01, high=1 if 'CRVD' in camera_type else 5))
    else:
        ratio = np.random.uniform(low=100, high=300)
    return {'K': K, 'sigTL': np.exp(log_sigTL), 'sigR': np.exp(log_sigR), 'sigGs': np.exp(log_sigGs),
            'bias': np.exp(log_bias), 'lam': params['lam'], 'q': q, 'ratio': ratio,
            'wp': params['wp'], 'bl': params['bl']}


def pack_raw_bayer(raw, wp=1023, clip=True):
    """process.py:40-64: pack a rawpy-like object (``raw_image_visible``, ``raw_pattern`` 2x2 with the
    colour indices 0..3 = R,G1,B,G2, ``black_level_per_channel``) to f32 [4,H/2,W/2]; float32
    arithmetic like the reference; runs on the HIP pack kernel, returns numpy like the reference."""
    im = np.ascontiguousarray(raw.raw_image_visible)
    pat = np.asarray(raw.raw_pattern)
    pos = (C.c_int * 4)()
    for c in range(4):
        r, q = np.where(pat == c)
        pos[c] = (int(r[0]) << 1) | int(q[0])
    black = (C.c_double * 4)(*[float(np.float32(b)) for b in raw.black_level_per_channel])
    is_f32 = im.dtype == np.float32
    if not is_f32:
        im = im.astype(np.uint16) if im.dtype != np.uint16 else im
    x = torch.from_numpy(im).cuda()
    H, W = x.shape
    out = torch.empty((4, H // 2, W // 2), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().pnnp_pack_bayer_pattern(_lib.ptr(x), int(is_f32), 1, H, W, C.c_int64(W), C.c_int64(H * W), _lib.ptr(out),
                                                  black, C.c_double(float(wp)), int(bool(clip)), pos, _lib.stream()), 'pack_bayer_pattern')
    return out.cpu().numpy()


# ---------------------------------------------------------------------------- device sampler
_FLAG = dict(p=0x01, g=0x02, r=0x04, q=0x08, d=0x10, b=0x20)
F_ORI, F_CLIP, F_TORCH = 0x100, 0x200, 0x1000
F_POST_MAX1, F_POST_MIN0 = 0x2000, 0x4000      # Trainer.preprocess clamp fused into the sampler
F_TORCH_TUKEY = 0x8000                         # extension: 'g' (Tukey-lambda read noise) in torch mode
_ORDER = ('K', 'sigGs', 'sigTL', 'lam', 'sigR', 'q', 'ratio', 'wp', 'bl')


class _RngState:
    """Counter-based RNG state of the HIP sampler: (seed, offset).  ``offset`` advances
    by one per sampler call, so successive calls draw independent noise and a run is
    reproducible from ``manual_seed``."""
    seed = 1997           # the reference seeds everything with 1997 (utils/utils.py:45-48)
    offset = 0


def manual_seed(seed, offset=0):
    _RngState.seed = int(seed) & (2 ** 64 - 1)
    _RngState.offset = int(offset)


def get_rng_state():
    return _RngState.seed, _RngState.offset


def noise_flags(noise_code, ori=False, clip=False, torch_mode=False):
    f = 0
    for ch in noise_code.lower():
        f |= _FLAG.get(ch, 0)
    return f | (F_ORI if ori else 0) | (F_CLIP if clip else 0) | (F_TORCH if torch_mode else 0)


def pack_params(plist, device):
    """list of param dicts (host scalars / numpy / 0-dim tensors on any device, as the
    reference trainer builds them, trainer_SID.py:455-459) -> f32 [B][NPARAM] on
    ``device`` without a device->host sync."""
    B = len(plist)
    if any(torch.is_tensor(v) and v.is_cuda for p in plist for v in p.values()):
        rows = []
        for p in plist:
            vals = [torch.as_tensor(p.get(k, 0.0), dtype=torch.float32, device=device).reshape(()) for k in _ORDER]
            bias = torch.as_tensor(p.get('bias', 0.0), dtype=torch.float32, device=device).reshape(-1)
            bias = bias.expand(4) if bias.numel() == 1 else bias[:4]
            rows.append(torch.cat([torch.stack(vals), bias, torch.zeros(NPARAM - 13, device=device)]))
        return torch.stack(rows).contiguous()
    rows = np.zeros((B, NPARAM), np.float32)
    for i, p in enumerate(plist):
        rows[i, :9] = [float(p.get(k, 0.0)) for k in _ORDER]
        bias = np.asarray(p.get('bias', 0.0), np.float64).reshape(-1)
        rows[i, 9:13] = bias if bias.size == 4 else bias[0]
    return torch.from_numpy(rows).to(device, non_blocking=True)


def noise_sample(y, params, flags, mfm=1.0, seed=None, offset=None, crop_base=0, out=None):
    """Batched entry: y f32 [B,C,H,W] (CUDA), params f32 [B,NPARAM] (CUDA)."""
    _lib.require_cuda(y, params)
    y = y.contiguous()
    B, Cc, H, W = y.shape
    if out is None:
        out = torch.empty_like(y)
    if seed is None:
        seed = _RngState.seed
    if offset is None:
        offset = _RngState.offset
        _RngState.offset += 1
    _lib.check(_lib.lib().pnnp_noise_sample_f32(_lib.ptr(y), _lib.ptr(out), B, Cc, H, W, _lib.ptr(params),
                                                C.c_uint(flags), C.c_float(mfm), C.c_uint64(seed), C.c_uint64(offset),
                                                C.c_uint32(crop_base), _lib.stream()), 'noise_sample')
    return out


def _as4d(y):
    if y.dim() == 3:
        return y[None], True
    if y.dim() == 4:
        return y, False
    raise ValueError('expected [C,H,W] or [B,C,H,W]')


def generate_noisy_torch(y, camera_type=None, noise_code='p', param=None, MultiFrameMean=1, ori=False, clip=False, tukey=False):
    """process.py:634-673 on the HIP sampler.  ``y`` is a CUDA tensor [C,H,W]; a 4-D
    batch shares one parameter set and, like the reference, draws ONE row-noise pattern
    per call only if you pass it as separate crops -- here every crop of a batch gets its
    own row noise (the reference's broadcast-over-batch is an artefact the trainers
    avoid by looping per crop, trainer_SID.py:451-462).
    Error behaviour kept: 'g' -> NotImplementedError (:654); no 'p' -> TypeError (:651).
    ``tukey=True`` (extension) lifts the first: 'g' then draws the Tukey-lambda read noise of
    generate_noisy_obs (:611, scale sigTL, shape lam) on the device, so the ELD/PMN codes ('pgrq') run on the
    GPU path."""
    code = noise_code.lower()
    if 'p' not in code:
        raise TypeError("Normal.__init__() missing 1 required positional argument: 'scale'")
    if 'g' in code and 'b' not in code and not tukey:
        raise NotImplementedError
    _lib.require_cuda(y)
    y4, squeeze = _as4d(y.float())
    if tukey and 'g' in code:
        flags = noise_flags(code, ori=ori, clip=bool(clip), torch_mode=True) | F_TORCH_TUKEY
    else:
        flags = noise_flags(code.replace('g', ''), ori=ori, clip=bool(clip), torch_mode=True)
    P = pack_params([param] * y4.shape[0], y.device)
    out = noise_sample(y4, P, flags, mfm=float(MultiFrameMean) ** 0.5)
    return out[0] if squeeze else out


def SNA_torch(gt, aug_wb, camera_type='IMX686', ratio=1, black_lr=False, ori=True, iso=None):
    """process.py:562-588 (shot-noise augmentation, the Mix_Dataset branch of Trainer.preprocess, trainer_SID.py:429-447)
    on the HIP sampler: gt [4,H,W] CUDA, aug_wb the four plane gains -> (dn, dy).  The gain draw K uses numpy's
    global RNG exactly like the reference; the Poisson draw uses the library's counter RNG (statistical parity)."""
    p = get_specific_noise_params(camera_type=camera_type, iso=iso)
    if p is None:
        assert camera_type == 'SonyA7S2'
        camera_type += '_lowISO' if iso <= 1600 else '_highISO'
        p = get_camera_noisy_params(camera_type=camera_type)
        p['K'] = 0.0009546 * iso * (1 + np.random.uniform(low=-0.01, high=+0.01)) - 0.00193
    else:
        p['K'] = p['Kmax'] * (1 + np.random.uniform(low=-0.01, high=+0.01))
    _lib.require_cuda(gt)
    g = gt.contiguous().float()
    Cc, H, W = g.shape
    dn = torch.empty_like(g); dy = torch.empty_like(g)
    aug = (C.c_float * 4)(*[float(v) for v in np.asarray(aug_wb, np.float32).reshape(-1)[:4]])
    off = _RngState.offset
    _RngState.offset += 1
    _lib.check(_lib.lib().pnnp_sna_f32(_lib.ptr(g), _lib.ptr(dn), _lib.ptr(dy), Cc, H, W, aug, C.c_float(float(p['K'])),
                                       C.c_float(float(p['wp'])), C.c_float(float(p['bl'])), C.c_float(float(ratio)), int(bool(black_lr)),
                                       int(bool(ori)), C.c_uint64(_RngState.seed), C.c_uint64(off), C.c_uint32(0), _lib.stream()), 'sna')
    return dn, dy


class HighBitRecovery:
    """process.py:675-751: maps low-bit raw values back to a high-bit distribution by re-drawing every integer-valued
    pixel inside its quantisation bin according to the calibrated read-noise law.  The LUT is built on the host exactly
    like the reference (scipy cdf per integer, same numpy draws in the same order); ``map`` runs on the device."""

    def __init__(self, camera_type='IMX686', noise_code='prq', param=None, perturb=True, factor=6, float=True):
        self.camera_type, self.noise_code, self.param = camera_type, noise_code, param
        self.perturb, self.factor, self.float = perturb, factor, float
        self.lut = {}

    def get_lut(self, iso_list, blc_mean=None):
        for iso in iso_list:
            bias = 0 if blc_mean is None else np.mean(blc_mean[iso])
            if self.perturb:
                bias += np.random.randn() * 0.1
            self.lut[iso] = self.HB2LB_LUT(iso, bias)

    def HB2LB_LUT(self, iso, bias=0, param=None):
        from scipy import stats
        p = sample_params_max(self.camera_type, iso=iso) if param is None else param
        info = {'param': p}
        if 'g' in self.noise_code.lower():
            dist, sigma, kind = stats.tukeylambda(p['lam'], loc=bias, scale=p['sigTL']), p['sigTL'], 1
        else:
            dist, sigma, kind = stats.norm(loc=bias, scale=p['sigGs']), p['sigGs'], 0
        low = max(int(-sigma * self.factor + bias), -p['bl'] + 1)
        high = int(sigma * self.factor + bias)
        xs = np.arange(low, high)
        c0, c1 = dist.cdf(xs - 0.5), dist.cdf(xs + 0.5)
        info.update(dist=dist, kind=kind, low=low, high=high, bias=bias, sigma=sigma, cdf=c0, range=c1 - c0)
        return info

    def map(self, data, iso=6400, norm=True, rand=None):
        """data: CUDA tensor (normalised if max <= 1, else DN); rand: optional float64 uniforms (testing)."""
        L = self.lut[iso]
        p = L['param']
        _lib.require_cuda(data)
        d = data.contiguous().float()
        span = float(p['wp'] - p['bl'])
        in_mul = span if float(d.max()) <= 1 else 1.0              # process.py:721 (one scalar sync, like np.max)
        key = ('dev', d.device)
        if key not in L:
            L[key] = (torch.from_numpy(np.ascontiguousarray(L['cdf'], np.float64)).to(d.device),
                      torch.from_numpy(np.ascontiguousarray(L['range'], np.float64)).to(d.device))
        cdf, rng = L[key]
        out = torch.empty_like(d)
        r = rand.contiguous().double() if rand is not None else None
        off = _RngState.offset
        _RngState.offset += 1
        lam = float(p.get('lam', 0.0)) if L['kind'] == 1 else 0.0
        _lib.check(_lib.lib().pnnp_hbr_map_f32(_lib.ptr(d), _lib.ptr(out), C.c_int64(d.numel()), _lib.ptr(cdf), _lib.ptr(rng), int(L['low']),
                                               int(L['high']), int(L['kind']), C.c_double(float(L['bias'])), C.c_double(float(L['sigma'])),
                                               C.c_double(lam), _lib.ptr(r), C.c_float(in_mul), C.c_float(span if norm else 0.0),
                                               C.c_float(float(p['bl'])), int(bool(self.float)), C.c_uint64(_RngState.seed), C.c_uint64(off),
                                               _lib.stream()), 'hbr_map')
        return out


def generate_noisy_obs(y, camera_type=None, wp=16383, noise_code='p', param=None, MultiFrameMean=1, ori=False, clip=False):
    """process.py:591-631 on the HIP sampler.  numpy in -> numpy out (staged through the
    GPU); CUDA tensor in -> CUDA tensor out."""
    host = not (torch.is_tensor(y) and y.is_cuda)
    yt = (torch.from_numpy(np.ascontiguousarray(y, np.float32)) if not torch.is_tensor(y) else y.float()).cuda() if host else y.float()
    y4, squeeze = _as4d(yt)
    flags = noise_flags(noise_code, ori=ori, clip=bool(clip), torch_mode=False)
    P = pack_params([param] * y4.shape[0], yt.device)
    out = noise_sample(y4, P, flags, mfm=float(MultiFrameMean) ** 0.5)
    out = out[0] if squeeze else out
    return out.cpu().numpy() if host else out
