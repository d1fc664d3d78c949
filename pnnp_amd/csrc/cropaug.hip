// Dataset-side crop / augment / white-balance gains on the device (SURVEY 8(f) rows f2 + f3):
//   init_random_crop_point + random_crop + data_aug   data_process/syn_datasets.py:69-107,162-173
//                                                     (8-way) and real_datasets.py:98-137 (4-way)
//   raw2bayer fused in front of the crop              utils/isp_ops.py:84-96 (float64 affine)
//   linear dark shading subtracted before the pack    real_datasets.py:360-368
//   random_gains white-balance augmentation           syn_datasets.py:313-322, unprocess.py:60-77
// so a frame crosses PCIe once as 2 B/px uint16 and every crop is cut, rotated, flipped, normalised and
// gained in one pass.  HBM-bound byte mover: a workgroup owns a 32x32 tile of SOURCE pixels of one crop
// (all planes), reads it row-wise (coalesced), turns it through LDS, and writes it row-wise in OUTPUT
// order (coalesced for every rotation).  Arithmetic is bit-exact with numpy: see pack_value below.
#include "common.h"

namespace {

struct CropArgs {
    double black[4];
    double den[4], rcp[4];   // wp - black and its correctly rounded reciprocal (host division)
    double wp;
    int norm, clip, post_clip;
    int H, W;            // raw frame (mode 0) or packed plane (mode 1) size
    int C;               // planes of the packed source (mode 1)
    int ps;              // crop side, packed pixels
    int ds_f64;          // dark shading map dtype
    double ds_add;       // added back after the subtraction (its mean for noise code 'd', + a random bias)
};

__device__ __forceinline__ float pack_value(float x, double black, double den, double rcp, int norm, int clip) {
    // numpy: float32 plane - int64/float64 black -> float64; / (wp - black); clip; -> float32.  The division by the per-plane
    // constant is the correctly rounded 3-operation sequence of csrc/pack.hip (div_const): bit-identical, no fp64 divide.
    if (!norm) return clip ? fminf(fmaxf(x, 0.f), 1.f) : x;
    const double n = (double)x - black, q = n * rcp;
    double v = fma(fma(-q, den, n), rcp, q);
    if (clip) v = fmin(fmax(v, 0.0), 1.0);
    return (float)v;
}

// desc[n] = {h0, w0, rot (np.rot90 k on the last two axes), flip (reverse the last axis, after the rotation)}
// gains[n] = {rgb, red, blue} or null: x = f32(x * rgb); plane 0: f32(x * red); plane 2: f32(x * blue)
// (computed in float64 and rounded once == numpy for float32 and float64 gain operands alike).
template <int MODE>   // 0: uint16 Bayer frame (+ optional dark shading), 1: packed float planes [C][H][W]
__global__ void __launch_bounds__(256)
crop_aug_kernel(const void* __restrict__ src_, const void* __restrict__ ds_, float* __restrict__ dst,
                const int* __restrict__ desc, const double* __restrict__ gains, CropArgs a) {
    __shared__ float t[4][32][33];
    const int n = blockIdx.z;
    const int h0 = desc[4 * n], w0 = desc[4 * n + 1], rot = desc[4 * n + 2] & 3, flip = desc[4 * n + 3];
    const int sa0 = blockIdx.y * 32, sb0 = blockIdx.x * 32, ps = a.ps;
    const int nplanes = MODE == 0 ? 4 : a.C;
    const bool odd = rot & 1;
    for (int c0 = 0; c0 < nplanes; c0 += 4) {
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int r = e >> 5, l = e & 31;
            const int si = sa0 + r, sj = sb0 + l;
            if (si >= ps || sj >= ps) continue;
            if constexpr (MODE == 0) {
                const uint16_t* src = (const uint16_t*)src_;
                const int64_t y = 2 * (int64_t)(h0 + si), x = 2 * (int64_t)(w0 + sj);
                float v[4];   // Bayer offsets (0,0) (0,1) (1,1) (1,0) = R, G1, B, G2
                const int64_t o00 = y * a.W + x, o10 = o00 + a.W;
                const int64_t off[4] = {o00, o00 + 1, o10 + 1, o10};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float raw = (float)src[off[c]];
                    if (ds_) {     // lr_raw - darkshading (+ mean/bias), then raw.astype(float32)
                        if (a.ds_f64) raw = (float)(((double)src[off[c]] - ((const double*)ds_)[off[c]]) + a.ds_add);
                        else raw = __fadd_rn(__fsub_rn(raw, ((const float*)ds_)[off[c]]), (float)a.ds_add);
                    }
                    v[c] = pack_value(raw, a.black[c], a.den[c], a.rcp[c], a.norm, a.clip);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c][r][l] = v[c];
            } else {
                const float* src = (const float*)src_;
                for (int c = 0; c < 4 && c0 + c < nplanes; ++c)
                    t[c][r][l] = src[((int64_t)(c0 + c) * a.H + (h0 + si)) * a.W + (w0 + sj)];
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 256) {
            const int r = e >> 5, l = e & 31;
            const int sa = odd ? l : r, sb = odd ? r : l;       // lanes run along the OUTPUT row
            const int si = sa0 + sa, sj = sb0 + sb;
            if (si >= ps || sj >= ps) continue;
            int i, j;
            switch (rot) {                                       // np.rot90(m, k)[i][j] inverted
                case 0: i = si; j = sj; break;
                case 1: i = ps - 1 - sj; j = si; break;
                case 2: i = ps - 1 - si; j = ps - 1 - sj; break;
                default: i = sj; j = ps - 1 - si; break;
            }
            if (flip) j = ps - 1 - j;
            for (int c = 0; c < 4 && c0 + c < nplanes; ++c) {
                float v = t[c][sa][sb];
                if (gains) {
                    v = (float)((double)v * gains[3 * n]);
                    if (c0 + c == 0) v = (float)((double)v * gains[3 * n + 1]);
                    if (c0 + c == 2) v = (float)((double)v * gains[3 * n + 2]);
                }
                if (a.post_clip) v = fminf(fmaxf(v, 0.f), 1.f);
                dst[(((int64_t)n * nplanes + c0 + c) * ps + i) * ps + j] = v;
            }
        }
        __syncthreads();
    }
}

int check_desc_args(int n, int ps, const void* src, const void* dst, const void* desc) {
    if (n < 0 || ps < 0) return PNNP_E_INVALID;
    if (n == 0 || ps == 0) return PNNP_OK + 1;       // nothing to do
    if (!src || !dst || !desc) return PNNP_E_INVALID;
    return PNNP_OK;
}

}  // namespace

extern "C" {

// Crops of the PACKED image of a uint16 Bayer frame: dst[n][4][ps][ps] = aug(raw2bayer(frame - ds)[:, h0:h0+ps, w0:w0+ps]) * gains.
// desc [device, n x 4 int32] crop origins in packed pixels + rot + flip; gains [device, n x 3 f64] or null;
// dark [device, H x W f32/f64] or null.  The caller guarantees h0+ps <= H/2, w0+ps <= W/2.
int pnnp_crop_pack_bayer_u16(const uint16_t* frame, int H, int W, const void* dark, int dark_is_f64, double dark_add,
                             float* dst, int n, int ps, const int* desc, const double* gains,
                             const double* black4, double wp, int norm, int clip, int post_clip, void* stream) {
    const int st = check_desc_args(n, ps, frame, dst, desc);
    if (st != PNNP_OK) return st < 0 ? st : PNNP_OK;
    if (!black4 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || 2 * ps > H || 2 * ps > W) return PNNP_E_INVALID;
    CropArgs a{};
    for (int c = 0; c < 4; ++c) { a.black[c] = black4[c]; a.den[c] = wp - black4[c]; a.rcp[c] = 1.0 / a.den[c]; }
    a.wp = wp; a.norm = norm; a.clip = clip; a.post_clip = post_clip; a.H = H; a.W = W; a.C = 4; a.ps = ps;
    a.ds_f64 = dark_is_f64; a.ds_add = dark_add;
    const dim3 grid((ps + 31) / 32, (ps + 31) / 32, n);
    hipLaunchKernelGGL(crop_aug_kernel<0>, grid, dim3(256), 0, as_stream(stream), (const void*)frame, dark, dst, desc, gains, a);
    return pnnp_launch_status();
}

// random_crop on an already packed float image [C][h][w]: dst[n][C][ps][ps].
int pnnp_crop_aug_f32(const float* img, int C, int h, int w, float* dst, int n, int ps, const int* desc,
                      const double* gains, int post_clip, void* stream) {
    const int st = check_desc_args(n, ps, img, dst, desc);
    if (st != PNNP_OK) return st < 0 ? st : PNNP_OK;
    if (C <= 0 || ps > h || ps > w) return PNNP_E_INVALID;
    CropArgs a{};
    a.H = h; a.W = w; a.C = C; a.ps = ps; a.post_clip = post_clip;
    const dim3 grid((ps + 31) / 32, (ps + 31) / 32, n);
    hipLaunchKernelGGL(crop_aug_kernel<1>, grid, dim3(256), 0, as_stream(stream), (const void*)img, (const void*)nullptr, dst, desc, gains, a);
    return pnnp_launch_status();
}

}  // extern "C"
