// Bayer pack / unpack kernels (HBM-bound byte movers, bit-exact with the reference).
//   raw2bayer  utils/isp_ops.py:84-96, bayer2raw :98-112, bayer2rggb/rggb2bayer :57-63,
//   bayer2rows/rows2bayer :65-81.
// Layout: one thread owns a 2x8 Bayer patch (two 16-byte row segments for u16) and
// writes one 16-byte float4 per packed plane, so loads and stores are fully coalesced.
#include "common.h"

namespace {

struct PackArgs {
    double black[4];
    double den[4], rcp[4];   // wp - black and its correctly rounded reciprocal (host division)
    double wp;
    int norm, clip;
    int pos[4];      // Bayer offset of plane c: (dy << 1) | dx; raw2bayer: {0, 1, 3, 2} = R,G1,B,G2
    int f32math;     // normalise in float32 (pack_raw_bayer, process.py:59-61) instead of float64 (raw2bayer)
};

// IEEE-exact double division by a constant in three operations (Markstein): with y = RN(1/d) from the host,
//   q = RN(n*y);  r = n - d*q exactly (FMA);  RN(q + r*y) = RN(n/d)   -- the correctly rounded quotient, i.e. bit-identical to
// numpy's float64 division.  The hardware's own fp64 division sequence (scale, rcp, four FMAs, fmas, fixup) made this
// byte mover ALU-bound (2.2 TB/s); d = wp - black is a per-plane constant.  tests: all 65536 uint16 codes, bit-exact.
__device__ __forceinline__ double div_const(double n, double d, double y) {
    const double q = n * y;
    const double r = fma(-q, d, n);
    return fma(r, y, q);
}

__device__ __forceinline__ float pack_value(float x, double black, double den, double rcp, int norm, int clip) {
    // numpy: float32 stack - float64 black -> float64; / (wp - black) float64; clip; -> f32
    if (!norm) {
        if (clip) x = fminf(fmaxf(x, 0.f), 1.f);
        return x;
    }
    double v = div_const((double)x - black, den, rcp);
    if (clip) v = fmin(fmax(v, 0.0), 1.0);
    return (float)v;
}

__device__ __forceinline__ float pack_value_f32(float x, float black, float wp, int clip) {
    // numpy: float32 stack - float32 black, / (python-int wp - float32 black) -> all float32
    float v = __fdiv_rn(__fsub_rn(x, black), __fsub_rn(wp, black));
    if (clip) v = fminf(fmaxf(v, 0.f), 1.f);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256)
pack_bayer_kernel(const T* __restrict__ src, float* __restrict__ dst, int B, int H, int W,
                  int64_t row_stride, int64_t batch_stride, PackArgs a) {
    const int h = H >> 1, w = W >> 1;
    const int wq = (w + 3) >> 2;                       // groups of 4 packed pixels per row
    // workgroup = 64 four-pixel segments x 4 packed rows; grid x along the row, y over rows (strided), z = images: no divisions
    const int b = blockIdx.z;
    const int xq = blockIdx.x * 64 + (threadIdx.x & 63);
    if (xq >= wq) return;
    for (int y = blockIdx.y * 4 + (threadIdx.x >> 6); y < h; y += gridDim.y * 4) {
        const T* r0 = src + b * batch_stride + (int64_t)(2 * y) * row_stride + 8 * xq;
        const T* r1 = r0 + row_stride;
        const int nx = min(4, w - 4 * xq);
        float top[8], bot[8];
        const bool vec = (nx == 4) && ((((uintptr_t)r0) | ((uintptr_t)r1)) % (8 * sizeof(T)) == 0);
        if (vec) {
            if constexpr (sizeof(T) == 2) {
                const uint4 u0 = *reinterpret_cast<const uint4*>(r0);
                const uint4 u1 = *reinterpret_cast<const uint4*>(r1);
                const unsigned a0[4] = {u0.x, u0.y, u0.z, u0.w}, a1[4] = {u1.x, u1.y, u1.z, u1.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    top[2 * i] = (float)(a0[i] & 0xffffu); top[2 * i + 1] = (float)(a0[i] >> 16);
                    bot[2 * i] = (float)(a1[i] & 0xffffu); bot[2 * i + 1] = (float)(a1[i] >> 16);
                }
            } else {
                const float4 f0 = reinterpret_cast<const float4*>(r0)[0], f1 = reinterpret_cast<const float4*>(r0)[1];
                const float4 g0 = reinterpret_cast<const float4*>(r1)[0], g1 = reinterpret_cast<const float4*>(r1)[1];
                top[0] = f0.x; top[1] = f0.y; top[2] = f0.z; top[3] = f0.w; top[4] = f1.x; top[5] = f1.y; top[6] = f1.z; top[7] = f1.w;
                bot[0] = g0.x; bot[1] = g0.y; bot[2] = g0.z; bot[3] = g0.w; bot[4] = g1.x; bot[5] = g1.y; bot[6] = g1.z; bot[7] = g1.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool in = i < 2 * nx;
                top[i] = in ? (float)r0[i] : 0.f;
                bot[i] = in ? (float)r1[i] : 0.f;
            }
        }
        float o[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {      // plane c <- Bayer offset pos[c] (default R,G1,B,G2 = (0,0),(0,1),(1,1),(1,0))
                const float te = (a.pos[c] & 1) ? top[2 * i + 1] : top[2 * i];
                const float be = (a.pos[c] & 1) ? bot[2 * i + 1] : bot[2 * i];
                const float v = (a.pos[c] & 2) ? be : te;
                o[c][i] = a.f32math ? pack_value_f32(v, (float)a.black[c], (float)a.wp, a.clip)
                                    : pack_value(v, a.black[c], a.den[c], a.rcp[c], a.norm, a.clip);
            }
        }
        const int64_t plane = (int64_t)h * w;
        float* d = dst + (int64_t)b * 4 * plane + (int64_t)y * w + 4 * xq;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float* dc = d + c * plane;
            if (nx == 4 && (((uintptr_t)dc) & 15) == 0) {
                *reinterpret_cast<float4*>(dc) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
            } else {
                for (int i = 0; i < nx; ++i) dc[i] = o[c][i];
            }
        }
    }
}

__global__ void __launch_bounds__(256)
unpack_bayer_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int B, int h, int w,
                    float span, float bl) {
    const int wq = (w + 3) >> 2;
    const int64_t total = (int64_t)B * h * wq;
    const int64_t plane = (int64_t)h * w;
    const int W = 2 * w;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int xq = (int)(t % wq);
        const int y = (int)((t / wq) % h);
        const int b = (int)(t / ((int64_t)wq * h));
        const int nx = min(4, w - 4 * xq);
        const float* s = src + (int64_t)b * 4 * plane + (int64_t)y * w + 4 * xq;
        unsigned short v[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float f[4];
            const float* sc = s + c * plane;
            if (nx == 4 && (((uintptr_t)sc) & 15) == 0) {
                const float4 q = *reinterpret_cast<const float4*>(sc);
                f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) f[i] = i < nx ? sc[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float x = fminf(fmaxf(f[i], 0.f), 1.f);       // np.clip(packed, 0, 1)
                x = __fmul_rn(x, span);                        // * (wp - bl)   (float32)
                x = __fadd_rn(x, bl);                          // + bl          (float32, no fma)
                v[c][i] = (unsigned short)(int)x;              // C-cast truncation into uint16
            }
        }
        uint16_t* d0 = dst + ((int64_t)b * 2 * h + 2 * y) * W + 8 * xq;
        uint16_t* d1 = d0 + W;
        if (nx == 4 && ((((uintptr_t)d0) | ((uintptr_t)d1)) & 15) == 0) {
            uint4 t0, t1;
            t0.x = v[0][0] | ((unsigned)v[1][0] << 16); t0.y = v[0][1] | ((unsigned)v[1][1] << 16);
            t0.z = v[0][2] | ((unsigned)v[1][2] << 16); t0.w = v[0][3] | ((unsigned)v[1][3] << 16);
            t1.x = v[3][0] | ((unsigned)v[2][0] << 16); t1.y = v[3][1] | ((unsigned)v[2][1] << 16);
            t1.z = v[3][2] | ((unsigned)v[2][2] << 16); t1.w = v[3][3] | ((unsigned)v[2][3] << 16);
            *reinterpret_cast<uint4*>(d0) = t0;
            *reinterpret_cast<uint4*>(d1) = t1;
        } else {
            for (int i = 0; i < nx; ++i) {
                d0[2 * i] = v[0][i]; d0[2 * i + 1] = v[1][i];
                d1[2 * i] = v[3][i]; d1[2 * i + 1] = v[2][i];
            }
        }
    }
}

// Generic index movers on elements of sizeof(T) bytes.  mode: 0 bayer->rggb, 1 rggb->bayer,
// 2 bayer->rows, 3 rows->bayer.  Indexed by the Bayer-domain element (y, x).
template <typename T>
__global__ void __launch_bounds__(256)
bayer_move_kernel(const T* __restrict__ src, T* __restrict__ dst, int H, int W, int mode) {
    const int64_t total = (int64_t)H * W;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(t % W), y = (int)(t / W);
        int64_t other;
        if (mode <= 1) other = ((int64_t)(y >> 1) * (W >> 1) + (x >> 1)) * 4 + ((y & 1) * 2 + (x & 1));
        else other = ((int64_t)(y & 1) * (H >> 1) + (y >> 1)) * W + x;
        if (mode == 0 || mode == 2) dst[other] = src[t];
        else dst[t] = src[other];
    }
}

int grid_for(int64_t threads) {
    int64_t blocks = (threads + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;     // 8 blocks per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <typename T>
int pack_impl(const T* src, int B, int H, int W, int64_t rs, int64_t bs, float* dst, const double* black4,
              double wp, int norm, int clip, void* stream, const int* pos4 = nullptr, int f32math = 0) {
    if (B < 0 || H < 0 || W < 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0 || H == 0 || W == 0) return PNNP_OK;          // empty input: nothing to do (pointers may be null)
    if (!src || !dst || !black4 || rs < W) return PNNP_E_INVALID;
    PackArgs a;
    for (int i = 0; i < 4; ++i) { a.black[i] = black4[i]; a.den[i] = wp - black4[i]; a.rcp[i] = 1.0 / a.den[i]; }
    a.wp = wp; a.norm = norm; a.clip = clip; a.f32math = f32math;
    static const int default_pos[4] = {0, 1, 3, 2};
    for (int i = 0; i < 4; ++i) a.pos[i] = (pos4 ? pos4[i] : default_pos[i]) & 3;
    const int wq = (W / 2 + 3) / 4, h = H / 2;
    if (B > 65535) return PNNP_E_UNSUPPORTED;
    hipLaunchKernelGGL(pack_bayer_kernel<T>, dim3((wq + 63) / 64, (h + 3) / 4 < 65535 ? (h + 3) / 4 : 65535, B), dim3(256), 0, as_stream(stream),
                       src, dst, B, H, W, rs, bs, a);
    return pnnp_launch_status();
}

template <typename T>
int move_impl(const void* src, void* dst, int H, int W, int mode, void* stream) {
    hipLaunchKernelGGL(bayer_move_kernel<T>, dim3(grid_for((int64_t)H * W)), dim3(256), 0, as_stream(stream),
                       (const T*)src, (T*)dst, H, W, mode);
    return pnnp_launch_status();
}

int move_dispatch(const void* src, void* dst, int H, int W, int eb, int mode, void* stream) {
    if (H < 0 || W < 0 || (H & 1) || ((mode <= 1) && (W & 1))) return PNNP_E_INVALID;
    if (H == 0 || W == 0) return PNNP_OK;
    if (!src || !dst) return PNNP_E_INVALID;
    switch (eb) {
        case 1: return move_impl<uint8_t>(src, dst, H, W, mode, stream);
        case 2: return move_impl<uint16_t>(src, dst, H, W, mode, stream);
        case 4: return move_impl<uint32_t>(src, dst, H, W, mode, stream);
        case 8: return move_impl<uint64_t>(src, dst, H, W, mode, stream);
        default: return PNNP_E_UNSUPPORTED;
    }
}

}  // namespace

extern "C" {

int pnnp_pack_bayer_u16(const uint16_t* src, int B, int H, int W, int64_t rs, int64_t bs, float* dst,
                        const double* black4, double wp, int norm, int clip, void* stream) {
    return pack_impl<uint16_t>(src, B, H, W, rs, bs, dst, black4, wp, norm, clip, stream);
}

int pnnp_pack_bayer_f32(const float* src, int B, int H, int W, int64_t rs, int64_t bs, float* dst,
                        const double* black4, double wp, int norm, int clip, void* stream) {
    return pack_impl<float>(src, B, H, W, rs, bs, dst, black4, wp, norm, clip, stream);
}

// pack_raw_bayer (process.py:40-64): the CFA-pattern-aware variant.  pos4 [host]: Bayer offset
// (dy << 1 | dx) of R, G1, B, G2 (from rawpy's raw_pattern); black4 [host] per-channel black level;
// arithmetic in float32 like the reference: (x - black) / (wp - black), optional clip to [0,1].
int pnnp_pack_bayer_pattern(const void* src, int is_f32, int B, int H, int W, int64_t rs, int64_t bs, float* dst,
                            const double* black4, double wp, int clip, const int* pos4, void* stream) {
    if (!pos4) return PNNP_E_INVALID;
    if (is_f32) return pack_impl<float>((const float*)src, B, H, W, rs, bs, dst, black4, wp, 1, clip, stream, pos4, 1);
    return pack_impl<uint16_t>((const uint16_t*)src, B, H, W, rs, bs, dst, black4, wp, 1, clip, stream, pos4, 1);
}

int pnnp_unpack_bayer_u16(const float* src, int B, int h, int w, uint16_t* dst, int wp, int bl, void* stream) {
    if (B < 0 || h < 0 || w < 0) return PNNP_E_INVALID;
    if (B == 0 || h == 0 || w == 0) return PNNP_OK;
    if (!src || !dst) return PNNP_E_INVALID;
    const int64_t total = (int64_t)B * h * ((w + 3) / 4);
    hipLaunchKernelGGL(unpack_bayer_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                       src, dst, B, h, w, (float)(wp - bl), (float)bl);
    return pnnp_launch_status();
}

int pnnp_bayer_to_rggb(const void* src, void* dst, int H, int W, int eb, void* stream) {
    return move_dispatch(src, dst, H, W, eb, 0, stream);
}
int pnnp_rggb_to_bayer(const void* src, void* dst, int h, int w, int eb, void* stream) {
    return move_dispatch(src, dst, 2 * h, 2 * w, eb, 1, stream);
}
int pnnp_bayer_to_rows(const void* src, void* dst, int H, int W, int eb, void* stream) {
    return move_dispatch(src, dst, H, W, eb, 2, stream);
}
int pnnp_rows_to_bayer(const void* src, void* dst, int h, int W, int eb, void* stream) {
    return move_dispatch(src, dst, 2 * h, W, eb, 3, stream);
}

}  // extern "C"
