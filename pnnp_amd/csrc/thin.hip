// The two thin ends of the networks (archs/Unet.py:31,80 conv1_1 / conv10_1; archs/ResUnet.py conv_in / conv10): a 3x3 convolution
// from the 4 Bayer planes to nf channels and a 1x1 convolution from nf channels to the 4 output planes.  At 4 channels on one side
// these are not matrix-core work -- 2 * 36 * nf and 2 * 4 * nf flops per pixel against 4 * nf bytes -- they are HBM streams over the
// full-resolution nf-channel map, and the generic implicit-GEMM kernels (which pad the 4 to a 32-wide tile) spent 8x the arithmetic
// and 3-7x the time on them.  Here, float32 on the vector ALUs (packed v_pk_fma_f32 where the data lies in pairs):
//
//   head forward   y[b][co][h][w] = bias[co] + sum_ci x[b][h][w][ci] * W[co][ci]  (+ residual), written NCHW directly
//                  (replaces the padded 1x1 GEMM + the NHWC->NCHW pass)
//   head backward  gx = (W^T g) * act'(x),  dW = sum_pix g (x) x,  db = sum_pix g   in ONE pass over x and g
//                  (replaces the padded backward-data GEMM + the 1x1 backward-weight kernel: x is read once, not twice)
//   first backward-weight   dW[co][ci][kh][kw] = sum_pix g[pix][co] * x[pix + (kh-1, kw-1)][ci],  db = sum_pix g
//
// Lane mapping of the head kernels: LPP = Cin / 4 consecutive lanes share a pixel, lane q of them owns channels 4q .. 4q+3, so one
// wave instruction reads 64 / LPP whole pixels = 1 KB contiguous.  Backward needs no cross-lane traffic at all (every lane has g of
// its pixel and its own four columns of W); forward sums the LPP partial dot products with three or four DPP adds.
// Weight gradients are summed per lane over the lane's pixels, then lanes -> waves -> workgroups -> thin_reduce_kernel, every stage
// in a fixed order: results are bit-reproducible run to run.
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the LPP lanes that share a pixel; every one of them ends with the same bits
template <int LPP>
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0xB1>(v);                      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                      // quad_perm [2,3,0,1]
    if (LPP >= 8) v = dpp_add<0x141>(v);       // row_half_mirror: lane i <- lane 7-i, the other quad of the 8
    if (LPP >= 16) v = dpp_add<0x140>(v);      // row_mirror: lane i <- lane 15-i, the other half of the row
    return v;
}

struct HeadFwd {
    const float* x; const float* w; const float* bias; const float* residual; float* out;
    int64_t npix, hw; int xcs, cin, cout;
};

template <int LPP>
__global__ void __launch_bounds__(256)
head_fwd_kernel(HeadFwd a) {
    constexpr int PPI = 64 / LPP;              // pixels per wave instruction
    const int lane = threadIdx.x & 63, q = lane % LPP, p = lane / LPP;
    float wq[4][4], bs[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
        bs[co] = co < a.cout ? a.bias[co] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[co][j] = co < a.cout ? a.w[co * a.cin + 4 * q + j] : 0.f;
    }
    const int64_t ngroups = a.npix >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t grp = wave0; grp < ngroups; grp += nwaves) {
        const int64_t base = grp << 6;
        f32x4 xv[LPP];
#pragma unroll
        for (int k = 0; k < LPP; ++k)
            xv[k] = *reinterpret_cast<const f32x4*>(a.x + (base + k * PPI + p) * a.xcs + 4 * q);
        float keep[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < LPP; ++k) {
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                float s = xv[k][0] * wq[co][0];
                s = fmaf(xv[k][1], wq[co][1], s); s = fmaf(xv[k][2], wq[co][2], s); s = fmaf(xv[k][3], wq[co][3], s);
                s = group_sum<LPP>(s);
                if (q == k) keep[co] = s;       // lane q keeps the pixel of step q: after LPP steps every lane holds one whole pixel
            }
        }
        const int64_t pix = base + q * PPI + p;
        const int64_t b = pix / a.hw, s = pix - b * a.hw;
#pragma unroll
        for (int co = 0; co < 4; ++co)
            if (co < a.cout) {
                const int64_t o = (b * a.cout + co) * a.hw + s;
                float v = keep[co] + bs[co];
                if (a.residual) v += a.residual[o];
                a.out[o] = v;
            }
    }
}

struct HeadBwd {
    const float* g; const float* x; const float* w; float* gx; float* partial;
    int64_t npix; int gcs, xcs, gxcs, cin, cout, mode;
    unsigned* amax;                     // max |gx| (csrc/h2.h) or null
};

// partial[block][4 * cin + 4]: dW[co][ci] for co < 4, then db[co]
template <int LPP>
__global__ void __launch_bounds__(256)
head_bwd_kernel(HeadBwd a) {
    constexpr int PPI = 64 / LPP;
    __shared__ float red[4][LPP][20];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, q = lane % LPP, p = lane / LPP;
    f32x4 wq[4];
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int j = 0; j < 4; ++j) wq[co][j] = co < a.cout ? a.w[co * a.cin + 4 * q + j] : 0.f;
    const float slope = a.mode == 1 ? 0.2f : (a.mode == 2 ? 0.f : 1.f);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[4] = {zero, zero, zero, zero};
    f32x4 db = zero;
    float amx = 0.f;
    const int64_t ngroups = a.npix >> 6;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    for (int64_t grp = wave0; grp < ngroups; grp += nwaves) {
        const int64_t base = grp << 6;
        f32x4 xv[LPP], gv[LPP];
#pragma unroll
        for (int k = 0; k < LPP; ++k) {
            const int64_t pix = base + k * PPI + p;
            xv[k] = *reinterpret_cast<const f32x4*>(a.x + pix * a.xcs + 4 * q);
            gv[k] = *reinterpret_cast<const f32x4*>(a.g + pix * a.gcs);
        }
#pragma unroll
        for (int k = 0; k < LPP; ++k) {
            f32x4 g4 = gv[k];
#pragma unroll
            for (int co = 0; co < 4; ++co) g4[co] = co < a.cout ? g4[co] : 0.f;      // padding channels of g may hold anything (0 x NaN would poison gx)
            f32x4 d = wq[0] * g4[0];
            d += wq[1] * g4[1]; d += wq[2] * g4[2]; d += wq[3] * g4[3];
            if (a.mode) {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[j] *= xv[k][j] > 0.f ? 1.f : slope;
            }
            *reinterpret_cast<f32x4*>(a.gx + (base + k * PPI + p) * a.gxcs + 4 * q) = d;
            amx = fmaxf(fmaxf(amx, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] += xv[k] * g4[co];
            db += g4;
        }
    }
    if (a.amax) pnnp_amax_commit(amx, a.amax);
    // lanes with the same q (other pixels) -> lane q; the bias sums only need the lanes of one q
#pragma unroll
    for (int m = LPP; m < 64; m <<= 1) {
#pragma unroll
        for (int co = 0; co < 4; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[co][j] += __shfl_xor(acc[co][j], m);
#pragma unroll
        for (int j = 0; j < 4; ++j) db[j] += __shfl_xor(db[j], m);
    }
    if (lane < LPP) {
#pragma unroll
        for (int co = 0; co < 4; ++co)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[wv][q][co * 4 + j] = acc[co][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wv][q][16 + j] = db[j];
    }
    __syncthreads();
    const int np = 4 * a.cin + 4;
    float* dst = a.partial + (int64_t)blockIdx.x * np;
    for (int o = threadIdx.x; o < np; o += 256) {
        int qq, e;
        if (o < 4 * a.cin) { const int co = o / a.cin, ci = o % a.cin; qq = ci >> 2; e = co * 4 + (ci & 3); }
        else { qq = 0; e = 16 + (o - 4 * a.cin); }
        dst[o] = (red[0][qq][e] + red[1][qq][e]) + (red[2][qq][e] + red[3][qq][e]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// First layer backward-weight.  A workgroup walks tiles of 2 NS rows x 48 columns (NS = 256 / CO row streams of CO lanes each,
// a stream owns two adjacent rows); lane = output channel.  The input tile with its halo sits in LDS as one float4 per pixel (the
// <= 4 real input channels); a stream slides a 4-row x 3-column window of float4 along its row pair -- four LDS reads per
// column step serve two pixels, and the three window columns rotate through three register slots with period 3, so the unrolled
// loop has no register moves -- and does 2 x 18 packed FMAs per step into its 36 accumulators (the accumulators are sums over
// pixels: more pixels per lane cost no registers).  g is read straight from global memory, CO lanes x 4 bytes = one or two whole
// cache lines per pixel, twelve columns (24 loads) ahead of their use.
constexpr int FW_PER_CU = 6;                     // workgroups per CU in the grid (3 are resident; measured flat from 4 to 12)
constexpr int FW_TW = 48, FW_LW = FW_TW + 4;     // tile width; LDS row width (columns -1 .. TW+2, the last two only ever loaded)

struct FirstW {
    const float* g; const float* x; float* partial;
    int B, H, W, gcs, xcs, tiles_w, ntiles;
};

template <int CO>
__global__ void __launch_bounds__(256)
first_wgrad_kernel(FirstW a) {
    constexpr int NS = 256 / CO, TR = 2 * NS;    // streams; tile rows
    constexpr int NX = (TR + 2) * FW_LW, NLD = (NX + 255) / 256;
    __shared__ f32x4 xt[NX];
    __shared__ float red[NS][37][CO];
    const int co = threadIdx.x % CO, st = threadIdx.x / CO;
    f32x2 acc[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t) { acc[t][0] = f32x2{0.f, 0.f}; acc[t][1] = f32x2{0.f, 0.f}; }
    float db = 0.f;
    const int tiles_h = a.H / TR;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tw = tile % a.tiles_w, th = (tile / a.tiles_w) % tiles_h, b = tile / (a.tiles_w * tiles_h);
        const int h0 = th * TR, w0 = tw * FW_TW;
        const int nw = min(FW_TW, a.W - w0);     // columns of this tile inside the image
        const float* g0 = a.g + (((int64_t)b * a.H + h0 + 2 * st) * a.W + w0) * a.gcs + co;
        const float* g1 = g0 + (int64_t)a.W * a.gcs;
        // g of twelve columns x two rows; columns outside the image read column 0 and count as zero
        auto load_g = [&](float (&gr)[2][12], int c0) {
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int64_t o = (int64_t)((c0 + u < nw) ? c0 + u : 0) * a.gcs;
                gr[0][u] = g0[o]; gr[1][u] = g1[o];
            }
        };
        float ga[2][12], gb[2][12];
        load_g(ga, 0);
        f32x4 stage[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int r = i / FW_LW, c = i % FW_LW;
            const int h = h0 - 1 + r, w = w0 - 1 + c;
            stage[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < NX && h >= 0 && h < a.H && w >= 0 && w < a.W)
                stage[k] = *reinterpret_cast<const f32x4*>(a.x + (((int64_t)b * a.H + h) * a.W + w) * a.xcs);
        }
        __syncthreads();                        // the previous tile's readers are done
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (threadIdx.x + 256 * k < NX) xt[threadIdx.x + 256 * k] = stage[k];
        __syncthreads();
        const f32x4* xr = xt + 2 * st * FW_LW;   // window rows 0..3 of the stream = image rows h-1 .. h+2 of its row pair (h, h+1)
        f32x4 s0[4], s1[4], s2[4];               // window columns; slot = (tile column + 1) % 3
#pragma unroll
        for (int r = 0; r < 4; ++r) { s2[r] = xr[r * FW_LW + 0]; s0[r] = xr[r * FW_LW + 1]; s1[r] = xr[r * FW_LW + 2]; }
        // one column step: the pixels (h, w) and (h+1, w) against window columns l, c, r
        auto pixels = [&](const f32x4 (&l)[4], const f32x4 (&c)[4], const f32x4 (&r)[4], float gv0, float gv1) {
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                const float gval = row ? gv1 : gv0;
                const f32x2 g2 = {gval, gval};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const f32x4 xl = l[row + kh], xc = c[row + kh], xg = r[row + kh];
                    acc[kh * 3 + 0][0] += f32x2{xl[0], xl[1]} * g2; acc[kh * 3 + 0][1] += f32x2{xl[2], xl[3]} * g2;
                    acc[kh * 3 + 1][0] += f32x2{xc[0], xc[1]} * g2; acc[kh * 3 + 1][1] += f32x2{xc[2], xc[3]} * g2;
                    acc[kh * 3 + 2][0] += f32x2{xg[0], xg[1]} * g2; acc[kh * 3 + 2][1] += f32x2{xg[2], xg[3]} * g2;
                }
                db += gval;
            }
        };
        auto chunk = [&](const float (&gr)[2][12], int c0) {
#pragma unroll
            for (int u = 0; u < 12; u += 3) {
                const int wc = c0 + u;           // tile column of the first of three steps; LDS column = tile column + 1
                const float m0 = wc < nw ? 1.f : 0.f, m1 = wc + 1 < nw ? 1.f : 0.f, m2 = wc + 2 < nw ? 1.f : 0.f;
                pixels(s2, s0, s1, gr[0][u] * m0, gr[1][u] * m0);
#pragma unroll
                for (int r = 0; r < 4; ++r) s2[r] = xr[r * FW_LW + wc + 3];
                pixels(s0, s1, s2, gr[0][u + 1] * m1, gr[1][u + 1] * m1);
#pragma unroll
                for (int r = 0; r < 4; ++r) s0[r] = xr[r * FW_LW + wc + 4];
                pixels(s1, s2, s0, gr[0][u + 2] * m2, gr[1][u + 2] * m2);
#pragma unroll
                for (int r = 0; r < 4; ++r) s1[r] = xr[r * FW_LW + wc + 5];
            }
        };
        load_g(gb, 12);
        chunk(ga, 0);
        load_g(ga, 24);
        chunk(gb, 12);
        load_g(gb, 36);
        chunk(ga, 24);
        chunk(gb, 36);
    }
    // streams -> workgroup: partial[block][(tap * 4 + ci) * CO + co], then [36 * CO + co] = db
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        red[st][t * 4 + 0][co] = acc[t][0][0]; red[st][t * 4 + 1][co] = acc[t][0][1];
        red[st][t * 4 + 2][co] = acc[t][1][0]; red[st][t * 4 + 3][co] = acc[t][1][1];
    }
    red[st][36][co] = db;
    __syncthreads();
    float* dst = a.partial + (int64_t)blockIdx.x * (37 * CO);
    for (int o = threadIdx.x; o < 37 * CO; o += 256) {
        const int k = o / CO, c = o % CO;
        float s = red[0][k][c];
#pragma unroll
        for (int i = 1; i < NS; ++i) s += red[i][k][c];
        dst[o] = s;
    }
}

// First layer FORWARD: y[pix][co] = act(bias[co] + sum_{tap, ci} x[pix + tap][ci] * w[co][ci][tap]).  Same tile, LDS image and sliding
// 4-row x 3-column window as the backward-weight kernel above; lane = output channel with its 36 weights in registers (as (ci0, ci1) /
// (ci2, ci3) pairs per tap), 18 packed FMAs per pixel into one pair of partial sums.  A pixel's CO outputs are one or two whole cache
// lines.  On the bf16x3 GEMM kernel the 4 input channels were padded to a 16-channel chunk: 4x the matrix work for this layer.
struct FirstF {
    const float* x; const float* w; const float* bias; float* y;
    int B, H, W, xcs, ycs, cin, act, tiles_w, ntiles;
    unsigned* amax;                     // max |y| (csrc/h2.h) or null
};

template <int CO>
__global__ void __launch_bounds__(256)
first_fwd_kernel(FirstF a) {
    constexpr int NS = 256 / CO, TR = 2 * NS;
    constexpr int NX = (TR + 2) * FW_LW, NLD = (NX + 255) / 256;
    __shared__ f32x4 xt[NX];
    const int co = threadIdx.x % CO, st = threadIdx.x / CO;
    f32x2 wq[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c0 = 2 * h, c1 = 2 * h + 1;
            wq[t][h] = f32x2{c0 < a.cin ? a.w[(co * a.cin + c0) * 9 + t] : 0.f, c1 < a.cin ? a.w[(co * a.cin + c1) * 9 + t] : 0.f};
        }
    const float bs = a.bias ? a.bias[co] : 0.f;
    const float slope = a.act == 1 ? 0.2f : (a.act == 2 ? 0.f : 1.f);
    const int tiles_h = a.H / TR;
    float amx = 0.f;
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int tw = tile % a.tiles_w, th = (tile / a.tiles_w) % tiles_h, b = tile / (a.tiles_w * tiles_h);
        const int h0 = th * TR, w0 = tw * FW_TW;
        const int nw = min(FW_TW, a.W - w0);
        f32x4 stage[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int r = i / FW_LW, c = i % FW_LW;
            const int h = h0 - 1 + r, w = w0 - 1 + c;
            stage[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < NX && h >= 0 && h < a.H && w >= 0 && w < a.W)
                stage[k] = *reinterpret_cast<const f32x4*>(a.x + (((int64_t)b * a.H + h) * a.W + w) * a.xcs);
        }
        __syncthreads();                        // the previous tile's readers are done
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (threadIdx.x + 256 * k < NX) xt[threadIdx.x + 256 * k] = stage[k];
        __syncthreads();
        float* y0 = a.y + (((int64_t)b * a.H + h0 + 2 * st) * a.W + w0) * a.ycs + co;
        float* y1 = y0 + (int64_t)a.W * a.ycs;
        const f32x4* xr = xt + 2 * st * FW_LW;
        f32x4 s0[4], s1[4], s2[4];               // window columns; slot = (tile column + 1) % 3
#pragma unroll
        for (int r = 0; r < 4; ++r) { s2[r] = xr[r * FW_LW + 0]; s0[r] = xr[r * FW_LW + 1]; s1[r] = xr[r * FW_LW + 2]; }
        auto pixels = [&](const f32x4 (&l)[4], const f32x4 (&c)[4], const f32x4 (&r)[4], int col) {
#pragma unroll
            for (int row = 0; row < 2; ++row) {
                f32x2 acc = {0.f, 0.f};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const f32x4 xl = l[row + kh], xc = c[row + kh], xg = r[row + kh];
                    acc += f32x2{xl[0], xl[1]} * wq[kh * 3 + 0][0]; acc += f32x2{xl[2], xl[3]} * wq[kh * 3 + 0][1];
                    acc += f32x2{xc[0], xc[1]} * wq[kh * 3 + 1][0]; acc += f32x2{xc[2], xc[3]} * wq[kh * 3 + 1][1];
                    acc += f32x2{xg[0], xg[1]} * wq[kh * 3 + 2][0]; acc += f32x2{xg[2], xg[3]} * wq[kh * 3 + 2][1];
                }
                float v = (acc[0] + acc[1]) + bs;
                v = fmaxf(v, slope * v);
                if (col < nw) { (row ? y1 : y0)[(int64_t)col * a.ycs] = v; amx = fmaxf(amx, fabsf(v)); }
            }
        };
        for (int w = 0; w < FW_TW; w += 3) {
            pixels(s2, s0, s1, w);
#pragma unroll
            for (int r = 0; r < 4; ++r) s2[r] = xr[r * FW_LW + w + 3];
            pixels(s0, s1, s2, w + 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) s0[r] = xr[r * FW_LW + w + 4];
            pixels(s1, s2, s0, w + 2);
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[r] = xr[r * FW_LW + w + 5];
        }
    }
    if (a.amax) pnnp_amax_commit(amx, a.amax);
}

// Sum of the per-workgroup partials in a fixed order, scattered to the torch layouts.
//   kind 0 (head):  o < 4*cin: dW[co][ci] with co = o / cin;  then db[o - 4*cin]                    (rows co >= cout dropped)
//   kind 1 (first): o = k * CO + co, k = tap * 4 + ci < 36: dW[co][ci][tap] (ci >= cin dropped);  k = 36: db[co]
__global__ void __launch_bounds__(1024)
thin_reduce_kernel(const float* __restrict__ partial, int nblocks, int np, int kind, int cin, int cout, float* __restrict__ dW,
                   float* __restrict__ dbias, int accumulate) {
    // 64 outputs x 16 slices per workgroup: a wave reads 64 consecutive floats of one partial row; 16 loads in flight per thread
    // (this kernel is pure latency: ~1-5 MB in total)
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, o = blockIdx.x * 64 + l, s = threadIdx.x >> 6;
    float v = 0.f;
    if (o < np) {
#pragma unroll 16
        for (int b = s; b < nblocks; b += 16) v += partial[(int64_t)b * np + o];
    }
    red[s][l] = v;
    __syncthreads();
    if (s != 0 || o >= np) return;
    v = red[0][l];
#pragma unroll
    for (int i = 1; i < 16; ++i) v += red[i][l];
    float* dst = nullptr;
    if (kind == 0) {
        if (o < 4 * cin) { const int co = o / cin, ci = o % cin; if (co < cout) dst = dW + co * cin + ci; }
        else if (o - 4 * cin < cout && dbias) dst = dbias + (o - 4 * cin);
    } else {
        const int k = o / cout, co = o % cout;
        if (k < 36) { const int tap = k >> 2, ci = k & 3; if (ci < cin) dst = dW + (co * cin + ci) * 9 + tap; }
        else if (dbias) dst = dbias + co;
    }
    if (dst) *dst = accumulate ? *dst + v : v;
}

int head_lpp(int cin) { return (cin == 16 || cin == 32 || cin == 64) ? cin / 4 : 0; }

int thin_blocks(int64_t units, int per_cu) {
    const int64_t cap = (int64_t)pnnp_device_cus() * per_cu;
    return (int)(units < cap ? (units < 1 ? 1 : units) : cap);
}

}  // namespace

extern "C" {

// 1 when the thin head kernels take this 1x1 layer (cin channels in, cout out)
int pnnp_head_supported(int cin, int cout, int64_t npix) { return head_lpp(cin) != 0 && cout >= 1 && cout <= 4 && npix > 0 && (npix & 63) == 0; }

int64_t pnnp_head_bwd_workspace_floats(int cin) { return (int64_t)pnnp_device_cus() * 4 * (4 * cin + 4); }

int pnnp_head_fwd_f32(const float* x, int xcs, int cin, const float* w, const float* bias, const float* residual, float* out,
                      int B, int H, int W, int cout, void* stream) {
    const int64_t npix = (int64_t)B * H * W;
    if (!x || !w || !bias || !out || B < 0 || H <= 0 || W <= 0 || (xcs & 3) || xcs < cin) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    if (!pnnp_head_supported(cin, cout, npix)) return PNNP_E_UNSUPPORTED;
    HeadFwd a{x, w, bias, residual, out, npix, (int64_t)H * W, xcs, cin, cout};
    const dim3 grid(thin_blocks((npix >> 6) / 4, 8)), blk(256);
    switch (head_lpp(cin)) {
        case 4: hipLaunchKernelGGL(head_fwd_kernel<4>, grid, blk, 0, as_stream(stream), a); break;
        case 8: hipLaunchKernelGGL(head_fwd_kernel<8>, grid, blk, 0, as_stream(stream), a); break;
        default: hipLaunchKernelGGL(head_fwd_kernel<16>, grid, blk, 0, as_stream(stream), a); break;
    }
    return pnnp_launch_status();
}

// gx[pix][0..cin) = (sum_co g[pix][co] W[co][ci]) * act'(x[pix][ci])   (mode 0: no activation, 1: LeakyReLU 0.2, 2: ReLU; x is the
// activation OUTPUT, whose sign is that of its input), dW[co][ci] (+)= sum_pix g x, dbias[co] (+)= sum_pix g.
int pnnp_head_bwd_f32(const float* g, int gcs, const float* x, int xcs, int cin, const float* w, float* gx, int gxcs, int mode,
                      float* dW, float* dbias, int B, int H, int W, int cout, int accumulate, float* ws, int64_t ws_floats,
                      void* stream) {
    return pnnp_head_bwd_amax_f32(g, gcs, x, xcs, cin, w, gx, gxcs, mode, dW, dbias, B, H, W, cout, accumulate, ws, ws_floats, nullptr, stream);
}
// ... and max |gx| into an amax slot (csrc/h2.h: gx is what the fp16x2 backward kernels of conv9_2 split)
int pnnp_head_bwd_amax_f32(const float* g, int gcs, const float* x, int xcs, int cin, const float* w, float* gx, int gxcs, int mode,
                           float* dW, float* dbias, int B, int H, int W, int cout, int accumulate, float* ws, int64_t ws_floats,
                           unsigned* amax_gx, void* stream) {
    const int64_t npix = (int64_t)B * H * W;
    if (!g || !x || !w || !gx || !dW || !ws || B < 0 || H <= 0 || W <= 0 || (xcs & 3) || (gcs & 3) || (gxcs & 3) || xcs < cin || gxcs < cin ||
        gcs < 4 || mode < 0 || mode > 2)
        return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    if (!pnnp_head_supported(cin, cout, npix)) return PNNP_E_UNSUPPORTED;
    const int np = 4 * cin + 4;
    const int blocks = thin_blocks((npix >> 6) / 4, 4);
    if (ws_floats < (int64_t)blocks * np) return PNNP_E_WORKSPACE;
    HeadBwd a{g, x, w, gx, ws, npix, gcs, xcs, gxcs, cin, cout, mode, amax_gx};
    const dim3 grid(blocks), blk(256);
    switch (head_lpp(cin)) {
        case 4: hipLaunchKernelGGL(head_bwd_kernel<4>, grid, blk, 0, as_stream(stream), a); break;
        case 8: hipLaunchKernelGGL(head_bwd_kernel<8>, grid, blk, 0, as_stream(stream), a); break;
        default: hipLaunchKernelGGL(head_bwd_kernel<16>, grid, blk, 0, as_stream(stream), a); break;
    }
    hipLaunchKernelGGL(thin_reduce_kernel, dim3(ceil_div(np, 64)), dim3(1024), 0, as_stream(stream), ws, blocks, np, 0, cin, cout, dW, dbias, accumulate);
    return pnnp_launch_status();
}

// 1 when the thin first-layer backward-weight kernel takes this 3x3 layer
int pnnp_first_wgrad_supported(int cin, int cout, int H, int W) {
    return cin >= 1 && cin <= 4 && (cout == 32 || cout == 64) && H > 0 && W > 0 && H % (512 / cout) == 0 && W % 16 == 0;
}

int64_t pnnp_first_wgrad_workspace_floats(int cout) { return (int64_t)pnnp_device_cus() * FW_PER_CU * 37 * cout; }

// dW[cout][cin][3][3] (+)= sum_pix g[pix][co] x[pix + tap][ci], dbias[co] (+)= sum_pix g;  x [B][H][W][xcs] with channels cin .. 3
// of every pixel ZERO (the engines' zero-padded NHWC copy of the input), g [B][H][W][gcs].
int pnnp_first_bwd_weight_f32(const float* g, int gcs, int cout, const float* x, int xcs, int cin, float* dW, float* dbias,
                              int B, int H, int W, int accumulate, float* ws, int64_t ws_floats, void* stream) {
    if (!g || !x || !dW || !ws || B < 0 || H <= 0 || W <= 0 || (xcs & 3) || xcs < 4 || gcs < cout) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    if (!pnnp_first_wgrad_supported(cin, cout, H, W)) return PNNP_E_UNSUPPORTED;
    const int tr = 512 / cout;
    FirstW a{g, x, ws, B, H, W, gcs, xcs, ceil_div(W, FW_TW), 0};
    a.ntiles = B * (H / tr) * a.tiles_w;
    const int blocks = thin_blocks(a.ntiles, FW_PER_CU);
    const int np = 37 * cout;
    if (ws_floats < (int64_t)blocks * np) return PNNP_E_WORKSPACE;
    const dim3 grid(blocks), blk(256);
    if (cout == 32) hipLaunchKernelGGL(first_wgrad_kernel<32>, grid, blk, 0, as_stream(stream), a);
    else hipLaunchKernelGGL(first_wgrad_kernel<64>, grid, blk, 0, as_stream(stream), a);
    hipLaunchKernelGGL(thin_reduce_kernel, dim3(ceil_div(np, 64)), dim3(1024), 0, as_stream(stream), ws, blocks, np, 1, cin, cout, dW, dbias, accumulate);
    return pnnp_launch_status();
}

// y [B][H][W][ycs] (first cout channels) = act(conv3x3(x; w [cout][cin][3][3]) + bias): the first layer's forward (archs/Unet.py:31-33).
// x [B][H][W][xcs] with channels cin .. 3 of every pixel ZERO; act 0 none / 1 LeakyReLU(0.2) / 2 ReLU.  pnnp_first_wgrad_supported()
// says whether the layer qualifies.
int pnnp_first_fwd_f32(const float* x, int xcs, int cin, const float* w, const float* bias, float* y, int ycs, int B, int H, int W, int cout,
                       int act, void* stream) {
    return pnnp_first_fwd_amax_f32(x, xcs, cin, w, bias, y, ycs, B, H, W, cout, act, nullptr, stream);
}
// ... and max |y| into an amax slot (csrc/h2.h)
int pnnp_first_fwd_amax_f32(const float* x, int xcs, int cin, const float* w, const float* bias, float* y, int ycs, int B, int H, int W, int cout,
                            int act, unsigned* amax_y, void* stream) {
    if (!x || !w || !y || B < 0 || H <= 0 || W <= 0 || (xcs & 3) || xcs < 4 || ycs < cout || act < 0 || act > 2) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    if (!pnnp_first_wgrad_supported(cin, cout, H, W)) return PNNP_E_UNSUPPORTED;
    const int tr = 512 / cout;
    FirstF a{x, w, bias, y, B, H, W, xcs, ycs, cin, act, ceil_div(W, FW_TW), 0, amax_y};
    a.ntiles = B * (H / tr) * a.tiles_w;
    const dim3 grid(thin_blocks(a.ntiles, FW_PER_CU)), blk(256);
    if (cout == 32) hipLaunchKernelGGL(first_fwd_kernel<32>, grid, blk, 0, as_stream(stream), a);
    else hipLaunchKernelGGL(first_fwd_kernel<64>, grid, blk, 0, as_stream(stream), a);
    return pnnp_launch_status();
}

}  // extern "C"
