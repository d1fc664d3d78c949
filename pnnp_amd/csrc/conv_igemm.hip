// Implicit-GEMM convolution kernels (forward / backward-data of conv3x3, ConvTranspose2d k2s2
// forward and backward-data) -- see igemm.h for the data layout.
//   reference ops: archs/Unet.py:16-51,54-94 (Conv2d 3x3 pad 1, ConvTranspose2d 2x2 s2,
//   LeakyReLU(0.2)), archs/ResUnet.py:15-44.
#include "igemm.h"

namespace {

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
struct IgemmCfg {
    static constexpr int P = (TAPS == 9) ? 1 : 0;
    static constexpr int TH = WM * MT;
    static constexpr int HR = TH + 2 * P, HC = 32 + 2 * P, NPIX = HR * HC;
    static constexpr int KQ = KC / 4;
    static constexpr int XS_F4 = KQ * NPIX;            // float4 count of the input tile
    static constexpr int WS_F4 = TAPS * KQ * BN;       // float4 count of the weight tile
    static constexpr int LDS_BYTES = (XS_F4 + WS_F4) * 16;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WN * NT * 32 == BN, "N tiling");
    static_assert(KC % 8 == 0, "KC multiple of 8");
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of logical
    // tiles so neighbours (shared halo rows / shared input tile) hit the same L2.  Bijective for any n.
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
__global__ void __launch_bounds__(256)
igemm_kernel(const IgemmArgs a) {
    using Cfg = IgemmCfg<TAPS, KC, BN, MT, NT, WM, WN>;
    constexpr int P = Cfg::P, TH = Cfg::TH, HC = Cfg::HC, NPIX = Cfg::NPIX, KQ = Cfg::KQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* xs = reinterpret_cast<float4*>(smem);
    float4* ws = xs + Cfg::XS_F4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int id = xcd_remap(blockIdx.x, total);
    const int nt_i = id % n_tiles;
    int m_i = id / n_tiles;
    const int tx = m_i % tiles_x; m_i /= tiles_x;
    const int ty = m_i % tiles_y;
    const int b = m_i / tiles_y;
    const int x0 = tx * 32, y0 = ty * TH, n0 = nt_i * BN;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = a.nseg * a.chunks_per_seg;
    const int K4 = nchunks * KQ;                       // Ktot / 4
    const float4* w4 = reinterpret_cast<const float4*>(a.w);

    for (int g = 0; g < nchunks; ++g) {
        const int si = g / a.chunks_per_seg;
        const IgemmSeg sg = a.seg[si];
        const int c0 = sg.coff + (g - si * a.chunks_per_seg) * KC;
        if (g) __syncthreads();
        // ---- stage the input halo tile: xs[cq][pix] <- src[b][gy][gx][c0 + 4cq ..]
        for (int i = tid; i < Cfg::XS_F4; i += 256) {
            const int cq = i % KQ, pix = i / KQ;
            const int r = pix / HC, q = pix - r * HC;
            const int gy = (y0 + r - P) * a.in_mul + sg.yoff, gx = (x0 + q - P) * a.in_mul + sg.xoff;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < a.IH && gx >= 0 && gx < a.IW)
                v = *reinterpret_cast<const float4*>(sg.ptr + (((int64_t)b * a.IH + gy) * a.IW + gx) * sg.cstride + c0 + 4 * cq);
            xs[cq * NPIX + pix] = v;
        }
        // ---- stage the weight tile: ws[t][cq][n] <- w[t][g*KQ + cq][n0 + n]
        for (int i = tid; i < Cfg::WS_F4; i += 256) {
            const int n = i % BN, rest = i / BN;
            const int cq = rest % KQ, t = rest / KQ;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + n < a.Ntot) v = w4[((int64_t)t * K4 + g * KQ + cq) * a.Ntot + n0 + n];
            ws[i] = v;
        }
        __syncthreads();
        // ---- MFMA over taps x channel octets
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dy = (TAPS == 9) ? t / 3 : 0, dx = (TAPS == 9) ? t % 3 : 0;
#pragma unroll
            for (int j = 0; j < KC / 8; ++j) {
                float4 av[MT], bv[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    av[i] = xs[(2 * j + half) * NPIX + (wm * MT + i + dy) * HC + dx + l31];
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    bv[i] = ws[(t * KQ + 2 * j + half) * BN + (wn * NT + i) * 32 + l31];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[k].x, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[k].y, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[k].z, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[k].w, acc[i][k], 0, 0, 0);
                    }
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&31 (channel), row = (r&3) + 8(r>>2) + 4(lane>>5) (pixel)
#pragma unroll
    for (int k = 0; k < NT; ++k) {
        const int n = n0 + (wn * NT + k) * 32 + l31;
        const bool n_ok = n < a.Ntot;
        const int d = (n >= a.n_split) ? 1 : 0;
        const int ch = n - (d ? a.n_split : 0);
        float* dst = a.dst[d];
        const float* msk = a.mask[d];
        const int cs = a.dst_cs[d], mmode = a.mask_mode[d], accum = a.accum[d];
        const float bias = (a.bias && n_ok) ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int py = y0 + wm * MT + i;
            if (py >= a.DH) continue;
            const int oy = py * a.out_mul + a.out_yoff;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (px >= a.DW || !n_ok) continue;
                const int ox = px * a.out_mul + a.out_xoff;
                const int64_t idx = (((int64_t)b * a.OH + oy) * a.OW + ox) * cs + ch;
                float v = acc[i][k][r] + bias;
                if (a.addsrc && d == 0) v += a.addsrc[idx];
                if (a.act == 1) v = v > 0.f ? v : 0.2f * v;
                else if (a.act == 2) v = fmaxf(v, 0.f);
                if (mmode) {
                    const float mv = msk[idx];
                    v *= (mv > 0.f) ? 1.f : (mmode == 1 ? 0.2f : 0.f);
                }
                if (accum) v += dst[idx];
                dst[idx] = v;
            }
        }
    }
}

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
int launch_cfg(const IgemmArgs& a, hipStream_t s) {
    using Cfg = IgemmCfg<TAPS, KC, BN, MT, NT, WM, WN>;
    auto kern = igemm_kernel<TAPS, KC, BN, MT, NT, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        if (Cfg::LDS_BYTES > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                Cfg::LDS_BYTES) != hipSuccess)
            return PNNP_E_LAUNCH;
        attr_set = true;
    }
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + Cfg::TH - 1) / Cfg::TH) * a.B * ((a.Ntot + BN - 1) / BN);
    if (tiles <= 0) return PNNP_OK;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

template <int TAPS>
int launch_taps(const IgemmArgs& a, int kc_chan, hipStream_t s) {
    // kc_chan: channels per segment (all segments equal).  BN from N, KC the largest of {32,16,8}
    // that divides the segment and keeps the weight tile <= 36 KB.
    const int bn = a.Ntot >= 128 ? 128 : (a.Ntot >= 64 ? 64 : 32);
    int kc = 1024 / bn;
    if (kc > 32) kc = 32;
    while (kc > 8 && (kc_chan % kc)) kc >>= 1;
    if (kc_chan % kc) return PNNP_E_UNSUPPORTED;
    if (bn == 32) {
        if (kc == 32) return launch_cfg<TAPS, 32, 32, 2, 1, 4, 1>(a, s);
        if (kc == 16) return launch_cfg<TAPS, 16, 32, 2, 1, 4, 1>(a, s);
        return launch_cfg<TAPS, 8, 32, 2, 1, 4, 1>(a, s);
    }
    if (bn == 64) {
        if (kc == 16) return launch_cfg<TAPS, 16, 64, 2, 2, 4, 1>(a, s);
        return launch_cfg<TAPS, 8, 64, 2, 2, 4, 1>(a, s);
    }
    return launch_cfg<TAPS, 8, 128, 2, 2, 2, 2>(a, s);
}

}  // namespace

int pnnp_igemm_launch(const IgemmArgs& a, int taps, int chan_per_seg, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 4 || chan_per_seg <= 0 || (chan_per_seg & 7) || a.Ntot <= 0) return PNNP_E_INVALID;
    IgemmArgs b = a;
    const int bn = a.Ntot >= 128 ? 128 : (a.Ntot >= 64 ? 64 : 32);
    int kc = 1024 / bn;
    if (kc > 32) kc = 32;
    while (kc > 8 && (chan_per_seg % kc)) kc >>= 1;
    b.chunks_per_seg = chan_per_seg / kc;
    if (taps == 9) return launch_taps<9>(b, chan_per_seg, s);
    if (taps == 1) return launch_taps<1>(b, chan_per_seg, s);
    return PNNP_E_UNSUPPORTED;
}
