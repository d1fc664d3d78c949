// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores, float32 operands split into three bf16
// pieces (the scheme of csrc/conv_x3.hip: a = hi + mid + lo exactly, six of the nine piece products, fp32 accumulation).
//
//   dW[m][n][t] = sum over pixels p of  G[p][m] * X[p + tap(t)][n]        (archs/Unet.py:16-50 via autograd;
//   G = dL/d(pre-activation output) [B][H][W][M = Cout], X = the layer's input [B][H][W][N = Cin], possibly the
//   un-materialised cat of two tensors), plus the bias gradient dbias[m] = sum_p G[p][m].
//
// GEMM view: M = Cout, N = Cin, K = pixels.  v_mfma_f32_32x32x16_bf16 wants, per lane, 8 consecutive k (pixels) of one
// channel, but the tensors are NHWC (a pixel's channels are contiguous).  The LDS images therefore stay pixel-major,
//   gs[32-channel block][piece][pixel][32 ch]      xs[32-channel block][piece][halo pixel][32 ch]          (bf16, 64-byte rows)
// and the operands are fetched with ds_read_b64_tr_b16, gfx950's transposing LDS read: per 16 lanes a 4-pixel x 16-channel
// block arrives channel-major, so two reads give a lane its 8 pixels of one channel of one piece (conflict-free on 64-byte
// rows: a 32-lane half touches 4 x 64 consecutive bytes).  A filter tap is a pixel offset into the halo image.
//
// One workgroup of 8 waves per CU owns an output tile of (32 WM) x (32 WN) channels x 9 taps and a strided share of the
// pixel tiles (TH rows x 32 px); a wave owns one 32 x 32 x 9 accumulator block (144 VGPRs) and 1/WK of a tile's 16-pixel
// k-steps.  Per k-step and wave: 6 transposed reads for G (shared by the 9 taps), 6 per tap for X, 54 MFMAs.
// Staging: fp32 tiles global -> registers one tile ahead (issued late in tile i for tile i+2, a full tile of flight time),
// split and written to the OTHER image between the MFMAs of the late part of tile i+1; one barrier per tile.
// Partial results go to per-workgroup slabs that a second kernel sums in a fixed order (deterministic, no float atomics).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef WX3_SPEC
#define WX3_SPEC 1               // 1: csrc/wgrad_x3s.hip (12 consumer + 4 producer waves; output tiles 64 x 64, 64 x 32, 32 x 64, 32 x 32); 0: the kernel below
#endif
struct Wx3sArgs { const float* G; int Gcs; const float* X[2]; int Xcs[2]; int n_split; int B, H, W, M, N; float* slab; float* bias_slab; int Z; };
int pnnp_wx3s_launch(const Wx3sArgs& a, hipStream_t s);           // csrc/wgrad_x3s.hip
int pnnp_wx3s_th(int M, int N);                                     // its pixel-tile height for (M, N)
struct Wh2sArgs { const float* G; int Gcs; const float* X[2]; int Xcs[2]; int n_split; int B, H, W, M, N; float* slab; float* bias_slab; int Z;
                  const unsigned* amax_g; const unsigned* amax_x[2]; };
int pnnp_wh2s_launch(const Wh2sArgs& a, hipStream_t s);           // csrc/wgrad_h2s.hip (fp16x2: same output tiles, same slabs, same reduce)
int pnnp_wh2s_th(int M, int N);                                     // its (taller) pixel tiles

namespace {

struct Wx3Args {
    const float* G; int Gcs;            // [B][H][W][Gcs], channels [0, M) used
    const float* X[2]; int Xcs[2];      // n < n_split -> X[0][n], else X[1][n - n_split]
    int n_split;
    int B, H, W, M, N;
    float* slab;                        // [Z][9][M][N]
    float* bias_slab;                   // [Z][M] or null
    int Z;
};

constexpr int NTHR = 512, NWAVE = 8;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {                  // f(integral_constant<int, I>) ... for I .. N - 1: every index a constant
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// Alternating signs over the pixel splits.  What was measured (tools/ubench/mfma_round.hip -> profiles/r3/mfma_round.txt, one MFMA with
// C = +-2^24, ulp 2; tools/x3_bias_probe.py -> profiles/r3/x3_bias_probe.txt), per instruction shape and operand case:
//   * equal small products (v_mfma_f32_32x32x16_bf16: 16 x v/16; v_mfma_f32_16x16x32_bf16: 32 x v/32): every PRODUCT is first rounded
//     to the nearest 1/8 ulp of the accumulator (0.25 here) -- 16 x 3/16 -> 16 x 0.25 = +4 (16777220), 32 x 3/32 -> 32 x 0 = +0
//     (16777216: the addend is dropped entirely), 32 x 5/32 -> 32 x 0.25 = +8 (16777224).  Both shapes follow the same rule; the
//     16 x 16 x 32 shape just has twice as many products per instruction to round.  Zero-mean for data of random size.
//   * one large + small products ("unequal" columns): the inexact SUM is rounded toward MINUS INFINITY whatever the signs
//     (16777219 -> 16777218, -16777219 -> -16777220, ties 16777217 -> 16777216, -16777217 -> -16777218); v_mfma_f32_32x32x2_f32 (an
//     fmaf chain) rounds to nearest.
// The second effect is a coherent downward drift over a long accumulation: nothing against a sum of K same-signed terms (4.4e-8), but
// against a gradient whose terms cancel (sum ~ sqrt(K) |term|) it reached 4e-6 relative at the top level's K = 16 x 512 x 512 pixels,
// 5x the fp32-MFMA kernel's error.
// The drift does not depend on the data's sign, so it cancels between two partial sums accumulated with OPPOSITE signs: workgroups
// with an odd pixel-split index z stage -G (one v_xor per value while splitting), their slabs hold -partial, and the reduce kernel
// adds the slabs with alternating signs.  No extra MFMA; the result is the same sum with the drift removed (to its fluctuation).
// (conv_x3's forward / backward-data reductions, K <= 9216, keep the drift: -1.4e-6 of the L2 norm at K = 4608, bounded by
// tests/test_gpu_x3.py::test_x3_forward_signed_mean_error_is_bounded.)
#ifndef WX3_ALT_SIGN
#define WX3_ALT_SIGN 1
#endif
#ifndef WX3_WIDE
#define WX3_WIDE 1               // 128 x 64 output tiles where the channel counts allow (round 4: -2 ... -3 % on those layers)
#endif
#ifndef WX3_M16
// Experiment of round 4, parity-green but 15-40 % SLOWER than the 32 x 32 x 16 path, so off: the 16 x 16 x 32
// shape needs twice the transposed reads (116 instead of 60 per k-step) and the 64 x 64 configuration has no registers for a deeper
// operand prefetch (256 with 16 spilled; B operands one 6-MFMA step = 96 cycles ahead): PMC on conv2_2 / conv4_2 -- matrix pipe busy
// 0.70 -> 0.42-0.50, SQ_WAIT_ANY 0.28 -> 0.43-0.52 of the wave cycles (waves parked on LDS latency), LDS bank conflicts 13 % of the active
// cycles (two-way on the staging writes of the split-half layout), LDS active 0.27 -> 0.32-0.37 (profiles/r4/wgrad_m16.txt).
#define WX3_M16 0
#endif

template <int WM, int WN, int TH>
struct Wx3Cfg {
    static constexpr int WK = NWAVE / (WM * WN);
    static constexpr int KS = TH * 2 / WK;                          // 16-pixel k-steps per wave and tile
    static constexpr int GPIX = TH * 32, XR = TH + 2, XC = 34, XPIX = XR * XC;
    static constexpr int G_BYTES = WM * 3 * GPIX * 64, X_BYTES = WN * 3 * XPIX * 64;
    static constexpr int IMG_BYTES = G_BYTES + X_BYTES;
    static constexpr int LDS_BYTES = 2 * IMG_BYTES;
    // staging slots (one float4 = 4 channels of a pixel): a 32-channel block of G / X is staged by the waves with wave % WM == block
    static constexpr int G_THR = NTHR / WM, X_THR = NTHR / WN;
    static constexpr int NG = (GPIX * 8 + G_THR - 1) / G_THR, NX = (XPIX * 8 + X_THR - 1) / X_THR;
    // the 32 x 32 output tile (one k-step of 9 groups per tile, six staging slices) measured 5 % faster with its staging slices as lumps
    // behind the MFMA groups than with everything placed between the individual MFMAs; the larger tiles 4-8 % slower
    static constexpr bool LUMPS = WM * WN == 1;
    // v_mfma_f32_16x16x32_bf16 with two PIECES concatenated along K (the trick of csrc/conv_x3.hip, X3_M16): the same multiply-adds per
    // cycle at less energy per FLOP, and the chip is power-limited in these loops (round 4: a kernel squatting on 32 of the 256 CUs costs a
    // convolution layer nothing -- the other 224 clock higher).  K = 32 of one instruction = 16 pixels of piece X ++ the same 16 pixels of
    // piece Y; a 32 x 32 block is four 16 x 16 blocks x three instructions.  The LDS images then keep the two 16-channel halves of a
    // 32-channel block in SEPARATE planes, [block][piece][half][pixel][32 B]: one ds_read_b64_tr_b16 instruction covers, in each 32-lane
    // half, 8 consecutive pixel rows of ONE 16-channel half -- 256 contiguous bytes, conflict-free (on 64-byte rows, pixel rows 4 apart
    // share their banks) -- by giving k-block (lane >> 4) & 1 the pixels {0-3, 8-11} or {4-7, 12-15} of the k-step (A and B agree on it).
    static constexpr bool M16 = WX3_M16 && !LUMPS;
    static_assert(WM * WN * WK == NWAVE && KS >= 1 && KS * WK == TH * 2, "wave layout");
    static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= NWAVE * 16 * 64 * 4, "LDS budget (images; reduction scratch aliases them)");
};

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // RNE, low half = a
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

template <int WM, int WN, int TH>
__global__ void __launch_bounds__(NTHR)
wgrad_x3_kernel(const Wx3Args a) {
    using Cfg = Wx3Cfg<WM, WN, TH>;
    constexpr int WK = Cfg::WK, KS = Cfg::KS, GPIX = Cfg::GPIX, XC = Cfg::XC, XPIX = Cfg::XPIX, NG = Cfg::NG, NX = Cfg::NX;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wk = wave % WK, wno = (wave / WK) % WN, wmo = wave / (WK * WN);

    const int n_tiles = a.N / (32 * WN);
    int id = blockIdx.x;
    const int z = id % a.Z; id /= a.Z;
    const int ni = id % n_tiles, mi = id / n_tiles;
    const int m0 = mi * 32 * WM, n0 = ni * 32 * WN;

    const int tiles_x = (a.W + 31) >> 5, tiles_y = (a.H + TH - 1) / TH;
    const int ntile = tiles_x * tiles_y * a.B;

    // ---- staging pattern.  G: block gb = wave % WM, slot j = (wave / WM) * 64 + lane + G_THR * k over (pixel j >> 3, quad j & 7);
    //      X: block xb = wave % WN likewise over the halo pixels.  Slots past the end repeat the thread's previous slot.
    const int gb = wave % WM, xb = wave % WN;
    const int q8 = lane & 7;                                        // this thread's channel quad inside its 32-channel block
    int g_r[NG], g_c[NG]; unsigned g_off[NG]; int g_dst[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        int j = (wave / WM) * 64 + lane + Cfg::G_THR * k;
        if (j >= GPIX * 8) j -= Cfg::G_THR;
        const int pix = j >> 3;
        g_r[k] = pix >> 5; g_c[k] = pix & 31;
        g_off[k] = (unsigned)((g_r[k] * a.W + g_c[k]) * a.Gcs + q8 * 4) * 4u;
        g_dst[k] = Cfg::M16 ? gb * 3 * GPIX * 64 + ((q8 >> 2) * GPIX + pix) * 32 + (q8 & 3) * 8      // [piece][half][pixel][32 B]
                            : (gb * 3 * GPIX + pix) * 64 + q8 * 8;                                    // byte offset in an image; + piece * GPIX * 64
    }
    const int xd = (n0 + 32 * xb >= a.n_split) ? 1 : 0;             // wave-uniform source of this wave's X block
    const int xch0 = n0 + 32 * xb - (xd ? a.n_split : 0);
    const int xcs = a.Xcs[xd];
    int x_r[NX], x_c[NX]; unsigned x_off[NX]; int x_dst[NX];
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        int j = (wave / WN) * 64 + lane + Cfg::X_THR * k;
        if (j >= XPIX * 8) j -= Cfg::X_THR;
        const int pix = j >> 3;
        x_r[k] = pix / XC; x_c[k] = pix - x_r[k] * XC;              // halo coordinates, 0-based: image pixel (y0 - 1 + r, x0 - 1 + c)
        x_off[k] = (unsigned)((x_r[k] * a.W + x_c[k]) * xcs + q8 * 4) * 4u;
        x_dst[k] = Cfg::G_BYTES + (Cfg::M16 ? xb * 3 * XPIX * 64 + ((q8 >> 2) * XPIX + pix) * 32 + (q8 & 3) * 8 : (xb * 3 * XPIX + pix) * 64 + q8 * 8);
    }
    const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.G + m0 + 32 * gb), 0, 0x7fffffff, 0x00020000);
    // (the X resource starts one row + one pixel BEFORE the tensor so that the scalar offset of a halo tile is never negative)
    const int xshift = (a.W + 1) * xcs;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X[xd] + xch0 - xshift), 0, 0x7fffffff, 0x00020000);
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };

    f32x4 rg[NG], rx[NX];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned sflip = (WX3_ALT_SIGN && (z & 1)) ? 0x80000000u : 0u;    // odd pixel splits accumulate -G * X (see WX3_ALT_SIGN)
    auto gsign = [&](f32x4 v) {
        return f32x4{__uint_as_float(__float_as_uint(v.x) ^ sflip), __uint_as_float(__float_as_uint(v.y) ^ sflip),
                     __uint_as_float(__float_as_uint(v.z) ^ sflip), __uint_as_float(__float_as_uint(v.w) ^ sflip)};
    };
    auto load_tile = [&](int tile) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int x0 = tx * 32, y0 = ty * TH;
        const int gso = (((b * a.H + y0) * a.W) + x0) * a.Gcs * 4;
        const int xso = ((((b * a.H + y0 - 1) * a.W) + x0 - 1) * xcs + xshift) * 4;
        const int rlim = a.H - y0, clim = a.W - x0;
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int bad = (rlim - 1 - g_r[k]) | (clim - 1 - g_c[k]);                 // sign bit set <=> pixel outside the image
            rg[k] = bload(rsg, bad < 0 ? OOB : g_off[k], gso);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int yy = y0 - 1 + x_r[k], xx = x0 - 1 + x_c[k];
            const int bad = yy | (a.H - 1 - yy) | xx | (a.W - 1 - xx);
            rx[k] = bload(rsx, bad < 0 ? OOB : x_off[k], xso);
        }
    };
    // one staging slice: split slot s (G slots first, then X) and write its three 8-byte words to image `img`
    auto stage_slice = [&](int s, int img) {
        char* ib = smem + img * Cfg::IMG_BYTES;
        const bool isg = s < NG;
        const f32x4 v = isg ? gsign(rg[isg ? s : 0]) : rx[isg ? 0 : s - NG];
        const int dst = isg ? g_dst[isg ? s : 0] : x_dst[isg ? 0 : s - NG];
        const int pstride = (isg ? GPIX : XPIX) * 64;
        unsigned h0, m0_, l0, h1, m1, l1;
        split2(v.x, v.y, h0, m0_, l0);
        split2(v.z, v.w, h1, m1, l1);
        const u32x2 h = {h0, h1}, m = {m0_, m1}, l = {l0, l1};
        *reinterpret_cast<u32x2*>(ib + dst) = h;
        *reinterpret_cast<u32x2*>(ib + dst + pstride) = m;
        *reinterpret_cast<u32x2*>(ib + dst + 2 * pstride) = l;
        if (isg && a.bias_slab) {                                    // bias gradient: column sums of G, gathered while it is staged
            // (a duplicated last slot would count twice: only configurations with exact G slot counts exist, see static_assert below)
            bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;
        }
    };
    static_assert((GPIX * 8) % Cfg::G_THR == 0, "G slots must divide evenly (bias sums count every pixel once)");
    // the same slice as dependent pieces of 2-4 VALU instructions (step 0 .. 9) and the stores (step 10, 11): one or two per MFMA gap
    f32x4 pv; unsigned ph[2], pm[2], pl[2];
    float bmul = 1.f;                                             // 0 while there is no next tile: the pieces then run on stale registers, branch-free
    auto stage_piece = [&](int sl, int step, int img) {
        const bool isg = sl < NG;
        switch (step) {
        case 0: pv = isg ? gsign(rg[isg ? sl : 0]) : rx[isg ? 0 : sl - NG]; ph[0] = cvt_pk_bf16(pv.x, pv.y); ph[1] = cvt_pk_bf16(pv.z, pv.w); break;
        case 1: if (isg) { bsum[0] = fmaf(pv.x, bmul, bsum[0]); bsum[1] = fmaf(pv.y, bmul, bsum[1]); bsum[2] = fmaf(pv.z, bmul, bsum[2]); bsum[3] = fmaf(pv.w, bmul, bsum[3]); } break;
        case 2: pv.x -= __uint_as_float(ph[0] << 16); pv.y -= __uint_as_float(ph[0] & 0xffff0000u); break;
        case 3: pv.z -= __uint_as_float(ph[1] << 16); pv.w -= __uint_as_float(ph[1] & 0xffff0000u); break;
        case 4: pm[0] = cvt_pk_bf16(pv.x, pv.y); pm[1] = cvt_pk_bf16(pv.z, pv.w); break;
        case 5: pv.x -= __uint_as_float(pm[0] << 16); pv.y -= __uint_as_float(pm[0] & 0xffff0000u); break;
        case 6: pv.z -= __uint_as_float(pm[1] << 16); pv.w -= __uint_as_float(pm[1] & 0xffff0000u); break;
        case 7: pl[0] = cvt_pk_bf16(pv.x, pv.y); pl[1] = cvt_pk_bf16(pv.z, pv.w); break;
        default: {
            char* ib = smem + img * Cfg::IMG_BYTES;
            const int dst = isg ? g_dst[isg ? sl : 0] : x_dst[isg ? 0 : sl - NG];
            const int pstride = (isg ? GPIX : XPIX) * 64;
            if (step == 8) *reinterpret_cast<u32x2*>(ib + dst) = u32x2{ph[0], ph[1]};
            if (step == 9) *reinterpret_cast<u32x2*>(ib + dst + pstride) = u32x2{pm[0], pm[1]};
            if (step == 10) *reinterpret_cast<u32x2*>(ib + dst + 2 * pstride) = u32x2{pl[0], pl[1]};
        } break;
        }
    };

    // accumulators of the wave's 32 x 32 x 9 block: 32 x 32 x 16 shape: acc[t][r]; 16 x 16 x 32 shape: acc[t][4 (2 mh + nh) + r] = the
    // 16 x 16 block (rows 16 mh .., columns 16 nh ..): lane l holds rows 4 (l >> 4) + r, column l & 15
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // transposed-read lane geometry: 16-lane group g = lane >> 4 reads channels 16 (g & 1) .., pixels 8 (g >> 1) ..; inside a
    // group lane 4 q + p supplies the address of pixel row q, channel chunk 4 p
    const int tr_lane = ((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto tr_read = [&](const char* base) {                           // 8 pixels x 1 channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * 64));
        u32x4 r;
        const u32x2 a0 = __builtin_bit_cast(u32x2, lo), a1 = __builtin_bit_cast(u32x2, hi);
        r.x = a0.x; r.y = a0.y; r.z = a1.x; r.w = a1.y;
        return r;
    };

    constexpr int NSL = NG + NX;                                    // staging slices per tile
    constexpr int NGRP = KS * 9;                                     // groups of six MFMAs per tile and wave
    constexpr int SL0 = NGRP - 1 - NSL;                              // first group that carries a slice (the last group carries the loads)
    static_assert(SL0 >= 0, "more staging slices than MFMA groups");

    if (z >= ntile) {                                               // (more pixel splits than tiles never happens: Z <= ntile)
        return;
    }
    // ---- prologue: the first tile goes straight into image 0, the second one into registers
    load_tile(z);
    __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int s = 0; s < NSL; ++s) stage_slice(s, 0);
    if (z + a.Z < ntile) load_tile(z + a.Z);
    int img = 0;
#ifdef WX3_STAMPS                 // debug build: cycle sums per wave (barrier wait / MFMA phase / tail), dumped into the slab at the end
    long long tb = 0, tm = 0, tw = 0, tall = clock64(), tlast_ = clock64(); int ntl = 0;
    long long gstamp[NGRP + 1] = {}, gt0 = 0;
#define WX3_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define WX3_T(v)
#endif
    for (int tile = z; tile < ntile; tile += a.Z) {
        WX3_T(tm)
        __syncthreads();                                            // image `img` is complete; every wave is done with the other one
        WX3_T(tb)
#ifdef WX3_STAMPS
        if (ntl == 5) gt0 = clock64();
        if (ntl == 6) gstamp[NGRP] = clock64();
#endif
        const char* gimg = smem + img * Cfg::IMG_BYTES;
        const char* ximg = gimg + Cfg::G_BYTES;
        const bool have_next = tile + a.Z < ntile, have_next2 = tile + 2 * a.Z < ntile;
        if constexpr (Cfg::M16) {
        // lane geometry of the transposed reads: 16-lane group g = lane >> 4 is k-block g: piece X (g < 2) or Y, pixels {0-3, 8-11} (g even)
        // or {4-7, 12-15} of the k-step; inside a group lane 4 q + p supplies the address of pixel row q, channel chunk 4 p
        const int lb = ((((lane >> 4) & 1) * 4 + ((lane & 15) >> 2)) * 32) + (lane & 3) * 8, hi2 = lane >> 5;
        constexpr int PSG = GPIX * 64, PSX = XPIX * 64;             // piece strides
        // operand forms: A0 = [hi | mid], A1 = [hi | lo];  B0 = [hi' | hi'], B1 = [mid' | mid'], B2 = [lo' | hi']
        //   A1 B2 = hi lo' + lo hi',  A0 B1 = hi mid' + mid mid',  A0 B0 = hi hi' + mid hi'   (smallest terms first)
        const int la0 = lb + hi2 * PSG, la1 = lb + hi2 * 2 * PSG, lb2 = lb + (1 - hi2) * 2 * PSX;
        auto half_read = [&](const char* base) {                     // 4 pixels x 1 channel per lane: one transposed read
            return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        };
        auto a_base = [&](int ks, int mh, int f) {
            const int kk = wk * KS + ks;
            return gimg + (wmo * 3) * PSG + (mh * GPIX + (kk >> 1) * 32 + (kk & 1) * 16) * 32 + (f ? la1 : la0);
        };
        auto b_base = [&](int ks, int t, int nh, int f) {
            const int kk = wk * KS + ks, dy = t / 3, dx = t - 3 * dy;
            return ximg + (wno * 3) * PSX + (nh * XPIX + ((kk >> 1) + dy) * XC + (kk & 1) * 16 + dx) * 32 + (f == 0 ? lb : (f == 1 ? lb + PSX : lb2));
        };
        bmul = have_next ? 1.f : 0.f;                                // (without a next tile the staging below rewrites the idle image from stale registers: harmless)
        u32x4 Av[2][2][2], Bv[2][3];                                 // [k-step parity][mh][form], [step parity][form]
        auto put_half = [](u32x4& d, int h, u32x2 v) { if (h == 0) { d.x = v.x; d.y = v.y; } else { d.z = v.x; d.w = v.y; } };
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int f = 0; f < 2; ++f) { put_half(Av[0][mh][f], 0, half_read(a_base(0, mh, f))); put_half(Av[0][mh][f], 1, half_read(a_base(0, mh, f) + 256)); }
#pragma unroll
        for (int f = 0; f < 3; ++f) { put_half(Bv[0][f], 0, half_read(b_base(0, 0, 0, f))); put_half(Bv[0][f], 1, half_read(b_base(0, 0, 0, f) + 256)); }
        constexpr int NSTEP = KS * 18;                               // steps (tap, n-half) of six MFMAs per tile and wave
        constexpr int SL0M = NSTEP - 1 - NSL;                        // first step that carries a staging slice
        static_assert(SL0M >= 0, "more staging slices than steps");
        static_for<0, NSTEP * 6>([&](auto GI) {
            constexpr int gi = decltype(GI)::value, S = gi / 6, G = gi % 6, ks = S / 18, st = S % 18, t = st / 2, nh = st % 2, mh = G / 3, pr = G % 3;
            constexpr int cur = S & 1;
            if constexpr (G == 0 && S == SL0M) __builtin_amdgcn_s_waitcnt(0x0f70);   // the next tile's loads were issued a full tile ago
            if constexpr (G == 0) __builtin_amdgcn_sched_barrier(0);
            f32x4 c4 = {acc[t][4 * (2 * mh + nh)], acc[t][4 * (2 * mh + nh) + 1], acc[t][4 * (2 * mh + nh) + 2], acc[t][4 * (2 * mh + nh) + 3]};
            if constexpr (pr == 0) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][1]), __builtin_bit_cast(bf16x8, Bv[cur][2]), c4, 0, 0, 0);
            else if constexpr (pr == 1) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][0]), __builtin_bit_cast(bf16x8, Bv[cur][1]), c4, 0, 0, 0);
            else c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][0]), __builtin_bit_cast(bf16x8, Bv[cur][0]), c4, 0, 0, 0);
            acc[t][4 * (2 * mh + nh)] = c4.x; acc[t][4 * (2 * mh + nh) + 1] = c4.y; acc[t][4 * (2 * mh + nh) + 2] = c4.z; acc[t][4 * (2 * mh + nh) + 3] = c4.w;
            // the next step's B operand: one transposed read per gap (form G / 2, half G % 2)
            if constexpr (S + 1 < NSTEP) {
                constexpr int S1 = S + 1, ks1 = S1 / 18, t1 = (S1 % 18) / 2, nh1 = S1 % 2;
                put_half(Bv[cur ^ 1][G >> 1], G & 1, half_read(b_base(ks1, t1, nh1, G >> 1) + (G & 1) * 256));
            }
            // the next k-step's A operands: eight reads in the gaps of its predecessor's last two steps
            if constexpr (ks + 1 < KS && st >= 16) {
                constexpr int j = (st - 16) * 6 + G;
                if constexpr (j < 8) put_half(Av[(ks + 1) & 1][j >> 2][(j >> 1) & 1], j & 1, half_read(a_base(ks + 1, j >> 2, (j >> 1) & 1) + (j & 1) * 256));
            }
            if constexpr (S >= SL0M && S < SL0M + NSL) { stage_piece(S - SL0M, 2 * G, img ^ 1); if constexpr (2 * G + 1 < 11) stage_piece(S - SL0M, 2 * G + 1, img ^ 1); }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (have_next2) load_tile(tile + 2 * a.Z);                  // registers are free again: the tile after next
        } else
        if constexpr (!Cfg::LUMPS) {
        // Operand reads and the staging pieces sit BETWEEN the individual MFMAs (fenced).  The SIMD's vector issue is the shared
        // resource: an MFMA holds it for 8 of its 32 cycles, a VALU instruction for 4-5, and DEPENDENT VALU instructions back to back
        // wait for each other -- a staging slice as one lump behind its group of MFMAs (30 dependent VALU + 3 stores) took ~300 cycles
        // during which the wave issued no MFMA (cycle stamps: 280 cycles per 6-MFMA group without a slice, 500-530 with one).
        auto g_addr = [&](int ks) {
            const int kk = wk * KS + ks;
            return gimg + ((wmo * 3) * GPIX + (kk >> 1) * 32 + (kk & 1) * 16) * 64 + tr_lane;
        };
        auto x_addr = [&](int ks, int t) {
            const int kk = wk * KS + ks, dy = t / 3, dx = t - 3 * dy;
            return ximg + ((wno * 3) * XPIX + ((kk >> 1) + dy) * XC + (kk & 1) * 16 + dx) * 64 + tr_lane;
        };
        bmul = have_next ? 1.f : 0.f;                                // (without a next tile the staging below rewrites the idle image from stale registers: harmless)
        u32x4 av[2][3], bv[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { av[0][p] = tr_read(g_addr(0) + p * GPIX * 64); bv[0][p] = tr_read(x_addr(0, 0) + p * XPIX * 64); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int grp = ks * 9 + t, cur = grp & 1;
                const bool more = grp + 1 < NGRP;                     // another (k-step, tap) follows in this tile
                const int nks = t + 1 < 9 ? ks : ks + 1, nt = t + 1 < 9 ? t + 1 : 0;
                const bool fill = grp >= SL0 && grp < SL0 + NSL;
#ifdef WX3_STAMPS
                if (ntl == 5) gstamp[grp] = clock64();
#endif
                if (grp == SL0) __builtin_amdgcn_s_waitcnt(0x0f70);        // the next tile's loads were issued a full tile ago
                __builtin_amdgcn_sched_barrier(0);
#define WX3_MFMA(PA, PB) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ks & 1][PA]), __builtin_bit_cast(bf16x8, bv[cur][PB]), acc[t], 0, 0, 0)
#define WX3_GAP(G) { if (more && G < 3) bv[cur ^ 1][G] = tr_read(x_addr(nks, nt) + G * XPIX * 64);                                          \
                     if (more && t == 8 && G >= 3) av[(ks + 1) & 1][G - 3] = tr_read(g_addr(ks + 1) + (G - 3) * GPIX * 64);                \
                     if (fill) { stage_piece(grp - SL0, 2 * G, img ^ 1); if (2 * G + 1 < 11) stage_piece(grp - SL0, 2 * G + 1, img ^ 1); } \
                     __builtin_amdgcn_sched_barrier(0); }
                // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
                WX3_MFMA(0, 2); WX3_GAP(0)
                WX3_MFMA(2, 0); WX3_GAP(1)
                WX3_MFMA(1, 1); WX3_GAP(2)
                WX3_MFMA(0, 1); WX3_GAP(3)
                WX3_MFMA(1, 0); WX3_GAP(4)
                WX3_MFMA(0, 0); WX3_GAP(5)
#undef WX3_GAP
#undef WX3_MFMA
                if (grp == NGRP - 1 && have_next2) load_tile(tile + 2 * a.Z);   // registers are free again: the tile after next
            }
        }
        } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int kk = wk * KS + ks;                            // k-step of the tile: pixel row kk >> 1, pixels 16 (kk & 1) ..
            const int r = kk >> 1, c0 = (kk & 1) * 16;
            const char* gbase = gimg + ((wmo * 3) * GPIX + r * 32 + c0) * 64 + tr_lane;
            u32x4 av[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) av[p] = tr_read(gbase + p * GPIX * 64);
            u32x4 bv[2][3];
            auto xload = [&](int t, u32x4 (&bx)[3]) {
                const int dy = t / 3, dx = t - 3 * dy;
                const char* xbase = ximg + ((wno * 3) * XPIX + (r + dy) * XC + c0 + dx) * 64 + tr_lane;
#pragma unroll
                for (int p = 0; p < 3; ++p) bx[p] = tr_read(xbase + p * XPIX * 64);
            };
            xload(0, bv[0]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) xload(t + 1, bv[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&bx)[3] = bv[t & 1];
                // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
#define WX3_MFMA(PA, PB) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[PA]), __builtin_bit_cast(bf16x8, bx[PB]), acc[t], 0, 0, 0)
                WX3_MFMA(0, 2); WX3_MFMA(2, 0); WX3_MFMA(1, 1); WX3_MFMA(0, 1); WX3_MFMA(1, 0); WX3_MFMA(0, 0);
#undef WX3_MFMA
                __builtin_amdgcn_sched_barrier(0);
                const int grp = ks * 9 + t;
#ifdef WX3_STAMPS
                if (ntl == 5) gstamp[grp] = clock64();
#endif
                if (grp >= SL0 && grp < SL0 + NSL) {                // the late groups carry the next tile's staging, one slice each
                    if (grp == SL0) {
#ifdef WX3_STAMPS
                        WX3_T(tm)
#endif
                        __builtin_amdgcn_s_waitcnt(0x0f70);                 // its loads were issued a full tile ago
#ifdef WX3_STAMPS
                        WX3_T(tw)
#endif
                    }
                    if (have_next) stage_slice(grp - SL0, img ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (grp == NGRP - 1 && have_next2) load_tile(tile + 2 * a.Z);   // registers are free again: the tile after next
            }
        }
        }
        img ^= 1;
#ifdef WX3_STAMPS
        ++ntl;
#endif
    }
#ifdef WX3_STAMPS
    WX3_T(tm)
    const long long tloop = clock64() - tall;
#endif

    // ---- reduce the WK pixel-split waves through LDS (the images are dead now), then write the slab [z][tap][m][n]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    const int64_t slab_base = (int64_t)z * a.M * a.N * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        f32x16 v = acc[t];
        if (WK > 1) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = v[r];
            __syncthreads();
            if (wk == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = 0.f;
#pragma unroll
                    for (int k = 0; k < WK; ++k) s += red[((wave + k) * 16 + r) * 64 + lane];
                    v[r] = s;
                }
            }
        }
        if (wk == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // 32 x 32 x 16 shape: column l31, row (r & 3) + 8 (r >> 2) + 4 half;  16 x 16 x 32 shape: block (mh, nh) = (r >> 3, (r >> 2) & 1),
                // column 16 nh + (lane & 15), row 16 mh + 4 (lane >> 4) + (r & 3)
                const int n = n0 + wno * 32 + (Cfg::M16 ? 16 * ((r >> 2) & 1) + (lane & 15) : l31);
                const int m = m0 + wmo * 32 + (Cfg::M16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * half);
                a.slab[slab_base + ((int64_t)t * a.M + m) * a.N + n] = v[r];
            }
        }
    }
    if (a.bias_slab && ni == 0) {                                   // block-uniform
        // a thread summed 4 channels (quad q8 of block gb) over its pixels: add up the threads that share (gb, q8)
        __syncthreads();
        float* bs = reinterpret_cast<float*>(smem);                 // [WM][8 quads][4] partial sums per contributing thread slot
        // threads with the same (gb, q8): tid' = all with wave % WM == gb and lane & 7 == q8: (NWAVE / WM) waves x 8 lanes
        const int slot = (wave / WM) * 8 + (lane >> 3);             // 0 .. (NWAVE / WM) * 8 - 1
        constexpr int NSLOTS = (NWAVE / WM) * 8;
#pragma unroll
        for (int c = 0; c < 4; ++c) bs[((gb * 8 + q8) * 4 + c) * NSLOTS + slot] = bsum[c];
        __syncthreads();
        if (tid < WM * 32) {
            const int blk = tid >> 5, ch = tid & 31;
            float s = 0.f;
            for (int k = 0; k < NSLOTS; ++k) s += bs[((blk * 8 + (ch >> 2)) * 4 + (ch & 3)) * NSLOTS + k];
            a.bias_slab[(int64_t)z * a.M + m0 + blk * 32 + ch] = s;
        }
    }
#ifdef WX3_STAMPS
    __syncthreads();
    if (lane == 0) {
        float* d = a.slab + ((int64_t)blockIdx.x * NWAVE + wave) * 8;
        d[0] = (float)tb; d[1] = (float)tm; d[2] = (float)tloop; d[3] = (float)(clock64() - tall); d[4] = (float)ntl; d[5] = (float)tw;
        if (blockIdx.x == 0) { float* e = a.slab + 256 * NWAVE * 8 + wave * 32; for (int i = 0; i <= NGRP; ++i) e[i] = (float)(gstamp[i] - gt0); }
    }
#endif
}

// out[o(i)] (+)= sum_z (-1)^z slab[z][i] (odd splits accumulated -G * X: WX3_ALT_SIGN); i = (t * M + m) * N + n  ->  o = (m * N + n) * taps + t
// (the parameter's own layout).  ONE launch reduces the weight slabs and, behind them (i >= n), the bias slabs [Z][nb] into bias_out.
__global__ void __launch_bounds__(256)
wx3_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int Z, int accumulate, int64_t mn, int taps,
                  const float* __restrict__ bias_slab, float* __restrict__ bias_out, int nb) {
    __shared__ float red[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float sg = (WX3_ALT_SIGN && (ty & 1)) ? -1.f : 1.f;      // a thread row sums splits ty, ty + 8, ...: one parity, one sign (odd splits hold -partial)
    const int64_t ntot = n + (bias_slab ? nb : 0);
    for (int64_t i0 = (int64_t)blockIdx.x * 32; i0 < ntot; i0 += (int64_t)gridDim.x * 32) {
        const int64_t i = i0 + tx;
        const bool isb = i >= n;                                    // (n is a multiple of 32: a 32-wide group never straddles the two parts)
        const float* src = isb ? bias_slab + (i - n) : slab + i;
        const int64_t zs = isb ? nb : n;
        // (eight independent loads in flight per thread: with two, a layer with few weights and many slabs -- conv1_2: 9216 x 256 --
        //  was one memory latency per pair, 11 us for 9 MB)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (i < ntot) {
            int zz = ty;
            for (; zz + 56 < Z; zz += 64) {
                const float v0 = src[(int64_t)zz * zs], v1 = src[(int64_t)(zz + 8) * zs], v2 = src[(int64_t)(zz + 16) * zs], v3 = src[(int64_t)(zz + 24) * zs];
                const float v4 = src[(int64_t)(zz + 32) * zs], v5 = src[(int64_t)(zz + 40) * zs], v6 = src[(int64_t)(zz + 48) * zs], v7 = src[(int64_t)(zz + 56) * zs];
                s0 += v0; s1 += v1; s2 += v2; s3 += v3; s0 += v4; s1 += v5; s2 += v6; s3 += v7;
            }
            for (; zz + 8 < Z; zz += 16) { s0 += src[(int64_t)zz * zs]; s1 += src[(int64_t)(zz + 8) * zs]; }
            for (; zz < Z; zz += 8) s0 += src[(int64_t)zz * zs];
        }
        red[ty][tx] = sg * ((s0 + s1) + (s2 + s3));
        __syncthreads();
        if (ty == 0 && i < ntot) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k][tx];
            float* o = isb ? bias_out + (i - n) : out + (taps == 1 ? i : (i % mn) * taps + i / mn);
            *o = accumulate ? *o + s : s;
        }
        __syncthreads();
    }
}

template <int WM, int WN, int TH>
int launch_wx3(const Wx3Args& a, hipStream_t s) {
    using Cfg = Wx3Cfg<WM, WN, TH>;
    auto kern = wgrad_x3_kernel<WM, WN, TH>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int blocks = (a.M / (32 * WM)) * (a.N / (32 * WN)) * a.Z;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

// output-tile shape: 64 x 64 (two pixel splits inside the workgroup), 64 x 32 / 32 x 64 (four), 32 x 32 (eight)
// shape 4 (WX3_WIDE, round 4): 128 x 64 with no pixel split inside the workgroup and one-row pixel tiles -- six staging slices instead of seven
// per 108 MFMAs, 253 registers without a spill (the 64 x 64 configuration: 256 with three): -2 ... -3 % on conv3 .. conv8 (profiles/r4/wgrad_wide.txt)
int wx3_shape(int M, int N) {
#if WX3_WIDE
    if (M % 128 == 0 && N % 64 == 0) return 4;
#endif
    return (M % 64 == 0 ? 1 : 0) + (N % 64 == 0 ? 2 : 0);
}
int wx3_th(int shape) { return shape == 4 ? 1 : (shape == 0 ? 4 : 2); }

bool wx3_spec(int, int) { return WX3_SPEC != 0; }

int wx3_splits(int B, int H, int W, int M, int N) {
    const int shape = wx3_shape(M, N);
    const bool spec = wx3_spec(M, N);                               // csrc/wgrad_x3s.hip: its own tiles
    const int bm = spec ? (M % 64 == 0 ? 64 : 32) : (shape == 4 ? 128 : ((shape & 1) ? 64 : 32)), bn = spec ? (N % 64 == 0 ? 64 : 32) : (shape == 4 ? 64 : ((shape & 2) ? 64 : 32));
    const int th = spec ? pnnp_wx3s_th(M, N) : wx3_th(shape);
    const int tiles = ((W + 31) / 32) * ((H + th - 1) / th) * B;
    const int out_tiles = (M / bm) * (N / bn);
    int cus = pnnp_device_cus();
    if (cus <= 0) cus = 256;
    int z = (cus + out_tiles - 1) / out_tiles;                      // one 8-wave workgroup per CU
    if (z > tiles) z = tiles;
    return z < 1 ? 1 : z;
}

}  // namespace

// slabs [Z][taps][M][N] (+ bias slabs [Z][nb] behind them) -> dW in the parameter's layout [M][N][taps] (+ dbias), alternating signs over z
// (also used by csrc/wgrad_x3g.hip)
int pnnp_wx3_reduce_launch(const float* slab, float* dW, int64_t mn, int taps, int Z, int accumulate,
                           const float* bias_slab, float* dbias, int nb, hipStream_t st) {
    const int64_t n = mn * taps, ntot = n + (bias_slab ? nb : 0);       // n % 32 == 0 (channels in multiples of 32)
    hipLaunchKernelGGL(wx3_reduce_kernel, dim3((unsigned)((ntot + 31) / 32 > 4096 ? 4096 : (ntot + 31) / 32)), dim3(256), 0, st,
                       slab, dW, n, Z, accumulate, mn, taps, bias_slab, dbias, nb);
    return pnnp_launch_status();
}

extern "C" {

int pnnp_x3_wgrad_supported(int H, int W, int Cout, int C1, int C2) {
    return (H > 0 && W > 0 && Cout > 0 && C1 > 0 && Cout % 32 == 0 && C1 % 32 == 0 && C2 % 32 == 0) ? 1 : 0;
}

// the backward-weight kernel addresses the WHOLE batch of a map through one 32-bit byte offset: does [B][H][W][cstride] fit?
int pnnp_x3_wgrad_fits(int B, int H, int W, int cstride) {
    return (B > 0 && H > 0 && W > 0 && cstride > 0 && ((int64_t)B * H + 2) * W * cstride * 4 < (1ll << 31)) ? 1 : 0;
}

int64_t pnnp_x3_wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin) {
    if (Cout % 32 || Cin % 32) return 0;
    return (int64_t)wx3_splits(B, H, W, Cout, Cin) * ((int64_t)9 * Cout * Cin + Cout);
}

// dW [Cout][C1+C2][3][3] (+ dbias [Cout]) of a 3x3 / stride 1 / pad 1 convolution; same contract as pnnp_conv_bwd_weight_f32
// with taps = 9 (g: dL/d(pre-activation output), x1 / x2: the layer's input(s)).
int pnnp_conv3x3_x3_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                                   const float* x2, int x2_cs, int C2, float* dW, float* dbias,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    if (!g || !x1 || !dW || !workspace || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    const int N = C1 + (x2 ? C2 : 0);
    if (!pnnp_x3_wgrad_supported(H, W, Cout, C1, x2 ? C2 : 0)) return PNNP_E_UNSUPPORTED;
    if (g_cs < Cout || x1_cs < C1 || (x2 && x2_cs < C2) || (g_cs & 3) || (x1_cs & 3) || (x2 && (x2_cs & 3))) return PNNP_E_INVALID;
    if ((((uintptr_t)g) | ((uintptr_t)x1) | ((uintptr_t)x2)) & 15) return PNNP_E_INVALID;
    // 32-bit byte offsets into the whole tensors (bit 31 marks "outside")
    const int cmax = g_cs > x1_cs ? g_cs : x1_cs;
    if (!pnnp_x3_wgrad_fits(B, H, W, cmax) || (x2 && !pnnp_x3_wgrad_fits(B, H, W, x2_cs))) return PNNP_E_UNSUPPORTED;
    if (workspace_floats < pnnp_x3_wgrad_workspace_floats(B, H, W, Cout, N)) return PNNP_E_WORKSPACE;
    hipStream_t st = as_stream(stream);
    Wx3Args a{};
    a.G = g; a.Gcs = g_cs;
    a.X[0] = x1; a.Xcs[0] = x1_cs; a.X[1] = x2 ? x2 : x1; a.Xcs[1] = x2 ? x2_cs : x1_cs; a.n_split = x2 ? C1 : (1 << 30);
    a.B = B; a.H = H; a.W = W; a.M = Cout; a.N = N;
    a.Z = wx3_splits(B, H, W, Cout, N);
    a.slab = workspace;
    a.bias_slab = dbias ? workspace + (int64_t)a.Z * 9 * Cout * N : nullptr;
    int rc;
    if (wx3_spec(Cout, N)) {
        Wx3sArgs b{};
        b.G = a.G; b.Gcs = a.Gcs; b.X[0] = a.X[0]; b.X[1] = a.X[1]; b.Xcs[0] = a.Xcs[0]; b.Xcs[1] = a.Xcs[1]; b.n_split = a.n_split;
        b.B = a.B; b.H = a.H; b.W = a.W; b.M = a.M; b.N = a.N; b.slab = a.slab; b.bias_slab = a.bias_slab; b.Z = a.Z;
        rc = pnnp_wx3s_launch(b, st);
    } else
    switch (wx3_shape(Cout, N)) {
        case 4: rc = launch_wx3<4, 2, 1>(a, st); break;
        case 3: rc = launch_wx3<2, 2, 2>(a, st); break;
        case 1: rc = launch_wx3<2, 1, 2>(a, st); break;
        case 2: rc = launch_wx3<1, 2, 2>(a, st); break;
        default: rc = launch_wx3<1, 1, 4>(a, st); break;
    }
    if (rc != PNNP_OK) return rc;
    const int64_t n = (int64_t)Cout * N * 9, ntot = n + (dbias ? Cout : 0);       // n % 32 == 0 (channels in multiples of 32)
    hipLaunchKernelGGL(wx3_reduce_kernel, dim3((unsigned)((ntot + 31) / 32 > 4096 ? 4096 : (ntot + 31) / 32)), dim3(256), 0, st,
                       a.slab, dW, n, a.Z, accumulate, (int64_t)Cout * N, 9, dbias ? a.bias_slab : nullptr, dbias, Cout);
    return pnnp_launch_status();
}

// The same weight gradient on the fp16 matrix cores (csrc/wgrad_h2s.hip, csrc/h2.h): both operands are split into two scaled fp16 pieces
// on the fly, so each comes with its amax slot.  Same tiles, slabs, workspace and reduce as the bf16x3 kernel above.
int pnnp_conv3x3_h2_bwd_weight_f32(const float* g, int g_cs, int Cout, const unsigned* amax_g, const float* x1, int x1_cs, int C1, const unsigned* amax_x1,
                                   const float* x2, int x2_cs, int C2, const unsigned* amax_x2, float* dW, float* dbias,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    if (!g || !x1 || !dW || !workspace || !amax_g || !amax_x1 || (x2 && !amax_x2) || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    const int N = C1 + (x2 ? C2 : 0);
    if (!pnnp_x3_wgrad_supported(H, W, Cout, C1, x2 ? C2 : 0)) return PNNP_E_UNSUPPORTED;
    if (g_cs < Cout || x1_cs < C1 || (x2 && x2_cs < C2) || (g_cs & 3) || (x1_cs & 3) || (x2 && (x2_cs & 3))) return PNNP_E_INVALID;
    if ((((uintptr_t)g) | ((uintptr_t)x1) | ((uintptr_t)x2)) & 15) return PNNP_E_INVALID;
    const int cmax = g_cs > x1_cs ? g_cs : x1_cs;
    if (!pnnp_x3_wgrad_fits(B, H, W, cmax) || (x2 && !pnnp_x3_wgrad_fits(B, H, W, x2_cs))) return PNNP_E_UNSUPPORTED;
    if (workspace_floats < pnnp_x3_wgrad_workspace_floats(B, H, W, Cout, N)) return PNNP_E_WORKSPACE;
    hipStream_t st = as_stream(stream);
    Wh2sArgs b{};
    b.G = g; b.Gcs = g_cs; b.X[0] = x1; b.Xcs[0] = x1_cs; b.X[1] = x2 ? x2 : x1; b.Xcs[1] = x2 ? x2_cs : x1_cs; b.n_split = x2 ? C1 : (1 << 30);
    b.B = B; b.H = H; b.W = W; b.M = Cout; b.N = N;
    b.Z = wx3_splits(B, H, W, Cout, N);                              // (same output tiles: never more splits than the workspace was sized for)
    {
        const int th = pnnp_wh2s_th(Cout, N);
        const int tiles = ((W + 31) / 32) * ((H + th - 1) / th) * B;    // ... and at most one per pixel tile of THIS kernel
        if (b.Z > tiles) b.Z = tiles;
    }
    b.slab = workspace;
    b.bias_slab = dbias ? workspace + (int64_t)b.Z * 9 * Cout * N : nullptr;
    b.amax_g = amax_g; b.amax_x[0] = amax_x1; b.amax_x[1] = x2 ? amax_x2 : nullptr;
    const int rc = pnnp_wh2s_launch(b, st);
    if (rc != PNNP_OK) return rc;
    const int64_t n = (int64_t)Cout * N * 9, ntot = n + (dbias ? Cout : 0);
    hipLaunchKernelGGL(wx3_reduce_kernel, dim3((unsigned)((ntot + 31) / 32 > 4096 ? 4096 : (ntot + 31) / 32)), dim3(256), 0, st,
                       b.slab, dW, n, b.Z, accumulate, (int64_t)Cout * N, 9, dbias ? b.bias_slab : nullptr, dbias, Cout);
    return pnnp_launch_status();
}

}  // extern "C"
