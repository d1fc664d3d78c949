// Weight-gradient kernel on the fp32 matrix cores.
//
//   dW[m][n][t] = sum over pixels p of  U[p][m] * S[p*s_mul + tap(t)][n]
//
//   Conv2d 3x3 (archs/Unet.py:16-50):        U = dL/d(pre-act output) (m = Cout), S = layer input
//     (n = Cin, possibly the un-materialised cat of two tensors), 9 taps with a 1-pixel halo
//     -> dW in the parameter's own layout [Cout][Cin][3][3].
//   ConvTranspose2d 2x2 s2 (archs/Unet.py:35-47): U = layer input (m = Cin), S = dL/d(output)
//     (n = Cout) read at (2y+a, 2x+c), 4 taps -> dW in layout [Cin][Cout][2][2].
//   taps = 1: Conv2d 1x1.
//
// GEMM view: M = m, N = n (per tap), K = pixels -- K is huge (B*H*W) and the output tiny, so
// the pixel range is split over Z workgroups per output tile; each keeps its [32 x 32 x taps]
// accumulators in registers across all its pixel tiles and writes ONE partial slab; a second
// tiny kernel sums the Z slabs in a fixed order (bitwise reproducible, no float atomics).
// MFMA operands come from LDS with conflict-free ds_read_b32: lanes 0-31 read 32 consecutive
// channels of pixel p, lanes 32-63 of pixel p+1 (the two k-slices of v_mfma_f32_32x32x2_f32).
// The bias gradient (column sums of U) rides along for free on the A operand.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WgradArgs {
    const float* U; int Ucs;            // [B][DH][DW][Ucs], channels [0, M) used
    const float* S[2]; int Scs[2];      // n < n_split -> S[0][n], else S[1][n - n_split]
    int n_split;
    int s_mul, SH, SW;                  // S pixel = U pixel * s_mul + tap offset
    int B, DH, DW;
    int M, N;
    float* slab;                        // [Z][M][N][TAPS]
    float* bias_slab;                   // [Z][M] or null: sum of U over the pixels (bias gradient of a Conv2d)
    float* bias_slab_n;                 // [Z][N] or null: sum of S over the pixels (TAPS == 4: bias gradient of a ConvTranspose2d)
    int Z;
};

namespace {

template <int TAPS, int WMO, int WNO, int WK, int TH, int SMUL = (TAPS == 4 ? 2 : 1)>
struct WgCfg {
    static constexpr int P = (TAPS == 9) ? 1 : 0;
    static constexpr int SM = SMUL;                        // S pixels per U pixel along each axis (2: ConvTranspose, stride-2 conv)
    static constexpr int BMO = 32 * WMO, BNO = 32 * WNO;
    static constexpr int UPIX = TH * 32;
    static constexpr int SR = TH * SM + 2 * P, SC = 32 * SM + 2 * P, SPIX = SR * SC;
    static constexpr int US_F = UPIX * BMO, SS_F = SPIX * BNO;
    static constexpr int RED_F = (WK > 1) ? 4 * 16 * 64 : 0;
    static constexpr int LDS_BYTES = (US_F + SS_F) * 4;
    static_assert(WMO * WNO * WK == 4, "4 waves");
    static_assert((US_F + SS_F) >= RED_F, "reduction scratch aliases the tiles");
};

template <int TAPS, int WMO, int WNO, int WK, int TH, int SMUL>
__global__ void __launch_bounds__(256, (TAPS == 9 && SMUL == 2) ? 1 : 2)   // the stride-2 halo tile needs more staging registers
wgrad_kernel(const WgradArgs a) {
    using Cfg = WgCfg<TAPS, WMO, WNO, WK, TH, SMUL>;
    constexpr int P = Cfg::P, SM = Cfg::SM, BMO = Cfg::BMO, BNO = Cfg::BNO, SC = Cfg::SC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* us = reinterpret_cast<float*>(smem);
    float* ss = us + Cfg::US_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wk = wave % WK, wno = (wave / WK) % WNO, wmo = wave / (WK * WNO);

    const int n_tiles = (a.N + BNO - 1) / BNO;
    int id = blockIdx.x;
    const int z = id % a.Z; id /= a.Z;
    const int ni = id % n_tiles, mi = id / n_tiles;
    const int m0 = mi * BMO, n0 = ni * BNO;

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int ntile = tiles_x * tiles_y * a.B;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f, bsn = 0.f;
    // every S pixel is visited exactly once per (pixel pair, tap) when TAPS == 4; only one M block / M wave adds it up
    const float bsn_w = (TAPS == 4 && a.bias_slab_n && mi == 0 && wmo == 0) ? 1.f : 0.f;

    // staging registers of the NEXT pixel tile (its loads fly while this tile's MFMAs run)
    constexpr int NU = (Cfg::US_F / 4 + 255) / 256, NS = (Cfg::SS_F / 4 + 255) / 256;
    float4 ru[NU], rs[NS];
    auto prefetch = [&](int tile) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int x0 = tx * 32, y0 = ty * TH;
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int i = tid + 256 * k;
            const int cq = i % (BMO / 4), pix = i / (BMO / 4);
            const int r = pix >> 5, c = pix & 31;
            const int gy = y0 + r, gx = x0 + c, m = m0 + 4 * cq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < Cfg::US_F / 4 && gy < a.DH && gx < a.DW && m < a.M)
                v = *reinterpret_cast<const float4*>(a.U + (((int64_t)b * a.DH + gy) * a.DW + gx) * a.Ucs + m);
            ru[k] = v;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = tid + 256 * k;
            const int cq = i % (BNO / 4), pix = i / (BNO / 4);
            const int r = pix / SC, c = pix - r * SC;
            const int gy = y0 * SM + r - P, gx = x0 * SM + c - P, n = n0 + 4 * cq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < Cfg::SS_F / 4 && gy >= 0 && gy < a.SH && gx >= 0 && gx < a.SW && n < a.N) {
                const int d = n >= a.n_split ? 1 : 0;
                const int ch = n - (d ? a.n_split : 0);
                v = *reinterpret_cast<const float4*>(a.S[d] + (((int64_t)b * a.SH + gy) * a.SW + gx) * a.Scs[d] + ch);
            }
            rs[k] = v;
        }
    };

    if (z < ntile) prefetch(z);
    for (int tile = z; tile < ntile; tile += a.Z) {
        if (tile != z) __syncthreads();
        // ---- write the staged tiles: us[pix][BMO], ss[spix][BNO] (linear float4 copies)
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            const int i = tid + 256 * k;
            if (i < Cfg::US_F / 4) reinterpret_cast<float4*>(us)[i] = ru[k];
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int i = tid + 256 * k;
            if (i < Cfg::SS_F / 4) reinterpret_cast<float4*>(ss)[i] = rs[k];
        }
        __syncthreads();
        if (tile + a.Z < ntile) prefetch(tile + a.Z);
        // ---- MFMA: k = pixel pairs along a row
        for (int r = wk; r < TH; r += WK) {
#pragma unroll 4
            for (int pp = 0; pp < 16; ++pp) {
                const int p = 2 * pp + half;
                const float av = us[(r * 32 + p) * BMO + wmo * 32 + l31];
                bsum += av;
#pragma unroll
                for (int t = 0; t < TAPS; ++t) {
                    int sp;
                    if (TAPS == 9) sp = (SM * r + t / 3) * SC + SM * p + t % 3;
                    else if (TAPS == 4) sp = (2 * r + (t >> 1)) * SC + 2 * p + (t & 1);
                    else sp = r * SC + p;
                    const float bv = ss[sp * BNO + wno * 32 + l31];
                    if (TAPS == 4) bsn = fmaf(bsn_w, bv, bsn);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }

    // ---- reduce the WK pixel-split waves through LDS (tiles are dead now), then write the slab
    float* red = reinterpret_cast<float*>(smem);
    const int64_t slab_base = (int64_t)z * a.M * a.N * TAPS;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
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
            const int n = n0 + wno * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wmo * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m < a.M && n < a.N) a.slab[slab_base + ((int64_t)t * a.M + m) * a.N + n] = v[r];   // [z][tap][m][n]: lanes (n) contiguous
            }
        }
    }
    if (a.bias_slab && ni == 0) {                        // block-uniform: every wave reaches the barriers
        bsum += __shfl_xor(bsum, 32);                    // even + odd pixels of the pair
        if (WK > 1) {
            __syncthreads();
            if (half == 0) red[wave * 32 + l31] = bsum;
            __syncthreads();
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WK; ++k) s += red[((wave - wk) + k) * 32 + l31];
            bsum = s;
        }
        const int m = m0 + wmo * 32 + l31;
        if (wno == 0 && wk == 0 && half == 0 && m < a.M) a.bias_slab[(int64_t)z * a.M + m] = bsum;
    }
    if (TAPS == 4 && a.bias_slab_n && mi == 0) {         // block-uniform
        bsn += __shfl_xor(bsn, 32);
        if (WK > 1) {
            __syncthreads();
            if (half == 0) red[wave * 32 + l31] = bsn;
            __syncthreads();
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < WK; ++k) s += red[((wave - wk) + k) * 32 + l31];
            bsn = s;
        }
        const int n = n0 + wno * 32 + l31;
        if (wmo == 0 && wk == 0 && half == 0 && n < a.N) a.bias_slab_n[(int64_t)z * a.N + n] = bsn;
    }
}

// out = sum_z slab[z] in a fixed order.  32 consecutive elements x 8 z-phases per block so the Z (up to
// 512) dependent loads per element are spread over 8 threads and many blocks.  The slabs are stored
// [tap][m][n] (what the accumulator lanes write coalesced -- the parameter's own [m][n][tap] order made
// every slab store a scattered 4-byte write: 8x write amplification, 7.8 GB per step); the transpose to
// [m][n][tap] happens here, once per element instead of once per slab.  taps == 1: plain sum.
__global__ void __launch_bounds__(256)
slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int Z, int accumulate,
                   int64_t mn, int taps) {
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int64_t i0 = (int64_t)blockIdx.x * 32; i0 < n; i0 += (int64_t)gridDim.x * 32) {
        const int64_t i = i0 + tx;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (i < n) {
            int z = ty;
            for (; z + 24 < Z; z += 32) {
                s0 += slab[(int64_t)z * n + i]; s1 += slab[(int64_t)(z + 8) * n + i];
                s2 += slab[(int64_t)(z + 16) * n + i]; s3 += slab[(int64_t)(z + 24) * n + i];
            }
            for (; z < Z; z += 8) s0 += slab[(int64_t)z * n + i];
        }
        red[ty][tx] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (ty == 0 && i < n) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k][tx];
            const int64_t o = taps == 1 ? i : (i % mn) * taps + i / mn;
            out[o] = accumulate ? out[o] + s : s;
        }
        __syncthreads();
    }
}

template <int TAPS, int WMO, int WNO, int WK, int TH, int SMUL = (TAPS == 4 ? 2 : 1)>
int launch_wg(const WgradArgs& a, hipStream_t s) {
    using Cfg = WgCfg<TAPS, WMO, WNO, WK, TH, SMUL>;
    auto kern = wgrad_kernel<TAPS, WMO, WNO, WK, TH, SMUL>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int blocks = ((a.M + Cfg::BMO - 1) / Cfg::BMO) * ((a.N + Cfg::BNO - 1) / Cfg::BNO) * a.Z;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

// output-tile shape from (M, N): 32x32 with the 4 waves splitting pixel rows (8-row tiles),
// 64x32 / 32x64 with 2 pixel splits (4-row tiles), 64x64 with none (2-row tiles, 3 workgroups/CU)
int pick_shape(int M, int N) { return (M > 32 ? 1 : 0) + (N > 32 ? 2 : 0); }
// rows per pixel tile: small tiles keep the prefetch registers (next tile in flight) + 144
// accumulator registers under 256 VGPRs without spilling
// taps == 18 stands for the 9 taps of a STRIDE-2 3x3 conv (S halo tile is (2 TH + 2) x 66 pixels)
int tile_rows(int taps, int shape) {
    if (taps == 18) return shape == 0 ? 4 : (shape == 3 ? 1 : 2);
    return taps == 4 ? 2 : (shape == 0 ? 4 : 2);
}

template <int TAPS>
int launch_shape(const WgradArgs& a, int shape, hipStream_t s) {
    if constexpr (TAPS == 4) {
        switch (shape) {
            case 0: return launch_wg<TAPS, 1, 1, 4, 2>(a, s);
            case 1: return launch_wg<TAPS, 2, 1, 2, 2>(a, s);
            case 2: return launch_wg<TAPS, 1, 2, 2, 2>(a, s);
            default: return launch_wg<TAPS, 2, 2, 1, 2>(a, s);
        }
    } else {
        switch (shape) {
            case 0: return launch_wg<TAPS, 1, 1, 4, 4>(a, s);
            case 1: return launch_wg<TAPS, 2, 1, 2, 2>(a, s);
            case 2: return launch_wg<TAPS, 1, 2, 2, 2>(a, s);
            default: return launch_wg<TAPS, 2, 2, 1, 2>(a, s);
        }
    }
}

int launch_s2(const WgradArgs& a, int shape, hipStream_t s) {
    switch (shape) {
        case 0: return launch_wg<9, 1, 1, 4, 4, 2>(a, s);
        case 1: return launch_wg<9, 2, 1, 2, 2, 2>(a, s);
        case 2: return launch_wg<9, 1, 2, 2, 2, 2>(a, s);
        default: return launch_wg<9, 2, 2, 1, 1, 2>(a, s);
    }
}

}  // namespace

extern "C" {

// Number of pixel splits Z the launch will use and the workspace it needs (floats).
int pnnp_wgrad_splits(int B, int H, int W, int M, int N, int taps) {
    const int shape = pick_shape(M, N);
    const int bmo = (shape & 1) ? 64 : 32, bno = (shape & 2) ? 64 : 32;
    const int th = tile_rows(taps, shape);
    const int tiles = ((W + 31) / 32) * ((H + th - 1) / th) * B;
    const int out_tiles = ((M + bmo - 1) / bmo) * ((N + bno - 1) / bno);
    int z = (2 * 256 + out_tiles - 1) / out_tiles;     // ~2 workgroups per CU overall
    if (z > tiles) z = tiles;
    if (z < 1) z = 1;
    return z;
}

int64_t pnnp_wgrad_workspace_floats(int B, int H, int W, int M, int N, int taps) {
    const int64_t z = pnnp_wgrad_splits(B, H, W, M, N, taps);
    return z * ((int64_t)M * N * (taps == 18 ? 9 : taps) + (M > N ? M : N));
}

// dW (+ optional dbias) of Conv2d 3x3 / 1x1 (taps = 9 / 1):
//   g [B][H][W][Cout] = dL/d(pre-activation output); x1 [..][C1], x2 [..][C2] (null if no concat)
//   dW [Cout][C1+C2][taps], dbias [Cout] (null to skip).  accumulate: dW += (gradient accumulation).
int pnnp_conv_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                             const float* x2, int x2_cs, int C2,
                             float* dW, float* dbias, int B, int H, int W, int taps, int accumulate,
                             float* workspace, int64_t workspace_floats, void* stream) {
    if (!g || !x1 || !dW || !workspace || B <= 0 || H <= 0 || W <= 0 || (taps != 9 && taps != 1)) return PNNP_E_INVALID;
    const int N = C1 + (x2 ? C2 : 0);
    if ((Cout & 3) || (C1 & 3) || (C2 & 3) || (g_cs & 3) || (x1_cs & 3) || (x2 && (x2_cs & 3))) return PNNP_E_UNSUPPORTED;
    if (g_cs < Cout || x1_cs < C1 || (x2 && x2_cs < C2)) return PNNP_E_INVALID;
    if (workspace_floats < pnnp_wgrad_workspace_floats(B, H, W, Cout, N, taps)) return PNNP_E_WORKSPACE;
    WgradArgs a{};
    a.U = g; a.Ucs = g_cs;
    a.S[0] = x1; a.Scs[0] = x1_cs; a.S[1] = x2 ? x2 : x1; a.Scs[1] = x2 ? x2_cs : x1_cs; a.n_split = x2 ? C1 : (1 << 30);
    a.s_mul = 1; a.SH = H; a.SW = W; a.B = B; a.DH = H; a.DW = W; a.M = Cout; a.N = N;
    a.Z = pnnp_wgrad_splits(B, H, W, Cout, N, taps);
    a.slab = workspace;
    a.bias_slab = dbias ? workspace + (int64_t)a.Z * Cout * N * taps : nullptr;
    const int shape = pick_shape(Cout, N);
    int rc;
    if (taps == 9) rc = launch_shape<9>(a, shape, as_stream(stream));
    else rc = launch_shape<1>(a, shape, as_stream(stream));
    if (rc != PNNP_OK) return rc;
    const int64_t n = (int64_t)Cout * N * taps;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 31) / 32 > 4096 ? 4096 : (n + 31) / 32)), dim3(256), 0,
                       as_stream(stream), a.slab, dW, n, a.Z, accumulate, (int64_t)a.M * a.N, (int)(n / ((int64_t)a.M * a.N)));
    if (dbias)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, as_stream(stream), a.bias_slab, dbias,
                           (int64_t)Cout, a.Z, accumulate, (int64_t)Cout, 1);
    return pnnp_launch_status();
}

// dW (+ dbias) of ConvTranspose2d(Cin, Cout, 2, stride=2):
//   x [B][H][W][Cin]; g [B][2H][2W][Cout]; dW [Cin][Cout][2][2]; dbias [Cout] = sum of g.
int pnnp_convt2x2_bwd_weight_f32(const float* x, int Cin, const float* g, int Cout, float* dW, float* dbias,
                                 int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats,
                                 void* stream) {
    if (!x || !g || !dW || !workspace || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if ((Cin & 3) || (Cout & 3)) return PNNP_E_UNSUPPORTED;
    if (workspace_floats < pnnp_wgrad_workspace_floats(B, H, W, Cin, Cout, 4)) return PNNP_E_WORKSPACE;
    WgradArgs a{};
    a.U = x; a.Ucs = Cin;
    a.S[0] = g; a.Scs[0] = Cout; a.S[1] = g; a.Scs[1] = Cout; a.n_split = 1 << 30;
    a.s_mul = 2; a.SH = 2 * H; a.SW = 2 * W; a.B = B; a.DH = H; a.DW = W; a.M = Cin; a.N = Cout;
    a.Z = pnnp_wgrad_splits(B, H, W, Cin, Cout, 4);
    a.slab = workspace; a.bias_slab = nullptr;
    a.bias_slab_n = dbias ? workspace + (int64_t)a.Z * Cin * Cout * 4 : nullptr;      // dbias = sum of g, gathered while g is staged anyway
    int rc = launch_shape<4>(a, pick_shape(Cin, Cout), as_stream(stream));
    if (rc != PNNP_OK) return rc;
    const int64_t n = (int64_t)Cin * Cout * 4;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 31) / 32 > 4096 ? 4096 : (n + 31) / 32)), dim3(256), 0,
                       as_stream(stream), a.slab, dW, n, a.Z, accumulate, (int64_t)a.M * a.N, (int)(n / ((int64_t)a.M * a.N)));
    if (dbias)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, as_stream(stream), a.bias_slab_n, dbias,
                           (int64_t)Cout, a.Z, accumulate, (int64_t)Cout, 1);
    return pnnp_launch_status();
}

// dW (+ dbias) of Conv2d 3x3 stride 2 pad 1 (ResUnet down-sampling, archs/modules.py:130-138):
//   g [B][H/2][W/2][Cout]; x [B][H][W][Cin]; dW [Cout][Cin][3][3].  Workspace: query with taps = 18
//   and the OUTPUT size (H/2, W/2).
int pnnp_conv3x3s2_bwd_weight_f32(const float* g, int Cout, const float* x, int Cin, float* dW, float* dbias,
                                  int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats,
                                  void* stream) {
    if (!g || !x || !dW || !workspace || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if ((Cout & 3) || (Cin & 3)) return PNNP_E_UNSUPPORTED;
    const int h = H / 2, w = W / 2;
    if (workspace_floats < pnnp_wgrad_workspace_floats(B, h, w, Cout, Cin, 18)) return PNNP_E_WORKSPACE;
    WgradArgs a{};
    a.U = g; a.Ucs = Cout;
    a.S[0] = x; a.Scs[0] = Cin; a.S[1] = x; a.Scs[1] = Cin; a.n_split = 1 << 30;
    a.s_mul = 2; a.SH = H; a.SW = W; a.B = B; a.DH = h; a.DW = w; a.M = Cout; a.N = Cin;
    a.Z = pnnp_wgrad_splits(B, h, w, Cout, Cin, 18);
    a.slab = workspace;
    a.bias_slab = dbias ? workspace + (int64_t)a.Z * Cout * Cin * 9 : nullptr;
    const int rc = launch_s2(a, pick_shape(Cout, Cin), as_stream(stream));
    if (rc != PNNP_OK) return rc;
    const int64_t n = (int64_t)Cout * Cin * 9;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 31) / 32 > 4096 ? 4096 : (n + 31) / 32)), dim3(256), 0,
                       as_stream(stream), a.slab, dW, n, a.Z, accumulate, (int64_t)a.M * a.N, (int)(n / ((int64_t)a.M * a.N)));
    if (dbias)
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, as_stream(stream), a.bias_slab, dbias,
                           (int64_t)Cout, a.Z, accumulate, (int64_t)Cout, 1);
    return pnnp_launch_status();
}

}  // extern "C"
