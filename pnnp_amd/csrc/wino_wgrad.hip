// Winograd F(2x2, 3x3) backward-weight of a 3x3 / stride 1 / pad 1 convolution on the fp32 matrix cores.
//
// With  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  the filter gradient is
//     dg = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G
// i.e. 16 independent GEMMs  dU_xi[m][n] = sum_tiles Z_xi[tile][m] * V_xi[tile][n]  whose reduction runs over
// the 2x2-output tiles of the whole batch: 16 multiply-adds per tile instead of the direct form's 36
// (2.25x fewer MFMA passes), then one 4x4 -> 3x3 transform per (m, n).
//   m = output channel (rows of g = dL/dy), n = input channel (x), dW [Cout][Cin][3][3]  (nn.Conv2d of archs/Unet.py).
//
// Workgroup (256 threads, waves 2x2) owns 64 m x 64 n and a contiguous range of "chunks" (2x4 tiles = 4x8 output
// pixels); every wave keeps all 16 xi accumulators of its 32x32 block (256 AGPRs).  Per chunk:
//   * the chunk's x patch (6x10 px) and g patch (4x8 px) travel global -> registers -> LDS as 16-byte loads
//     (6 per thread and chunk; a first version loaded 72 dwords per lane straight into registers and was bound by
//     the CU's vector-memory instruction rate), three chunks ahead of the MFMAs
//   * lane = channel, wave = row index i of the transforms: Z = A dY A^T and V = B^T d B are formed from the raw
//     LDS patches and written as [xi][tile row][channel][4 tiles], so one ds_read_b128 feeds the four MFMAs of a
//     step (k = the 8 tiles of the chunk)
//   * all of that sits, slot by slot, in the shadow of the chunk's 64 MFMAs per wave: the source order is the
//     schedule (-pre-RA-sched=source + sched_barrier fences).  Zt/Vt are double buffered; two barriers per chunk
//     (raw patch consumed / next stage complete).
// Split-K partials go to slabs [z][tap][m][n]; a deterministic reduce transposes them into dW [m][n][tap].
// The bias gradient (sum of g) falls out of the Z transform of wave 1 (rows y0 + y1).
#include "common.h"
#include <type_traits>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef WW_EXPERIMENT
#define WW_EXPERIMENT 0
#endif

namespace {

constexpr int BM = 64, BN = 64;
constexpr int PLANE = 64 * 4;                  // [channel][4 tiles]
constexpr int STAGE = 16 * 2 * PLANE;          // [xi][tile row][channel][4] = 8192 floats
constexpr int XRAW = 6 * 10 * 64, GRAW = 4 * 8 * 64;
constexpr int SMEM_FLOATS = 4 * STAGE + XRAW + GRAW;   // Zt, Vt x 2 stages (128 KB) + raw patches (23 KB)

struct WwArgs {
    const float* g; int g_cs;                  // [B][H][W][g_cs], M channels used
    const float* x[2]; int x_cs[2]; int C1;    // n < C1 from x[0], else x[1] (channel n - C1)
    int M, N, B, H, W;
    int chunks_x, chunks_y, nchunks, Z;        // chunk grid per image, total chunks, splits
    float* slab;                               // [Z][9][M][N]
    float* bias_slab;                          // [Z][M] or null
};

__global__ void __launch_bounds__(256, 1) wino_wgrad_kernel(WwArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* Zt = smem;                          // 2 stages
    float* Vt = smem + 2 * STAGE;
    float* xraw = smem + 4 * STAGE;            // [6 rows][10 px][64 ch]
    float* graw = xraw + XRAW;                 // [4 rows][8 px][64 ch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;

    const int mblocks = a.M / BM, nblocks = a.N / BN;
    int blk = blockIdx.x;
    const int nb = blk % nblocks; blk /= nblocks;
    const int mb = blk % mblocks;
    const int z = blk / mblocks;
    const int per = (a.nchunks + a.Z - 1) / a.Z;
    const int c_begin = z * per, c_end = min(a.nchunks, c_begin + per);
    const int nloc = c_end - c_begin;                                 // >= 1 by construction of Z

    const int n0 = nb * BN;
    const bool second = n0 >= a.C1;
    const float* xsrc = second ? a.x[1] + (n0 - a.C1) : a.x[0] + n0;
    const int xcs = second ? a.x_cs[1] : a.x_cs[0];
    const float* gsrc = a.g + mb * BM;
    const int gcs = a.g_cs;

    // row of the transforms owned by this wave (in an SGPR: everything derived from it stays scalar)
    const int i = __builtin_amdgcn_readfirstlane(wave);
    const int ra = (i == 0) ? 0 : (i == 2 ? 2 : 1);                   // B^T rows: d0-d2, d1+d2, d2-d1, d1-d3
    const int rb = (i == 0) ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
    const float sg = (i == 1) ? 1.f : -1.f;
    const float za = (i == 3) ? 0.f : 1.f;                            // A rows: y0, y0+y1, y0-y1, -y1
    const float zb = (i == 0) ? 0.f : (i == 1 ? 1.f : -1.f);

    // ---- staging slots: the chunk's x patch (6x10 px) and g patch (4x8 px), 64 channels each, as float4 per
    // (pixel, channel quad): 960 + 512 float4 = 4 + 2 per thread, global -> registers -> LDS.  Out-of-image x
    // pixels load a clamped address and are zeroed at the LDS store; slots past the 960th duplicate the last one.
    // Loads go through buffer resources (base = the image, shifted one row + one pixel up-left for x so that chunk offsets are
    // non-negative): lane offset = loop-invariant VGPR, chunk offset = SGPR.  A chunk's patch leaves the image only through its
    // first/last row or column, so out-of-image lanes are four loop-invariant lane masks per slot, combined per chunk on the
    // scalar unit; such lanes get an offset beyond num_records and the hardware returns zeros.  (Was: clamped 64-bit addresses,
    // 16 min/max, 16 selects and ~30 address VALU per chunk -- every VALU instruction costs matrix-pipe time here.)
    unsigned OOB = 0x80000000u;
    asm("" : "+v"(OOB));                     // pinned in one VGPR (else re-materialised with a v_mov per use)
    int xdst[4], gdst[2];
    unsigned xvoff[4], gvoff[2];
    unsigned long long m_top[4], m_bot[4], m_left[4], m_right[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int q = min(tid + 256 * s, 959), px = q >> 4;
        const int prow = px / 10, pcol = px % 10;
        xdst[s] = px * 64 + (q & 15) * 4;
        xvoff[s] = (unsigned)(((prow * a.W + pcol) * xcs + (q & 15) * 4) * 4);
        m_top[s] = __ballot(prow == 0); m_bot[s] = __ballot(prow == 5);
        m_left[s] = __ballot(pcol == 0); m_right[s] = __ballot(pcol == 9);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int q = tid + 256 * s, px = q >> 4;
        gdst[s] = px * 64 + (q & 15) * 4;
        gvoff[s] = (unsigned)((((px >> 3) * a.W + (px & 7)) * gcs + (q & 15) * 4) * 4);
    }
    f32x4 xr[4], gr[2];
    auto chunk_pos = [&](int c, int& b, int& y0, int& x0) {
        const int cx = c % a.chunks_x; c /= a.chunks_x;
        const int cy = c % a.chunks_y;
        b = c / a.chunks_y; y0 = cy * 4; x0 = cx * 8;
    };
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto gload = [&](int c) {
        int b, y0, x0; chunk_pos(c, b, y0, x0);
        const int64_t img = (int64_t)b * a.H * a.W;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + (img - a.W - 1) * xcs), 0,
                                                                            (a.H * a.W + a.W + 1) * xcs * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gsrc + img * gcs), 0, a.H * a.W * gcs * 4, 0x00020000);
        const int sx = (y0 * a.W + x0) * xcs * 4, sgo = (y0 * a.W + x0) * gcs * 4;
        const bool top = y0 == 0, bot = y0 + 4 == a.H, left = x0 == 0, right = x0 + 8 == a.W;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const unsigned long long bad = (top ? m_top[s] : 0ull) | (bot ? m_bot[s] : 0ull) | (left ? m_left[s] : 0ull) | (right ? m_right[s] : 0ull);
            unsigned vo;
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(vo) : "v"(xvoff[s]), "v"(OOB), "s"(bad));
            xr[s] = bload(rx, vo, sx);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) gr[s] = bload(rg, gvoff[s], sgo);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<f32x4*>(xraw + xdst[s]) = xr[s];
#pragma unroll
        for (int s = 0; s < 2; ++s) *reinterpret_cast<f32x4*>(graw + gdst[s]) = gr[s];
    };

    // ---- transforms: lane = channel; Z = A dY A^T, V = B^T d B for one tile row, raw LDS -> [xi][tq][lane][4 tiles]
    float y0r[8], y1r[8], dar[10], dbr[10];
    float bsum = 0.f;
    auto z_read = [&](int tq) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) { y0r[cc] = graw[((2 * tq) * 8 + cc) * 64 + lane]; y1r[cc] = graw[((2 * tq + 1) * 8 + cc) * 64 + lane]; }
    };
    auto z_write = [&](float* zt, int tq, float wsum) {
        float r[8];
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) r[cc] = za * y0r[cc] + zb * y1r[cc];
        bsum = fmaf(wsum, ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])), bsum);   // wave 1: sum of g
        float* o = zt + ((i * 4) * 2 + tq) * PLANE + lane * 4;
        *reinterpret_cast<float4*>(o) = make_float4(r[0], r[2], r[4], r[6]);
        *reinterpret_cast<float4*>(o + 2 * PLANE) = make_float4(r[0] + r[1], r[2] + r[3], r[4] + r[5], r[6] + r[7]);
        *reinterpret_cast<float4*>(o + 4 * PLANE) = make_float4(r[0] - r[1], r[2] - r[3], r[4] - r[5], r[6] - r[7]);
        *reinterpret_cast<float4*>(o + 6 * PLANE) = make_float4(-r[1], -r[3], -r[5], -r[7]);
    };
    auto v_read = [&](int tq) {
#pragma unroll
        for (int cc = 0; cc < 10; ++cc) { dar[cc] = xraw[((2 * tq + ra) * 10 + cc) * 64 + lane]; dbr[cc] = xraw[((2 * tq + rb) * 10 + cc) * 64 + lane]; }
    };
    auto v_write = [&](float* vt, int tq) {
        float t[10];
#pragma unroll
        for (int cc = 0; cc < 10; ++cc) t[cc] = fmaf(sg, dbr[cc], dar[cc]);
        float* o = vt + ((i * 4) * 2 + tq) * PLANE + lane * 4;
        *reinterpret_cast<float4*>(o) = make_float4(t[0] - t[2], t[2] - t[4], t[4] - t[6], t[6] - t[8]);
        *reinterpret_cast<float4*>(o + 2 * PLANE) = make_float4(t[1] + t[2], t[3] + t[4], t[5] + t[6], t[7] + t[8]);
        *reinterpret_cast<float4*>(o + 4 * PLANE) = make_float4(t[2] - t[1], t[4] - t[3], t[6] - t[5], t[8] - t[7]);
        *reinterpret_cast<float4*>(o + 6 * PLANE) = make_float4(t[1] - t[3], t[3] - t[5], t[5] - t[7], t[7] - t[9]);
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    // ---- prologue: chunk 0 -> Zt/Vt stage 0, chunk 1 -> raw, chunk 2 -> registers (indices park on the last chunk)
    gload(c_begin); lstore();
    __syncthreads();
    z_read(0); z_write(Zt, 0, 1.f); z_read(1); z_write(Zt, 1, 1.f);
    v_read(0); v_write(Vt, 0); v_read(1); v_write(Vt, 1);
    __syncthreads();
    gload(min(c_begin + 1, c_end - 1)); lstore();
    gload(min(c_begin + 2, c_end - 1));
    __syncthreads();

    const int aoff = (lane >> 5) * PLANE + (wm * 32 + (lane & 31)) * 4;
    const int boff = (lane >> 5) * PLANE + (wn * 32 + (lane & 31)) * 4;
    for (int p = 0; p < nloc; ++p) {
        const int stage = p & 1;
        const float* zb_ = Zt + stage * STAGE + aoff;
        const float* vb_ = Vt + stage * STAGE + boff;
        float* zt = Zt + (stage ^ 1) * STAGE;
        float* vt = Vt + (stage ^ 1) * STAGE;
        const int cnext = min(c_begin + p + 3, c_end - 1);            // parks on the last chunk (re-loads, never used)
        const float wsum = p + 1 < nloc ? 1.f : 0.f;                  // the parked copy of the last chunk does not count
        float4 av[2], bv[2];
        av[0] = *reinterpret_cast<const float4*>(zb_);
        bv[0] = *reinterpret_cast<const float4*>(vb_);
        // Program order IS the schedule (built with -pre-RA-sched=source, every slot fenced by sched_barrier):
        // slot id = 4*step + k sits right after the k-th MFMA of transformed position `step`.  In the MFMA shadow:
        //   steps 0..4   transforms of chunk p+1 out of the raw LDS patches (reads one step ahead of the math)
        //   step  5      barrier: every wave is done reading raw
        //   steps 6,7    chunk p+2: registers -> raw;  steps 8,9  chunk p+3: global -> registers
        auto slot = [&](auto IC) {
            constexpr int id = decltype(IC)::value;
            constexpr int st = (WW_EXPERIMENT & 1) ? 99 : (id >> 2), k = id & 3;
            if constexpr (st == 0 && k == 0) z_read(0);
            if constexpr (st == 1 && k == 0) z_write(zt, 0, wsum);
            if constexpr (st == 1 && k == 2) z_read(1);
            if constexpr (st == 2 && k == 0) z_write(zt, 1, wsum);
            if constexpr (st == 2 && k == 2) v_read(0);
            if constexpr (st == 3 && k == 0) v_write(vt, 0);
            if constexpr (st == 3 && k == 2) v_read(1);
            if constexpr (st == 4 && k == 0) v_write(vt, 1);
            if constexpr (st == 5 && k == 0) __syncthreads();
            if constexpr (st == 6 && k == 0) lstore();
            if constexpr (st == 8 && k == 0) gload(cnext);
        };
#define WW_STEP(XI)                                                                                                    \
        {                                                                                                              \
            constexpr int xi = XI;                                                                                     \
            if constexpr (xi + 1 < 16) {                                                                               \
                av[(xi + 1) & 1] = *reinterpret_cast<const float4*>(zb_ + (xi + 1) * 2 * PLANE);                       \
                bv[(xi + 1) & 1] = *reinterpret_cast<const float4*>(vb_ + (xi + 1) * 2 * PLANE);                       \
            }                                                                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            const float4 A = av[xi & 1], Bv = bv[xi & 1];                                                              \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.x, acc[xi], 0, 0, 0);                               \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            slot(std::integral_constant<int, 4 * xi + 0>{});                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.y, acc[xi], 0, 0, 0);                               \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            slot(std::integral_constant<int, 4 * xi + 1>{});                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, Bv.z, acc[xi], 0, 0, 0);                               \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            slot(std::integral_constant<int, 4 * xi + 2>{});                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, Bv.w, acc[xi], 0, 0, 0);                               \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
            slot(std::integral_constant<int, 4 * xi + 3>{});                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                         \
        }
        WW_STEP(0) WW_STEP(1) WW_STEP(2) WW_STEP(3) WW_STEP(4) WW_STEP(5) WW_STEP(6) WW_STEP(7)
        WW_STEP(8) WW_STEP(9) WW_STEP(10) WW_STEP(11) WW_STEP(12) WW_STEP(13) WW_STEP(14) WW_STEP(15)
#undef WW_STEP
        __syncthreads();
    }

    // ---- dg = G^T dU G in registers, partial sums to the slab [z][tap][m][n]
    const int n = n0 + wn * 32 + (lane & 31);
    const int64_t mn = (int64_t)a.M * a.N;
    float* slab = a.slab + (int64_t)z * 9 * mn;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mb * BM + wm * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
        float h[3][4];                                                 // G^T applied to the first index
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float u0 = acc[j][e], u1 = acc[4 + j][e], u2 = acc[8 + j][e], u3 = acc[12 + j][e];
            h[0][j] = u0 + 0.5f * (u1 + u2);
            h[1][j] = 0.5f * (u1 - u2);
            h[2][j] = 0.5f * (u1 + u2) + u3;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float w0 = h[r][0] + 0.5f * (h[r][1] + h[r][2]);
            const float w1 = 0.5f * (h[r][1] - h[r][2]);
            const float w2 = 0.5f * (h[r][1] + h[r][2]) + h[r][3];
            float* o = slab + (int64_t)(r * 3) * mn + (int64_t)m * a.N + n;
            o[0] = w0; o[mn] = w1; o[2 * mn] = w2;
        }
    }
    if (a.bias_slab && nb == 0 && wave == 1) a.bias_slab[(int64_t)z * a.M + mb * BM + lane] = bsum;
}

// out[m][n][tap] (+)= sum_z slab[z][tap][m][n]   (taps == 1: plain sum over z)
__global__ void __launch_bounds__(256)
ww_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int Z, int accumulate, int64_t mn, int taps) {
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int64_t i0 = (int64_t)blockIdx.x * 32; i0 < n; i0 += (int64_t)gridDim.x * 32) {
        const int64_t i = i0 + tx;
        float s = 0.f;
        if (i < n)
            for (int zz = ty; zz < Z; zz += 8) s += slab[(int64_t)zz * n + i];
        red[ty][tx] = s;
        __syncthreads();
        if (ty == 0 && i < n) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) t += red[k][tx];
            const int64_t o = taps == 1 ? i : (i % mn) * taps + i / mn;
            out[o] = accumulate ? out[o] + t : t;
        }
        __syncthreads();
    }
}

int ww_splits(int B, int H, int W, int M, int N) {
    const int chunks = B * (H / 4) * (W / 8);
    const int blocks = (M / BM) * (N / BN);
    int cus = pnnp_device_cus();
    if (cus <= 0) cus = 256;
    int z = (cus + blocks - 1) / blocks;
    if (z > chunks) z = chunks;
    if (z < 1) z = 1;
    const int per = (chunks + z - 1) / z;             // no empty split
    return (chunks + per - 1) / per;
}

}  // namespace

extern "C" {

// 1 when pnnp_conv3x3_wino_bwd_weight_f32 accepts the layer
int pnnp_wino_wgrad_supported(int H, int W, int Cout, int C1, int C2) {
    return (H > 0 && W > 0 && H % 4 == 0 && W % 8 == 0 && Cout % BM == 0 && C1 % BN == 0 && C2 % BN == 0) ? 1 : 0;
}

int64_t pnnp_wino_wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin) {
    if (!pnnp_wino_wgrad_supported(H, W, Cout, Cin, 0)) return 0;
    const int64_t z = ww_splits(B, H, W, Cout, Cin);
    return z * ((int64_t)9 * Cout * Cin + Cout);
}

// dW [Cout][C1+C2][3][3] (+ dbias [Cout]) of a 3x3 / stride 1 / pad 1 convolution; same contract as
// pnnp_conv_bwd_weight_f32 with taps = 9 (g: dL/d(pre-activation output), x1/x2: the layer's input(s)).
int pnnp_conv3x3_wino_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                                     const float* x2, int x2_cs, int C2, float* dW, float* dbias,
                                     int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats,
                                     void* stream) {
    if (!g || !x1 || !dW || !workspace || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    const int N = C1 + (x2 ? C2 : 0);
    if (!pnnp_wino_wgrad_supported(H, W, Cout, C1, x2 ? C2 : 0)) return PNNP_E_UNSUPPORTED;
    if (g_cs < Cout || x1_cs < C1 || (x2 && x2_cs < C2) || (g_cs & 3) || (x1_cs & 3) || (x2 && (x2_cs & 3))) return PNNP_E_INVALID;
    if ((int64_t)B * H * W * (g_cs > x1_cs ? g_cs : x1_cs) >= (1ll << 31) || (x2 && (int64_t)B * H * W * x2_cs >= (1ll << 31))) return PNNP_E_UNSUPPORTED;
    if (workspace_floats < pnnp_wino_wgrad_workspace_floats(B, H, W, Cout, N)) return PNNP_E_WORKSPACE;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, wino_wgrad_kernel, SMEM_FLOATS * 4) != PNNP_OK) return PNNP_E_LAUNCH;
    hipStream_t st = as_stream(stream);
    WwArgs a{};
    a.g = g; a.g_cs = g_cs;
    a.x[0] = x1; a.x_cs[0] = x1_cs; a.x[1] = x2 ? x2 : x1; a.x_cs[1] = x2 ? x2_cs : x1_cs; a.C1 = x2 ? C1 : (1 << 30);
    a.M = Cout; a.N = N; a.B = B; a.H = H; a.W = W;
    a.chunks_x = W / 8; a.chunks_y = H / 4; a.nchunks = B * a.chunks_x * a.chunks_y;
    a.Z = ww_splits(B, H, W, Cout, N);
    a.slab = workspace;
    a.bias_slab = dbias ? a.slab + (int64_t)a.Z * 9 * Cout * N : nullptr;
    const unsigned grid = (unsigned)((Cout / BM) * (N / BN) * a.Z);
    hipLaunchKernelGGL(wino_wgrad_kernel, dim3(grid), dim3(256), SMEM_FLOATS * 4, st, a);
    const int64_t n = (int64_t)Cout * N * 9;
    hipLaunchKernelGGL(ww_reduce_kernel, dim3((unsigned)((n + 31) / 32 > 4096 ? 4096 : (n + 31) / 32)), dim3(256), 0, st,
                       a.slab, dW, n, a.Z, accumulate, (int64_t)Cout * N, 9);
    if (dbias)
        hipLaunchKernelGGL(ww_reduce_kernel, dim3((Cout + 31) / 32), dim3(256), 0, st, a.bias_slab, dbias, (int64_t)Cout, a.Z,
                           accumulate, (int64_t)Cout, 1);
    return pnnp_launch_status();
}

}  // extern "C"
