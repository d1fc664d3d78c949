// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores, NHWC, stride 1, pad 1.
//
//   Y = A^T [ sum_k (G g_k G^T) (.) (B^T d_k B) ] A        per 2x2 output tile, 4x4 input tile
//
// so the 9 multiply-adds per (pixel, cin, cout) of the direct form (archs/Unet.py:16-52 via nn.Conv2d)
// become 16 per FOUR pixels: 2.25x fewer MFMA passes for the same layer.  The 16 element-wise products
// are 16 independent GEMMs  M_xi[tile][n] = sum_k V_xi[tile][k] * U_xi[k][n]  (xi = position in the 4x4
// transformed tile), run on v_mfma_f32_32x32x2_f32.
//
// Workgroup (256 threads, 4 waves as 2x2) owns 8x8 tiles (16x16 output pixels) x 64 output channels:
//   * K loop in chunks of 8 channels.  The 18x18x8 halo patch goes global -> registers -> LDS (raws),
//     every thread transforms two (tile, 4-channel, row) items  B^T d B  into  Vs[xi][k/4][tile][4],
//     the pre-transformed weights U (packed once per weight update) stream global -> registers -> Us.
//   * each wave keeps ALL 16 xi accumulators of its 32 tiles x 32 channels in registers (256 AGPRs),
//     so the inverse transform A^T M A runs in registers and the epilogue (bias, activation, act'
//     mask, split destinations) stores straight to HBM -- no LDS round trip for the output.
//   * Vs/Us are double buffered; global loads for chunk c+1 are issued before the MFMAs of chunk c.
// LDS: 2 x (36 KB Vs + 32 KB Us) + 10 KB raws = 146 KB of the CU's 160 KB, one workgroup per CU.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

#ifndef WINO_FENCED
#define WINO_FENCED 1
#endif
#ifndef WINO_STEPTIME
#define WINO_STEPTIME 0     // profiling build: clock64() at every step of chunk 2 (tools/wino_phases.py --steps)
#endif
#ifndef PNNP_WINO_DEBUG
#define PNNP_WINO_DEBUG WINO_STEPTIME   // profiling builds only (tools/ab_build.sh -DPNNP_WINO_DEBUG=1): exports pnnp_wino_set_debug;
#endif                                  // the shipped library has no debug hook and no mutable global
constexpr int KC = 8, BN = 64, TT = 64, PATCH = 18;
constexpr int VPLANE = TT * 4 + 32;                    // 288: the +32 keeps the two k-quads of a b128 store on disjoint banks
constexpr int VS_STAGE = 16 * 2 * VPLANE;              // floats
constexpr int UPLANE = BN * 4;
constexpr int US_STAGE = 16 * 2 * UPLANE;              // 8192 floats
constexpr int RAW_ROW = 2 * 9 * KC;                    // a patch row: [column parity][9][8 ch]
constexpr int RAW_FLOATS = PATCH * RAW_ROW;
constexpr int SMEM_FLOATS = 2 * VS_STAGE + 2 * US_STAGE + 2 * RAW_FLOATS;

struct WinoArgs {
    const float* src[2]; int src_cs[2]; int C1;        // K channels: [0,C1) from src[0], the rest from src[1]
    int K, N;
    const float* u;                                    // [N/64][K/8][16][2][64][4]
    int B, H, W, tiles_x, tiles_y;
    float* dst[2]; int dst_cs[2]; int n_split;         // column n < n_split -> dst[0][n] else dst[1][n - n_split]
    const float* bias; int act;
    const float* mask[2]; int mask_mode[2]; int accum[2];
    const float* addsrc;
    long long* dbg;                                    // profiling hook (pnnp_wino_set_debug): per-workgroup cycle stamps, or null
};

__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 b) {      // b + s*a
    return make_float4(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z), fmaf(s, a.w, b.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// B^T d B for two (tile, channel quad, row i) items per thread: raws -> Vs stage
__device__ __forceinline__ void input_transform(const float* __restrict__ raws, float* __restrict__ vs, int tid) {
    const int cg = tid & 1, tx = (tid >> 1) & 7, i = (tid >> 4) & 3, tyb = tid >> 6;
    const int ra = (i == 0) ? 0 : (i == 2 ? 2 : 1);
    const int rb = (i == 0) ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3));
    const float sg = (i == 1) ? 1.f : -1.f;            // rows of B^T: d0-d2, d1+d2, d2-d1, d1-d3
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int ty = tyb + 4 * rep;
        float4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = ((c & 1) * 9 + tx + (c >> 1)) * KC + cg * 4;
            const float4 da = *reinterpret_cast<const float4*>(raws + (2 * ty + ra) * RAW_ROW + col);
            const float4 db = *reinterpret_cast<const float4*>(raws + (2 * ty + rb) * RAW_ROW + col);
            t[c] = f4_fma(db, sg, da);
        }
        float* o = vs + (i * 8 + cg) * VPLANE + (ty * 8 + tx) * 4;
        *reinterpret_cast<float4*>(o) = f4_sub(t[0], t[2]);
        *reinterpret_cast<float4*>(o + 2 * VPLANE) = f4_add(t[1], t[2]);
        *reinterpret_cast<float4*>(o + 4 * VPLANE) = f4_sub(t[2], t[1]);
        *reinterpret_cast<float4*>(o + 6 * VPLANE) = f4_sub(t[1], t[3]);
    }
}

// Packed fp32 VALU (two floats of an even-aligned register pair per issue slot).  hipcc expands <4 x float> / <2 x float> fma
// and fadd into scalar v_fma_f32 / v_add_f32 in this kernel, so the three forms the input transform needs are written out.
// Their operands come from LDS reads or from results at least one fenced step old, and their results go to LDS stores or to a
// later step: no back-to-back VALU dependency the hazard recogniser would have to pad.
__device__ __forceinline__ f32x2 pk_fma2(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
    f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ f32x4 pk_fma(f32x4 a, f32x2 s, f32x4 b) { f32x4 r; r.lo = pk_fma2(a.lo, s, b.lo); r.hi = pk_fma2(a.hi, s, b.hi); return r; }
__device__ __forceinline__ f32x4 pk_add(f32x4 a, f32x4 b) { f32x4 r; r.lo = pk_add2(a.lo, b.lo); r.hi = pk_add2(a.hi, b.hi); return r; }
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) { f32x4 r; r.lo = pk_sub2(a.lo, b.lo); r.hi = pk_sub2(a.hi, b.hi); return r; }

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.2f * v;
    if (act == 2) return fmaxf(v, 0.f);
    return v;
}

__global__ void __launch_bounds__(256, 1) wino_kernel(WinoArgs a) {
    extern __shared__ __align__(16) float smem[];
    float* Vs = smem;
    float* Us = smem + 2 * VS_STAGE;
    float* raws = Us + 2 * US_STAGE;                   // two patch buffers
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const long long t_start = a.dbg ? clock64() : 0;

    int mt = blockIdx.x;
    const int nb = blockIdx.y;
    const int txi = mt % a.tiles_x; mt /= a.tiles_x;
    const int tyi = mt % a.tiles_y;
    const int b = mt / a.tiles_y;
    const int y0 = tyi * 16, x0 = txi * 16;

    // halo patch staging slots: 18*18 pixels x 2 channel quads = 648 float4, three per thread.
    // Global loads go through buffer resources (one per source image, one for this channel block's U): the lane offset is a
    // loop-invariant 32-bit VGPR, the chunk offset an SGPR, and an out-of-image halo pixel gets an offset beyond num_records, for
    // which the hardware returns zeros -- no per-chunk address arithmetic and no selects on the VALU (instruction count is the
    // currency here: every non-MFMA instruction in the main loop costs ~7 cycles of matrix-pipe time).
    constexpr unsigned OOB = 0x80000000u;              // host guarantees every image is < 2 GB
    unsigned roff0[3], roff1[3]; int rdst[3];          // byte offsets inside image b of source 0 / 1
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int q = tid + 256 * s, pl = min(q >> 1, PATCH * PATCH - 1);
        const int py = pl / PATCH, px = pl % PATCH;
        const int y = y0 - 1 + py, x = x0 - 1 + px;
        const bool ok = y >= 0 && y < a.H && x >= 0 && x < a.W;
        const unsigned pix = (unsigned)(y * a.W + x);
        roff0[s] = ok ? (pix * (unsigned)a.src_cs[0] + (tid & 1) * 4) * 4u : OOB;
        roff1[s] = ok ? (pix * (unsigned)a.src_cs[1] + (tid & 1) * 4) * 4u : OOB;
        rdst[s] = py * RAW_ROW + ((px & 1) * 9 + (px >> 1)) * KC + (tid & 1) * 4;
    }
    const int64_t img0 = (int64_t)b * a.H * a.W;
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.src[0] + img0 * a.src_cs[0]), 0, a.H * a.W * a.src_cs[0] * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)((a.src[1] ? a.src[1] : a.src[0]) + img0 * a.src_cs[1]), 0,
                                                                         a.src[1] ? a.H * a.W * a.src_cs[1] * 4 : 0, 0x00020000);
    const int nchunks = a.K / KC;
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)(a.u + (int64_t)nb * nchunks * US_STAGE), 0,
                                                                         nchunks * US_STAGE * 4, 0x00020000);
    const unsigned utid = tid * 16;                    // bytes

    f32x4 rr[3], ur[8];
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto gload_raw = [&](int c) {
        const int k0 = c * KC;
        const bool s = k0 >= a.C1;
        const int so = (k0 - (s ? a.C1 : 0)) * 4;
#pragma unroll
        for (int j = 0; j < 3; ++j) rr[j] = bload(s ? rs1 : rs0, s ? roff1[j] : roff0[j], so);
    };
    auto gload_u = [&](int c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ur[j] = bload(rsu, utid, (c * US_STAGE + j * 1024) * 4);
    };
    auto store_raw = [&](int buf) {
        float* r = raws + buf * RAW_FLOATS;
        *reinterpret_cast<f32x4*>(r + rdst[0]) = rr[0];
        *reinterpret_cast<f32x4*>(r + rdst[1]) = rr[1];
        *reinterpret_cast<f32x4*>(r + rdst[2]) = rr[2];                      // slots past the 648th duplicate the last one
    };
    auto store_u = [&](int stage) {
        float* us = Us + stage * US_STAGE + tid * 4;
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(us + j * 1024) = ur[j];
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[xi][e] = 0.f;

    // Pipeline (one barrier per chunk).  Iteration i runs  MFMA(i) || transform(i+1)  after storing U(i+1) and
    // patch(i+2) from registers and issuing the global loads of patch(i+3) / U(i+2).
    gload_raw(0); gload_u(0);
    store_raw(0); store_u(0);
    if (nchunks > 1) { gload_raw(1); store_raw(1); }
    __syncthreads();
    input_transform(raws, Vs, tid);
    if (nchunks > 2) gload_raw(2);
    if (nchunks > 1) gload_u(1);
    __syncthreads();

    const int voff = (lane >> 5) * VPLANE + (wm * 32 + (lane & 31)) * 4;
    const int uoff = (lane >> 5) * UPLANE + (wn * 32 + (lane & 31)) * 4;
    // per-thread constants of the two input-transform items (tile row ty = tyb, tyb+4; channel quad cg; row i of B^T)
    const int it_cg = tid & 1, it_tx = (tid >> 1) & 7, it_i = (tid >> 4) & 3, it_ty = tid >> 6;
    const int it_ra = (it_i == 0) ? 0 : (it_i == 2 ? 2 : 1);
    const int it_rb = (it_i == 0) ? 2 : (it_i == 1 ? 2 : (it_i == 2 ? 1 : 3));
    const float it_sg = (it_i == 1) ? 1.f : -1.f;
    const f32x2 it_sg2 = {it_sg, it_sg};
    const int it_a = (2 * it_ty + it_ra) * RAW_ROW + it_tx * KC + it_cg * 4;      // + item*8*RAW_ROW + column offset
    const int it_b = (2 * it_ty + it_rb) * RAW_ROW + it_tx * KC + it_cg * 4;
    const int it_o = (it_i * 8 + it_cg) * VPLANE + (it_ty * 8 + it_tx) * 4;        // + item*32*4 + j*2*VPLANE
    const int us_w = tid * 4;

    // One branch-free basic block per chunk: the 64 MFMAs of chunk c with, in their shadow, the input transform of
    // chunk c+1, the LDS stores of U(c+1)/patch(c+2) and the global loads of U(c+2)/patch(c+3).  Past the last
    // chunk the same instructions run on clamped (valid) addresses and write LDS buffers nobody reads.
    const long long t_loop = a.dbg ? clock64() : 0;
    long long stp[17] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int c = 0; c < nchunks; ++c) {
        const int stage = c & 1;
        const float* vb = Vs + stage * VS_STAGE + voff;
        const float* ub = Us + stage * US_STAGE + uoff;
        const float* rsrc = raws + (stage ^ 1) * RAW_FLOATS;         // patch(c+1)
        float* vdst = Vs + (stage ^ 1) * VS_STAGE + it_o;
        float* usdst = Us + (stage ^ 1) * US_STAGE + us_w;
        float* rdstb = raws + stage * RAW_FLOATS;                    // receives patch(c+2)
        const int unext = min(c + 2, nchunks - 1) * (US_STAGE * 4);      // byte offset of U(c+2)
        const int k3 = min(c + 3, nchunks - 1) * KC;
        const bool s3 = k3 >= a.C1;
        const int rnext = (k3 - (s3 ? a.C1 : 0)) * 4;                   // byte offset of the channels of patch(c+3)
        const __amdgpu_buffer_rsrc_t rs3 = s3 ? rs1 : rs0;
        f32x4 av[2], bv[2];
        f32x4 da[2], db[2], tt[4];
        av[0] = *reinterpret_cast<const f32x4*>(vb);
        bv[0] = *reinterpret_cast<const f32x4*>(ub);
        // one step = the 4 MFMAs of transformed position xi plus a fixed slice of the other work
#define WINO_STEP(XI)                                                                                                  \
        {                                                                                                              \
            constexpr int xi = XI;                                                                                     \
            if constexpr (WINO_STEPTIME) { if (c == 2) stp[xi] = clock64(); }                                            \
            constexpr int ur_ = (xi < 4) ? xi : ((xi >= 6 && xi < 10) ? xi - 2 : -1);    /* unit whose LDS reads go here */ \
            constexpr int uf_ = (xi >= 1 && xi < 5) ? xi - 1 : ((xi >= 7 && xi < 11) ? xi - 3 : -1);   /* unit whose fma goes here */ \
            constexpr bool out_ = (xi == 5 || xi == 11), ust_ = xi < 8, pst_ = (xi >= 12 && xi < 15);                  \
            constexpr int n_read = (xi + 1 < 16 ? 2 : 0) + (ur_ >= 0 ? 2 : 0);                                         \
            constexpr int n_valu = (uf_ >= 0 ? 4 : 0) + (ust_ ? 2 : 0) + (out_ ? 16 : 0) + (pst_ ? 1 : 0);             \
            constexpr int n_write = (ust_ ? 1 : 0) + (out_ ? 4 : 0) + (pst_ ? 1 : 0);                                  \
            constexpr int n_vmem = (ust_ ? 1 : 0) + (pst_ ? 1 : 0);                                                    \
            if constexpr (xi + 1 < 16) {                                                                               \
                av[(xi + 1) & 1] = *reinterpret_cast<const f32x4*>(vb + (xi + 1) * 2 * VPLANE);                        \
                bv[(xi + 1) & 1] = *reinterpret_cast<const f32x4*>(ub + (xi + 1) * 2 * UPLANE);                        \
            }                                                                                                          \
            if constexpr (ur_ >= 0) {                                                                                  \
                constexpr int col = (((ur_ & 3) & 1) * 9 + ((ur_ & 3) >> 1)) * KC + (ur_ >> 2) * 8 * RAW_ROW;          \
                da[ur_ & 1] = *reinterpret_cast<const f32x4*>(rsrc + it_a + col);                                      \
                db[ur_ & 1] = *reinterpret_cast<const f32x4*>(rsrc + it_b + col);                                      \
            }                                                                                                          \
            if constexpr (WINO_FENCED) __builtin_amdgcn_sched_barrier(0);   /* the reads above lead the step */          \
            const f32x4 A = av[xi & 1], Bv = bv[xi & 1];                                                               \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.x, Bv.x, acc[xi], 0, 0, 0);                               \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.y, Bv.y, acc[xi], 0, 0, 0);                               \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.z, Bv.z, acc[xi], 0, 0, 0);                               \
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(A.w, Bv.w, acc[xi], 0, 0, 0);                               \
            if constexpr (uf_ >= 0) tt[uf_ & 3] = pk_fma(db[uf_ & 1], it_sg2, da[uf_ & 1]);                            \
            if constexpr (ust_) {                                           /* U(c+1) -> LDS, U(c+2) -> registers */   \
                *reinterpret_cast<f32x4*>(usdst + (xi & 7) * 1024) = ur[xi & 7];                                       \
                ur[xi & 7] = bload(rsu, utid, unext + (xi & 7) * 4096);                                                \
            }                                                                                                          \
            if constexpr (out_) {                                           /* outputs of item 0 / item 1 */           \
                float* o = vdst + (xi == 11 ? 32 * 4 : 0);                                                             \
                *reinterpret_cast<f32x4*>(o) = pk_sub(tt[0], tt[2]);                                            \
                *reinterpret_cast<f32x4*>(o + 2 * VPLANE) = pk_add(tt[1], tt[2]);                               \
                *reinterpret_cast<f32x4*>(o + 4 * VPLANE) = pk_sub(tt[2], tt[1]);                               \
                *reinterpret_cast<f32x4*>(o + 6 * VPLANE) = pk_sub(tt[1], tt[3]);                               \
            }                                                                                                          \
            if constexpr (pst_) {                                           /* patch(c+2) -> LDS, patch(c+3) -> registers */ \
                constexpr int j = pst_ ? xi - 12 : 0;                                                                  \
                *reinterpret_cast<f32x4*>(rdstb + rdst[j]) = rr[j];                                                    \
                rr[j] = bload(rs3, s3 ? roff1[j] : roff0[j], rnext);                                                   \
            }                                                                                                          \
            if constexpr (WINO_FENCED) {                                                                               \
                __builtin_amdgcn_sched_barrier(0);      /* nothing crosses a step boundary */                          \
            } else {                                                                                                   \
            /* pin the order: this step's LDS reads, then the MFMAs with the other work in their shadow */             \
            if constexpr (n_read > 0) __builtin_amdgcn_sched_group_barrier(0x100, n_read, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            if constexpr (n_valu > 0) __builtin_amdgcn_sched_group_barrier(0x002, (n_valu + 1) / 2, 0);                \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            if constexpr (n_write > 0) __builtin_amdgcn_sched_group_barrier(0x200, n_write, 0);                        \
            if constexpr (n_vmem > 0) __builtin_amdgcn_sched_group_barrier(0x020, n_vmem, 0);                          \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            if constexpr (n_valu > 1) __builtin_amdgcn_sched_group_barrier(0x002, n_valu / 2, 0);                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                         \
            }                                                                                                          \
        }
        WINO_STEP(0) WINO_STEP(1) WINO_STEP(2) WINO_STEP(3) WINO_STEP(4) WINO_STEP(5) WINO_STEP(6) WINO_STEP(7)
        WINO_STEP(8) WINO_STEP(9) WINO_STEP(10) WINO_STEP(11) WINO_STEP(12) WINO_STEP(13) WINO_STEP(14) WINO_STEP(15)
#undef WINO_STEP
        if constexpr (WINO_STEPTIME) { if (c == 2) stp[16] = clock64(); }
        __syncthreads();
    }
    const long long t_epi = a.dbg ? clock64() : 0;

    // ---- inverse transform A^T M A in registers, then a 16-byte epilogue.
    // The accumulator layout has the channel on the lane and pixels in registers: stored directly that is 64 dword stores
    // per lane and tile, and dword stores run at < 1 TB/s on this chip (they were HALF of the time of a K = 64 layer).
    // Each wave therefore turns its 128 pixels x 32 channels through a private LDS patch (the chunk buffers are free
    // now) so that a lane owns 4 consecutive channels of a pixel: mask / residual loads and the store are dwordx4
    // covering whole 128-byte pixel rows, 16 per lane instead of 64.
    constexpr int TSTR = 36;                                           // floats per patch row (32 channels + pad, 16-byte aligned)
    float* T = smem + wave * (128 * TSTR);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);       // tile of this wave's 32
        float s[4], t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = acc[j][e] + acc[4 + j][e] + acc[8 + j][e];
            t[j] = acc[4 + j][e] - acc[8 + j][e] - acc[12 + j][e];
        }
        float* o = T + (m * 4) * TSTR + (lane & 31);
        o[0] = s[0] + s[1] + s[2];                                     // output pixel (0,0) of the tile
        o[TSTR] = s[1] - s[2] - s[3];                                  // (0,1)
        o[2 * TSTR] = t[0] + t[1] + t[2];                              // (1,0)
        o[3 * TSTR] = t[1] - t[2] - t[3];                              // (1,1)
    }
    const int cq = lane & 7;                                           // channel quad of this lane (fixed over the 16 rows it handles)
    const int n4 = nb * BN + wn * 32 + cq * 4;
    // wave-uniform (n_split is a multiple of 32): readfirstlane tells the compiler so, else the buffer resources below are 'divergent'
    const int d = __builtin_amdgcn_readfirstlane((a.n_split > 0 && n4 >= a.n_split) ? 1 : 0);
    const int nc = n4 - (d ? a.n_split : 0);
    float* dst = d ? a.dst[1] : a.dst[0];
    const int dcs = d ? a.dst_cs[1] : a.dst_cs[0];
    const float* mask = d ? a.mask[1] : a.mask[0];
    const int mmode = d ? a.mask_mode[1] : a.mask_mode[0], accum = d ? a.accum[1] : a.accum[0];
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n4);
    // Epilogue traffic through buffer resources on image b of the destination geometry: a lane whose pixel is outside the image
    // gets an out-of-range offset (loads return zeros, stores are dropped), so there is no per-store branch and no 64-bit address
    // arithmetic; the activation is branch-free (max(v, slope*v) covers none / LeakyReLU / ReLU with slope 1 / 0.2 / 0).
    const int64_t img = (int64_t)b * a.H * a.W * dcs;
    const int ibytes = a.H * a.W * dcs * 4;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dst + img), 0, ibytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)((mmode ? mask : dst) + img), 0, ibytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)((a.addsrc ? a.addsrc : dst) + img), 0, ibytes, 0x00020000);
    unsigned off[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = (k * 64 + lane) >> 3;                            // patch row = tile * 4 + sub-pixel
        const int m = wm * 32 + (q >> 2), p = q & 3;
        const int y = y0 + 2 * (m >> 3) + (p >> 1), x = x0 + 2 * (m & 7) + (p & 1);
        off[k] = (y < a.H && x < a.W) ? (unsigned)(((y * a.W + x) * dcs + nc) * 4) : 0x80000000u;
    }
    f32x4 mk[16], ad[16];                                              // batched: one HBM latency for all 16
    if (mmode) {
#pragma unroll
        for (int k = 0; k < 16; ++k) mk[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, off[k], 0, 0));
    }
    if (a.addsrc) {
#pragma unroll
        for (int k = 0; k < 16; ++k) ad[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off[k], 0, 0));
    }
    const float slope = mmode == 1 ? 0.2f : 0.f;
    const float aslope = a.act == 1 ? 0.2f : (a.act == 2 ? 0.f : 1.f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int q = (k * 64 + lane) >> 3;
        f32x4 v = *reinterpret_cast<const f32x4*>(T + q * TSTR + cq * 4) + bias4;
        if (a.addsrc) v += ad[k];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], aslope * v[c]);
        if (mmode) {
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] *= (mk[k][c] > 0.f) ? 1.f : slope;
        }
        if (accum) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, off[k], 0, 0));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rd, off[k], 0, 0);
    }
    if (a.dbg && tid == 0) {
        long long* d = a.dbg + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4;
        d[0] = t_start; d[1] = t_loop; d[2] = t_epi; d[3] = clock64();
        if constexpr (WINO_STEPTIME) {        // second region of the buffer: 17 stamps per workgroup
            long long* e = a.dbg + (int64_t)gridDim.x * gridDim.y * 4 + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 17;
#pragma unroll
            for (int i = 0; i < 17; ++i) e[i] = stp[i];
        }
    }
}

static_assert(KC == 8 && BN == 64 && UPLANE == 256 && US_STAGE == 8192, "csrc/pack_jobs.hip restates the U layout");

#if PNNP_WINO_DEBUG
long long* g_wino_dbg = nullptr;       // set by pnnp_wino_set_debug; profiling builds only
#endif

int wino_launch(WinoArgs& a, hipStream_t st) {
#if PNNP_WINO_DEBUG
    a.dbg = g_wino_dbg;
#else
    a.dbg = nullptr;
#endif
    if (a.K % KC || a.N % BN || a.C1 % KC || (a.n_split % 32)) return PNNP_E_UNSUPPORTED;
    if ((int64_t)a.B * a.H * a.W * (a.src_cs[0] > a.src_cs[1] ? a.src_cs[0] : a.src_cs[1]) >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    // buffer-resource addressing: byte offsets inside one image (and inside one channel block of U) are 32-bit, 2^31 marks out-of-range
    if ((int64_t)a.H * a.W * (a.src_cs[0] > a.src_cs[1] ? a.src_cs[0] : a.src_cs[1]) * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    if ((int64_t)a.K * 16 * BN * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    a.tiles_x = (a.W + 15) / 16; a.tiles_y = (a.H + 15) / 16;
    // 16-byte epilogue accesses
    if ((a.dst_cs[0] & 3) || (a.dst_cs[1] & 3) ||
        ((((uintptr_t)a.dst[0]) | ((uintptr_t)a.dst[1]) | ((uintptr_t)a.bias) | ((uintptr_t)a.mask[0]) | ((uintptr_t)a.mask[1]) | ((uintptr_t)a.addsrc)) & 15))
        return PNNP_E_INVALID;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, wino_kernel, SMEM_FLOATS * 4) != PNNP_OK) return PNNP_E_LAUNCH;
    const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.B), (unsigned)(a.N / BN));
    hipLaunchKernelGGL(wino_kernel, grid, dim3(256), SMEM_FLOATS * 4, st, a);
    return pnnp_launch_status();
}

}  // namespace

extern "C" {

// floats of a Winograd-packed weight: 16 * K * N  (K, N multiples of 8 / 64)
int64_t pnnp_wino_weight_floats(int Cout, int Cin) { return (int64_t)16 * Cout * Cin; }

// 1 when pnnp_conv3x3_wino_* accepts the layer (K = channels read, N = channels written)
int pnnp_wino_supported(int K, int N) { return (K % KC == 0 && N % BN == 0) ? 1 : 0; }

int pnnp_pack_conv_weight_wino_f32(const float* w, float* fwd, float* dgrad, int Cout, int Cin, void* stream) {
    PnnpPackJob jobs[2]; int n = 0;
    const int rc = pnnp_pack_jobs_add_wino(jobs, &n, 2, w, fwd, dgrad, Cout, Cin);      // the filter transform lives in csrc/pack_jobs.hip
    return rc != PNNP_OK ? rc : pnnp_pack_jobs_f32(jobs, n, stream);
}

// y = act(conv3x3(cat[x1,x2]) + bias), same contract as pnnp_conv_fwd_f32 with taps = 9.
int pnnp_conv3x3_wino_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* u_fwd, const float* bias,
                              const float* residual, float* y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !u_fwd || !y || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 <= 0)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    WinoArgs a{};
    a.src[0] = x1; a.src_cs[0] = C1; a.src[1] = x2 ? x2 : x1; a.src_cs[1] = x2 ? C2 : C1; a.C1 = C1;
    a.K = C1 + (x2 ? C2 : 0); a.N = Cout; a.u = u_fwd; a.B = B; a.H = H; a.W = W;
    a.dst[0] = y; a.dst[1] = y; a.dst_cs[0] = a.dst_cs[1] = Cout; a.bias = bias; a.act = act; a.addsrc = residual;
    return wino_launch(a, as_stream(stream));
}

// backward-data, same contract as pnnp_conv_bwd_data_f32 with taps = 9.
int pnnp_conv3x3_wino_bwd_data_f32(const float* g, int Cout, const float* u_dgrad,
                                   float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                   float* dx2, int C2, const float* mask2, int mode2, int accum2,
                                   int B, int H, int W, void* stream) {
    if (!g || !u_dgrad || !dx1 || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    WinoArgs a{};
    a.src[0] = g; a.src[1] = g; a.src_cs[0] = a.src_cs[1] = Cout; a.C1 = Cout;
    a.K = Cout; a.N = C1 + (dx2 ? C2 : 0); a.u = u_dgrad; a.B = B; a.H = H; a.W = W;
    a.dst[0] = dx1; a.dst_cs[0] = C1; a.mask[0] = mask1; a.mask_mode[0] = mask1 ? mode1 : 0; a.accum[0] = accum1;
    a.dst[1] = dx1; a.dst_cs[1] = C1;
    if (dx2) {
        a.n_split = C1;
        a.dst[1] = dx2; a.dst_cs[1] = C2; a.mask[1] = mask2; a.mask_mode[1] = mask2 ? mode2 : 0; a.accum[1] = accum2;
    }
    return wino_launch(a, as_stream(stream));
}

// backward-data through an identity shortcut: dx = (conv_bwd_data(g) + addsrc) * act'(mask), same contract as
// pnnp_conv_bwd_data_res_f32 with taps = 9.
int pnnp_conv3x3_wino_bwd_data_res_f32(const float* g, int Cout, const float* u_dgrad, float* dx, int C1,
                                       const float* addsrc, const float* mask, int mode, int B, int H, int W, void* stream) {
    if (!g || !u_dgrad || !dx || !addsrc || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    WinoArgs a{};
    a.src[0] = g; a.src[1] = g; a.src_cs[0] = a.src_cs[1] = Cout; a.C1 = Cout;
    a.K = Cout; a.N = C1; a.u = u_dgrad; a.B = B; a.H = H; a.W = W;
    a.dst[0] = a.dst[1] = dx; a.dst_cs[0] = a.dst_cs[1] = C1; a.addsrc = addsrc;
    a.mask[0] = a.mask[1] = mask; a.mask_mode[0] = a.mask_mode[1] = mask ? mode : 0;
    return wino_launch(a, as_stream(stream));
}

#if PNNP_WINO_DEBUG
// Profiling builds only: when buf is non-null every later Winograd launch writes, per workgroup w (= blockIdx.y*gridDim.x+blockIdx.x),
// buf[4w..4w+3] = clock64() at kernel entry, main-loop entry, epilogue entry and exit.  buf must hold 4 * workgroups entries.
int pnnp_wino_set_debug(long long* buf) { g_wino_dbg = buf; return PNNP_OK; }
#endif

}  // extern "C"
