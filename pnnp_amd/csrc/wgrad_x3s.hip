// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on the bf16 matrix cores with SPECIALISED waves (round 4) -- the arithmetic, the LDS
// images, the slabs and the reduce of csrc/wgrad_x3.hip, for layers whose channel counts are multiples of 64.
//
// wgrad_x3.hip's eight waves each carry a 32 x 32 x 9-tap accumulator block (144 registers) and split their share of the next pixel tile
// between their own MFMAs; a wave's own vector instructions are ADDED to its MFMA time (tools/ubench/mfma_valu_coissue.hip), and 144 + operands
// leave no room for a third wave per SIMD.  Here a workgroup is
//   12 CONSUMER waves (three per SIMD): wave (mo, no, tr) owns the 32 x 32 block (mo, no) of a 64 x 64 output tile for the three taps of filter
//     ROW tr -- 48 accumulator registers -- and issues transposed LDS reads + MFMAs only;
//   4 PRODUCER waves (one per SIMD): the next pixel tile (G: 2 rows x 32 px x 64 channels, X: its 4 x 34 halo x 64 channels) fp32 global ->
//     registers (one tile ahead) -> hi / mid / lo split -> the other LDS image; the bias gradient (column sums of G) on the way.
// 1024 threads, <= 128 registers each, one barrier per pixel tile (72 MFMAs per consumer).
#include "common.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Wx3sArgs {                       // (= Wx3Args of csrc/wgrad_x3.hip)
    const float* G; int Gcs;            // [B][H][W][Gcs], channels [0, M) used
    const float* X[2]; int Xcs[2];      // n < n_split -> X[0][n], else X[1][n - n_split]
    int n_split;
    int B, H, W, M, N;
    float* slab;                        // [Z][9][M][N]
    float* bias_slab;                   // [Z][M] or null
    int Z;
};
int pnnp_wx3s_launch(const Wx3sArgs& a, hipStream_t s);

namespace {

constexpr int NCW = 12, NPW = 4, NTHR = 64 * (NCW + NPW);
constexpr int TH = 2, KS = TH * 2;                                 // pixel tile: 2 rows x 32 px = four 16-pixel k-steps
constexpr int GPIX = TH * 32, XR = TH + 2, XC = 34, XPIX = XR * XC; // 64 / 136 pixels
constexpr int G_BYTES = 2 * 3 * GPIX * 64, X_BYTES = 2 * 3 * XPIX * 64, IMG_BYTES = G_BYTES + X_BYTES, LDS_BYTES = 2 * IMG_BYTES;   // 24576 + 52224; 153600
constexpr int NG = GPIX * 8 / 128, NX = (XPIX * 8 + 127) / 128;    // float4 staging slots per producer thread: 4 of G, 9 of X (two waves per 32-channel block)
static_assert(LDS_BYTES <= 160 * 1024 && (GPIX * 8) % 128 == 0, "LDS budget; G slots divide evenly (bias sums count every pixel once)");
constexpr unsigned OOB = 0x80000000u;
#define WXS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifndef WXS_M16
#define WXS_M16 0                      // 1: v_mfma_f32_16x16x32_bf16 with two pieces concatenated along K (parity-green, but 4-5 % SLOWER than 0 =
#endif                                 //    v_mfma_f32_32x32x16_bf16: 44 transposed reads per k-step instead of 24; profiles/r4/ab_wgrad_specialised.txt)
#ifndef WX3_ALT_SIGN
#define WX3_ALT_SIGN 1                 // odd pixel splits accumulate -G * X (csrc/wgrad_x3.hip: the matrix core's accumulation rounds toward minus infinity)
#endif

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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__global__ void __launch_bounds__(NTHR, 1)
wgrad_x3s_kernel(const Wx3sArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 11 consumers, 12 .. 15 producers

    const int n_tiles = a.N / 64;
    int id = blockIdx.x;
    const int z = id % a.Z; id /= a.Z;
    const int ni = id % n_tiles, mi = id / n_tiles;
    const int m0 = mi * 64, n0 = ni * 64;
    const int tiles_x = (a.W + 31) >> 5, tiles_y = (a.H + TH - 1) / TH;
    const int ntile = tiles_x * tiles_y * a.B;
    if (z >= ntile) return;                                          // (Z <= ntile: never)

    if (wave >= NCW) {
        // =============================================== PRODUCER ===============================================
        const int pw = wave - NCW, blk = pw >> 1;                    // this wave stages 32-channel block blk of G and of X, with one other wave
        const int lt = (pw & 1) * 64 + lane;                         // 0 .. 127
        const int q8 = lane & 7;                                     // channel quad of the block (8 lanes read a pixel's 128 contiguous bytes)
        unsigned g_off[NG]; int g_r[NG], g_c[NG], g_dst[NG];
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int pix = (lt + 128 * k) >> 3;
            g_r[k] = pix >> 5; g_c[k] = pix & 31;
            g_off[k] = (unsigned)((g_r[k] * a.W + g_c[k]) * a.Gcs + q8 * 4) * 4u;
            g_dst[k] = WXS_M16 ? blk * 3 * GPIX * 64 + ((q8 >> 2) * GPIX + pix) * 32 + (q8 & 3) * 8      // [block][piece][16-channel half][pixel][32 B]
                               : (blk * 3 * GPIX + pix) * 64 + q8 * 8;                                   // [block][piece][pixel][64 B]; + piece * GPIX * 64
        }
        const int xd = (n0 + 32 * blk >= a.n_split) ? 1 : 0;         // wave-uniform source of this wave's X block
        const int xch0 = n0 + 32 * blk - (xd ? a.n_split : 0);
        const int xcs = a.Xcs[xd];
        unsigned x_off[NX]; int x_r[NX], x_c[NX], x_dst[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            int j = lt + 128 * k;
            if (j >= XPIX * 8) j -= 128;                             // a slot past the end repeats the thread's previous one
            const int pix = j >> 3;
            x_r[k] = pix / XC; x_c[k] = pix - x_r[k] * XC;          // halo coordinates: image pixel (y0 - 1 + r, x0 - 1 + c)
            x_off[k] = (unsigned)((x_r[k] * a.W + x_c[k]) * xcs + q8 * 4) * 4u;
            x_dst[k] = G_BYTES + (WXS_M16 ? blk * 3 * XPIX * 64 + ((q8 >> 2) * XPIX + pix) * 32 + (q8 & 3) * 8 : (blk * 3 * XPIX + pix) * 64 + q8 * 8);
        }
        const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.G + m0 + 32 * blk), 0, 0x7fffffff, 0x00020000);
        const int xshift = (a.W + 1) * xcs;                          // the X resource starts one row + one pixel BEFORE the tensor
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X[xd] + xch0 - xshift), 0, 0x7fffffff, 0x00020000);
        f32x4 rg[NG], rx[NX];
        float bsum[4] = {0.f, 0.f, 0.f, 0.f};
        const unsigned sflip = (WX3_ALT_SIGN && (z & 1)) ? 0x80000000u : 0u;
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
                rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, bad < 0 ? OOB : g_off[k], gso, 0));
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) {
                const int yy = y0 - 1 + x_r[k], xx = x0 - 1 + x_c[k];
                const int bad = yy | (a.H - 1 - yy) | xx | (a.W - 1 - xx);
                rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, bad < 0 ? OOB : x_off[k], xso, 0));
            }
        };
        auto stage = [&](f32x4 v, char* ib, int dst, int pstride) {
            unsigned h0, m0_, l0, h1, m1, l1;
            split2(v.x, v.y, h0, m0_, l0);
            split2(v.z, v.w, h1, m1, l1);
            *reinterpret_cast<u32x2*>(ib + dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(ib + dst + pstride) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(ib + dst + 2 * pstride) = u32x2{l0, l1};
        };
        auto stage_tile = [&](int img) {
            char* ib = smem + img * IMG_BYTES;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                f32x4 v = rg[k];
                bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;     // bias gradient: column sums of G (unsigned)
                v = f32x4{__uint_as_float(__float_as_uint(v.x) ^ sflip), __uint_as_float(__float_as_uint(v.y) ^ sflip),
                          __uint_as_float(__float_as_uint(v.z) ^ sflip), __uint_as_float(__float_as_uint(v.w) ^ sflip)};
                stage(v, ib, g_dst[k], GPIX * 64);
            }
#pragma unroll
            for (int k = 0; k < NX; ++k) stage(rx[k], ib, x_dst[k], XPIX * 64);
        };
        // the first tile straight into image 0, the second into the registers
        load_tile(z);
        stage_tile(0);
        if (z + a.Z < ntile) load_tile(z + a.Z);
        int img = 0;
        for (int tile = z; tile < ntile; tile += a.Z) {
            WXS_BARRIER();                                          // image img is complete; every consumer is done with the other one
            if (tile + a.Z < ntile) {
                stage_tile(img ^ 1);                                // the next tile (requested a whole tile ago)
                if (tile + 2 * a.Z < ntile) load_tile(tile + 2 * a.Z);
            }
            img ^= 1;
        }
        // ---- bias gradient of this pixel split: add up the threads that share (blk, q8) through LDS (the images are dead)
        WXS_BARRIER();
        if (a.bias_slab && ni == 0) {                               // block-uniform
            float* bs = reinterpret_cast<float*>(smem);             // [2 blocks][8 quads][4][16 slots]
            const int slot = (pw & 1) * 8 + (lane >> 3);
#pragma unroll
            for (int c = 0; c < 4; ++c) bs[((blk * 8 + q8) * 4 + c) * 16 + slot] = bsum[c];
        }
        WXS_BARRIER();
        if (a.bias_slab && ni == 0 && pw < 1) {                      // one producer wave: 64 channels
            float* bs = reinterpret_cast<float*>(smem);
            const int b2 = lane >> 5, ch = lane & 31;
            float s = 0.f;
            for (int k = 0; k < 16; ++k) s += bs[((b2 * 8 + (ch >> 2)) * 4 + (ch & 3)) * 16 + k];
            a.bias_slab[(int64_t)z * a.M + m0 + b2 * 32 + ch] = (WX3_ALT_SIGN && (z & 1)) ? -s : s;      // (the reduce kernel adds odd splits with a minus sign)
        }
        return;
    }

    // =============================================== CONSUMER ===============================================
    const int tr = wave % 3, no = (wave / 3) & 1, mo = wave / 6;    // filter row, 32-column block, 32-row block
    const int l31 = lane & 31, half = lane >> 5;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // transposed-read lane geometry (csrc/wgrad_x3.hip): 16-lane group g reads channels 16 (g & 1) .., pixels 8 (g >> 1) ..; inside a group lane
    // 4 q + p supplies the address of pixel row q, channel chunk 4 p
    const int tr_lane = ((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto tr_read = [&](const char* base) {                           // 8 pixels x 1 channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * 64));
        const u32x2 a0 = __builtin_bit_cast(u32x2, lo), a1 = __builtin_bit_cast(u32x2, hi);
        return u32x4{a0.x, a0.y, a1.x, a1.y};
    };
    int img = 0;
    for (int tile = z; tile < ntile; tile += a.Z) {
        WXS_BARRIER();
        const char* gimg = smem + img * IMG_BYTES;
        const char* ximg = gimg + G_BYTES;
#if WXS_M16
        // 16 x 16 x 32 with two PIECES concatenated along K (csrc/wgrad_x3.hip WX3_M16, csrc/conv_x3.hip): the LDS images keep the two 16-channel
        // halves of a block in separate planes; k-block (lane >> 4) & 1 takes pixels {0-3, 8-11} or {4-7, 12-15} of the k-step; lanes 0-31 read the
        // first piece of a form, lanes 32-63 the second.  Forms: A0 = [hi | mid], A1 = [hi | lo]; B0 = [hi' | hi'], B1 = [mid' | mid'], B2 = [lo' | hi'].
        const int lb = ((((lane >> 4) & 1) * 4 + ((lane & 15) >> 2)) * 32) + (lane & 3) * 8, hi2 = lane >> 5;
        constexpr int PSG = GPIX * 64, PSX = XPIX * 64;
        const int la0 = lb + hi2 * PSG, la1 = lb + hi2 * 2 * PSG, lb2 = lb + (1 - hi2) * 2 * PSX;
        auto half_read = [&](const char* base) {
            return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        };
        auto a_base = [&](int ks, int mh, int f) { return gimg + (mo * 3) * PSG + (mh * GPIX + (ks >> 1) * 32 + (ks & 1) * 16) * 32 + (f ? la1 : la0); };
        auto b_base = [&](int ks, int dx, int nh, int f) {
            return ximg + (no * 3) * PSX + (nh * XPIX + ((ks >> 1) + tr) * XC + (ks & 1) * 16 + dx) * 32 + (f == 0 ? lb : (f == 1 ? lb + PSX : lb2));
        };
        auto put_half = [](u32x4& d, int h, u32x2 v) { if (h == 0) { d.x = v.x; d.y = v.y; } else { d.z = v.x; d.w = v.y; } };
        u32x4 Av[2][2][2], Bv[2][3];                                 // [k-step parity][16-row half][form], [step parity][form]
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int f = 0; f < 2; ++f) { put_half(Av[0][mh][f], 0, half_read(a_base(0, mh, f))); put_half(Av[0][mh][f], 1, half_read(a_base(0, mh, f) + 256)); }
#pragma unroll
        for (int f = 0; f < 3; ++f) { put_half(Bv[0][f], 0, half_read(b_base(0, 0, 0, f))); put_half(Bv[0][f], 1, half_read(b_base(0, 0, 0, f) + 256)); }
        constexpr int NSTEP = KS * 6;                                // steps (k-step, tap, 16-column half) of six MFMAs
        static_for<0, NSTEP * 6>([&](auto GI) {
            constexpr int gi = decltype(GI)::value, S = gi / 6, Gq = gi % 6, ks = S / 6, st = S % 6, dx = st / 2, nh = st % 2, mh = Gq / 3, pr = Gq % 3;
            constexpr int cur = S & 1;
            if constexpr (Gq == 0) __builtin_amdgcn_sched_barrier(0);
            f32x4 c4 = {acc[dx][4 * (2 * mh + nh)], acc[dx][4 * (2 * mh + nh) + 1], acc[dx][4 * (2 * mh + nh) + 2], acc[dx][4 * (2 * mh + nh) + 3]};
            if constexpr (pr == 0) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][1]), __builtin_bit_cast(bf16x8, Bv[cur][2]), c4, 0, 0, 0);
            else if constexpr (pr == 1) c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][0]), __builtin_bit_cast(bf16x8, Bv[cur][1]), c4, 0, 0, 0);
            else c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Av[ks & 1][mh][0]), __builtin_bit_cast(bf16x8, Bv[cur][0]), c4, 0, 0, 0);
            acc[dx][4 * (2 * mh + nh)] = c4.x; acc[dx][4 * (2 * mh + nh) + 1] = c4.y; acc[dx][4 * (2 * mh + nh) + 2] = c4.z; acc[dx][4 * (2 * mh + nh) + 3] = c4.w;
            // the next step's B words: one transposed read per gap (form Gq / 2, half Gq % 2)
            if constexpr (S + 1 < NSTEP) {
                constexpr int S1 = S + 1, ks1 = S1 / 6, dx1 = (S1 % 6) / 2, nh1 = S1 % 2;
                put_half(Bv[cur ^ 1][Gq >> 1], Gq & 1, half_read(b_base(ks1, dx1, nh1, Gq >> 1) + (Gq & 1) * 256));
            }
            // the next k-step's A words: eight reads in the gaps of its predecessor's last two steps
            if constexpr (ks + 1 < KS && st >= 4) {
                constexpr int j = (st - 4) * 6 + Gq;
                if constexpr (j < 8) put_half(Av[(ks + 1) & 1][j >> 2][(j >> 1) & 1], j & 1, half_read(a_base(ks + 1, j >> 2, (j >> 1) & 1) + (j & 1) * 256));
            }
            __builtin_amdgcn_sched_barrier(0);
        });
#else
        u32x4 av[2][3], bv[2][3];
        auto gload = [&](int ks, u32x4 (&ax)[3]) {
            const char* gbase = gimg + ((mo * 3) * GPIX + (ks >> 1) * 32 + (ks & 1) * 16) * 64 + tr_lane;
#pragma unroll
            for (int p = 0; p < 3; ++p) ax[p] = tr_read(gbase + p * GPIX * 64);
        };
        auto xload = [&](int ks, int dx, u32x4 (&bx)[3]) {
            const char* xbase = ximg + ((no * 3) * XPIX + ((ks >> 1) + tr) * XC + (ks & 1) * 16 + dx) * 64 + tr_lane;
#pragma unroll
            for (int p = 0; p < 3; ++p) bx[p] = tr_read(xbase + p * XPIX * 64);
        };
        gload(0, av[0]);
        xload(0, 0, bv[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int s = ks * 3 + dx;
                // the next step's X words (and, at a k-step's last tap, the next k-step's G words) one step ahead
                if (s + 1 < KS * 3) xload((s + 1) / 3, (s + 1) % 3, bv[(s + 1) & 1]);
                if (dx == 2 && ks + 1 < KS) gload(ks + 1, av[(ks + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const u32x4 (&ax)[3] = av[ks & 1];
                const u32x4 (&bx)[3] = bv[s & 1];
                // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
#define WXS_MFMA(PA, PB) acc[dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax[PA]), __builtin_bit_cast(bf16x8, bx[PB]), acc[dx], 0, 0, 0)
                WXS_MFMA(0, 2); WXS_MFMA(2, 0); WXS_MFMA(1, 1); WXS_MFMA(0, 1); WXS_MFMA(1, 0); WXS_MFMA(0, 0);
#undef WXS_MFMA
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        img ^= 1;
    }
    // ---- the slab [z][tap][m][n]
    const int64_t slab_base = (int64_t)z * a.M * a.N * 9;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int t = tr * 3 + dx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // 32 x 32 x 16: column l31, row (r & 3) + 8 (r >> 2) + 4 half;  16 x 16 x 32: block (mh, nh) = (r >> 3, (r >> 2) & 1), column lane & 15, row 4 (lane >> 4) + (r & 3)
            const int n = n0 + no * 32 + (WXS_M16 ? 16 * ((r >> 2) & 1) + (lane & 15) : l31);
            const int m = m0 + mo * 32 + (WXS_M16 ? 16 * (r >> 3) + 4 * (lane >> 4) + (r & 3) : (r & 3) + 8 * (r >> 2) + 4 * half);
            a.slab[slab_base + ((int64_t)t * a.M + m) * a.N + n] = acc[dx][r];
        }
    }
    WXS_BARRIER();                                                  // (the producers' bias reduction: two more barriers for every wave)
    WXS_BARRIER();
}

}  // namespace

int pnnp_wx3s_launch(const Wx3sArgs& a, hipStream_t s) {
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, wgrad_x3s_kernel, LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int blocks = (a.M / 64) * (a.N / 64) * a.Z;
    hipLaunchKernelGGL(wgrad_x3s_kernel, dim3(blocks), dim3(NTHR), LDS_BYTES, s, a);
    return pnnp_launch_status();
}
