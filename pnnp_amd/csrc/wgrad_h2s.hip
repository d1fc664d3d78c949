// Weight gradient of a 3x3 / stride 1 / pad 1 convolution on the fp16 matrix cores, float32 operands split into two scaled fp16 pieces
// (csrc/h2.h) -- the workgroup of csrc/wgrad_x3s.hip (12 consumer waves: block (mo, no) x filter row tr x pixel split wk; 4 producer waves;
// pixel-major LDS images read through ds_read_b64_tr_b16; slabs with alternating signs, one reduce) with THREE v_mfma_f32_32x32x16_f16 per
// (16-pixel k-step, tap) where the bf16x3 kernel issues six: (hi, lo') (lo, hi') (hi, hi').  Both operands are activations / gradients split
// on the fly: G with 2^se_g (odd pixel splits: -2^se_g, the alternating sign costs nothing), X with 2^se_x from the amax slots of the
// tensors; the slab values are multiplied by 2^-(se_g + se_x) where they leave the accumulators.  The LDS images are two planes instead of
// three, so pixel tiles could be taller; only the 32 x 64 tile's (3 rows instead of 2) fit the producers' registers without spilling.
#include "common.h"
#include "h2.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct Wh2sArgs {                       // (Wh2sArgs of csrc/wgrad_x3s.hip + the amax slots)
    const float* G; int Gcs;            // [B][H][W][Gcs], channels [0, M) used
    const float* X[2]; int Xcs[2];      // n < n_split -> X[0][n], else X[1][n - n_split]
    int n_split;
    int B, H, W, M, N;
    float* slab;                        // [Z][9][M][N]
    float* bias_slab;                   // [Z][M] or null
    int Z;
    const unsigned* amax_g; const unsigned* amax_x[2];      // amax slots of G and of the X tensor(s) (csrc/h2.h); amax_x[1] null without a second one
};
int pnnp_wh2s_launch(const Wh2sArgs& a, hipStream_t s);
int pnnp_wh2s_th(int M, int N);

namespace {

constexpr int NCW = 12, NPW = 4, NTHR = 64 * (NCW + NPW);
constexpr int XC = 34;
constexpr unsigned OOB = 0x80000000u;
#define WHS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef WHS_STAMPS                     // debug build: cycle sums per wave, dumped into the slab (tools/whs_stamps.py; the results are then garbage)
#define WHS_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define WHS_T(v)
#endif
#ifndef WHS_G_AUX
#define WHS_G_AUX 2                    // cache-policy bits of the producers' G loads: 2 = nt -- G is read exactly once by this kernel (X's halo rows are shared between
#endif                                 // tiles and keep the default): step +0.3 % on two boxes, config 5 equal (profiles/r6/ab_stream_load_policy.txt)
#ifndef WHS_ROLL
#define WHS_ROLL 1                     // producers: rolling refill of the staging registers (see roll_tile; 0 = round 5's order)
#endif
#ifndef WX3_ALT_SIGN
#define WX3_ALT_SIGN 1                 // odd pixel splits accumulate -G * X (csrc/wgrad_x3.hip: the matrix core's accumulation rounds toward minus infinity)
#endif

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int MO, int NO, int TH, int MW = 1> struct WsCfg {
    // MW: 32 x 32 blocks along M that ONE consumer owns (1, or 2 = both of a 64-row tile: the X words of a tap then feed two blocks, 20 transposed
    // reads per 18 MFMAs instead of 16 per 9 -- this kernel is LDS-bandwidth-bound with one block per wave: 12 waves x 16 reads x 512 B per k-step
    // are 89 % of the LDS's 128 B / clk at full matrix rate)
    static constexpr int WK = 4 * MW / (MO * NO);                  // consumers that share a (block, filter row): pixel split inside the workgroup
    static constexpr int KS = TH * 2, KSW = KS / WK;               // 16-pixel k-steps per pixel tile; per consumer
    static constexpr int GPIX = TH * 32, XPIX = (TH + 2) * XC;
    static constexpr int G_BYTES = MO * 2 * GPIX * 64, X_BYTES = NO * 2 * XPIX * 64, IMG_BYTES = G_BYTES + X_BYTES, LDS_BYTES = 2 * IMG_BYTES;
    static constexpr int GT = 256 / MO, XT = 256 / NO;             // producer threads per 32-channel block of G / X
    static constexpr int NG = GPIX * 8 / GT, NX = (XPIX * 8 + XT - 1) / XT;      // float4 staging slots per producer thread
    static_assert(MO * NO * WK == 4 * MW && KSW * WK == KS && (MW == 1 || MW == MO), "wave layout");
    static_assert((GPIX * 8) % GT == 0, "G slots divide evenly (the bias sums count every pixel once)");
    static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= NCW * 16 * 64 * 4, "LDS budget (images; the final reduction aliases them, one accumulator block at a time)");
};

// hi = f16(a s), lo = f16(a s - hi) of two values, packed (csrc/conv_h2s.hip)
__device__ __forceinline__ void split_h2(float a0, float a1, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(s), "v"(h));
    hi = h; lo = l;
}

template <int MO, int NO, int TH, int MW>
__global__ void __launch_bounds__(NTHR, 1)
wgrad_h2s_kernel(const Wh2sArgs a) {
    using Cfg = WsCfg<MO, NO, TH, MW>;
    constexpr int WK = Cfg::WK, KSW = Cfg::KSW, GPIX = Cfg::GPIX, XPIX = Cfg::XPIX, G_BYTES = Cfg::G_BYTES, IMG_BYTES = Cfg::IMG_BYTES, NG = Cfg::NG, NX = Cfg::NX;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 11 consumers, 12 .. 15 producers

    unsigned axb = a.amax_x[0] ? a.amax_x[0][0] : 0u;
    if (a.amax_x[1]) { const unsigned a2 = a.amax_x[1][0]; axb = a2 > axb ? a2 : axb; }
    const int se_x = __builtin_amdgcn_readfirstlane(pnnp_h2_scale_exp(axb));
    const int se_g = __builtin_amdgcn_readfirstlane(a.amax_g ? pnnp_h2_scale_exp(a.amax_g[0]) : 0);
    const int n_tiles = a.N / (32 * NO);
    int id = blockIdx.x;
    const int z = id % a.Z; id /= a.Z;
    const int ni = id % n_tiles, mi = id / n_tiles;
    const int m0 = mi * 32 * MO, n0 = ni * 32 * NO;
    const int tiles_x = (a.W + 31) >> 5, tiles_y = (a.H + TH - 1) / TH;
    const int ntile = tiles_x * tiles_y * a.B;
    if (z >= ntile) return;                                          // (Z <= ntile: never)

    if (wave >= NCW) {
        // =============================================== PRODUCER ===============================================
        const int pw = wave - NCW;
        const int q8 = lane & 7;                                     // channel quad of the block (8 lanes read a pixel's 128 contiguous bytes)
        // the waves that stage 32-channel block gblk of G (GT threads) / xblk of X (XT threads): wave-uniform, like the tensors behind them
        const int gblk = MO == 2 ? pw >> 1 : 0, lg = MO == 2 ? (pw & 1) * 64 + lane : pw * 64 + lane;
        const int xblk = NO == 2 ? pw >> 1 : 0, lx = NO == 2 ? (pw & 1) * 64 + lane : pw * 64 + lane;
        // Staging slots WITHOUT per-slot address registers (with them -- 4 per slot -- the taller pixel tiles that two planes per operand leave room
        // for did not fit 128 registers).  G: slot k of a thread is pixel gp0 + GSTEP k of the 32-wide tile, i.e. row (GSTEP k) >> 5 (a compile-time
        // number) and column gp0 + (GSTEP k & 31): one base offset, the rest is a scalar.  X: the (TH + 2) x 34 halo does not divide that way: one
        // packed (row << 8 | column) per slot, everything else derived per tile.
        constexpr int GSTEP = Cfg::GT / 8, XSTEP = Cfg::XT / 8;
        const int gp0 = lg >> 3;                                     // < GSTEP <= 32
        const unsigned g_base = (unsigned)(gp0 * a.Gcs + q8 * 4) * 4u;
        const int g_dst0 = (gblk * 2 * GPIX + gp0) * 64 + q8 * 8;   // byte offset in an image; + GSTEP k * 64; + piece * GPIX * 64
        const int xd = (n0 + 32 * xblk >= a.n_split) ? 1 : 0;        // wave-uniform source of this wave's X block
        const int xch0 = n0 + 32 * xblk - (xd ? a.n_split : 0);
        const int xcs = a.Xcs[xd];
        int x_rc[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            int j = lx + Cfg::XT * k;
            if (j >= XPIX * 8) j -= Cfg::XT;                         // a slot past the end repeats the thread's previous one
            const int pix = j >> 3;
            const int r = pix / XC;
            x_rc[k] = (r << 8) | (pix - r * XC);                     // halo coordinates: image pixel (y0 - 1 + r, x0 - 1 + c)
        }
        (void)XSTEP;
        const __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)(a.G + m0 + 32 * gblk), 0, 0x7fffffff, 0x00020000);
        const int xshift = (a.W + 1) * xcs;                          // the X resource starts one row + one pixel BEFORE the tensor
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X[xd] + xch0 - xshift), 0, 0x7fffffff, 0x00020000);
        f32x4 rg[NG], rx[NX];
        float bsum[4] = {0.f, 0.f, 0.f, 0.f};
        // the scales as floats; odd pixel splits stage -G (the matrix core's accumulation rounds toward minus infinity: csrc/wgrad_x3.hip)
        const float sgs = __uint_as_float(((unsigned)(se_g + 127) << 23) | ((WX3_ALT_SIGN && (z & 1)) ? 0x80000000u : 0u));
        const float sxs = __uint_as_float((unsigned)(se_x + 127) << 23);
        // one tile's scalars, then per staging slot: request (global -> registers) and stage (registers -> both planes of an image)
        struct TileSc { int gso, xso, rlim, clim, y0, x0; };
        auto tile_sc = [&](int tile) {
            int q = tile;
            const int tx = q % tiles_x; q /= tiles_x;
            const int ty = q % tiles_y;
            const int b = q / tiles_y;
            TileSc t;
            t.x0 = tx * 32; t.y0 = ty * TH;
            t.gso = (((b * a.H + t.y0) * a.W) + t.x0) * a.Gcs * 4;
            t.xso = ((((b * a.H + t.y0 - 1) * a.W) + t.x0 - 1) * xcs + xshift) * 4;
            t.rlim = a.H - t.y0; t.clim = a.W - t.x0;
            return t;
        };
        auto load_g = [&](auto ktag, const TileSc& t) {
            constexpr int k = decltype(ktag)::value;
            const int gr = (GSTEP * k) >> 5, gc = gp0 + ((GSTEP * k) & 31);
            const int bad = (t.rlim - 1 - gr) | (t.clim - 1 - gc);                             // sign bit set <=> pixel outside the image
            rg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsg, bad < 0 ? OOB : g_base, t.gso + (gr * a.W + ((GSTEP * k) & 31)) * a.Gcs * 4, WHS_G_AUX));
        };
        auto load_x = [&](auto ktag, const TileSc& t) {
            constexpr int k = decltype(ktag)::value;
            const int xr = x_rc[k] >> 8, xc = x_rc[k] & 255;
            const int yy = t.y0 - 1 + xr, xx = t.x0 - 1 + xc;
            const int bad = yy | (a.H - 1 - yy) | xx | (a.W - 1 - xx);
            rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, bad < 0 ? OOB : (unsigned)((xr * a.W + xc) * xcs + q8 * 4) * 4u, t.xso, 0));
        };
        auto load_tile = [&](int tile) {
            const TileSc t = tile_sc(tile);
            static_for<0, NG>([&](auto kt) { load_g(kt, t); });
            static_for<0, NX>([&](auto kt) { load_x(kt, t); });
        };
        auto stage = [&](f32x4 v, float sc, char* ib, int dst, int pstride) {
            unsigned h0, l0, h1, l1;
            split_h2(v.x, v.y, sc, h0, l0);
            split_h2(v.z, v.w, sc, h1, l1);
            *reinterpret_cast<u32x2*>(ib + dst) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(ib + dst + pstride) = u32x2{l0, l1};
        };
        auto stage_g = [&](auto ktag, char* ib) {
            constexpr int k = decltype(ktag)::value;
            const f32x4 v = rg[k];
            bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;         // bias gradient: column sums of G (unscaled, unsigned)
            stage(v, sgs, ib, g_dst0 + GSTEP * k * 64, GPIX * 64);
        };
        auto stage_x = [&](auto ktag, char* ib) {
            constexpr int k = decltype(ktag)::value;
            stage(rx[k], sxs, ib, G_BYTES + (xblk * 2 * XPIX + (x_rc[k] >> 8) * XC + (x_rc[k] & 255)) * 64 + q8 * 8, XPIX * 64);
        };
        auto stage_tile = [&](int img) {
            char* ib = smem + img * IMG_BYTES;
            static_for<0, NG>([&](auto kt) { stage_g(kt, ib); });
            static_for<0, NX>([&](auto kt) { stage_x(kt, ib); });
        };
        // ROLLING refill (WHS_ROLL, round 6): slot k of the next tile is staged and the SAME registers immediately re-requested for the tile after it, slot by
        // slot -- every load is then in flight for a whole period (stage-everything-then-request-everything left them the barrier wait only: the
        // producers' period was load latency + staging + issue, 4 100 cycles where the consumers need 1 800-2 900: profiles/r6/wgrad_stamps.txt).
        // Buffer loads return in order, so slot k has landed when at most NG + NX - 1 later requests are outstanding (the compiler counts them:
        // s_waitcnt vmcnt(NG + NX - 1) in front of every slot; the scheduling barriers keep it from regrouping the requests).
        auto roll_tile = [&](int img, int tile_after) {
            char* ib = smem + img * IMG_BYTES;
            const TileSc t = tile_sc(tile_after);
            static_for<0, NG>([&](auto kt) { stage_g(kt, ib); load_g(kt, t); __builtin_amdgcn_sched_barrier(0); });
            static_for<0, NX>([&](auto kt) { stage_x(kt, ib); load_x(kt, t); __builtin_amdgcn_sched_barrier(0); });
        };
        // the first tile straight into image 0, the second into the registers
        load_tile(z);
        stage_tile(0);
        if (z + a.Z < ntile) load_tile(z + a.Z);
        int img = 0;
#ifdef WHS_STAMPS
        long long t_stage = 0, t_issue = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int ntl = 0;
#endif
        for (int tile = z; tile < ntile; tile += a.Z) {
            WHS_BARRIER();                                          // image img is complete; every consumer is done with the other one
            WHS_T(t_bar)
#ifdef WHS_STAMPS
            ++ntl;
#endif
            if (WHS_ROLL && tile + 2 * a.Z < ntile) {
                roll_tile(img ^ 1, tile + 2 * a.Z);
                WHS_T(t_stage)
            } else if (tile + a.Z < ntile) {
                stage_tile(img ^ 1);                                // the next tile (requested a whole tile ago)
                WHS_T(t_stage)
                if (tile + 2 * a.Z < ntile) load_tile(tile + 2 * a.Z);
                WHS_T(t_issue)
            }
            img ^= 1;
        }
#ifdef WHS_STAMPS
        if (lane == 0) {
            float* d = a.slab + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
            d[0] = (float)t_stage; d[1] = (float)t_issue; d[2] = (float)t_bar; d[4] = (float)(clock64() - tall); d[5] = (float)ntl;
        }
#endif
        // ---- (the consumers' pixel-split reduction: 2 barriers per tap of a row when WK > 1) then the bias gradient of this pixel split: add up
        // the threads that share (block, q8) through LDS (the images are dead)
        if (WK > 1) {
#pragma unroll
            for (int i = 0; i < 6 * MW; ++i) WHS_BARRIER();
        }
        WHS_BARRIER();
        constexpr int NSLOTS = Cfg::GT / 8;                           // threads per (block, quad)
        if (a.bias_slab && ni == 0) {                               // block-uniform
            float* bs = reinterpret_cast<float*>(smem);             // [MO blocks][8 quads][4][NSLOTS]
            const int slot = lg >> 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) bs[((gblk * 8 + q8) * 4 + c) * NSLOTS + slot] = bsum[c];
        }
        WHS_BARRIER();
        if (a.bias_slab && ni == 0 && pw < 1 && lane < 32 * MO) {    // one producer wave: 32 MO channels
            float* bs = reinterpret_cast<float*>(smem);
            const int b2 = lane >> 5, ch = lane & 31;
            float s = 0.f;
            for (int k = 0; k < NSLOTS; ++k) s += bs[((b2 * 8 + (ch >> 2)) * 4 + (ch & 3)) * NSLOTS + k];
            a.bias_slab[(int64_t)z * a.M + m0 + b2 * 32 + ch] = (WX3_ALT_SIGN && (z & 1)) ? -s : s;      // (the reduce kernel adds odd splits with a minus sign)
        }
        return;
    }

    // =============================================== CONSUMER ===============================================
    const int tr = wave % 3, rest = wave / 3;                        // filter row; (block, pixel split)
    const int wk = rest % WK, no = (rest / WK) % NO, mo = MW == 1 ? rest / (WK * NO) : 0;      // (MW = 2: the wave owns blocks mo = 0 and 1)
    const int l31 = lane & 31, half = lane >> 5;
    f32x16 acc[MW][3];
#pragma unroll
    for (int mb = 0; mb < MW; ++mb)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][t][r] = 0.f;
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
#ifdef WHS_STAMPS
    long long t_mfma = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int ntl = 0;
    const long long t_entry = tlast_;
    long long t_first = 0;
#endif
    for (int tile = z; tile < ntile; tile += a.Z) {
        WHS_BARRIER();
        WHS_T(t_bar)
#ifdef WHS_STAMPS
        if (!ntl) t_first = tlast_ - t_entry;                       // kernel entry -> the first tile is staged
        ++ntl;
#endif
        const char* gimg = smem + img * IMG_BYTES;
        const char* ximg = gimg + G_BYTES;
        // MW = 1: operand words double-buffered (the next step's are read while this step's MFMAs issue).  MW = 2: 96 accumulator registers
        // leave room for ONE set (128 registers per wave at 16 waves): the next step's words are requested right BEHIND this step's MFMAs --
        // the matrix core has captured its operands by then -- and the LDS latency is covered by the SIMD's other two consumer waves.
        constexpr int NBUF = MW == 1 ? 2 : 1;
        u32x4 av[NBUF][MW][2], bv[NBUF][2];
        auto gload = [&](int kl, u32x4 (&ax)[MW][2]) {               // kl: this consumer's kl-th k-step of the tile
            const int ks = wk * KSW + kl;
#pragma unroll
            for (int mb = 0; mb < MW; ++mb) {
                const char* gbase = gimg + (((mo + mb) * 2) * GPIX + (ks >> 1) * 32 + (ks & 1) * 16) * 64 + tr_lane;
#pragma unroll
                for (int p = 0; p < 2; ++p) ax[mb][p] = tr_read(gbase + p * GPIX * 64);
            }
        };
        auto xload = [&](int kl, int dx, u32x4 (&bx)[2]) {
            const int ks = wk * KSW + kl;
            const char* xbase = ximg + ((no * 2) * XPIX + ((ks >> 1) + tr) * XC + (ks & 1) * 16 + dx) * 64 + tr_lane;
#pragma unroll
            for (int p = 0; p < 2; ++p) bx[p] = tr_read(xbase + p * XPIX * 64);
        };
        gload(0, av[0]);
        xload(0, 0, bv[0]);
#pragma unroll
        for (int kl = 0; kl < KSW; ++kl) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int s = kl * 3 + dx;
                if constexpr (NBUF == 2) {
                    // the next step's X words (and, at a k-step's last tap, the next k-step's G words) one step ahead
                    if (s + 1 < KSW * 3) xload((s + 1) / 3, (s + 1) % 3, bv[(s + 1) & 1]);
                    if (dx == 2 && kl + 1 < KSW) gload(kl + 1, av[(kl + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const u32x4 (&ax)[MW][2] = av[NBUF == 2 ? (kl & 1) : 0];
                const u32x4 (&bx)[2] = bv[NBUF == 2 ? (s & 1) : 0];
                // smallest terms first: (hi, lo') (lo, hi') (hi, hi')
#define WHS_MFMA(MB, PA, PB) acc[MB][dx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ax[MB][PA]), __builtin_bit_cast(f16x8, bx[PB]), acc[MB][dx], 0, 0, 0)
#pragma unroll
                for (int mb = 0; mb < MW; ++mb) { WHS_MFMA(mb, 0, 1); WHS_MFMA(mb, 1, 0); WHS_MFMA(mb, 0, 0); }
#undef WHS_MFMA
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (NBUF == 1) {
                    if (s + 1 < KSW * 3) xload((s + 1) / 3, (s + 1) % 3, bv[0]);
                    if (dx == 2 && kl + 1 < KSW) gload(kl + 1, av[0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        img ^= 1;
#ifdef WHS_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 0" ::: "memory");
#endif
        WHS_T(t_mfma)
    }
#ifdef WHS_STAMPS
    const long long t_loop_end = clock64();
#endif
    // ---- the pixel splits of a (block, row) are added up through LDS (the images are dead; one accumulator block at a time), then the slab
    // [z][tap][m][n]: 32 x 32 x 16 accumulator layout: column l31, row (r & 3) + 8 (r >> 2) + 4 half
    float* red = reinterpret_cast<float*>(smem);
    const int dexp = -(se_g + se_x);                                 // undo the operand scales (exact: a power of two)
    float* const slab_z = a.slab + (int64_t)z * a.M * a.N * 9;      // this split's slab: 9 M N < 2^31 floats (the launcher checks), so the index inside it is 32-bit
#pragma unroll
    for (int mb = 0; mb < MW; ++mb)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        f32x16 v = acc[mb][dx];
        const int t = tr * 3 + dx;
        if constexpr (WK > 1) {
            // Through LDS in the accumulator layout, back out ROW-major: a lane of the storing wave takes 4 consecutive columns of rows (lane >> 3) + 8 i, adds the WK
            // partial sums in the order k = 0, 1, ... (as before: bit-identical) and stores 16 bytes -- 4 store instructions of 8 rows x 128 bytes per (block, tap) where the
            // accumulator layout needed 16 of 2 x 128 bytes (the epilogue was ~13 000 cycles per workgroup, most of it these stores: profiles/r6/wgrad_stamps.txt)
            WHS_BARRIER();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = v[r];
            WHS_BARRIER();
            if (wk == 0) {
                const int c4 = (lane & 7) * 4;
                const unsigned n_u = (unsigned)a.N;
                const unsigned base = (unsigned)((t * a.M + m0 + (mo + mb) * 32) * a.N + n0 + no * 32 + c4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (lane >> 3) + 8 * i;             // row (r & 3) + 8 (r >> 2) + 4 half of the 32 x 32 block
                    const int r = ((row >> 3) << 2) | (row & 3), hf = (row >> 2) & 1;
                    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < WK; ++k) sum += *reinterpret_cast<const f32x4*>(red + ((wave + 3 * k) * 16 + r) * 64 + hf * 32 + c4);      // (the split index steps the wave number by 3)
#pragma unroll
                    for (int c = 0; c < 4; ++c) sum[c] = __builtin_ldexpf(sum[c], dexp);
                    *reinterpret_cast<f32x4*>(slab_z + base + (unsigned)row * n_u) = sum;
                }
            }
        } else if (wk == 0) {
            // one base index per (block, tap); the 16 rows of the accumulator layout are compile-time multiples of N behind it (with a 64-bit index per
            // element the compiler hoisted 16 address pairs and spilled: 12 bytes of scratch in the 64 x 64 kernel)
            const unsigned base = (unsigned)((t * a.M + m0 + (mo + mb) * 32 + 4 * half) * a.N + n0 + no * 32 + l31);
            const unsigned n_u = (unsigned)a.N;
#pragma unroll
            for (int r = 0; r < 16; ++r) slab_z[base + (unsigned)((r & 3) + 8 * (r >> 2)) * n_u] = __builtin_ldexpf(v[r], dexp);
        }
    }
    WHS_BARRIER();                                                  // (the producers' bias reduction: two more barriers for every wave)
    WHS_BARRIER();
#ifdef WHS_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0f70);                             // (the slab stores have left: what follows overwrites a corner of the slab)
    if (lane == 0) {
        float* d = a.slab + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
        d[0] = (float)t_mfma; d[1] = (float)t_first; d[2] = (float)t_bar; d[3] = (float)(clock64() - t_loop_end); d[4] = (float)(t_loop_end - tall); d[5] = (float)ntl;
    }
#endif
}

template <int MO, int NO, int TH, int MW = 1>
int launch_whs(const Wh2sArgs& a, hipStream_t s) {
    using Cfg = WsCfg<MO, NO, TH, MW>;
    auto kern = wgrad_h2s_kernel<MO, NO, TH, MW>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int blocks = (a.M / (32 * MO)) * (a.N / (32 * NO)) * a.Z;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

}  // namespace

// pixel-tile height of the configuration for (M, N)
#ifndef WH2S_MW2
#define WH2S_MW2 1                     // 64-row output tiles: one consumer owns both 32-row blocks (see WsCfg)
#endif
#ifndef WH2S_TH22
#define WH2S_TH22 2                    // A/B knobs: the tile heights of the four configurations.  Measured (profiles/r5/ab_wgrad_tile_heights.txt): 3 / 4 / 6 rows for the
                                       // 64x64 / 64x32 / 32x32 tiles spill 11-18 registers in the producers and are 8-25 % SLOWER; 32x64 at 3 rows fits: -8 %
#endif
#ifndef WH2S_TH21
#define WH2S_TH21 3
#endif
#ifndef WH2S_TH12
#define WH2S_TH12 3
#endif
#ifndef WH2S_TH11
#define WH2S_TH11 4
#endif
int pnnp_wh2s_th(int M, int N) { return (M % 64 == 0) ? ((N % 64 == 0) ? (WH2S_MW2 ? 2 : WH2S_TH22) : WH2S_TH21) : ((N % 64 == 0) ? WH2S_TH12 : WH2S_TH11); }

int pnnp_wh2s_launch(const Wh2sArgs& a, hipStream_t s) {
    if ((int64_t)a.M * a.N * 9 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;      // the kernel indexes one slab with 32 bits
#if WH2S_MW2                          // (measured, profiles/r5/ab_wgrad_two_blocks_per_wave.txt: 64 x 64 tiles -3 ... -6 % per layer; 64 x 32 tiles +6 %: they keep one block per wave)
    if (a.M % 64 == 0 && a.N % 64 == 0) return launch_whs<2, 2, 2, 2>(a, s);
#endif
    if (a.M % 64 == 0) return a.N % 64 == 0 ? launch_whs<2, 2, WH2S_TH22>(a, s) : launch_whs<2, 1, WH2S_TH21>(a, s);
    return a.N % 64 == 0 ? launch_whs<1, 2, WH2S_TH12>(a, s) : launch_whs<1, 1, WH2S_TH11>(a, s);
}
