// Weight gradients of the POINTWISE / STRIDED layers on the fp16 matrix cores (round 5: csrc/wgrad_x3g.hip -- the exact bf16 three-way split --
// re-done on the fp16x2 scheme of csrc/h2.h): ConvTranspose2d(2, stride 2), Conv2d 3x3 stride 2 and Conv2d 1x1.
//
//   dW[m][n][t] = sum over pixels p of  U[p][m] * S[SM p + off(t)][n]          (geometries, tiles, staging layout, slabs: csrc/wgrad_x3g.hip)
//
// What changes: both operands are scaled by a power of two from their tensors' amax slots (U: amax_u; S: the larger of amax_s[0..1]) and split
// into TWO fp16 pieces on the way into LDS (images [32-channel block][piece 2][pixel][32 ch]); a (tap, 32 x 32 block) is THREE
// v_mfma_f32_32x32x16_f16 -- (hi, lo') (lo, hi') (hi, hi') -- instead of six; a staging slice is 5 steps instead of 11; the slab store
// multiplies by 2^-(se_u + se_s).  Odd pixel splits still stage -U (alternating-sign slabs), the bias sums use the unscaled values.
#include "h2.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

int pnnp_wx3_reduce_launch(const float* slab, float* dW, int64_t mn, int taps, int Z, int accumulate,
                           const float* bias_slab, float* dbias, int nb, hipStream_t st);

namespace {

struct WhgArgs {
    const float* U; int Ucs;            // [B][UH][UW][Ucs], channels [0, M) used
    const float* S[2]; int Scs[2];      // [B][SH][SW][Scs]; n < n_split -> S[0][n], else S[1][n - n_split]
    int n_split;
    int B, UH, UW, SH, SW, M, N;
    float* slab;                        // [Z][TAPS][M][N]
    float* bias_u;                      // [Z][M] or null: sum of U over the pixels (bias gradient of a Conv2d: U = g)
    float* bias_s;                      // [Z][N] or null: sum of S over the pixels (bias gradient of a ConvTranspose2d: S = g)
    int Z;
    const unsigned* amax_u; const unsigned* amax_s[2];   // amax slots of U and of the S segment(s) ([1] null without a second one)
};

#ifndef WHG_ROLL
#define WHG_ROLL 1                     // rolling refill of the staging registers (see stage_piece; 0 = round 5's order)
#endif
constexpr int NTHR = 512, NWAVE = 8;
enum { GEO_PW = 0, GEO_CT = 1, GEO_S2 = 2 };

template <int GEO, int MB, int NB, int WM, int WN, int WK, int TW, int TR>
struct WhgCfg {
    static constexpr int TAPS = GEO == GEO_PW ? 1 : (GEO == GEO_CT ? 4 : 9);
    static constexpr int SM = GEO == GEO_PW ? 1 : 2, P = GEO == GEO_S2 ? 1 : 0;
    static constexpr int NPL = SM, PLW = TW + P, SROWS = SM * TR + P;          // parity planes, entries per plane, S rows of a tile
    static constexpr int UPIX = TR * TW, SPIX = SROWS * NPL * PLW;
    static constexpr int MBT = WM * MB, NBT = WN * NB;                          // 32-channel blocks of the workgroup's output tile
    static constexpr int U_BYTES = MBT * 2 * UPIX * 64, S_BYTES = NBT * 2 * SPIX * 64, IMG_BYTES = U_BYTES + S_BYTES;
    static constexpr int RED_BYTES = NWAVE * 16 * 64 * 4;                        // the WK-wave reduction scratch aliases the (dead) images
    static constexpr int LDS_BYTES = 2 * IMG_BYTES > RED_BYTES ? 2 * IMG_BYTES : RED_BYTES;
    static constexpr int NU = MBT * UPIX * 8 / NTHR, NS = (NBT * SPIX * 8 + NTHR - 1) / NTHR;   // staging slots (float4) per thread
    static constexpr int NGRP = MB * NB * TAPS, NSL = NU + NS;                   // groups of three MFMAs / staging slices per wave and tile
    static constexpr int NSTEP = 5;                                               // steps of a staging slice (stage_piece)
    static constexpr int UNITS = NSL * NSTEP, GAPS = NGRP * 3, UPG = (UNITS + GAPS - 1) / GAPS;   // staging units (slice, step) per MFMA gap
    static_assert(WM * WN * WK == NWAVE && TR * TW / 16 == WK, "one 16-pixel k-step per wave and tile");
    static_assert((MBT * UPIX * 8) % NTHR == 0, "U slots must divide evenly (the bias sums count every pixel once)");
    static_assert(GEO != GEO_CT || (NBT * SPIX * 8) % NTHR == 0, "ConvTranspose2d: S slots must divide evenly (bias sums)");
    static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES >= RED_BYTES, "LDS budget (images; reduction scratch aliases them)");
    static_assert(MB * NB * TAPS * 16 <= 144, "accumulator registers");
};

// hi = f16(a s) of two values packed (low half = a0); lo = f16(a s - hi): csrc/conv_h2s.hip split_h2, as two steps
__device__ __forceinline__ unsigned h2_hi(float a0, float a1, float s) {
    unsigned h;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    return h;
}
__device__ __forceinline__ unsigned h2_lo(float a0, float a1, float s, unsigned h) {
    unsigned l;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(s), "v"(h));
    return l;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int GEO, int MB, int NB, int WM, int WN, int WK, int TW, int TR>
__global__ void __launch_bounds__(NTHR)
wgrad_h2g_kernel(const WhgArgs a) {
    using Cfg = WhgCfg<GEO, MB, NB, WM, WN, WK, TW, TR>;
    constexpr int TAPS = Cfg::TAPS, SM = Cfg::SM, P = Cfg::P, NPL = Cfg::NPL, PLW = Cfg::PLW, UPIX = Cfg::UPIX, SPIX = Cfg::SPIX;
    constexpr int NU = Cfg::NU, NS = Cfg::NS, MBT = Cfg::MBT, NBT = Cfg::NBT;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    // ---- scales (csrc/h2.h)
    unsigned as_ = a.amax_s[0] ? a.amax_s[0][0] : 0u;
    if (a.amax_s[1]) { const unsigned a2 = a.amax_s[1][0]; as_ = a2 > as_ ? a2 : as_; }
    const int se_u = __builtin_amdgcn_readfirstlane(pnnp_h2_scale_exp(a.amax_u ? a.amax_u[0] : 0u));
    const int se_s = __builtin_amdgcn_readfirstlane(pnnp_h2_scale_exp(as_));
    const float scu = __uint_as_float((unsigned)(se_u + 127) << 23), scs = __uint_as_float((unsigned)(se_s + 127) << 23);
    const int wk = wave % WK, wno = (wave / WK) % WN, wmo = wave / (WK * WN);

    const int n_tiles = a.N / (32 * NBT);
    int id = blockIdx.x;
    const int z = id % a.Z; id /= a.Z;
    const int ni = id % n_tiles, mi = id / n_tiles;
    const int m0 = mi * 32 * MBT, n0 = ni * 32 * NBT;
    const int tiles_x = (a.UW + TW - 1) / TW, tiles_y = (a.UH + TR - 1) / TR;
    const int ntile = tiles_x * tiles_y * a.B;
    if (z >= ntile) return;                                         // (Z <= ntile: never)

    // ---- staging pattern: slot j = tid + 512 k over (32-channel block, pixel, channel quad); S slots past the end repeat the previous one
    const int q8 = lane & 7;
    int u_r[NU], u_c[NU]; unsigned u_off[NU]; int u_dst[NU]; bool u_blk[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int j = tid + NTHR * k;
        const int blk = j / (UPIX * 8), pix = (j % (UPIX * 8)) >> 3;
        u_r[k] = pix / TW; u_c[k] = pix % TW;
        u_off[k] = (unsigned)((u_r[k] * a.UW + u_c[k]) * a.Ucs + blk * 32 + q8 * 4) * 4u;
        u_dst[k] = (blk * 2 * UPIX + pix) * 64 + q8 * 8;             // + piece * UPIX * 64
        u_blk[k] = m0 + blk * 32 < a.M;
    }
    int s_r[NS], s_x[NS]; unsigned s_off[NS]; int s_dst[NS]; bool s_blk[NS]; int s_seg[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        int j = tid + NTHR * k;
        if (j >= NBT * SPIX * 8) j -= NTHR;
        const int blk = j / (SPIX * 8), pix = (j % (SPIX * 8)) >> 3;
        const int jr = pix / (NPL * PLW), e = (pix / PLW) % NPL, idx = pix % PLW;
        s_r[k] = jr; s_x[k] = NPL == 2 ? 2 * idx + e : idx;          // tile-relative S row / column (image origin = SM * (y0, x0) - P)
        const int nb = n0 + blk * 32;
        s_seg[k] = nb >= a.n_split ? 1 : 0;                         // (wave-uniform where there are two segments: 1x1 only, SPIX % 8 == 0)
        const int ch = nb - (s_seg[k] ? a.n_split : 0);
        s_off[k] = (unsigned)((jr * a.SW + s_x[k]) * a.Scs[s_seg[k]] + ch + q8 * 4) * 4u;
        s_dst[k] = Cfg::U_BYTES + (blk * 2 * SPIX + pix) * 64 + q8 * 8;
        s_blk[k] = nb < a.N;
    }
    const __amdgpu_buffer_rsrc_t rsu = __builtin_amdgcn_make_buffer_rsrc((void*)(a.U + m0), 0, 0x7fffffff, 0x00020000);
    // (the S resources start one row + one pixel BEFORE the tensor so that the scalar offset of a halo tile is never negative)
    const int sshift0 = (a.SW + 1) * a.Scs[0], sshift1 = (a.SW + 1) * a.Scs[1];
    const __amdgpu_buffer_rsrc_t rss0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S[0] - sshift0), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S[1] - sshift1), 0, 0x7fffffff, 0x00020000);
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };

    f32x4 ru[NU], rs[NS];
    float bsu[NU][4], bss[GEO == GEO_CT ? NS : 1][4];
#pragma unroll
    for (int k = 0; k < NU; ++k) bsu[k][0] = bsu[k][1] = bsu[k][2] = bsu[k][3] = 0.f;
#pragma unroll
    for (int k = 0; k < (GEO == GEO_CT ? NS : 1); ++k) bss[k][0] = bss[k][1] = bss[k][2] = bss[k][3] = 0.f;
    const unsigned sflip = (z & 1) ? 0x80000000u : 0u;              // odd pixel splits accumulate -U * S (WX3_ALT_SIGN of csrc/wgrad_x3.hip)
    auto usign = [&](f32x4 v) {
        return f32x4{__uint_as_float(__float_as_uint(v.x) ^ sflip), __uint_as_float(__float_as_uint(v.y) ^ sflip),
                     __uint_as_float(__float_as_uint(v.z) ^ sflip), __uint_as_float(__float_as_uint(v.w) ^ sflip)};
    };
    // one tile's scalars, then one request per staging slot (global -> registers)
    struct TileSc { int uso, sso0, sso1, rlim, clim, sy0, sx0, dead; };      // dead: -1 = no such tile (every request out of range: zeros, no traffic), else 0
    auto tile_sc = [&](int tile, bool exists) {
        int q = tile;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y;
        const int b = q / tiles_y;
        const int x0 = tx * TW, y0 = ty * TR;
        TileSc t;
        t.uso = ((b * a.UH + y0) * a.UW + x0) * a.Ucs * 4;
        t.sy0 = SM * y0 - P; t.sx0 = SM * x0 - P;
        t.sso0 = (((b * a.SH + t.sy0) * a.SW + t.sx0) * a.Scs[0] + sshift0) * 4; t.sso1 = (((b * a.SH + t.sy0) * a.SW + t.sx0) * a.Scs[1] + sshift1) * 4;
        t.rlim = a.UH - y0; t.clim = a.UW - x0;
        t.dead = exists ? 0 : -1;
        if (!exists) { t.uso = 0; t.sso0 = 0; t.sso1 = 0; }
        return t;
    };
    auto load_u = [&](int k, const TileSc& t) {
        const int bad = (t.rlim - 1 - u_r[k]) | (t.clim - 1 - u_c[k]) | (u_blk[k] ? 0 : -1) | t.dead;      // sign bit set <=> outside
        ru[k] = bload(rsu, bad < 0 ? OOB : u_off[k], t.uso);
    };
    auto load_s = [&](int k, const TileSc& t) {
        const int yy = t.sy0 + s_r[k], xx = t.sx0 + s_x[k];
        const int bad = yy | (a.SH - 1 - yy) | xx | (a.SW - 1 - xx) | (s_blk[k] ? 0 : -1) | t.dead;
        const unsigned vo = bad < 0 ? OOB : s_off[k];
        if (GEO == GEO_PW && __builtin_amdgcn_readfirstlane(s_seg[k])) rs[k] = bload(rss1, vo, t.sso1);
        else rs[k] = bload(rss0, vo, t.sso0);
    };
    auto load_tile = [&](int tile) {
        const TileSc t = tile_sc(tile, true);
#pragma unroll
        for (int k = 0; k < NU; ++k) load_u(k, t);
#pragma unroll
        for (int k = 0; k < NS; ++k) load_s(k, t);
    };
    // one staging slice as a whole (prologue) ...
    float bmul = 1.f;                                             // 0 while there is no next tile: the pieces then run on stale registers, branch-free
    auto stage_slice = [&](int s, int img) {
        char* ib = smem + img * Cfg::IMG_BYTES;
        const bool isu = s < NU;
        const f32x4 v = isu ? usign(ru[isu ? s : 0]) : rs[isu ? 0 : s - NU];
        const int dst = isu ? u_dst[isu ? s : 0] : s_dst[isu ? 0 : s - NU];
        const int pstride = (isu ? UPIX : SPIX) * 64;
        const float sc = isu ? scu : scs;
        const unsigned h0 = h2_hi(v.x, v.y, sc), h1 = h2_hi(v.z, v.w, sc);
        const unsigned l0 = h2_lo(v.x, v.y, sc, h0), l1 = h2_lo(v.z, v.w, sc, h1);
        *reinterpret_cast<u32x2*>(ib + dst) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(ib + dst + pstride) = u32x2{l0, l1};
        if (isu) { bsu[isu ? s : 0][0] += v.x; bsu[isu ? s : 0][1] += v.y; bsu[isu ? s : 0][2] += v.z; bsu[isu ? s : 0][3] += v.w; }
        else if (GEO == GEO_CT) {
            const f32x4 w = usign(v); const int k = isu ? 0 : s - NU;
            bss[k][0] += w.x; bss[k][1] += w.y; bss[k][2] += w.z; bss[k][3] += w.w;
        }
    };
    // ... and as dependent pieces (step 0: hi; 1: bias sums; 2: lo) and the stores (step 3, 4), dealt over the MFMA gaps
    f32x4 pv; unsigned ph[2], pl[2];
    // ROLLING refill (WHG_ROLL, round 6; csrc/wgrad_h2s.hip roll_tile): a slot's registers are re-requested for the tile after next as soon as step 2 has
    // used them last, so that every load is in flight for a whole tile (requesting the whole tile behind the last MFMA left them the barrier wait only)
    auto stage_piece = [&](int sl, int step, int img, const TileSc& t2) {
        const bool isu = sl < NU;
        const int ku = isu ? sl : 0, ks = isu ? 0 : sl - NU;
        const float sc = isu ? scu : scs;
        switch (step) {
        case 0:
            pv = isu ? usign(ru[ku]) : rs[ks]; ph[0] = h2_hi(pv.x, pv.y, sc); ph[1] = h2_hi(pv.z, pv.w, sc);
            break;
        case 1:
            if (isu) { bsu[ku][0] = fmaf(pv.x, bmul, bsu[ku][0]); bsu[ku][1] = fmaf(pv.y, bmul, bsu[ku][1]); bsu[ku][2] = fmaf(pv.z, bmul, bsu[ku][2]); bsu[ku][3] = fmaf(pv.w, bmul, bsu[ku][3]); }
            else if (GEO == GEO_CT) {
                const float sm = (z & 1) ? -bmul : bmul;
                bss[GEO == GEO_CT ? ks : 0][0] = fmaf(pv.x, sm, bss[GEO == GEO_CT ? ks : 0][0]); bss[GEO == GEO_CT ? ks : 0][1] = fmaf(pv.y, sm, bss[GEO == GEO_CT ? ks : 0][1]);
                bss[GEO == GEO_CT ? ks : 0][2] = fmaf(pv.z, sm, bss[GEO == GEO_CT ? ks : 0][2]); bss[GEO == GEO_CT ? ks : 0][3] = fmaf(pv.w, sm, bss[GEO == GEO_CT ? ks : 0][3]);
            }
            break;
        case 2:
            pl[0] = h2_lo(pv.x, pv.y, sc, ph[0]); pl[1] = h2_lo(pv.z, pv.w, sc, ph[1]);
            // (behind the LAST use of the slot's value: the request can then land in the same registers -- issued at step 0 the compiler had to give it
            // others and copy them at the end of the loop, which waits for every load)
            if constexpr (WHG_ROLL) { if (isu) load_u(ku, t2); else load_s(ks, t2); }
            break;
        default: {
            char* ib = smem + img * Cfg::IMG_BYTES;
            const int dst = isu ? u_dst[ku] : s_dst[ks];
            const int pstride = (isu ? UPIX : SPIX) * 64;
            if (step == 3) *reinterpret_cast<u32x2*>(ib + dst) = u32x2{ph[0], ph[1]};
            if (step == 4) *reinterpret_cast<u32x2*>(ib + dst + pstride) = u32x2{pl[0], pl[1]};
        } break;
        }
    };

    f32x16 acc[MB][NB][TAPS];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int t = 0; t < TAPS; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][t][r] = 0.f;

    // transposed-read lane geometry (csrc/wgrad_x3.hip): 16-lane group g = lane >> 4 reads channels 16 (g & 1) .., pixels 8 (g >> 1) ..;
    // inside a group lane 4 q + p supplies the address of pixel row q, channel chunk 4 p
    const int tr_lane = ((8 * (lane >> 5) + ((lane & 15) >> 2)) * 64) + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto tr_read = [&](const char* base) {                           // 8 pixels x 1 channel per lane: two transposed reads
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * 64));
        u32x4 r;
        const u32x2 a0 = __builtin_bit_cast(u32x2, lo), a1 = __builtin_bit_cast(u32x2, hi);
        r.x = a0.x; r.y = a0.y; r.z = a1.x; r.w = a1.y;
        return r;
    };
    // this wave's k-step: U pixels (row kr, columns kc0 .. kc0 + 15) of the tile
    const int kr = wk / (TW / 16), kc0 = (wk % (TW / 16)) * 16;

    // ---- prologue: the first tile goes straight into image 0, the second one into registers
    load_tile(z);
    __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int s = 0; s < Cfg::NSL; ++s) stage_slice(s, 0);
    if (z + a.Z < ntile) load_tile(z + a.Z);
    int img = 0;
    for (int tile = z; tile < ntile; tile += a.Z) {
        // image `img` is complete; every wave is done with the other one.  (Not __syncthreads(): its fence waits for vmcnt(0) -- the loads in flight for the
        // tile after next -- where only the LDS writes have to be visible.)
        if constexpr (WHG_ROLL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
        const char* uimg = smem + img * Cfg::IMG_BYTES;
        const char* simg = uimg + Cfg::U_BYTES;
        const bool have_next = tile + a.Z < ntile, have_next2 = tile + 2 * a.Z < ntile;
        bmul = have_next ? 1.f : 0.f;
        auto u_addr = [&](int mb) { return uimg + (((wmo * MB + mb) * 2) * UPIX + kr * TW + kc0) * 64 + tr_lane; };
        // tap t of N block nb: S row SM kr + oy, parity plane ox & (NPL - 1), first entry kc0 + (ox >> (NPL - 1))
        auto s_addr = [&](int nb, int t) {
            const int oy = GEO == GEO_CT ? t >> 1 : (GEO == GEO_S2 ? t / 3 : 0), ox = GEO == GEO_CT ? t & 1 : (GEO == GEO_S2 ? t % 3 : 0);
            const int jr = SM * kr + oy, e = ox & (NPL - 1), idx = kc0 + (NPL == 2 ? ox >> 1 : ox);
            return simg + (((wno * NB + nb) * 2) * SPIX + (jr * NPL + e) * PLW + idx) * 64 + tr_lane;
        };
        u32x4 av[MB][2], bv[2][2];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int p = 0; p < 2; ++p) av[mb][p] = tr_read(u_addr(mb) + p * UPIX * 64);
#pragma unroll
        for (int p = 0; p < 2; ++p) bv[0][p] = tr_read(s_addr(0, 0) + p * SPIX * 64);
        if constexpr (!WHG_ROLL) __builtin_amdgcn_s_waitcnt(0x0f70);  // (rolling: the compiler's own vmcnt in front of every slot's step 0)
        const TileSc t2 = tile_sc(have_next2 ? tile + 2 * a.Z : tile, WHG_ROLL && have_next2);
        __builtin_amdgcn_sched_barrier(0);
        // groups: B-step bs = (N block, tap), M block innermost; the next B-step's operand is read in the first two gaps of a B-step's
        // first group; the staging units (slice, step) are dealt evenly over all gaps
        static_for<0, Cfg::GAPS>([&](auto GI) {
            constexpr int gi = decltype(GI)::value, grp = gi / 3, G = gi % 3, bs = grp / MB, mb = grp % MB, nb = bs / TAPS, t = bs % TAPS;
            constexpr int cur = bs & 1;
            constexpr int PA = G == 1 ? 1 : 0, PB = G == 0 ? 1 : 0;            // smallest terms first: (hi, lo') (lo, hi') (hi, hi')
            acc[mb][nb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av[mb][PA]), __builtin_bit_cast(f16x8, bv[cur][PB]), acc[mb][nb][t], 0, 0, 0);
            if constexpr (mb == 0 && G < 2 && bs + 1 < NB * TAPS)
                bv[cur ^ 1][G] = tr_read(s_addr((bs + 1) / TAPS, (bs + 1) % TAPS) + G * SPIX * 64);
            constexpr int n0u = (gi * Cfg::UNITS + Cfg::GAPS - 1) / Cfg::GAPS, n1u = ((gi + 1) * Cfg::UNITS + Cfg::GAPS - 1) / Cfg::GAPS;
            static_for<n0u, (n1u < Cfg::UNITS ? n1u : Cfg::UNITS)>([&](auto UI) { constexpr int u = decltype(UI)::value; stage_piece(u / Cfg::NSTEP, u % Cfg::NSTEP, img ^ 1, t2); });
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (!WHG_ROLL) { if (have_next2) load_tile(tile + 2 * a.Z); }      // registers are free again: the tile after next
        img ^= 1;
    }

    // ---- reduce the WK pixel-split waves through LDS (the images are dead now), then write the slab [z][tap][m][n]
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    const int64_t slab_base = (int64_t)z * a.M * a.N * TAPS;
    const int dexp = -(se_u + se_s);                                 // undo the operand scales (exact: a power of two)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                f32x16 v = acc[mb][nb][t];
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
                const int mblk = m0 + (wmo * MB + mb) * 32, n = n0 + (wno * NB + nb) * 32 + l31;
                if (wk == 0 && mblk < a.M && n < a.N) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mblk + (r & 3) + 8 * (r >> 2) + 4 * half;
                        a.slab[slab_base + ((int64_t)t * a.M + m) * a.N + n] = __builtin_ldexpf(v[r], dexp);
                    }
                }
            }
    // ---- bias sums: a thread summed 4 channels (block, quad q8) over the pixels of its slots; slot j's partial goes to red[j][4] and the
    // first 32 * blocks threads add up the pixels of their channel in a fixed order
    if ((a.bias_u && ni == 0) || (GEO == GEO_CT && a.bias_s && mi == 0)) {   // block-uniform
        __syncthreads();
        if (a.bias_u && ni == 0) {
#pragma unroll
            for (int k = 0; k < NU; ++k) *reinterpret_cast<f32x4*>(red + (tid + NTHR * k) * 4) = f32x4{bsu[k][0], bsu[k][1], bsu[k][2], bsu[k][3]};
            __syncthreads();
            if (tid < MBT * 32 && m0 + tid < a.M) {
                const int blk = tid >> 5, ch = tid & 31;
                float s = 0.f;
                for (int px = 0; px < UPIX; ++px) s += red[(((blk * UPIX + px) * 8) + (ch >> 2)) * 4 + (ch & 3)];
                a.bias_u[(int64_t)z * a.M + m0 + tid] = s;
            }
            __syncthreads();
        }
        if constexpr (GEO == GEO_CT) {
            if (a.bias_s && mi == 0) {
#pragma unroll
                for (int k = 0; k < NS; ++k) *reinterpret_cast<f32x4*>(red + (tid + NTHR * k) * 4) = f32x4{bss[k][0], bss[k][1], bss[k][2], bss[k][3]};
                __syncthreads();
                if (tid < NBT * 32 && n0 + tid < a.N) {
                    const int blk = tid >> 5, ch = tid & 31;
                    float s = 0.f;
                    for (int px = 0; px < SPIX; ++px) s += red[(((blk * SPIX + px) * 8) + (ch >> 2)) * 4 + (ch & 3)];
                    a.bias_s[(int64_t)z * a.N + n0 + tid] = s;
                }
            }
        }
    }
}

template <int GEO, int MB, int NB, int WM, int WN, int WK, int TW, int TR>
struct WhgLaunch {
    using Cfg = WhgCfg<GEO, MB, NB, WM, WN, WK, TW, TR>;
    static constexpr int BM = 32 * Cfg::MBT, BN = 32 * Cfg::NBT;
    static int out_tiles(int M, int N) { return ((M + BM - 1) / BM) * (N / BN); }
    static int pixel_tiles(int B, int UH, int UW) { return ((UW + TW - 1) / TW) * ((UH + TR - 1) / TR) * B; }
    static int launch(const WhgArgs& a, hipStream_t s) {
        auto kern = wgrad_h2g_kernel<GEO, MB, NB, WM, WN, WK, TW, TR>;
        static PnnpPerDevice lds_once;
        if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(out_tiles(a.M, a.N) * a.Z), dim3(NTHR), Cfg::LDS_BYTES, s, a);
        return pnnp_launch_status();
    }
};

// Configurations (output tile M x N of the workgroup, U pixels per tile):
//   ConvTranspose2d  A 256 x 64 (16 px): two M blocks per wave against one staged S block   B 128 x 64 (16 px)   C 64 x 32 (64 px)
//   3x3 stride 2     128 x 32 (32 px; M = Cout in multiples of 128: pool2 .. pool4 of the ResUnet)
//   1x1              128 x 128 (32 px; two M and two N blocks per wave)   64 x 128 (32 px)
using CtA = WhgLaunch<GEO_CT, 2, 1, 4, 2, 1, 16, 1>;
using CtB = WhgLaunch<GEO_CT, 1, 1, 4, 2, 1, 16, 1>;
using CtC = WhgLaunch<GEO_CT, 1, 1, 2, 1, 4, 32, 2>;
using S2A = WhgLaunch<GEO_S2, 1, 1, 4, 1, 2, 32, 1>;
using S2B = WhgLaunch<GEO_S2, 1, 1, 2, 1, 4, 32, 2>;          // 64 x 32 (64 px; M = Cout = 64: pool1 of the ResUnet -- no bf16x3 twin, it ran on fp32 MFMA)
using PwA = WhgLaunch<GEO_PW, 2, 2, 2, 2, 2, 32, 1>;
using PwB = WhgLaunch<GEO_PW, 1, 2, 2, 2, 2, 32, 1>;
using PwC = WhgLaunch<GEO_PW, 1, 1, 1, 2, 4, 32, 2>;          // 32 x 64 (64 px; M = Cout = 32: sc9 of the ResUnet -- round 6: it ran on fp32 MFMA)

// which configuration a (geometry, M, N) runs on: 0 = not supported
int wxg_config(int geo, int M, int N) {
    if (M <= 0 || N <= 0 || (M & 31) || (N & 31)) return 0;
    if (geo == GEO_CT) return (M % 256 == 0 && N % 64 == 0) ? 1 : ((M % 128 == 0 && N % 64 == 0) ? 2 : ((M % 64 == 0) ? 3 : 0));
    if (geo == GEO_S2) return (M % 128 == 0) ? 4 : ((M % 64 == 0) ? 7 : 0);
    if (geo == GEO_PW) return (N % 128 == 0 && M % 64 == 0) ? ((M % 128 == 0) ? 5 : 6) : ((N % 64 == 0) ? 8 : 0);
    return 0;
}
template <class F> auto wxg_dispatch(int cfg, F&& f) {
    switch (cfg) {
        case 1: return f(CtA{});
        case 2: return f(CtB{});
        case 3: return f(CtC{});
        case 4: return f(S2A{});
        case 5: return f(PwA{});
        case 7: return f(S2B{});
        case 8: return f(PwC{});
        default: return f(PwB{});
    }
}
int wxg_splits(int cfg, int B, int UH, int UW, int M, int N) {
    return wxg_dispatch(cfg, [&](auto L) {
        using LT = decltype(L);
        const int tiles = LT::pixel_tiles(B, UH, UW), ot = LT::out_tiles(M, N);
        int cus = pnnp_device_cus();
        if (cus <= 0) cus = 256;
        int z = (cus + ot - 1) / ot;                                 // one 8-wave workgroup per CU
        if (z > tiles) z = tiles;
        return z < 1 ? 1 : z;
    });
}

int wxg_run(int geo, const float* U, int Ucs, int M, const unsigned* amax_u, const float* S0, int S0cs, int N0, const unsigned* amax_s0,
            const float* S1, int S1cs, int N1, const unsigned* amax_s1,
            int B, int UH, int UW, int SH, int SW, float* dW, float* dbias, bool bias_from_s, int accumulate,
            float* workspace, int64_t workspace_floats, hipStream_t st) {
    if (!amax_u || !amax_s0 || (S1 && !amax_s1)) return PNNP_E_INVALID;
    const int N = N0 + (S1 ? N1 : 0);
    const int cfg = wxg_config(geo, M, N);
    if (!cfg) return PNNP_E_UNSUPPORTED;
    if (S1 && (N0 & 31)) return PNNP_E_UNSUPPORTED;
    if ((Ucs & 3) || (S0cs & 3) || (S1 && (S1cs & 3)) || Ucs < M || S0cs < N0 || (S1 && S1cs < N1)) return PNNP_E_INVALID;
    if ((((uintptr_t)U) | ((uintptr_t)S0) | ((uintptr_t)S1)) & 15) return PNNP_E_INVALID;
    // 32-bit byte offsets into the whole tensors (bit 31 marks "outside")
    if (((int64_t)B * UH + 2) * UW * Ucs * 4 >= (1ll << 31) || ((int64_t)B * SH + 2) * SW * (S0cs > S1cs ? S0cs : S1cs) * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    const int taps = geo == GEO_PW ? 1 : (geo == GEO_CT ? 4 : 9);
    WhgArgs a{};
    a.U = U; a.Ucs = Ucs;
    a.S[0] = S0; a.Scs[0] = S0cs; a.S[1] = S1 ? S1 : S0; a.Scs[1] = S1 ? S1cs : S0cs; a.n_split = S1 ? N0 : (1 << 30);
    a.B = B; a.UH = UH; a.UW = UW; a.SH = SH; a.SW = SW; a.M = M; a.N = N;
    a.Z = wxg_splits(cfg, B, UH, UW, M, N);
    const int nbias = bias_from_s ? N : M;
    if (workspace_floats < (int64_t)a.Z * ((int64_t)taps * M * N + nbias)) return PNNP_E_WORKSPACE;
    a.slab = workspace;
    float* bslab = dbias ? workspace + (int64_t)a.Z * taps * M * N : nullptr;
    a.bias_u = bias_from_s ? nullptr : bslab;
    a.bias_s = bias_from_s ? bslab : nullptr;
    a.amax_u = amax_u; a.amax_s[0] = amax_s0; a.amax_s[1] = S1 ? amax_s1 : nullptr;
    const int rc = wxg_dispatch(cfg, [&](auto L) { return decltype(L)::launch(a, st); });
    if (rc != PNNP_OK) return rc;
    return pnnp_wx3_reduce_launch(a.slab, dW, (int64_t)M * N, taps, a.Z, accumulate, bslab, dbias, nbias, st);
}

}  // namespace

extern "C" {

// Contracts of the _x3_ entries of csrc/wgrad_x3g.hip, plus the amax slots of the two tensors that are split on the fly (csrc/h2.h).  Shapes and
// workspace: pnnp_h2g_wgrad_supported / pnnp_h2g_wgrad_workspace_floats -- everything pnnp_x3g_wgrad_* takes and, in addition, the stride-2
// layer with Cout % 64 == 0 (a 64 x 32 tile: two pieces per operand leave the LDS room).
/* kind: 0 = Conv2d 1x1 (M = Cout, N = Cin), 1 = ConvTranspose2d 2x2 s2 (M = Cin, N = Cout), 2 = Conv2d 3x3 s2 (M = Cout, N = Cin) */
int pnnp_h2g_wgrad_supported(int kind, int M, int N) { return wxg_config(kind, M, N) ? 1 : 0; }
int64_t pnnp_h2g_wgrad_workspace_floats(int kind, int B, int UH, int UW, int M, int N) {
    const int cfg = wxg_config(kind, M, N);
    if (!cfg) return 0;
    const int taps = kind == GEO_PW ? 1 : (kind == GEO_CT ? 4 : 9);
    return (int64_t)wxg_splits(cfg, B, UH, UW, M, N) * ((int64_t)taps * M * N + (M > N ? M : N));
}
int pnnp_convt2x2_h2_bwd_weight_f32(const float* x, int Cin, const unsigned* amax_x, const float* g, int Cout, const unsigned* amax_g, float* dW, float* dbias,
                                    int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    if (!x || !g || !dW || !workspace || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    return wxg_run(GEO_CT, x, Cin, Cin, amax_x, g, Cout, Cout, amax_g, nullptr, 0, 0, nullptr, B, H, W, 2 * H, 2 * W, dW, dbias, true, accumulate,
                   workspace, workspace_floats, as_stream(stream));
}
int pnnp_conv3x3s2_h2_bwd_weight_f32(const float* g, int Cout, const unsigned* amax_g, const float* x, int Cin, const unsigned* amax_x, float* dW, float* dbias,
                                     int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    if (!g || !x || !dW || !workspace || B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    return wxg_run(GEO_S2, g, Cout, Cout, amax_g, x, Cin, Cin, amax_x, nullptr, 0, 0, nullptr, B, H / 2, W / 2, H, W, dW, dbias, false, accumulate,
                   workspace, workspace_floats, as_stream(stream));
}
int pnnp_conv1x1_h2_bwd_weight_f32(const float* g, int g_cs, int Cout, const unsigned* amax_g, const float* x1, int x1_cs, int C1, const unsigned* amax_x1,
                                   const float* x2, int x2_cs, int C2, const unsigned* amax_x2, float* dW, float* dbias,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream) {
    if (!g || !x1 || !dW || !workspace || B <= 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    return wxg_run(GEO_PW, g, g_cs, Cout, amax_g, x1, x1_cs, C1, amax_x1, x2, x2_cs, C2, amax_x2, B, H, W, H, W, dW, dbias, false, accumulate,
                   workspace, workspace_floats, as_stream(stream));
}

}  // extern "C"
