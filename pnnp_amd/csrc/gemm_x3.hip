// Pointwise convolutions as a float32 GEMM on the bf16 matrix cores (the 3-way split of csrc/conv_x3.hip): everything the
// network has with ONE tap per K segment --
//   ConvTranspose2d(k2, s2) forward   (archs/Unet.py:35-47):   N = 4 x Cout sub-pixel columns, output scattered with stride 2
//   ConvTranspose2d backward-data:                              K = 4 segments reading g at (2y+a, 2x+c)
//   Conv2d 1x1 (ResidualBlock shortcuts, archs/modules.py:176-197) forward / backward-data, with concat K segments
//   Conv2d 3x3 stride 2 (archs/modules.py:130-138) forward as 9 strided taps, backward-data per input-pixel parity class
// -- all through the IgemmArgs description of csrc/igemm.h (K segments with pixel offsets, in_mul, out_mul / offsets, n_sub).
//
// Without a halo there is no 9-fold reuse of a staged activation, so the activation side dominates: per 16-channel k-step a
// pixel feeds only N/32 x 6 MFMAs.  Tile = 8 rows x 32 px (one row per wave, 8 waves = one workgroup per CU), N = 128 (or 64)
// columns per workgroup (4 / 2 accumulator blocks per wave), work item = 32 channels of one K segment (two k-steps: 48 / 24
// MFMAs per wave between barriers).  Activations: fp32 global -> registers one item ahead (two register sets, so a request
// has a whole item of flight time) -> split into three bf16 pieces -> the other of two LDS images, between the MFMA groups.
// Weights: pre-split packs (pack_jobs kind 2 with one tap) by LDS-DMA into two stages, one item ahead.
#include "igemm.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int pnnp_gemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);
int pnnp_gemm_x3s_launch(const IgemmArgs& a, hipStream_t s);        // csrc/gemm_x3s.hip: the same GEMMs with specialised waves
#ifndef GX_SPEC
#define GX_SPEC 1                  // 1: the pointwise layers run on csrc/gemm_x3s.hip (producer / consumer waves); 0: on the kernel below
#endif

namespace {

constexpr int NWAVE = 8, NTHR = 512, TH = 8, NPIX = TH * 32;        // 256 pixels per tile
constexpr int KI = 32;                                              // channels per work item (two 16-channel k-steps)
// one image in 16-byte words: [k-step 2][octet 2] slots of [piece 3][pixel], each slot padded by 64 bytes: the staging writes of 8
// consecutive lanes go to 2 pixels x 4 slots, and with an unpadded slot stride (a multiple of 256 bytes) the four slots fell on the
// same banks -- a 4-way conflict on every ds_write_b128 that made the kernel LDS-bound
constexpr int SLOT_F4 = 3 * NPIX + 4;
constexpr int XS_F4 = 2 * 2 * SLOT_F4;
constexpr int XS_BYTES = XS_F4 * 16;                                // 49408
constexpr int WBLK = 2 * 3 * 32 * 16;                               // one k-step of one 32-column block: [octet 2][piece 3][32][16 B] = 3072
constexpr unsigned OOB = 0x80000000u;

template <int NT> struct GxCfg {
    static constexpr int BN = 32 * NT;
    static constexpr int WS_STAGE = NT * 2 * WBLK;                  // [block][k-step][3072]: 24576 / 12288
    static constexpr int NDMA = WS_STAGE / 1024, DPW = (NDMA + NWAVE - 1) / NWAVE;
    static constexpr int EPI = NWAVE * 2048;
    static constexpr bool EPI_ALIAS = WS_STAGE >= EPI;
    static constexpr int LDS_BYTES = 2 * XS_BYTES + 2 * WS_STAGE + (EPI_ALIAS ? 0 : EPI);
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

template <int NT>
__global__ void __launch_bounds__(NTHR, 1)
gemm_x3_kernel(const IgemmArgs a) {
    using Cfg = GxCfg<NT>;
    constexpr int BN = Cfg::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);
    char* wsb = smem + 2 * XS_BYTES;
    float* epi_sep = reinterpret_cast<float*>(smem + 2 * XS_BYTES + 2 * Cfg::WS_STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int G = gridDim.x;
    const int nitems = a.nseg * a.chunks_per_seg;                   // 32-channel items of K
    const int K16 = nitems * 2;

    // ---- staging: 256 px x 4 octets = 1024 (pixel, octet) slots of 8 channels, two per thread: slot s = tid + 512 k ->
    // pixel s >> 2, octet s & 3 (4 consecutive lanes read the 128 contiguous bytes of a pixel's 32 channels)
    const int oc = tid & 3;                                         // octet of the item (k-step oc >> 1, octet oc & 1)
    int prow[2], pcol[2], xdst[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pix = (tid + NTHR * k) >> 2;
        prow[k] = pix >> 5; pcol[k] = pix & 31;
        xdst[k] = ((oc >> 1) * 2 + (oc & 1)) * SLOT_F4 + pix;      // + piece * NPIX (+ image * XS_F4)
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    struct Tile { int b, y0, x0, n0; };
    auto decode = [&](int t) {
        Tile o;
        const int nt_i = t % n_tiles;
        int m_i = t / n_tiles;
        const int tx = m_i % tiles_x; m_i /= tiles_x;
        o.x0 = tx * 32; o.y0 = (m_i % tiles_y) * TH; o.b = m_i / tiles_y; o.n0 = nt_i * BN;
        return o;
    };

    f32x4 ra[2][2][2];                                              // [register set][slot][half of the 8 channels]
    // global loads of item g of tile tl into register set `set`
    auto load_item = [&](const Tile& tl, int g, int set) {
        const int si = g / a.chunks_per_seg, cc = g - si * a.chunks_per_seg;
        const IgemmSeg sg = a.seg[si];
        const int c0 = sg.coff + cc * KI;
        const int mul = a.in_mul;
        // the resource starts `shift` elements before the image so that the scalar offset below is never negative (offsets >= -1 pixel)
        const int shift = (a.IW + 1) * sg.cstride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(sg.ptr + ((int64_t)tl.b * a.IH * a.IW * sg.cstride - shift)), 0, 0x7fffffff, 0x00020000);
        const int soff = (((tl.y0 * mul + sg.yoff) * a.IW + tl.x0 * mul + sg.xoff) * sg.cstride + c0 + shift) * 4;
        const unsigned cs4 = (unsigned)sg.cstride * 4u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int iy = (tl.y0 + prow[k]) * mul + sg.yoff, ix = (tl.x0 + pcol[k]) * mul + sg.xoff;
            const int bad = iy | (a.IH - 1 - iy) | ix | (a.IW - 1 - ix) | (a.DH - 1 - tl.y0 - prow[k]) | (a.DW - 1 - tl.x0 - pcol[k]);
            const unsigned vo = bad < 0 ? OOB : __umul24((unsigned)((prow[k] * a.IW + pcol[k]) * mul), cs4) + oc * 32;
            ra[set][k][0] = bload(rs, vo, soff);
            ra[set][k][1] = bload(rs, vo, soff + 16);
        }
    };
    // one of the two staging slots of an item, in quarters: quarter p splits one float pair of slot k of register set `set`;
    // after the fourth the three 16-byte words go to image `img`
    u32x4 sH, sM, sL;
    float qa0[2], qa1[2];                                           // a quarter in flight (up to two per MFMA group)
    // step 0..4 of splitting float pair p of slot k: five dependent pieces of 1-4 VALU instructions, one per MFMA gap
    auto stage_piece = [&](int k, int set, int p, int q, int step) {
        if (step == 0) {
            const f32x4 v = ra[set][k][p >> 1];
            qa0[q] = v[(p & 1) * 2]; qa1[q] = v[(p & 1) * 2 + 1];
            sH[p] = cvt_pk_bf16(qa0[q], qa1[q]);
        } else if (step == 1) {
            qa0[q] -= __uint_as_float(sH[p] << 16); qa1[q] -= __uint_as_float(sH[p] & 0xffff0000u);
        } else if (step == 2) {
            sM[p] = cvt_pk_bf16(qa0[q], qa1[q]);
        } else if (step == 3) {
            qa0[q] -= __uint_as_float(sM[p] << 16); qa1[q] -= __uint_as_float(sM[p] & 0xffff0000u);
        } else if (step == 4) {
            sL[p] = cvt_pk_bf16(qa0[q], qa1[q]);
        }
    };
    auto stage_quarter = [&](int k, int set, int p) {
#pragma unroll
        for (int st = 0; st < 5; ++st) stage_piece(k, set, p, 0, st);
    };
    auto stage_write = [&](int k, int img) {
        u32x4* d = xs + img * XS_F4 + xdst[k];
        d[0] = sH; d[NPIX] = sM; d[2 * NPIX] = sL;
    };
    auto stage_slot = [&](int k, int set, int img) {
#pragma unroll
        for (int p = 0; p < 4; ++p) stage_quarter(k, set, p);
        stage_write(k, img);
    };
    // LDS-DMA of the weights of item g (tile columns n0 ..): per 32-column block two consecutive k-step blocks of the pack
    auto dma_weights = [&](const Tile& tl, int g, int st, bool valid = true) {
#pragma unroll
        for (int i = 0; i < Cfg::DPW; ++i) {
            const int ins = min(wave + NWAVE * i, Cfg::NDMA - 1);   // 1 KB pieces: [block j][6 pieces of its 6144 bytes]
            const int j = ins / 6, r = ins - 6 * j;
            const int nb = (tl.n0 >> 5) + j;
#ifdef GX_SKIP_DMA
            const bool ok = false;
#else
            const bool ok = valid && nb * 32 < a.Ntot;
#endif
            const int soff = ok ? ((nb * K16 + 2 * g) * WBLK + r * 1024) : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wsb + st * Cfg::WS_STAGE + ins * 1024),
                                                     16, ok ? (unsigned)lane * 16u : OOB, soff, 0, 0);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // the two k-steps of an item; the staging slot ks of register set `set` is split behind each k-step into image img ^ 1.
    // Operand reads run one 32-column block ahead of the MFMAs (two rotating B register sets; the next k-step's A with the last
    // block): only the first six reads of an item are exposed.  (All 15 reads of a k-step in front of its 24 MFMAs cost a full
    // LDS round trip per k-step: 35 % of the kernel's time.)
    auto mfma_item = [&](int st, int img, int set, auto&& requests) {
        const char* wst = wsb + st * Cfg::WS_STAGE;
        const u32x4* xim = xs + img * XS_F4;
        u32x4 av[2][3], bv[2][3];
        auto load_a = [&](int ks, u32x4 (&d)[3]) {
#pragma unroll
            for (int p = 0; p < 3; ++p) d[p] = xim[(ks * 2 + half) * SLOT_F4 + p * NPIX + wave * 32 + l31];
        };
        auto load_b = [&](int ks, int j, u32x4 (&d)[3]) {
#pragma unroll
            for (int p = 0; p < 3; ++p) d[p] = *reinterpret_cast<const u32x4*>(wst + (j * 2 + ks) * WBLK + ((half * 3 + p) * 32 + l31) * 16);
        };
        load_a(0, av[0]);
        load_b(0, 0, bv[0]);
        constexpr int QG = 4 / NT;                                  // staging quarters per 6-MFMA group
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cur = (ks * NT + j) & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < NT) load_b(ks, j + 1, bv[cur ^ 1]);
                else if (ks == 0) { load_b(1, 0, bv[cur ^ 1]); load_a(1, av[1]); }
                // The split of one staging slot is spread over the k-step's MFMAs: QG quarters per group, and inside the group one
                // dependent piece (1-4 VALU instructions) goes BETWEEN each two MFMAs, fenced so that it stays there.  The six MFMAs
                // of a group depend on each other through the accumulator: each leaves 32 cycles of issue slots, which a lump of
                // VALU work behind the group does not use (a wave alone on its SIMD then runs MFMAs and VALU back to back).
#define GX_MFMA(PA, PB) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[ks][PA]), __builtin_bit_cast(bf16x8, bv[cur][PB]), acc[j], 0, 0, 0)
#define GX_GAP(STEP) { _Pragma("unroll") for (int q = 0; q < QG; ++q) stage_piece(ks, set, j * QG + q, q, STEP); __builtin_amdgcn_sched_barrier(0); }
                GX_MFMA(0, 2); GX_GAP(0)
                GX_MFMA(2, 0); GX_GAP(1)
                GX_MFMA(1, 1); GX_GAP(2)
                GX_MFMA(0, 1); GX_GAP(3)
                GX_MFMA(1, 0); GX_GAP(4)
                GX_MFMA(0, 0);
#undef GX_GAP
#undef GX_MFMA
                if (j == NT - 1) stage_write(ks, img ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == 0 && j == 0) {
                    requests();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // ---- epilogue (csrc/conv_igemm.hip's fast path incl. the sub-pixel scatter of ConvTranspose2d forward)
    auto epilogue = [&](const Tile& tl, float* epi) {
        const int b = tl.b, x0 = tl.x0, y0 = tl.y0, n0 = tl.n0;
        float* eb = epi + wave * 512;
        const int q4 = (lane & 7) * 4, pr = lane >> 3;
        const int py = y0 + wave;
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
            const bool n_ok = nwv + q4 < a.Ntot;
            const int du = nwv >= a.n_split ? 1 : 0;
            const int subu = a.n_sub ? nwv / a.n_sub : 0;
            const int chw = nwv - subu * a.n_sub - (du ? a.n_split : 0);
            const int yo2 = a.out_yoff + (subu >> 1), xo2 = a.out_xoff + (subu & 1);
            const int cs2 = a.dst_cs[du], mm2 = a.mask_mode[du], acc2 = a.accum[du];
            const int oy2 = py * a.out_mul + yo2;
            const bool rowok = py < a.DH && oy2 >= 0 && oy2 < a.OH;
            const int64_t imgo = (int64_t)b * a.OH * a.OW * cs2;
            const int ibytes = a.OH * a.OW * cs2 * 4;
            float* dstb = a.dst[du] + imgo;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dstb, 0, ibytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(mm2 ? a.mask[du] + imgo : dstb), 0, ibytes, 0x00020000);
            const bool use_add2 = a.addsrc && du == 0;
            const __amdgpu_buffer_rsrc_t rad = __builtin_amdgcn_make_buffer_rsrc((void*)(use_add2 ? a.addsrc + imgo : dstb), 0, ibytes, 0x00020000);
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (a.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(a.bias + nwv - subu * a.n_sub + q4);
            const float aslope = a.act == 1 ? 0.2f : (a.act == 2 ? 0.f : 1.f), mslope = mm2 == 1 ? 0.2f : 0.f;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    eb[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[k][8 * h2 + r];
                    acc[k][8 * h2 + r] = 0.f;
                }
                unsigned vo[2];
                f32x4 v2[2], m2[2], ad2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = pr + 8 * e, px = x0 + 16 * h2 + p;
                    const int oxp = px * a.out_mul + xo2;
                    const bool ok2 = rowok && px < a.DW && n_ok && oxp >= 0 && oxp < a.OW;
                    vo[e] = ok2 ? (unsigned)(((oy2 * a.OW + oxp) * cs2 + chw + q4) * 4) : OOB;
                    v2[e] = *reinterpret_cast<const f32x4*>(eb + p * 32 + q4);
                }
                if (mm2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) m2[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, vo[e], 0, 0));
                }
                if (use_add2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) ad2[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rad, vo[e], 0, 0));
                }
                if (acc2) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) ad2[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vo[e], 0, 0));
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    f32x4 o = v2[e] + bias4;
                    if (use_add2) o += ad2[e];
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], aslope * o[c]);
                    if (mm2) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] *= (m2[e][c] > 0.f) ? 1.f : mslope;
                    }
                    if (acc2) o += ad2[e];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[e], 0, 0);
                }
            }
        }
    };

    // ---- main loop over (tile, item).  Register set of item g's activations: g & 1 ... but items are counted per workgroup
    // (`it`), so that the sets alternate across tile boundaries.  Issue order, all exact (in-order completion):
    //   item it (after its barrier):  [weights of item it+1 -> stage (it+1)&1] [activations of item it+2 -> set it&1]
    //   during item it the MFMA groups split set (it+1)&1 (requested during item it-1) into image (it+1)&1.
    int t = xcd_remap(blockIdx.x, G);
    if (t >= total) return;
    Tile cur = decode(t), nxt = decode(t + G < total ? t + G : t);
    int g = 0;
    struct Ck { Tile tile; int g; bool ok; };
    auto item_at = [&](int k) {
        int gk = g + k, tk = t;
        while (gk >= nitems) { gk -= nitems; tk += G; }
        Ck c;
        c.ok = tk < total;
        c.g = c.ok ? gk : g;
        c.tile = !c.ok || tk == t ? cur : (tk == t + G ? nxt : decode(tk));
        return c;
    };
    constexpr int D = Cfg::DPW, AL = 4;                             // vmcnt units: weight requests per item, activation loads per item
    // prologue: item 0 -> image 0 directly; item 1 -> register set 1; weights of item 0
    dma_weights(cur, 0, 0);
    load_item(cur, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
    stage_slot(0, 0, 0); stage_slot(1, 0, 0);
    {
        const Ck n1 = item_at(1);
        load_item(n1.tile, n1.g, 1);
    }
    int it = 0;
#ifdef GX_STAMPS                  // debug build: where does an item's time go?  (cycle sums per wave, dumped into dst[0] at the end)
    long long tw = 0, tb = 0, tm = 0, te = 0, to = 0, tall = clock64(), tlast_ = clock64();
#define GX_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define GX_T(v)
#endif
    for (;;) {
        const int st = it & 1, img = it & 1;
        const Ck n1 = item_at(1), n2 = item_at(2);
        GX_T(to)
        // outstanding, in issue order: [weights it] [activations it+1], both requested during item it-1.  The weights are needed now;
        // the activations only when the first staging slot is split (after the first k-step's MFMAs: the compiler's own count
        // waits there), so they get another third of an item of flight time
#ifdef GX_WAIT_ALL
        __builtin_amdgcn_s_waitcnt(0x0f70);
#else
        __builtin_amdgcn_s_waitcnt(0x0f70 | AL);
#endif
        GX_T(tw)
        __syncthreads();
        GX_T(tb)
        if (it & 1) mfma_item(st, img, 0, [&] { dma_weights(n1.tile, n1.g, st ^ 1, n1.ok); load_item(n2.tile, n2.g, 1); });
        else        mfma_item(st, img, 1, [&] { dma_weights(n1.tile, n1.g, st ^ 1, n1.ok); load_item(n2.tile, n2.g, 0); });
        GX_T(tm)
        if (g == nitems - 1) {
            if constexpr (Cfg::EPI_ALIAS) {
                __syncthreads();
                epilogue(cur, reinterpret_cast<float*>(wsb + st * Cfg::WS_STAGE));
            } else {
                epilogue(cur, epi_sep);
            }
        }
        GX_T(te)
        if (!n1.ok) break;
        if (g == nitems - 1) { t += G; cur = nxt; nxt = decode(t + G < total ? t + G : t); g = 0; } else ++g;
        ++it;
    }
#ifdef GX_STAMPS
    if (lane == 0) {
        __syncthreads();
        float* d = a.dst[0] + ((int64_t)blockIdx.x * NWAVE + wave) * 8;
        d[0] = (float)tw; d[1] = (float)tb; d[2] = (float)tm; d[3] = (float)te; d[4] = (float)to; d[5] = (float)(clock64() - tall); d[6] = (float)(it + 1);
    }
#endif
    (void)D; (void)AL;
}

template <int NT>
int launch_gx(const IgemmArgs& a, hipStream_t s) {
    using Cfg = GxCfg<NT>;
    auto kern = gemm_x3_kernel<NT>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * ((a.Ntot + Cfg::BN - 1) / Cfg::BN);
    if (tiles <= 0) return PNNP_OK;
    const int wgs = pnnp_persistent_grid(tiles);                     // one workgroup per CU, or n of 1/n share (pnnp_set_persistent_split)
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

}  // namespace

// a.w: x3 pack with ONE tap: [N/32][K/16][octet 2][piece 3][32][8 bf16].  chan_per_seg: channels per K segment, a multiple of 32.
int pnnp_gemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 9 || chan_per_seg <= 0 || (chan_per_seg & 31) || a.Ntot <= 0) return PNNP_E_INVALID;
    if ((a.Ntot & 31) || a.in_mul < 1 || a.in_mul > 2 || (a.n_sub & 31) || (a.dst[1] && (a.n_split & 31))) return PNNP_E_UNSUPPORTED;
    if (a.addsrc && a.accum[0]) return PNNP_E_UNSUPPORTED;
    if ((a.dst_cs[0] & 3) || (a.dst[1] && (a.dst_cs[1] & 3))) return PNNP_E_UNSUPPORTED;
    if ((((uintptr_t)a.dst[0]) | ((uintptr_t)a.dst[1]) | ((uintptr_t)a.bias) | ((uintptr_t)a.mask[0]) | ((uintptr_t)a.mask[1]) |
         ((uintptr_t)a.addsrc) | ((uintptr_t)a.w)) & 15) return PNNP_E_INVALID;
    for (int i = 0; i < a.nseg; ++i) {
        if (a.seg[i].yoff < -1 || a.seg[i].xoff < -1 || (a.seg[i].cstride & 3) || (((uintptr_t)a.seg[i].ptr) & 15)) return PNNP_E_UNSUPPORTED;
        if (((int64_t)a.IH + 4) * a.IW * a.seg[i].cstride * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    }
    for (int d = 0; d < 2; ++d)
        if (a.dst[d] && (int64_t)a.OH * a.OW * a.dst_cs[d] * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    IgemmArgs b = a;
    b.chunks_per_seg = chan_per_seg / KI;
    b.seg_channels = chan_per_seg;
    const int64_t wbytes = (int64_t)(a.Ntot / 32) * b.nseg * b.chunks_per_seg * 2 * WBLK;
    if (wbytes >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    if (GX_SPEC) {
        b.chunks_per_seg = chan_per_seg / 16;                       // csrc/gemm_x3s.hip walks K in 16-channel items
        return pnnp_gemm_x3s_launch(b, s);
    }
    if (a.amax_out[0] || a.amax_out[1]) return PNNP_E_UNSUPPORTED;  // (only the specialised kernel reports amax)
    // 128-column tiles unless they leave CUs idle (single-crop forwards): then 64-column tiles, twice as many
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t tiles128 = (int64_t)((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * (a.Ntot / 128);
    return (a.Ntot % 128 == 0 && tiles128 * 4 >= (int64_t)cus * 3) ? launch_gx<4>(b, s) : launch_gx<2>(b, s);
}
