// Pointwise convolutions as a float32 GEMM on the fp16 matrix cores (round 5; the fp16x2 scheme of csrc/h2.h applied to what csrc/gemm_x3s.hip
// runs on the exact bf16 three-way split: ConvTranspose2d(k2, s2) forward / backward-data, 1x1 shortcuts, the stride-2 3x3 convolution as 9
// strided taps and its backward-data per input-pixel parity class -- one tap per K segment, described by IgemmArgs).
//
// Same workgroup as gemm_x3s: 8 CONSUMER waves (ds_read_b128 + v_mfma_f32_16x16x32_f16 only, epilogue straight from the accumulators) and 4
// PRODUCER waves (weights by LDS-DMA, activations fp32 global -> registers -> x 2^se, hi / lo by v_fma_mix*_f16 -> the other of two LDS
// images), one s_barrier per item, the same cursor over (tile, K segment, item) and the same three epilogues.  What changes:
//   * an ITEM is 32 channels (two 16-channel halves h0, h1), so that the `hi lo'` products of the two halves share one K = 32 instruction:
//       pixels [hi h0 | lo h0] x weights [hi' h0 | hi' h0]     = hi hi' + lo hi' of h0
//       pixels [hi h1 | lo h1] x weights [hi' h1 | hi' h1]     =   ...           of h1
//       pixels [hi h0 | hi h1] x weights [lo' h0 | lo' h1]     = hi lo' of both
//     3 instructions per 32 channels and 16 x 16 block where bf16x3 needs 6: 3 executed FLOP per algorithmic FLOP instead of 6;
//   * images [piece 2][octet 4][pixel] 16-byte words, weights [piece 2][octet 4][32][8] fp16 = 4096 bytes per item and 32-column block
//     (csrc/pack_jobs.hip kind 6), scaled with the weight tensor's amax slot; the activations' scale comes from the amax slots of the K
//     segments' tensors; the epilogue multiplies by 2^-(se_x + se_w) first;
//   * an item carries twice the bytes of a bf16x3 item at the same matrix-pipe time, so the lookahead is 3 items (96 KB per CU in flight).
// Tiles: 256 px x 128 columns (4 x 2 consumer waves of 64 px x 64), 256 px x 64 (4 x 2 waves of 64 px x 32) and, for GEMMs with 32 (mod 64) columns, 256 px x 32
// (4 x 1 waves of 64 px x 32; the other four consumer waves only keep the barriers -- these layers move 32 KB per item for 384 matrix-pipe cycles: HBM-bound).
#include "h2.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

int pnnp_gemm_h2s_launch(const H2Args& a, hipStream_t s);

namespace {

constexpr int NCW = 8, NPW = 4, NTHR = 64 * (NCW + NPW), PTHR = 64 * NPW;
constexpr int MT = 2;                                              // pixel rows (of 32 px) per consumer wave
constexpr int WBLK1 = 2 * 4 * 32 * 16;                             // one item of one 32-column block: [piece 2][octet 4][32][16 B] = 4096
constexpr unsigned OOB = 0x80000000u;
#ifndef GHS_STORE_AUX
#define GHS_STORE_AUX 2              // cache-policy bits of the epilogue's stores: 2 = nt (non-temporal, as in csrc/conv_x3s.hip: config 3 +0.3 %, config 5 +0.6 %,
                                   // three alternating same-box pairs: profiles/r4/ab_store_policy.txt)
#endif
#define GHS_VMCNT(N) (0x0f70 | ((N) & 15) | (((N) >> 4) << 14))
#define GHS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")     // (see csrc/conv_x3s.hip: not __syncthreads())

template <int BN, int WN_> struct GCfg {
    static constexpr int WN = WN_;                                 // columns per consumer wave: 64 or 32
    static constexpr int NWN = BN / WN, NWM = BN == 32 ? 4 : NCW / NWN;      // consumer waves along N (2; 1 for the 32-column tile) and along the pixels (4)
    static constexpr int NACT = NWN * NWM;                         // consumer waves that compute (the 32-column tile: 4 of the 8 -- its layers are HBM-bound several times over)
    static constexpr int TH = NWM * MT, PT = TH * 32;              // tile: 8 rows of 32 px = 256 pixels
    static constexpr int NB = WN / 16, NTW = WN / 32;              // 16-column accumulator blocks / 32-column blocks per wave
    static constexpr int XS_F4 = 2 * 4 * PT, XS_BYTES = XS_F4 * 16; // one image: [piece 2][octet 4][pixel] 16-byte words: 32768
    static constexpr int WS_STAGE = (BN / 32) * WBLK1;             // 16384 / 8192
    static constexpr int NDMA = WS_STAGE / 1024, DPW = (NDMA + NPW - 1) / NPW;
    static constexpr int NSL = 4 * PT / PTHR;                      // (pixel, octet) staging slots per producer thread and item: 4
    static constexpr int A = 3;                                    // items of lookahead (= register sets of the producers); ring of A + 1 weight stages
    static constexpr int NSTAGE = A + 1;
    static constexpr int LDS_BYTES = 2 * XS_BYTES + NSTAGE * WS_STAGE;
    static_assert(NWN * WN == BN && PT == 256 && NACT <= NCW, "tile shapes");
    static_assert(LDS_BYTES <= 160 * 1024, "a workgroup's LDS");
    static_assert(A * 2 * NSL + (A - 1) * DPW <= 63, "the producers' vmcnt");
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
// hi = f16(a s), lo = f16(a s - hi) of two values, packed (low half = a0): csrc/conv_h2s.hip split_h2
__device__ __forceinline__ void split_h2(float a0, float a1, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(s), "v"(h));
    hi = h; lo = l;
}

#ifdef GHS_STAMPS                 // debug build: cycle sums per wave, dumped into dst[0] (tools/gx_stamps.py --spec)
#define GHS_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define GHS_T(v)
#endif
enum { EK_FWD = 0, EK_BWD = 1, EK_GEN = 2, EK_BWDB = 3 };          // the epilogue a kernel carries: plain / act' masks (float32) / residual + accumulation / act' masks as the
                                                                   // 3x3 forward kernel's SIGN BITS (round 6: ConvTranspose2d backward-data read 503 MB of float32 activations per step as masks)

template <int BN, int WN, int EK>
__global__ void __launch_bounds__(NTHR, 1)
gemm_h2s_kernel(const H2Args ha) {
    const IgemmArgs& a = ha.g;
    using Cfg = GCfg<BN, WN>;
    constexpr int TH = Cfg::TH, PT = Cfg::PT, NB = Cfg::NB, NTW = Cfg::NTW, NSL = Cfg::NSL, D = Cfg::DPW, XS_F4 = Cfg::XS_F4, A = Cfg::A, NSTAGE = Cfg::NSTAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                     // two activation images
    char* wsb = smem + 2 * Cfg::XS_BYTES;                           // the weight ring

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 7 consumers, 8 .. 11 producers
    // ---- scales (csrc/h2.h): se_x from the largest amax of the K segments' tensors, se_w from the weight tensor's
    unsigned ax = ha.amax_in[0] ? ha.amax_in[0][0] : 0u;
    if (ha.amax_in[1]) { const unsigned a2 = ha.amax_in[1][0]; ax = a2 > ax ? a2 : ax; }
    const int se_x = __builtin_amdgcn_readfirstlane(pnnp_h2_scale_exp(ax));
    const int se_w = __builtin_amdgcn_readfirstlane(ha.amax_w ? pnnp_h2_scale_exp(ha.amax_w[0]) : 0);

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int G = gridDim.x;
    const int nitems = a.nseg * a.chunks_per_seg;                   // 32-channel items of K
    struct Tile { int b, y0, x0, n0; };
    auto decode = [&](int t) {
        Tile o;
        const int nt_i = t % n_tiles;
        int m_i = t / n_tiles;
        const int tx = m_i % tiles_x; m_i /= tiles_x;
        o.x0 = tx * 32; o.y0 = (m_i % tiles_y) * TH; o.b = m_i / tiles_y; o.n0 = nt_i * BN;
        return o;
    };
    // A cursor walks this workgroup's items (tile t, t + G, ...; inside a tile the K segments, inside a segment its 16-channel chunks) ONE
    // item at a time with additions and carries only: decoding a tile number costs six integer divisions by run-time values, and with
    // one decode per lookahead per item (first version) the producers spent 1750 of an item's 3100 cycles on bookkeeping.
    const Tile gstep = decode(G);
    auto advance = [&](Tile o) {
        o.n0 += gstep.n0; if (o.n0 >= n_tiles * BN) { o.n0 -= n_tiles * BN; o.x0 += 32; }
        o.x0 += gstep.x0; if (o.x0 >= tiles_x * 32) { o.x0 -= tiles_x * 32; o.y0 += TH; }
        o.y0 += gstep.y0; if (o.y0 >= tiles_y * TH) { o.y0 -= tiles_y * TH; o.b += 1; }
        o.b += gstep.b;
        return o;
    };
    struct It { Tile tile; int t, g, si, cc; bool ok; };
    auto step = [&](It& c) {
        ++c.g;
        if (++c.cc == a.chunks_per_seg) { c.cc = 0; ++c.si; }
        if (c.g == nitems) { c.g = 0; c.si = 0; c.cc = 0; c.t += G; c.tile = advance(c.tile); c.ok = c.t < total; }
    };
    It cu;                                                           // the current item
    cu.t = xcd_remap(blockIdx.x, G);
    if (cu.t >= total) return;
    cu.tile = decode(cu.t); cu.g = 0; cu.si = 0; cu.cc = 0; cu.ok = true;

    if (wave >= NCW) {
        // =============================================== PRODUCER ===============================================
#ifdef GHS_PPRIO
        __builtin_amdgcn_s_setprio(GHS_PPRIO);                        // experiment: the producers win the issue arbitration
#endif
        const int pw = wave - NCW, ptid = tid - 64 * NCW;
        // staging slots: s = ptid + 256 k -> (pixel s >> 2, channel octet s & 3): four consecutive lanes read the 128 contiguous bytes of a pixel
        const int oct = ptid & 3;
        const float sx = __uint_as_float((unsigned)(se_x + 127) << 23);      // 2^se_x
        int prow[NSL], pcol[NSL], xdst[NSL];
#pragma unroll
        for (int k = 0; k < NSL; ++k) {
            const int pix = (ptid + PTHR * k) >> 2;
            prow[k] = pix >> 5; pcol[k] = pix & 31;
            xdst[k] = oct * PT + pix;                               // + piece * 4 PT (+ image * XS_F4)
        }
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
        f32x4 ra[A][NSL][2];                                        // A register sets: the activations of the next A items
        // What a request needs is computed once per (tile, K segment) -- resource, scalar offset, the slots' lane offsets and validity -- and
        // once per tile for the weights; inside a segment the next item is 16 channels (64 bytes) on, the next k-step of the pack 3072 bytes
        // on.  (A producer shares its SIMD with two MFMA waves and gets an issue slot every ~8 cycles: the ~150 instructions of a request
        // computed from scratch took 1200 cycles of a 1536-cycle item.)
        const float* la_base = a.w; int la_soff = 0; unsigned la_vo[NSL];      // (the resource is re-made from the pointer: 4 scalar moves)
        static_assert(D <= 4, "LDS-DMA pieces per producer wave");
        int la_wsoff[4]; unsigned la_wvo[4];                        // (a literal size: with [D] the host pass of hipcc 7.2 silently drops the kernel stubs)
#pragma unroll
        for (int k = 0; k < NSL; ++k) la_vo[k] = OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) { la_wsoff[i] = 0; la_wvo[i] = OOB; }
        auto load_item = [&](const It& q, auto set_tag) {
            constexpr int set = decltype(set_tag)::value;
            if (q.cc == 0) {                                        // first item of a segment (of a tile)
                const Tile& tl = q.tile;
                const IgemmSeg sg = a.seg[q.si];
                const int mul = a.in_mul;
                const int shift = (a.IW + 1) * sg.cstride;         // the resource starts before the image: offsets >= -1 pixel stay >= 0
                la_base = sg.ptr + ((int64_t)tl.b * a.IH * a.IW * sg.cstride - shift);
                la_soff = (((tl.y0 * mul + sg.yoff) * a.IW + tl.x0 * mul + sg.xoff) * sg.cstride + sg.coff + shift) * 4;
                const unsigned cs4 = (unsigned)sg.cstride * 4u;
#pragma unroll
                for (int k = 0; k < NSL; ++k) {
                    const int iy = (tl.y0 + prow[k]) * mul + sg.yoff, ix = (tl.x0 + pcol[k]) * mul + sg.xoff;
                    const int bad = iy | (a.IH - 1 - iy) | ix | (a.IW - 1 - ix) | (a.DH - 1 - tl.y0 - prow[k]) | (a.DW - 1 - tl.x0 - pcol[k]) | (q.ok ? 0 : -1);
                    la_vo[k] = bad < 0 ? OOB : __umul24((unsigned)((prow[k] * a.IW + pcol[k]) * mul), cs4) + oct * 32;
                }
            }
            const __amdgpu_buffer_rsrc_t la_rs = __builtin_amdgcn_make_buffer_rsrc((void*)la_base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                ra[set][k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(la_rs, la_vo[k], la_soff, 0));
                ra[set][k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(la_rs, la_vo[k], la_soff + 16, 0));
            }
            la_soff += 128;
        };
        auto stage_set = [&](auto set_tag, int img) {
            constexpr int set = decltype(set_tag)::value;
#pragma unroll
            for (int k = 0; k < NSL; ++k) {
                u32x4 sh, sl;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 v = ra[set][k][p >> 1];
                    unsigned h, l;
                    split_h2(v[(p & 1) * 2], v[(p & 1) * 2 + 1], sx, h, l);
                    sh[p] = h; sl[p] = l;
                }
                u32x4* d = xs + img * XS_F4 + xdst[k];
                d[0] = sh; d[4 * PT] = sl;
            }
        };
        // LDS-DMA of the weights of item q into stage st: per 32-column block the 4096 contiguous bytes of its k-step, as 1 KB pieces
        const int K32 = nitems;
        auto dma_weights = [&](const It& q, int st) {
            if (q.g == 0) {                                         // first item of a tile
#pragma unroll
                for (int i = 0; i < D; ++i) {
                    const int ins = min(pw + NPW * i, Cfg::NDMA - 1);
                    const int j = ins >> 2, r = ins & 3;
                    const int nb = (q.tile.n0 >> 5) + j;
                    const bool ok = q.ok && nb * 32 < a.Ntot;
                    la_wsoff[i] = ok ? (nb * K32 * WBLK1 + r * 1024) : 0;
                    la_wvo[i] = ok ? (unsigned)lane * 16u : OOB;
                }
            }
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int ins = min(pw + NPW * i, Cfg::NDMA - 1);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wsb + st * Cfg::WS_STAGE + ins * 1024),
                                                         16, la_wvo[i], la_wsoff[i], 0, 0);
                la_wsoff[i] += WBLK1;
            }
        };
        // ---- prologue: weights and activations of items 0 .. A - 1 (stage j, set j); item 0 straight into image 0
        It la = cu;                                                  // the lookahead cursor: A items ahead of cu
        static_for<0, A>([&](auto J) { constexpr int j = decltype(J)::value; dma_weights(la, j); load_item(la, J); step(la); });
        stage_set(std::integral_constant<int, 0>{}, 0);
        GHS_BARRIER();                                              // barrier 0 (the wait for set 0 covered the weights of item 0)
        int it = 0, st = 0;                                          // items since the start (image it & 1); stage of item it
        // Item it, S = it mod A: [weights of item it + A -> the stage item it - 1 left] [activations of item it + A -> set S, split during
        // item it - 1] [split the set of item it + 1 into image (it + 1) & 1]; in front of the barrier the weights of item it + 1
        // (requested A - 1 blocks ago) must have landed: vmcnt(what was issued behind them = A x the loads of an item + (A - 1) x its LDS-DMAs).
#ifdef GHS_STAMPS
        long long t_req = 0, t_split = 0, t_wait = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_;
#endif
        auto block = [&](auto s_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value;
            dma_weights(la, st == 0 ? NSTAGE - 1 : st - 1);
            load_item(la, s_tag);
            step(la);
            GHS_T(t_req)
            stage_set(std::integral_constant<int, (S + 1) % A>{}, (it + 1) & 1);
            GHS_T(t_split)
            __builtin_amdgcn_s_waitcnt(GHS_VMCNT(A * 2 * NSL + (A - 1) * D));
            GHS_T(t_wait)
            step(cu);
            if (!cu.ok) return false;
            GHS_BARRIER();
            GHS_T(t_bar)
            ++it; st = st == NSTAGE - 1 ? 0 : st + 1;
            return true;
        };
        for (;;) {
            bool go = true;
            static_for<0, A>([&](auto S) { if (go) go = block(S); });
            if (!go) break;
        }
#ifdef GHS_STAMPS
        if (lane == 0) {
            float* d = a.dst[0] + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
            d[0] = (float)t_req; d[1] = (float)t_split; d[2] = (float)t_wait; d[3] = (float)t_bar; d[4] = (float)(clock64() - tall); d[5] = (float)(it + 1);
        }
#endif
        return;
    }

    // =============================================== CONSUMER ===============================================
    if constexpr (Cfg::NACT < NCW) {
        if (wave >= Cfg::NACT) {                                     // (32-column tile) a consumer wave without a share: the barriers of the item loop, nothing else
            GHS_BARRIER();
            for (;;) {
                step(cu);
                if (!cu.ok) break;
                GHS_BARRIER();
            }
            return;
        }
    }
    const int wn = wave % Cfg::NWN, wm = wave / Cfg::NWN;             // this wave's column group / pixel-row pair
    constexpr int MB = 2 * MT;
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q16 = lane >> 4;                      // pixel / column of the block; the lane's K group of 8
    // operand forms (k-group q16 of an instruction): pixels 0 = [hi h0 | lo h0], 1 = [hi h1 | lo h1], 2 = [hi h0 | hi h1];
    // weights 0 = [hi' h0 | hi' h0], 1 = [hi' h1 | hi' h1], 2 = [lo' h0 | lo' h1]   (h0 / h1 = octets 0,1 / 2,3 of the 32-channel item)
    const int pb = wm * (MT * 32) + r16;
    const int aoff0 = ((q16 >> 1) * 4 + (q16 & 1)) * PT + pb, aoff1 = ((q16 >> 1) * 4 + 2 + (q16 & 1)) * PT + pb, aoff2 = q16 * PT + pb;
    const int boff0 = ((q16 & 1) * 32 + r16) * 16, boff1 = ((2 + (q16 & 1)) * 32 + r16) * 16, boff2 = ((4 + q16) * 32 + r16) * 16;
    auto mfma_item = [&](int st, int img) {
        const char* wst = wsb + st * Cfg::WS_STAGE + wn * NTW * WBLK1;
        const u32x4* xim = xs + img * XS_F4;
        u32x4 A[MB][3], Bv[2][3];
        auto a_read = [&](int mb, int f) { A[mb][f] = xim[(f == 0 ? aoff0 : (f == 1 ? aoff1 : aoff2)) + (mb >> 1) * 32 + 16 * (mb & 1)]; };
        auto b_read = [&](int j, int f, int buf) {
            Bv[buf][f] = *reinterpret_cast<const u32x4*>(wst + (j >> 1) * WBLK1 + (f == 0 ? boff0 : (f == 1 ? boff1 : boff2)) + (j & 1) * 256);
        };
#pragma unroll
        for (int f = 0; f < 3; ++f) b_read(0, 2 - f, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { a_read(mb, 2); a_read(mb, 1); a_read(mb, 0); }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, NB * MB * 3>([&](auto Gc) {
            constexpr int gi = decltype(Gc)::value;
            constexpr int j = gi / (MB * 3), w = gi % (MB * 3), mb = w / 3, sp = w % 3, buf = j & 1;
            constexpr int F = 2 - sp;                               // hi lo' of both halves first, then the two halves' hi hi' + lo hi'
            acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Bv[buf][F]), __builtin_bit_cast(f16x8, A[mb][F]), acc[mb][j], 0, 0, 0);
            if constexpr (w < 3 && j + 1 < NB) b_read(j + 1, 2 - w, buf ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- epilogue, straight from the accumulators (a lane holds 4 consecutive channels of one pixel): sub-pixel scatter of ConvTranspose2d
    // forward (n_sub), strided / offset outputs (out_mul, out_yoff / out_xoff), two destinations (n_split), bias, activation, act' mask,
    // residual, accumulation -- csrc/gemm_x3.hip's contract
    float amx0 = 0.f;                                                // max |stored value| of this lane, destination 0 (a.amax_out[0]; the launcher refuses [1])
    auto epilogue = [&](const Tile& tl) __attribute__((always_inline)) {
        const int b = tl.b;
        // (general epilogue: the lane number as an OPAQUE value, csrc/conv_h2s.hip -- what is derived from it is computed here, per tile, instead of being
        //  hoisted in front of the item loop and carried through it; the other two epilogues sit at exactly 168 registers without spilling as they are)
        int lane_e = lane;
        if constexpr (EK == EK_GEN) asm volatile("" : "+v"(lane_e));
        const int p16 = lane_e & 15, c4 = (lane_e >> 4) * 4;
        const int py0 = tl.y0 + wm * MT, px0 = tl.x0 + p16;
        int du_[NTW], cs_[NTW], bch_[NTW]; bool blk_[NTW];
        unsigned vo[NTW][MT][2];
        // (the general epilogue derives the offsets and the bias words of ONE 32-column block at a time, inside its own loop: its 128-column kernel
        //  has no registers to spare; the other two keep everything up front)
        auto block_setup = [&](int k) __attribute__((always_inline)) {
            const int nwv = __builtin_amdgcn_readfirstlane(tl.n0 + wn * WN + k * 32);
            const int du = nwv >= a.n_split ? 1 : 0;
            const int subu = a.n_sub ? nwv / a.n_sub : 0;
            const int chw = nwv - subu * a.n_sub - (du ? a.n_split : 0);
            const int yo2 = a.out_yoff + (subu >> 1), xo2 = a.out_xoff + (subu & 1);
            du_[k] = du; cs_[k] = a.dst_cs[du]; blk_[k] = nwv < a.Ntot; bch_[k] = nwv - subu * a.n_sub;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int py = py0 + i, px = px0 + 16 * h;
                    const int oy = py * a.out_mul + yo2, ox = px * a.out_mul + xo2;
                    const bool ok = blk_[k] && py < a.DH && px < a.DW && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
                    vo[k][i][h] = ok ? (unsigned)(((oy * a.OW + ox) * cs_[k] + chw + c4) * 4) : OOB;
                }
        };
        if constexpr (EK != EK_GEN) {
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(tl.n0 + wn * WN + k * 32);
            const int du = nwv >= a.n_split ? 1 : 0;
            const int subu = a.n_sub ? nwv / a.n_sub : 0;
            const int chw = nwv - subu * a.n_sub - (du ? a.n_split : 0);
            const int yo2 = a.out_yoff + (subu >> 1), xo2 = a.out_xoff + (subu & 1);
            du_[k] = du; cs_[k] = a.dst_cs[du]; blk_[k] = nwv < a.Ntot; bch_[k] = nwv - subu * a.n_sub;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int py = py0 + i, px = px0 + 16 * h;
                    const int oy = py * a.out_mul + yo2, ox = px * a.out_mul + xo2;
                    const bool ok = blk_[k] && py < a.DH && px < a.DW && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
                    vo[k][i][h] = ok ? (unsigned)(((oy * a.OW + ox) * cs_[k] + chw + c4) * 4) : OOB;
                }
        }
        }
        auto rsrc = [&](const float* base, int k) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)b * a.OH * a.OW * cs_[k]), 0, a.OH * a.OW * cs_[k] * 4, 0x00020000);
        };
        const float aslope = a.act == 1 ? 0.2f : (a.act == 2 ? 0.f : 1.f);
        f32x4 bias4[NB];
        auto bias_load = [&](int j) __attribute__((always_inline)) {
            bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && blk_[j >> 1]) bias4[j] = *reinterpret_cast<const f32x4*>(a.bias + bch_[j >> 1] + 16 * (j & 1) + c4);
        };
        if constexpr (EK != EK_GEN) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias && blk_[j >> 1]) bias4[j] = *reinterpret_cast<const f32x4*>(a.bias + bch_[j >> 1] + 16 * (j & 1) + c4);
        }
        }
        auto act4 = [&](f32x4 o) {
            const f32x4 t = o * aslope;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], t[c]);
            return o;
        };
        // Undoing the operand scales: x 2^dexp (exact) as one multiplier while 2^dexp is a normal float32; tensors so small / large that it is
        // not first take the remainder in a pass over the accumulators (csrc/conv_h2s.hip)
        const int dexp = -(se_x + se_w);
        const int dexp_c = dexp < -120 ? -120 : (dexp > 120 ? 120 : dexp);      // (|.| <= 120: 0.2 x 2^dexp_c stays a normal float32: mask_scale)
        const float dsc = __uint_as_float((unsigned)(dexp_c + 127) << 23);
        if (dexp != dexp_c) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[mb][j][c] = __builtin_ldexpf(acc[mb][j][c], dexp - dexp_c);
        }
        auto take = [&](int mb, int j) { const f32x4 v = acc[mb][j]; acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; return v * dsc; };
        // max |.| of a stored block into the lane's running maximum of destination du (uniform); lanes whose store is dropped do not count
        auto track = [&](f32x4 o, bool valid, int du) {
            const float m = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
            amx0 = fmaxf(amx0, (valid && !du) ? m : 0.f);
        };
        if constexpr (EK == EK_FWD || EK == EK_BWD || EK == EK_BWDB) {
            constexpr bool MASKED = EK == EK_BWD, BITS = EK == EK_BWDB;
            // FULL-LINE memory pattern (csrc/conv_x3s.hip): the two 16-column blocks of a 32-column block trade halves between lanes p and p + 8
            // of a 16-lane row, so that each 16-byte store instruction writes 8 pixels x 128 bytes (whole lines) instead of 16 x 64; the
            // act' masks come in by the same pattern and are traded back.
            const bool lo8 = p16 < 8;
            auto ror8 = [&](f32x4 v) {
                float r0, r1, r2, r3;
                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                             "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
                             : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
                return f32x4{r0, r1, r2, r3};
            };
            auto sel = [&](bool c, f32x4 x, f32x4 y) { return f32x4{c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w}; };
            // this lane's byte offsets in instruction 1 (pixel p16 & 7 of the half) and 2 (eight pixels on) of block k
            unsigned wo[NTW][MT][2][2];
            unsigned mbits[BITS ? NTW : 1];                          // (EK_BWDB) this lane's word of the tile-private sign-bit image (csrc/h2.h) per 32-column block
            const int pxl = tl.x0 + (p16 & 7);
#pragma unroll
            for (int k = 0; k < NTW; ++k) {
                const int nwv = __builtin_amdgcn_readfirstlane(tl.n0 + wn * WN + k * 32);
                const int subu = a.n_sub ? nwv / a.n_sub : 0;
                const int chw = nwv - subu * a.n_sub - (du_[k] ? a.n_split : 0);
                if constexpr (BITS) {
                    // The 3x3 kernel's word (16-row tile ty, tile column, 32-channel block, its consumer wave w16, lane) holds rows 2 w16 + i, pixels 16 h + (lane & 15),
                    // channels 16 jj + 4 (lane >> 4) + c of the block -- exactly the 32 values THIS lane holds of the block (this wave's rows py0, py0 + 1: py0 even),
                    // in the order the loop below walks them.  Plain output geometry only (the launcher checks): out pixel = tile pixel.
                    const int tiles_x16 = (a.OW + 31) >> 5, tiles_y16 = (a.OH + 15) >> 4, nblk = ha.bits_nblk[0];
                    const int tile_id = (b * tiles_y16 + (py0 >> 4)) * tiles_x16 + (tl.x0 >> 5);
                    const unsigned word = (unsigned)((((tile_id * nblk + (chw >> 5)) * 8 + ((py0 & 15) >> 1)) * 64 + lane) * 4);
                    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)ha.bits_in[0], 0, a.B * tiles_y16 * tiles_x16 * nblk * 8 * 64 * 4, 0x00020000);
                    mbits[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, (blk_[k] && a.mask_mode[du_[k]] && !du_[k] && py0 < a.DH) ? word : OOB, 0, 0);
                }
                const int yo2 = a.out_yoff + (subu >> 1), xo2 = a.out_xoff + (subu & 1);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int py = py0 + i, px = pxl + 16 * h + 8 * e;
                            const int oy = py * a.out_mul + yo2, ox = px * a.out_mul + xo2;
                            const bool ok = blk_[k] && py < a.DH && px < a.DW && oy >= 0 && oy < a.OH && ox >= 0 && ox < a.OW;
                            wo[k][i][h][e] = ok ? (unsigned)(((oy * a.OW + ox) * cs_[k] + chw + (lo8 ? 0 : 16) + c4) * 4) : OOB;
                        }
            }
            f32x4 mk[MASKED ? MB : 1][MASKED ? NB : 1];              // [.][2 k] = what instruction 1 fetched, [.][2 k + 1] = instruction 2
            if constexpr (MASKED) {
#pragma unroll
                for (int k = 0; k < NTW; ++k) {
                    const int mm = a.mask_mode[du_[k]];
                    const __amdgpu_buffer_rsrc_t rm = rsrc(mm ? a.mask[du_[k]] : a.dst[du_[k]], k);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int e = 0; e < 2; ++e)
                                mk[2 * i + h][2 * k + e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm ? wo[k][i][h][e] : OOB, 0, 0));
                }
            }
#pragma unroll
            for (int k = 0; k < NTW; ++k) {
                const __amdgpu_buffer_rsrc_t rd = rsrc(a.dst[du_[k]], k);
                const int mm = a.mask_mode[du_[k]];
                const float msl = mm == 1 ? 0.2f : (mm == 0 ? 1.f : 0.f);      // act'(x <= 0); 1 for a destination without a mask
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 o0, o1;
                        if constexpr (BITS) {
                            // scale and mask in three instructions per element (csrc/conv_h2s.hip mask_scale): next bit -> vcc, 2^dexp or msl 2^dexp, multiply
                            const f32x4 v0 = acc[2 * i + h][2 * k], v1 = acc[2 * i + h][2 * k + 1];
                            acc[2 * i + h][2 * k] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * i + h][2 * k + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                            const float fneg = dsc * msl;
                            auto ms = [&](float v) __attribute__((always_inline)) {
                                float f, o;
                                asm("v_add_co_u32 %2, vcc, %2, %2\n\tv_cndmask_b32 %1, %4, %3, vcc\n\tv_mul_f32 %0, %1, %5" : "=v"(o), "=&v"(f), "+v"(mbits[k]) : "v"(dsc), "v"(fneg), "v"(v) : "vcc");
                                return o;
                            };
#pragma unroll
                            for (int c = 0; c < 4; ++c) o0[c] = ms(v0[c]);
#pragma unroll
                            for (int c = 0; c < 4; ++c) o1[c] = ms(v1[c]);
                        } else if constexpr (!MASKED) {
                            // forward: scale and bias in ONE fma per element; the activation only where the layer has one (ConvTranspose2d has none: it used to pay a
                            // multiply and a maximum per element for max(o, 1.0 o))
                            const f32x4 v0 = acc[2 * i + h][2 * k], v1 = acc[2 * i + h][2 * k + 1];
                            acc[2 * i + h][2 * k] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[2 * i + h][2 * k + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int c = 0; c < 4; ++c) { o0[c] = __builtin_fmaf(v0[c], dsc, bias4[2 * k][c]); o1[c] = __builtin_fmaf(v1[c], dsc, bias4[2 * k + 1][c]); }
                            if (a.act) { o0 = act4(o0); o1 = act4(o1); }
                        } else {
                            o0 = take(2 * i + h, 2 * k); o1 = take(2 * i + h, 2 * k + 1);
                        }
                        if constexpr (MASKED) {
                            const f32x4 m1 = mk[2 * i + h][2 * k], m2 = mk[2 * i + h][2 * k + 1], mx = ror8(sel(lo8, m2, m1));
                            const f32x4 q0 = sel(lo8, m1, mx), q1 = sel(lo8, mx, m2);
                            const f32x4 t0 = o0 * msl, t1 = o1 * msl;
#pragma unroll
                            for (int c = 0; c < 4; ++c) { o0[c] = q0[c] > 0.f ? o0[c] : t0[c]; o1[c] = q1[c] > 0.f ? o1[c] : t1[c]; }
                        }
                        track(o0, vo[k][i][h] != OOB, du_[k]); track(o1, vo[k][i][h] != OOB, du_[k]);
                        const f32x4 ox = ror8(sel(lo8, o1, o0));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, o0, ox)), rd, wo[k][i][h][0], 0, GHS_STORE_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, ox, o1)), rd, wo[k][i][h][1], 0, GHS_STORE_AUX);
                    }
            }
            return;
        }
        // the general case (residual, accumulation), branch-free: what a block does not use is requested out of range (zeros, no traffic)
#pragma unroll
        for (int k = 0; k < NTW; ++k) {
            block_setup(k); bias_load(2 * k); bias_load(2 * k + 1);
            const int du = du_[k], mm2 = a.mask_mode[du], acc2 = a.accum[du];
            const bool use_add2 = a.addsrc && du == 0;
            const __amdgpu_buffer_rsrc_t rd = rsrc(a.dst[du], k);
            const __amdgpu_buffer_rsrc_t rm = rsrc(mm2 ? a.mask[du] : a.dst[du], k);
            const __amdgpu_buffer_rsrc_t rad = rsrc(use_add2 ? a.addsrc : a.dst[du], k);
            const float msl = mm2 == 1 ? 0.2f : 0.f;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                // (one pixel row at a time: with the mask / residual / previous-value words of BOTH rows in flight -- 48 registers -- the 128-column kernel
                //  spilled 7 registers into a 32-byte scratch frame)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    f32x4 m2[2], ad2[2], pr2[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        m2[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm2 ? vo[k][i][h] : OOB, jj * 64, 0));
                        ad2[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rad, use_add2 ? vo[k][i][h] : OOB, jj * 64, 0));
                        pr2[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, acc2 ? vo[k][i][h] : OOB, jj * 64, 0));
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 o = act4(take(2 * i + h, 2 * k + jj) + bias4[2 * k + jj] + ad2[h]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] *= (m2[h][c] > 0.f || !mm2) ? 1.f : msl;
                        o += pr2[h];
                        track(o, vo[k][i][h] != OOB, du);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[k][i][h], jj * 64, 0);
                    }
                }
            }
        }
    };

    int it = 0, st = 0;
#ifdef GHS_STAMPS
    long long t_mfma = 0, t_epi = 0, t_bar = 0, t_top = 0, tlast_ = clock64(), tall = tlast_;
#endif
    GHS_BARRIER();                                                  // barrier 0
    GHS_T(t_bar)
    for (;;) {
        GHS_T(t_top)
        mfma_item(st, it & 1);
        GHS_T(t_mfma)
        if (cu.g == nitems - 1) epilogue(cu.tile);
        GHS_T(t_epi)
        step(cu);
        if (!cu.ok) break;
        GHS_BARRIER();
        GHS_T(t_bar)
        ++it; st = st == NSTAGE - 1 ? 0 : st + 1;
    }
    if (a.amax_out[0]) pnnp_amax_commit(amx0, a.amax_out[0]);
#ifdef GHS_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (lane == 0) {
        float* d = a.dst[0] + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
        d[0] = (float)t_mfma; d[1] = (float)t_epi; d[2] = (float)t_bar; d[3] = (float)t_top; d[4] = (float)(clock64() - tall); d[5] = (float)(it + 1);
    }
#endif
}

template <int BN, int WN, int EK>
int launch_ghs(const H2Args& ha, hipStream_t s) {
    using Cfg = GCfg<BN, WN>;
    const IgemmArgs& a = ha.g;
    auto kern = gemm_h2s_kernel<BN, WN, EK>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + Cfg::TH - 1) / Cfg::TH) * a.B * ((a.Ntot + BN - 1) / BN);
    if (tiles <= 0) return PNNP_OK;
    const int wgs = pnnp_persistent_grid(tiles);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), Cfg::LDS_BYTES, s, ha);
    return pnnp_launch_status();
}

template <int BN, int WN>
int launch_ghs_ek(const H2Args& ha, hipStream_t s) {
    const IgemmArgs& b = ha.g;
    const bool two = b.dst[1] != nullptr;
    const bool plain = !b.addsrc && !b.accum[0] && !(two && b.accum[1]);
    const bool any_mask = b.mask_mode[0] || (two && b.mask_mode[1]);
    if (plain && !any_mask) return launch_ghs<BN, WN, EK_FWD>(ha, s);
    if (ha.bits_in[0]) {
        // sign-bit masks: one plain destination in the tile domain's own geometry (ConvTranspose2d backward-data), the bit image must fit a buffer resource
        if (!plain || two || b.act || b.bias || !b.mask_mode[0] || b.n_sub || b.out_mul != 1 || b.out_yoff || b.out_xoff || b.OH != b.DH || b.OW != b.DW ||
            ha.bits_nblk[0] * 32 != b.dst_cs[0] || (int64_t)b.B * ((b.OH + 15) / 16) * ((b.OW + 31) / 32) * ha.bits_nblk[0] * 2048 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
        return launch_ghs<BN, WN, EK_BWDB>(ha, s);
    }
    if (plain && !b.act && !b.bias) return launch_ghs<BN, WN, EK_BWD>(ha, s);
    return launch_ghs<BN, WN, EK_GEN>(ha, s);
}

}  // namespace

// `ha.g`: validated like pnnp_gemm_x3_launch's argument (csrc/gemm_x3.hip), with chunks_per_seg = 32-channel items per K segment (every
// segment a multiple of 32 channels) and Ntot a multiple of 64; weights = the kind-6 pack (csrc/pack_jobs.hip), amax slots as in csrc/h2.h.
int pnnp_gemm_h2s_launch(const H2Args& ha, hipStream_t s) {
    const IgemmArgs& b = ha.g;
    if (!ha.amax_in[0] || !ha.amax_w || b.amax_out[1] || ha.bits_out || ha.bits_in[1]) return PNNP_E_INVALID;
    if (b.Ntot % 32 || b.chunks_per_seg <= 0) return PNNP_E_UNSUPPORTED;
    if (b.Ntot % 128 == 0) return launch_ghs_ek<128, 64>(ha, s);
    if (b.Ntot % 64 == 0) return launch_ghs_ek<64, 32>(ha, s);
    return launch_ghs_ek<32, 32>(ha, s);                            // (round 6) 256 px x 32 columns: ResUnet's 32-column layers at 512 x 512 (pool1 backward-data, sc9, upv9 beside them)
}
