// 3x3 convolution (forward / backward-data) on the fp16 matrix cores, float32 operands split into TWO scaled fp16 pieces (csrc/h2.h: the
// scheme, its error, the amax slots) -- the workgroup of csrc/conv_x3s.hip (8 MFMA-only consumer waves, 2 pixel rows x BN channels each, + 4
// producer waves; 16-row x 32-px tiles; persistent, XCD-aware tile order) with HALF the matrix instructions:
//
//   per 16-channel chunk and (16 px x 16 ch) block the nine taps need  hi hi' + lo hi'  (one v_mfma_f32_16x16x32_f16 per tap: pixels
//   [hi | lo] concatenated along K against weights [hi' | hi']) and  hi lo'  (half an instruction per tap: TWO TAPS share one, pixels
//   [hi @ tap t | hi @ tap t'] against [lo' @ t | lo' @ t'] -- the K halves of an instruction are fetched per lane anyway, so the second half
//   simply reads the halo image at the other tap's offset): 9 + 5 = 14 instructions where the bf16x3 kernel issues 27.
//
// Everything a chunk needs is resident at once -- halo image 2 pieces x 2 octets x 624 px x 16 B = 39 KB, weights 18 KB per 32-channel block
// -- so the work item is a whole CHUNK (one s_barrier per chunk instead of three), the weights ring has two chunk stages (LDS-DMA, one chunk
// ahead) and the halo tile two images.  Producers: fp32 NHWC global -> registers (one chunk ahead) -> x 2^se, hi / lo by v_fma_mix*_f16 (4
// vector instructions per pixel pair where the 3-way bf16 split took 11) -> LDS.  Consumers: ds_read_b128 + MFMA only inside the K loop; the
// epilogue works on the accumulators (x 2^-(se_x + se_w) by v_ldexp_f32, bias, activation, act' mask, residual, accumulate) and additionally
//   * tracks max|stored value| per destination (one atomicMax per wave at the end of the launch: the next layer's scale),
//   * forward: writes the SIGN BITS of the activated output, 32 per lane, in a tile-private layout; backward-data reads them back as the
//     LeakyReLU' / ReLU' mask -- 4 bytes per lane and 32-channel block where the float32 activation cost 8 x 16.
#include "h2.h"
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int NCW = 8, NPW = 4, NTHR = 64 * (NCW + NPW);           // consumer / producer waves
constexpr int MT = 2, TH = NCW * MT, HR = TH + 2, HC = 34, NPIX = HR * HC;     // 16-row x 32-px tile, 612 halo pixels
// halo image in 16-byte words: [piece 2][k-octet 2][pixel, plane padded to a multiple of 16 words]; the 12 padding words of the hi planes are
// ZERO (written once; see the prologue)
constexpr int NPIXP = (NPIX + 15) / 16 * 16;                       // 624
constexpr int XS_F4 = 2 * 2 * NPIXP, XS_BYTES = XS_F4 * 16;        // 2496 words, 39936 bytes
#define XS_PLANE(piece, oct) (((piece) * 2 + (oct)) * NPIXP)
constexpr int WPLANE = 9 * 2 * 32 * 16;                            // the hi' piece of one 32-channel block of one chunk: [tap 9][octet 2][32][16 B] = 9216
constexpr int WBLK = 2 * WPLANE + 1024;                            // [hi' 9 taps][lo' 9 taps + one tap of ZEROS: the unpaired ninth tap's partner] = 19456
constexpr int PTHR = 64 * NPW;                                     // producer threads
constexpr int NSLOT = (2 * NPIX + PTHR - 1) / PTHR;                // halo staging slots per producer thread: 1224 (pixel, octet) pairs / 256 -> 5
constexpr unsigned OOB = 0x80000000u;
#define H2S_VMCNT(N) (0x0f70 | ((N) & 15) | (((N) >> 4) << 14))    // s_waitcnt vmcnt(N) alone
#define H2S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")      // (csrc/conv_x3s.hip: why not __syncthreads())

constexpr int HEAD_LDS_FLOATS = 256;                               // (EK_HEAD) [4][32] head weights + [4] biases, padded
template <int BN> struct SCfg {
    static constexpr int NT = BN / 32;
    static constexpr int WS_STAGE = NT * WBLK;                     // 19456 / 38912: one chunk's weights
    static constexpr int NDMA = WS_STAGE / 1024;                   // 1 KB LDS-DMA pieces per stage: 19 / 38
    static constexpr int DPW = (NDMA + NPW - 1) / NPW;             // LDS-DMA instructions per producer wave and chunk: 5 / 10
    static constexpr int NSTAGE = 2;
    static constexpr int BIAS_MAX = 1024;                          // the layer's bias vector lives in LDS: at most this many output channels (the launcher checks)
    static constexpr int LDS_BYTES = 2 * XS_BYTES + NSTAGE * WS_STAGE + (BIAS_MAX + 64) * 4 + (BN == 32 ? HEAD_LDS_FLOATS * 4 : 0);      // 162048 / 124160
    static_assert(LDS_BYTES <= 160 * 1024, "a workgroup's LDS");
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}
// hi = f16(a s), lo = f16(a s - hi) of two values, packed (low half = a0): v_fma_mix*_f16 computes the fma in float32 and rounds ONCE to fp16
// (nearest even); a s is exact (power of two), a s - hi is exact in float32 (the residual of a 24-bit significand after its top 11 bits), so
// both pieces are correctly rounded (tools/ubench/h2_probe.hip: 65536 values bit for bit against the host)
__device__ __forceinline__ void split_h2(float a0, float a1, float s, unsigned& hi, unsigned& lo) {
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(s), "v"(h));
    hi = h; lo = l;
}

#ifndef H2S_STORE_AUX
#define H2S_STORE_AUX 2              // cache-policy bits of the epilogue's full-resolution stores: 2 = nt (csrc/conv_x3s.hip, profiles/r4/ab_store_policy.txt)
#endif
#ifndef H2S_TURNS
#define H2S_TURNS 0                  // 1: the two consumer waves of a SIMD take turns with a tile's epilogue (see the consumers' loop): measured 1-3 % SLOWER, off
#endif
#ifndef H2S_ABL
#define H2S_ABL 0                    // timing ablations of the FWD / BWDB epilogue (results are WRONG with any bit set; tools/scratch builds only):
#endif                               // 1 no line trade, 2 no sign bits, 4 no amax tracking, 8 accumulators not zeroed, 16 no stores, 32 no activation / mask
#ifndef H2S_PREBITS
#define H2S_PREBITS 1                // backward-data with bit masks: the tile's mask words are requested in front of its last chunk (see prefetch_bits)
#endif
#ifndef H2S_NSETS
#define H2S_NSETS 2                  // producer register sets for the halo tile: 1 = a chunk's halo is requested at the END of the period before the one that splits it (round 5),
#endif                               // 2 / 3 = at the START of that period / a period earlier still (profiles/r6/ab_producer_sets.txt)
#ifdef H2S_STAMPS                 // debug build: cycle sums per wave, dumped into dst[0] (tools/x3s_stamps.py)
#define H2S_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define H2S_T(v)
#endif
// the epilogue a kernel carries (one straight-line path each): forward (no mask, no accumulation, no residual; writes sign bits when asked),
// masked backward-data with float32 masks / with bit masks, the general one, forward + MaxPool2d(2)
enum { EK_FWD = 0, EK_BWD = 1, EK_GEN = 2, EK_POOL = 3, EK_BWDB = 4, EK_HEAD = 5, EK_RES = 6 };
// EK_RES (round 6): a plain layer + a residual tensor of the destination's geometry, no activation, no mask (ResUnet: the second convolution of every
// ResidualBlock, forward `conv + bias + shortcut` and backward-data `dgrad + g`, archs/modules.py:176-197) -- the general epilogue took 3.2 x the
// forward epilogue's cycles for them (16-pixel x 64-byte stores, three loads per block: profiles/r6/gen_epilogue_stamps.txt).  Here: the residual
// words are requested up front in the full-line pattern the stores use and added BEHIND the line trade; scale + bias in one fma (bias from LDS).
// ((v 2^dexp + bias) + res in this order, as the general epilogue computes it: bit-identical.)
// EK_HEAD (32-column kernel only): the forward epilogue + the network's 1x1 head (archs/Unet.py:94: conv10_1, 32 -> 4 channels, no activation) computed from
// the activated accumulators -- a lane holds 8 of a pixel's 32 channels, the 4 lanes of a pixel add their partial sums through two butterfly exchanges --
// and written as the NCHW output planes (+ the `res` networks' input residual).  The 32-channel map itself is stored only when the caller asks for it
// (a training forward: backward needs it); an eval forward never writes or re-reads it.


// The 14 matrix instructions of a (chunk, 16 x 16 block): kind 0 = tap t0, pixels [hi | lo] x weights [hi' | hi'];  kind 1 / 2 = taps (t0, t0 + 1),
// pixels [hi @ t0 | hi @ t0 + 1] x weights [lo' @ t0 | lo' @ t0 + 1] (the second tap's halo offset is + 1 pixel, or + 32 from tap 2 to tap 3);
// tap 8 has no partner: its pair is (8, "9"), where the pack holds a tenth lo' tap of zeros (the pixel half reads one pixel on: finite x 0).
constexpr int NGRP = 14;
__device__ constexpr int grp_t0(int g) { constexpr int t[NGRP] = {0, 1, 0, 2, 3, 2, 4, 5, 4, 6, 7, 6, 8, 8}; return t[g]; }
__device__ constexpr int grp_kind(int g) { constexpr int k[NGRP] = {0, 0, 1, 0, 0, 2, 0, 0, 1, 0, 0, 1, 0, 1}; return k[g]; }

template <int BN, int EK>
__global__ void __launch_bounds__(NTHR, 1)
igemm_h2s_kernel(const H2Args ha) {
    const IgemmArgs& a = ha.g;
    constexpr bool POOL = EK == EK_POOL;
    using Cfg = SCfg<BN>;
    constexpr int NT = Cfg::NT, D = Cfg::DPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                     // two halo images
    char* wsb = smem + 2 * XS_BYTES;                                // the weight ring: two chunk stages
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * XS_BYTES + Cfg::NSTAGE * Cfg::WS_STAGE);      // bias[0 .. Ntot) (zeros without a bias)
    float* head_lds = bias_lds + Cfg::BIAS_MAX + 64;                // (EK_HEAD) the head's weights [4][32] and biases [4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 7 consumers, 8 .. 11 producers

    // ---- scales (csrc/h2.h): se_x from the largest amax of the K segments, se_w from the weight tensor's
    unsigned ax = ha.amax_in[0] ? ha.amax_in[0][0] : 0u;
    if (ha.amax_in[1]) { const unsigned a2 = ha.amax_in[1][0]; ax = a2 > ax ? a2 : ax; }
    const int se_x = __builtin_amdgcn_readfirstlane(pnnp_h2_scale_exp(ax));
    const int se_w = __builtin_amdgcn_readfirstlane(ha.amax_w ? pnnp_h2_scale_exp(ha.amax_w[0]) : 0);

    // ---- the workgroup's tiles t, t + G, ...: decoded once, then stepped by mixed-radix addition (both roles walk the same sequence)
    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    // split-K (general epilogue only; ha.ksplit = S > 1): S workgroups share an output tile, slice ks owning chunks [ks, ks + 1) nchunks / S of K and
    // writing its raw partial sums into image ks B + b of a [S][B][OH][OW][cs] slab tensor; h2_splitk_reduce_kernel adds the slabs in a fixed order
    constexpr bool SPLITK = EK == EK_GEN;
    const int KSPL = SPLITK ? ha.ksplit : 1;
    const int total = tiles_x * tiles_y * a.B * n_tiles * KSPL;
    const int G = gridDim.x;
    const int nchunks = a.nseg * a.chunks_per_seg / KSPL;           // 16-channel chunks of K a workgroup walks per tile (its slice)
    struct Tile { int b, y0, x0, n0, ks; };
    auto decode = [&](int t) {
        Tile o;
        const int nt_i = t % n_tiles;
        int m_i = t / n_tiles;
        const int tx = m_i % tiles_x; m_i /= tiles_x;
        o.x0 = tx * 32; o.y0 = (m_i % tiles_y) * TH; o.b = m_i / tiles_y; o.n0 = nt_i * BN; o.ks = 0;
        if constexpr (SPLITK) { o.ks = o.b / a.B; o.b -= o.ks * a.B; }
        return o;
    };
    auto pick = [](bool c, const Tile& x, const Tile& y) {
        Tile o; o.b = c ? x.b : y.b; o.y0 = c ? x.y0 : y.y0; o.x0 = c ? x.x0 : y.x0; o.n0 = c ? x.n0 : y.n0; o.ks = 0;
        if constexpr (SPLITK) o.ks = c ? x.ks : y.ks;
        return o;
    };
    const Tile gstep = decode(G);
    auto advance = [&](Tile o) {
        o.n0 += gstep.n0; if (o.n0 >= n_tiles * BN) { o.n0 -= n_tiles * BN; o.x0 += 32; }
        o.x0 += gstep.x0; if (o.x0 >= tiles_x * 32) { o.x0 -= tiles_x * 32; o.y0 += TH; }
        o.y0 += gstep.y0; if (o.y0 >= tiles_y * TH) { o.y0 -= tiles_y * TH; o.b += 1; }
        o.b += gstep.b;
        if constexpr (SPLITK) { if (o.b >= a.B) { o.b -= a.B; o.ks += 1; } o.ks += gstep.ks; }
        return o;
    };
    int t = xcd_remap(blockIdx.x, G);
    if (t >= total) return;
    // lookahead of LA tiles: the producers request halo chunks up to max(2, H2S_NSETS) chunks ahead, which is that many TILES ahead for a one-chunk layer
    constexpr int NS = H2S_NSETS, LA = NS > 2 ? NS : 2;
    static_assert(NS >= 1 && NS <= 3, "producer register sets");
    Tile cur = decode(t), ahead[LA];
    ahead[0] = pick(t + G < total, advance(cur), cur);
#pragma unroll
    for (int i = 1; i < LA; ++i) ahead[i] = pick(t + (i + 1) * G < total, advance(ahead[i - 1]), ahead[i - 1]);
    int g = 0;                                                       // chunk of the current tile
    // the k-th chunk after the current one, k = 1 .. LA: (tile, chunk, exists); past the end of this workgroup's work it falls back to the
    // current chunk (requests stay branch-free; weights are then requested with valid = false)
    struct Ck { Tile tile; int g; bool ok; };
    auto chunk_at = [&](auto ktag) {
        constexpr int k = decltype(ktag)::value;
        static_assert(k >= 1 && k <= LA, "lookahead");
        int gk = g + k, hop = 0;
#pragma unroll
        for (int i = 0; i < k; ++i)
            if (gk >= nchunks) { gk -= nchunks; ++hop; }
        Ck c;
        c.ok = t + hop * G < total;
        c.g = c.ok ? gk : g;
        Tile far = ahead[0];
#pragma unroll
        for (int i = 1; i < k; ++i) far = pick(hop > i, ahead[i], far);
        c.tile = pick(!c.ok || hop == 0, cur, far);
        return c;
    };
    auto next_tile = [&]() {
        t += G; cur = ahead[0];
#pragma unroll
        for (int i = 0; i + 1 < LA; ++i) ahead[i] = ahead[i + 1];
        ahead[LA - 1] = pick(t + LA * G < total, advance(ahead[LA - 2]), ahead[LA - 2]);
        g = 0;
    };
    using K1 = std::integral_constant<int, 1>;

    if (wave >= NCW) {
        // =============================================== PRODUCER ===============================================
        const int pw = wave - NCW, ptid = tid - 64 * NCW;            // 0 .. 3, 0 .. 255
        const float sx = __uint_as_float((unsigned)(se_x + 127) << 23);      // 2^se_x
        // staging slots: s = ptid + 256 k -> (pixel s >> 1, channel octet s & 1); a slot past the end repeats the previous one of the thread
        int rk[NSLOT], qk[NSLOT]; unsigned pixk[NSLOT]; int xdst[NSLOT];
        const int oct = ptid & 1;
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) {
            int s = ptid + PTHR * k;
            if (s >= 2 * NPIX) s -= PTHR;
            const int pix = s >> 1;
            const int r = pix / HC, q = pix - r * HC;
            rk[k] = r - 1; qk[k] = q - 1;
            pixk[k] = (unsigned)(r * a.IW + q);
            xdst[k] = XS_PLANE(0, oct) + pix;                       // + 2 NPIXP for the lo plane (+ image * XS_F4)
        }
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
        f32x4 ra[NS][NSLOT][2];                                     // halo chunks in flight / waiting to be split, 8 channels per slot
        // global loads of the halo tile of (tile, chunk gq) -> register set S: hardware zero for pixels outside the image and channels past the segment
        auto load_halo = [&](auto stag, const Tile& tl, int gq) {
            constexpr int S = decltype(stag)::value;
            if constexpr (SPLITK) gq += tl.ks * nchunks;            // (this slice's chunks of K)
            const int si = gq / a.chunks_per_seg, cc = gq - si * a.chunks_per_seg;
            const IgemmSeg sg = a.seg[si];
            const int c0 = sg.coff + cc * 16;
            const int rlo = -tl.y0, rhi = a.IH - tl.y0, qlo = -tl.x0, qhi = a.IW - tl.x0;
            const int shift = (2 * a.IW + 2) * sg.cstride;         // the resource starts before the image: the scalar offset below stays >= 0
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.ptr + ((int64_t)tl.b * a.IH * a.IW * sg.cstride - shift)), 0, 0x7fffffff, 0x00020000);
            const int soff = (((tl.y0 - 1) * a.IW + tl.x0 - 1) * sg.cstride + c0 + shift) * 4;
            const unsigned cs4 = (unsigned)sg.cstride * 4u;
            const int cvalid = a.seg_channels - cc * 16 - oct * 8;  // > 0: this thread's octet exists
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                const int bad = (rk[k] - rlo) | (rhi - 1 - rk[k]) | (qk[k] - qlo) | (qhi - 1 - qk[k]) | (cvalid - 1);     // sign bit set <=> outside
                const unsigned vo = bad < 0 ? OOB : __umul24(pixk[k], cs4) + oct * 32;
                ra[S][k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
                ra[S][k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff + 16, 0));
            }
        };
        // register set S -> the two 16-byte words (8 channels of hi, of lo) per slot in halo image img
        auto stage = [&](auto stag, int img) {
            constexpr int S = decltype(stag)::value;
#pragma unroll
            for (int k = 0; k < NSLOT; ++k) {
                u32x4 sh, sl;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x4 v = ra[S][k][p >> 1];
                    unsigned h, l;
                    split_h2(v[(p & 1) * 2], v[(p & 1) * 2 + 1], sx, h, l);
                    sh[p] = h; sl[p] = l;
                }
                u32x4* d = xs + img * XS_F4 + xdst[k];
                d[0] = sh; d[2 * NPIXP] = sl;
            }
        };
        // LDS-DMA of the weights of (tile n0, chunk gq) into stage st: per 32-channel block 18432 contiguous bytes of the pack, as 1 KB pieces
        // dealt over the 4 producer waves; past the end a wave repeats the last piece (same bytes, same place)
        const int K16 = a.nseg * a.chunks_per_seg;
        auto dma_weights = [&](const Tile& tl, int gq, int st, bool valid) {
            if constexpr (SPLITK) gq += tl.ks * nchunks;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int ins = min(pw + NPW * i, Cfg::NDMA - 1);
                const int j = ins / 19, r = ins - 19 * j;
                const int nb = (tl.n0 >> 5) + j;
                const bool ok = valid && nb * 32 < a.Ntot;        // (an invalid request still issues: the vmcnt counts below count instructions)
                const int soff = ok ? ((nb * K16 + gq) * WBLK + r * 1024) : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wsb + st * Cfg::WS_STAGE + ins * 1024),
                                                         16, ok ? (unsigned)lane * 16u : OOB, soff, 0, 0);
            }
        };
        // ---- prologue: the bias vector, the padding words of the hi planes (the unpaired ninth tap's second half reads one pixel past tap 8: word 612
        // for the last lane of the last row -- multiplied by the pack's zero tap, so it must be FINITE, not whatever bit pattern the LDS held), the weights
        // of chunk 0, chunk 0's halo straight into image 0, the halos of chunks 1 .. max(1, NS - 1) into the register sets
        for (int i = ptid; i < Cfg::BIAS_MAX + 64; i += PTHR) bias_lds[i] = (a.bias && i < a.Ntot) ? a.bias[i] : 0.f;
        if constexpr (EK == EK_HEAD) {
            if (ptid < 132) head_lds[ptid] = ptid < 128 ? ha.head_w[ptid] : (ha.head_b ? ha.head_b[ptid - 128] : 0.f);
        }
        if (ptid < 48) xs[(ptid / 24) * XS_F4 + XS_PLANE(0, (ptid / 12) & 1) + NPIX + ptid % 12] = u32x4{0u, 0u, 0u, 0u};
        dma_weights(cur, 0, 0, true);
        load_halo(std::integral_constant<int, 0>{}, cur, 0);
        stage(std::integral_constant<int, 0>{}, 0);
        static_for<1, (NS > 1 ? NS : 2)>([&](auto jt) {             // (past the end: the current chunk again, harmless)
            constexpr int j = decltype(jt)::value;
            const Ck nj = chunk_at(jt);
            load_halo(std::integral_constant<int, j % NS>{}, nj.tile, nj.g);
        });
        __builtin_amdgcn_s_waitcnt(H2S_VMCNT(2 * NSLOT * (NS > 1 ? NS - 1 : 1)));      // the weights; the halos stay in flight
        H2S_BARRIER();                                            // barrier 0: chunk 0 may start
        int img = 0, st = 0;                                        // image / weight stage of the current chunk
#ifdef H2S_STAMPS
        long long t_work = 0, t_wait = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int nch = 0;
#endif
        // One period = the consumers run chunk c: the weights of chunk c + 1 into the other stage (the consumers left it at the last barrier);
        //   NS = 1: the halo of chunk c + 1 (requested at the end of period c - 1) is split into the other image, then chunk c + 2's is requested;
        //   NS > 1: chunk c + NS's halo is requested FIRST, into the register set chunk c was split out of a period ago, then chunk c + 1's
        //           (set (c + 1) % NS, in flight for NS - 1 whole periods) is split.
        // In front of the barrier the weights must have landed: vmcnt(what was issued behind them).  The loop is unrolled NS times (P = c % NS).
        auto period = [&](auto ptag) {
            constexpr int P = decltype(ptag)::value;
            const Ck n1 = chunk_at(K1{}), nl = chunk_at(std::integral_constant<int, (NS > 1 ? NS : 2)>{});
#ifdef H2S_STAMPS
            ++nch;
#endif
            dma_weights(n1.tile, n1.g, st ^ 1, n1.ok);
            if constexpr (NS == 1) {
                stage(std::integral_constant<int, 0>{}, img ^ 1);
                load_halo(std::integral_constant<int, 0>{}, nl.tile, nl.g);
            } else {
                load_halo(ptag, nl.tile, nl.g);
                stage(std::integral_constant<int, (P + 1) % NS>{}, img ^ 1);
            }
            H2S_T(t_work)
            __builtin_amdgcn_s_waitcnt(H2S_VMCNT(2 * NSLOT));
            H2S_T(t_wait)
            if (!n1.ok) return false;                               // (the consumers' epilogue and exit need no barrier)
            H2S_BARRIER();
            H2S_T(t_bar)
            if (g == nchunks - 1) next_tile(); else ++g;
            img ^= 1; st ^= 1;
            return true;
        };
        for (;;) {
            if (!period(std::integral_constant<int, 0>{})) break;
            if constexpr (NS > 1) { if (!period(std::integral_constant<int, 1>{})) break; }
            if constexpr (NS > 2) { if (!period(std::integral_constant<int, 2>{})) break; }
        }
#ifdef H2S_STAMPS
        if (lane == 0) {
            float* d = a.dst[0] + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
            d[0] = (float)t_work; d[1] = (float)t_wait; d[2] = (float)t_bar; d[3] = 0.f; d[4] = (float)(clock64() - tall); d[5] = (float)nch;
        }
#endif
        return;
    }

    // =============================================== CONSUMER ===============================================
    // 16 x 16 accumulator blocks, WEIGHTS as the instruction's first operand: acc[2 i + h][j] = pixel row i of the wave, 16-pixel half h,
    // channels 16 j .. 16 j + 15; lane l holds channels 4 (l >> 4) .. + 3 of pixel l & 15
    constexpr int MB = 2 * MT, NB = BN / 16;
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // MFMAs of one chunk (halo image img, weight stage st): group by group (grp_*), per group pass j (16 output channels) x pixel block mb.
    // The 4 pixel words of the NEXT group and the weight word of the NEXT pass are read between the MFMAs into the other register set.
    auto mfma_chunk = [&](int st, int img) {
        // lane bases (16-byte words of an image / bytes of a weight block); k-block q16 of an instruction = (octet q16 & 1, K half q16 >> 1).
        // Derived from an OPAQUE copy of the lane number per chunk: five registers that would otherwise live (spilled, in the 64-column forward
        // kernel) across the epilogue
        int lane_m = lane;
        asm volatile("" : "+v"(lane_m));
        const int r16 = lane_m & 15, q16 = lane_m >> 4, oct16 = q16 & 1, ps16 = q16 >> 1;
        const int rowbase = wave * MT * HC + r16;
        const int a_main = XS_PLANE(ps16, oct16) + rowbase;          // [hi | lo]
        const int a_hi = XS_PLANE(0, oct16) + rowbase;
        const int a_p1 = a_hi + ps16, a_p32 = a_hi + 32 * ps16;      // [hi @ t | hi @ t + 1]: the second tap is one pixel (taps 2 -> 3: 32 pixels) on
        const int b_main = (oct16 * 32 + r16) * 16, b_p1 = b_main + ps16 * 1024;
        const char* wst = wsb + st * Cfg::WS_STAGE;
        const u32x4* xim = xs + img * XS_F4;
        u32x4 A[2][MB], Bv[2];
        auto a_read = [&](auto gtag, int mb) {
            constexpr int gg = decltype(gtag)::value, t0 = grp_t0(gg), kind = grp_kind(gg);
            const int off = ((mb >> 1) + t0 / 3) * HC + t0 % 3 + 16 * (mb & 1);
            A[gg & 1][mb] = xim[(kind == 0 ? a_main : (kind == 1 ? a_p1 : a_p32)) + off];
        };
        auto b_read = [&](auto ptag) {
            constexpr int pp = decltype(ptag)::value, gg = pp / NB, j = pp % NB, t0 = grp_t0(gg), kind = grp_kind(gg);
            const int base = kind == 0 ? b_main : b_p1;
            Bv[pp & 1] = *reinterpret_cast<const u32x4*>(wst + base + (kind == 0 ? 0 : WPLANE) + (j >> 1) * WBLK + t0 * 1024 + (j & 1) * 256);
        };
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a_read(std::integral_constant<int, 0>{}, mb);
        b_read(std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
        constexpr int GM = NB * MB;                                  // MFMAs per group
        static_for<0, NGRP * GM>([&](auto Gc) {
            constexpr int gi = decltype(Gc)::value;
            constexpr int gg = gi / GM, m = gi % GM, j = m / MB, mb = m % MB, pass = gg * NB + j;
            acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, Bv[pass & 1]), __builtin_bit_cast(f16x8, A[gg & 1][mb]), acc[mb][j], 0, 0, 0);
            if constexpr (mb == 0 && pass + 1 < NGRP * NB) b_read(std::integral_constant<int, pass + 1>{});
            if constexpr (gg + 1 < NGRP) {
                // the next group's pixel words: NB = 4: one per pass (behind its second MFMA); NB = 2: two per pass
                if constexpr (NB >= MB) { if constexpr (mb == 1 && j < MB) a_read(std::integral_constant<int, gg + 1>{}, j); }
                else { if constexpr (mb == 1 || mb == 2) a_read(std::integral_constant<int, gg + 1>{}, j * 2 + mb - 1); }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- the epilogue's kernel arguments, cached in ONE vector register (lane i = argument i) and fetched with v_readlane (csrc/conv_x3s.hip:
    // the unrolled K loop leaves the compiler no scalar registers for them).  The layer's bias vector sits in LDS for the same reason.
    enum { E_OH, E_OW, E_DH, E_DW, E_NTOT, E_NSPLIT, E_ACT, E_POOLCS, E_CS0, E_CS1, E_MM0, E_MM1, E_AC0, E_AC1, E_DEXP, E_NBLK0, E_NBLK1,
           E_DST0, E_DST1 = E_DST0 + 2, E_MASK0 = E_DST1 + 2, E_MASK1 = E_MASK0 + 2, E_ADD = E_MASK1 + 2, E_PDST = E_ADD + 2,
           E_PCODE = E_PDST + 2, E_BOUT = E_PCODE + 2, E_BIN0 = E_BOUT + 2, E_BIN1 = E_BIN0 + 2, E_HOUT = E_BIN1 + 2, E_HRES = E_HOUT + 2, E_COUNT = E_HRES + 2 };
    static_assert(E_COUNT <= 64, "one lane per cached argument");
    unsigned argv = 0;
    {
        auto put = [&](int idx, unsigned v) { argv = lane == idx ? v : argv; };
        auto putp = [&](int idx, const void* q) { put(idx, (unsigned)(uintptr_t)q); put(idx + 1, (unsigned)((uintptr_t)q >> 32)); };
        put(E_OH, a.OH); put(E_OW, a.OW); put(E_DH, a.DH); put(E_DW, a.DW); put(E_NTOT, a.Ntot); put(E_NSPLIT, a.n_split); put(E_ACT, a.act);
        put(E_POOLCS, a.pool_cs); put(E_CS0, a.dst_cs[0]); put(E_CS1, a.dst_cs[1]); put(E_MM0, a.mask_mode[0]); put(E_MM1, a.mask_mode[1]);
        put(E_AC0, a.accum[0]); put(E_AC1, a.accum[1]); put(E_DEXP, (unsigned)(-(se_x + se_w))); put(E_NBLK0, ha.bits_nblk[0]); put(E_NBLK1, ha.bits_nblk[1]);
        putp(E_DST0, a.dst[0]); putp(E_DST1, a.dst[1]); putp(E_MASK0, a.mask[0]); putp(E_MASK1, a.mask[1]); putp(E_ADD, a.addsrc);
        putp(E_PDST, a.pool_dst); putp(E_PCODE, a.pool_codes); putp(E_BOUT, ha.bits_out); putp(E_BIN0, ha.bits_in[0]); putp(E_BIN1, ha.bits_in[1]);
        putp(E_HOUT, ha.head_out); putp(E_HRES, ha.head_res);
    }
    struct EpiArgs {
        int OH, OW, DH, DW, Ntot, n_split, act, pool_cs, cs0, cs1, mm0, mm1, ac0, ac1, dexp, nblk0, nblk1;
        float *dst0, *dst1, *pool_dst; const float *mask0, *mask1, *addsrc; unsigned char* pool_codes;
        unsigned* bits_out; const unsigned *bin0, *bin1;
        float* head_out; const float* head_res;
        __device__ int dst_cs(int du) const { return du ? cs1 : cs0; }
        __device__ int mask_mode(int du) const { return du ? mm1 : mm0; }
        __device__ int accum(int du) const { return du ? ac1 : ac0; }
        __device__ float* dst(int du) const { return du ? dst1 : dst0; }
        __device__ const float* mask(int du) const { return du ? mask1 : mask0; }
        __device__ const unsigned* bits_in(int du) const { return du ? bin1 : bin0; }
        __device__ int nblk(int du) const { return du ? nblk1 : nblk0; }
    };
    auto epi_args = [&]() {
        auto rl = [&](int idx) { return (int)__builtin_amdgcn_readlane((int)argv, idx); };
        auto rp = [&](int idx) { return (uintptr_t)(unsigned)rl(idx) | ((uintptr_t)(unsigned)rl(idx + 1) << 32); };
        EpiArgs e;
        e.OH = rl(E_OH); e.OW = rl(E_OW); e.DH = rl(E_DH); e.DW = rl(E_DW); e.Ntot = rl(E_NTOT); e.n_split = rl(E_NSPLIT); e.act = rl(E_ACT);
        e.pool_cs = rl(E_POOLCS); e.cs0 = rl(E_CS0); e.cs1 = rl(E_CS1); e.mm0 = rl(E_MM0); e.mm1 = rl(E_MM1); e.ac0 = rl(E_AC0); e.ac1 = rl(E_AC1);
        e.dexp = rl(E_DEXP); e.nblk0 = rl(E_NBLK0); e.nblk1 = rl(E_NBLK1);
        e.dst0 = (float*)rp(E_DST0); e.dst1 = (float*)rp(E_DST1); e.mask0 = (const float*)rp(E_MASK0); e.mask1 = (const float*)rp(E_MASK1);
        e.addsrc = (const float*)rp(E_ADD); e.pool_dst = (float*)rp(E_PDST); e.pool_codes = (unsigned char*)rp(E_PCODE);
        e.bits_out = (unsigned*)rp(E_BOUT); e.bin0 = (const unsigned*)rp(E_BIN0); e.bin1 = (const unsigned*)rp(E_BIN1);
        if constexpr (EK == EK_HEAD) { e.head_out = (float*)rp(E_HOUT); e.head_res = (const float*)rp(E_HRES); }
        return e;
    };
    float amx0 = 0.f, amx1 = 0.f;                                    // max |stored value| of this lane, per destination
    f32x4 hw[EK == EK_HEAD ? 8 : 1];                                 // (EK_HEAD) head weights of this lane's 8 channels: [output o][16-column block jj] (loaded behind barrier 0)

    // (EK_BWDB) the tile's mask words, requested in FRONT of its last chunk (H2S_PREBITS, round 6): requested at the start of the epilogue, the first
    // mask_scale waited a memory latency for them with the matrix pipe idle -- per tile, ~2 000 cycles of a 13 000 ... 40 000-cycle tile on the shallow layers
    unsigned mbits_pre[EK == EK_BWDB ? NT : 1];
    auto prefetch_bits = [&](const Tile& tl) __attribute__((always_inline)) {
        const EpiArgs ea = epi_args();
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
        const int tile_id = (tl.b * tiles_y + tl.y0 / TH) * tiles_x + (tl.x0 >> 5);
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(tl.n0 + k * 32);
            const int du = nwv >= ea.n_split ? 1 : 0, chw = nwv - (du ? ea.n_split : 0), nblk = ea.nblk(du);
            const unsigned* bp = ea.bits_in(du);
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)(bp ? bp : ea.bin0), 0, a.B * tiles_y * tiles_x * nblk * NCW * 64 * 4, 0x00020000);
            const unsigned off = (unsigned)((((tile_id * nblk + (chw >> 5)) * NCW + wave) * 64 + lane_p) * 4);
            mbits_pre[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, (ea.mask_mode(du) && bp && nwv < ea.Ntot) ? off : OOB, 0, 0);
        }
    };

    // ---- epilogue of tile `tl`, straight from the accumulators: x 2^dexp (undo the operand scales), bias, activation, act' mask, residual and
    // accumulation are float4 arithmetic on the accumulator registers; stores cover whole 128-byte lines (csrc/conv_x3s.hip).  The fused
    // MaxPool2d(2) takes the other pixel of a pair from the neighbouring lane (DPP) and the other row from the wave's second accumulator row.
    auto epilogue = [&](const Tile& tl) __attribute__((always_inline)) {
        const EpiArgs ea = epi_args();
        const int b = SPLITK ? tl.b + tl.ks * a.B : tl.b, n0 = tl.n0;     // (split-K: slab image ks B + b; the launcher allows no mask / residual / bits there)
        // the lane number as an OPAQUE value: everything the epilogue derives from it is then computed here, per tile (a dozen instructions),
        // instead of being hoisted in front of the K loop and carried through it in ~50 registers (19 of them spilled to scratch in the
        // 64-column forward kernel: cycle stamps 14 600 cycles per forward tile against 10 000 for the spill-free backward-data kernel)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int p16 = lane_e & 15, c4 = (lane_e >> 4) * 4;
        const int py0 = tl.y0 + wave * MT, px0 = tl.x0 + p16;
        int du_[NT], chw_[NT], cs_[NT]; bool blk_[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
            du_[k] = nwv >= ea.n_split ? 1 : 0; chw_[k] = nwv - (du_[k] ? ea.n_split : 0); cs_[k] = ea.dst_cs(du_[k]); blk_[k] = nwv < ea.Ntot;
        }
        // is this lane's own pixel (row i, 16-pixel half h) inside the map?  (lane masks: scalar registers)
        bool okp[MT][2];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) okp[i][h] = py0 + i < ea.DH && px0 + 16 * h < ea.DW;
        // (general epilogue) byte offset of that pixel and the lane's channel quad in the destination of 32-column block k, or out of range;
        // the 16-column block inside it (+ 64 bytes) goes through the instruction's scalar offset
        unsigned vo[EK == EK_GEN ? NT : 1][MT][2];
        if constexpr (EK == EK_GEN) {
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        vo[k][i][h] = (blk_[k] && okp[i][h]) ? (unsigned)((((py0 + i) * ea.OW + px0 + 16 * h) * cs_[k] + chw_[k] + c4) * 4) : OOB;
        }
        auto rsrc = [&](const float* base, int k) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)b * ea.OH * ea.OW * cs_[k]), 0, ea.OH * ea.OW * cs_[k] * 4, 0x00020000);
        };
        // this lane's word of the tile-private bit layout (csrc/h2.h) for 32-column block k of a tensor with nblk channel blocks, in bytes
        const int tile_id = (b * tiles_y + tl.y0 / TH) * tiles_x + (tl.x0 >> 5);
        auto bits_off = [&](int k, int nblk) { return (unsigned)((((tile_id * nblk + (chw_[k] >> 5)) * NCW + wave) * 64 + lane_e) * 4); };
        auto bits_rsrc = [&](const unsigned* base, int nblk) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, a.B * tiles_y * tiles_x * nblk * NCW * 64 * 4, 0x00020000);
        };
        const float aslope = ea.act == 1 ? 0.2f : (ea.act == 2 ? 0.f : 1.f);
        f32x4 bias4[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bias4[j] = *reinterpret_cast<const f32x4*>(bias_lds + n0 + 16 * j + c4);      // (columns past Ntot: zeros, and their stores are dropped anyway)
        }
        auto act4 = [&](f32x4 o) {                                   // LeakyReLU(0.2) / ReLU / none as max(o, slope * o)
            const f32x4 t = o * aslope;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], t[c]);
            return o;
        };
        // Undoing the operand scales: x 2^dexp (exact).  As ONE multiplier (fused with the bias add where there is one) while 2^dexp is a normal
        // float32 with room for the LeakyReLU slope; tensors so small / large that it is not (|dexp| > 120: max |x| max |w| beyond 2^+-92) first take the remainder in a pass over the
        // accumulators -- a wave-uniform branch that the networks' tensors never take (tests/test_gpu_h2.py::test_h2_dynamic_range does).
        const int dexp_c = ea.dexp < -120 ? -120 : (ea.dexp > 120 ? 120 : ea.dexp);      // (|.| <= 120: 0.2 x 2^dexp_c stays a normal float32, see mask_scale)
        const float dsc = __uint_as_float((unsigned)(dexp_c + 127) << 23);
        if (ea.dexp != dexp_c) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[mb][j][c] = __builtin_ldexpf(acc[mb][j][c], ea.dexp - dexp_c);
        }
        auto take = [&](int mb, int j) {                             // the block, un-scaled, and the accumulator zeroed for the next tile
            const f32x4 v = acc[mb][j];
            if constexpr (!(H2S_ABL & 8)) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return v * dsc;
        };
        auto take_bias = [&](int mb, int j, f32x4 bias) {            // ... with the bias: one fma per element
            const f32x4 v = acc[mb][j];
            if constexpr (!(H2S_ABL & 8)) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return f32x4{__builtin_fmaf(v.x, dsc, bias.x), __builtin_fmaf(v.y, dsc, bias.y), __builtin_fmaf(v.z, dsc, bias.z), __builtin_fmaf(v.w, dsc, bias.w)};
        };
        // max |.| of the blocks of one destination: a running maximum per 32-column block (two v_max3_f32 per float4), merged into the lane's
        // maximum of that destination once per block.  Lanes whose store is dropped (pixels / columns outside the tensor) count too: what
        // they hold are finite sums over zero padding, and an amax slot may over-estimate (csrc/h2.h) -- masking them was a compare and
        // two selects per float4.
        float amk = 0.f;
        auto track = [&](f32x4 o) {
            if constexpr (H2S_ABL & 4) return;
            amk = fmaxf(fmaxf(amk, fabsf(o.x)), fabsf(o.y));
            amk = fmaxf(fmaxf(amk, fabsf(o.z)), fabsf(o.w));
        };
        auto track_done = [&](int du) {
            if (du) amx1 = fmaxf(amx1, amk); else amx0 = fmaxf(amx0, amk);
            amk = 0.f;
        };
        // Sign bits: element n = ((i 2 + h) 2 + jj) 4 + c of a 32-column block sits at bit 31 - n of the lane's word -- the order in which the
        // forward epilogue produces the values, so that it can SHIFT them in (sb = 2 sb + (o > 0): a compare and an add-with-carry per element) and
        // the backward epilogue can shift them out (carry of sb + sb).  signs4: the generic form (pool and tests).
        auto signs4 = [&](f32x4 o, int pos) {
            return ((o.x > 0.f ? 8u : 0u) | (o.y > 0.f ? 4u : 0u) | (o.z > 0.f ? 2u : 0u) | (o.w > 0.f ? 1u : 0u)) << (28 - pos);
        };
        // one element of the forward epilogue: shift (o > 0) into sb; with an activation o = (o > 0) ? o : slope o on the same compare
        auto act_sign = [](float& o, unsigned& sb, float slope_, auto act_tag) __attribute__((always_inline)) {
            if constexpr (H2S_ABL & 32) return;
            if constexpr (H2S_ABL & 2) { if constexpr (decltype(act_tag)::value) o = fmaxf(o, o * slope_); return; }
            unsigned long long cout_;
            if constexpr (decltype(act_tag)::value) {
                float t;
                asm("v_cmp_lt_f32 vcc, 0, %0\n\tv_addc_co_u32 %1, %2, %1, %1, vcc\n\tv_mul_f32 %3, %4, %0\n\tv_cndmask_b32 %0, %3, %0, vcc"
                    : "+v"(o), "+v"(sb), "=s"(cout_), "=&v"(t) : "v"(slope_) : "vcc");
            } else {
                asm("v_cmp_lt_f32 vcc, 0, %2\n\tv_addc_co_u32 %0, %1, %0, %0, vcc" : "+v"(sb), "=s"(cout_) : "v"(o) : "vcc");
            }
        };
        // one element of the bit-masked backward epilogue, scale included: the next bit of mb out (carry of mb + mb), o = v x (bit ? 2^dexp : msl 2^dexp) --
        // three instructions where scaling first and masking afterwards took four; bit-identical: 2^dexp is a power of two, so (v 2^dexp) msl == v (2^dexp msl)
        auto mask_scale = [](float v, unsigned& mb_, float fpos, float fneg) __attribute__((always_inline)) {
            float f, o;
            asm("v_add_co_u32 %2, vcc, %2, %2\n\tv_cndmask_b32 %1, %4, %3, vcc\n\tv_mul_f32 %0, %1, %5" : "=v"(o), "=&v"(f), "+v"(mb_) : "v"(fpos), "v"(fneg), "v"(v) : "vcc");
            return o;
        };
        auto take_raw = [&](int mb, int j) {                         // the accumulator block as it is (the caller scales), zeroed for the next tile
            const f32x4 v = acc[mb][j];
            if constexpr (!(H2S_ABL & 8)) acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            return v;
        };
        // ---- full-line memory pattern (FWD / BWD / POOL: see csrc/conv_x3s.hip)
        const bool lo8 = p16 < 8;
        auto ror8 = [&](f32x4 v) {                               // (inline assembly: see the pool path about __builtin_amdgcn_update_dpp)
            float r0, r1, r2, r3;
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            return f32x4{r0, r1, r2, r3};
        };
        auto sel = [&](bool c, f32x4 x, f32x4 y) { return f32x4{c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w}; };
        // The trade of the two 16-column blocks of a pair IN PLACE: afterwards o0 is what store instruction 1 writes (lanes p < 8: their own lower
        // quad, lanes p >= 8: the upper quad of pixel p - 8) and o1 what instruction 2 writes.  A DPP row rotation by 8 whose bank mask enables
        // only the receiving half of each 16-lane row: two moves per register pair + one copy (the select-rotate-select form took four + four).
        auto trade = [&](f32x4& o0, f32x4& o1) __attribute__((always_inline)) {
            float a0 = o0.x, a1 = o0.y, a2 = o0.z, a3 = o0.w, b0 = o1.x, b1 = o1.y, b2 = o1.z, b3 = o1.w;
            const float t0 = b0, t1 = b1, t2 = b2, t3 = b3;
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
                         "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0x3"
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
                         "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
            o0 = f32x4{a0, a1, a2, a3}; o1 = f32x4{b0, b1, b2, b3};
        };
        // this lane's byte offset in instruction 1 of block k: pixel (row i, half h, p16 & 7), quad q16 of the lower / upper 16 columns
        unsigned wo[NT][MT][2];
        const int pxl = tl.x0 + (p16 & 7);
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = blk_[k] && py0 + i < ea.DH && pxl + 16 * h < ea.DW;
                    wo[k][i][h] = ok ? (unsigned)((((py0 + i) * ea.OW + pxl + 16 * h) * cs_[k] + chw_[k] + (lo8 ? 0 : 16) + c4) * 4) : OOB;
                }
        auto wo2 = [&](int k, int i, int h) {                    // instruction 2: eight pixels on
            return (wo[k][i][h] != OOB && pxl + 16 * h + 8 < ea.DW) ? wo[k][i][h] + (unsigned)(8 * cs_[k] * 4) : OOB;
        };
        if constexpr (POOL) {
            // Forward layer in front of MaxPool2d(2) (archs/Unet.py:35,41,47,53): single destination, bias + activation only.  A wave owns rows
            // 2w, 2w + 1 of its 32 columns: a lane's two accumulator rows + the same two of lane ^ 1 are one 2x2 window of 4 channels; the even
            // lane writes the pooled float4 and the four codes (bits 0-1 first maximum in the order (0,0) (0,1) (1,0) (1,1), bits 2-5 the signs)
            // of csrc/misc.hip maxpool_fwd_codes_kernel.
            static_assert(MT == 2, "a wave owns one row pair");
            const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(0), 0);
            const __amdgpu_buffer_rsrc_t rb = bits_rsrc(ea.bits_out, ea.nblk0);
            const int ph = ea.OH >> 1, pwd = ea.OW >> 1;
            const int64_t pimg = (int64_t)b * ph * pwd * ea.pool_cs;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_dst + pimg), 0, ph * pwd * ea.pool_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_codes + pimg), 0, ph * pwd * ea.pool_cs, 0x00020000);
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                unsigned sb = 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 wn[2][2];                                  // [16-column block of the pair][row]
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            wn[jj][i] = act4(take_bias(2 * i + h, 2 * k + jj, bias4[2 * k + jj]));      // (one fma for scale + bias: bit-identical, v 2^dexp is exact)
                            track(wn[jj][i]);
                            sb |= signs4(wn[jj][i], ((i * 2 + h) * 2 + jj) * 4);
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {                     // full resolution: whole lines (the halves of the block pair traded)
                        const f32x4 ox = ror8(sel(lo8, wn[1][i], wn[0][i]));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, wn[0][i], ox)), rd, wo[k][i][h], 0, H2S_STORE_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, ox, wn[1][i])), rd, wo2(k, i, h), 0, H2S_STORE_AUX);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * k + jj;
                        const f32x4 (&win)[2] = wn[jj];
                        f32x4 nbr[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            // the pixel to the right (even lanes) / left (odd lanes): quad_perm [1, 0, 3, 2] (inline assembly: csrc/conv_x3s.hip)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                float nv; const float sv = win[i][c];
                                asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(nv) : "v"(sv));
                                nbr[i][c] = nv;
                            }
                        }
                        f32x4 mx;
                        unsigned code = 0;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float w0 = win[0][c], w1 = nbr[0][c], w2 = win[1][c], w3 = nbr[1][c];
                            unsigned arg = 0; float best = w0;
                            if (w1 > best) { best = w1; arg = 1; }                  // first maximum wins
                            if (w2 > best) { best = w2; arg = 2; }
                            if (w3 > best) { best = w3; arg = 3; }
                            const unsigned cj = arg | (w0 > 0.f ? 4u : 0u) | (w1 > 0.f ? 8u : 0u) | (w2 > 0.f ? 16u : 0u) | (w3 > 0.f ? 32u : 0u);
                            mx[c] = fmaxf(fmaxf(w0, w1), fmaxf(w2, w3));
                            code |= cj << (8 * c);
                        }
                        const int px = px0 + 16 * h;
                        const bool ok2 = !(lane_e & 1) && blk_[k] && py0 < ea.DH && px < ea.DW;      // even sizes: the whole window is inside or outside
                        const unsigned po = (unsigned)(((py0 >> 1) * pwd + (px >> 1)) * ea.pool_cs + n0 + 16 * j + c4);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mx), rp, ok2 ? po * 4u : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(code, rc, ok2 ? po : OOB, 0, 0);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b32(sb, rb, (ea.bits_out && blk_[k]) ? bits_off(k, ea.nblk0) : OOB, 0, 0);
                track_done(0);
            }
            return;
        }
        // ---- FWD: no mask, no accumulation, no residual (every forward layer; sign bits on request);  BWD / BWDB: act' masks as float32
        // activations / as the forward kernel's bits (a destination without one requests them out of range: zeros come back, no memory traffic).
        if constexpr (EK == EK_FWD || EK == EK_BWD || EK == EK_BWDB || EK == EK_HEAD || EK == EK_RES) {
            constexpr bool RES = EK == EK_RES;
            constexpr bool MASKED = EK == EK_BWD, BITS = EK == EK_BWDB, FWDL = EK == EK_FWD || EK == EK_HEAD;
            f32x4 mk[(MASKED || RES) ? MB : 1][(MASKED || RES) ? NB : 1];      // [.][2 k] = what instruction 1 fetched, [.][2 k + 1] = instruction 2 (EK_RES: the residual words)
            unsigned mbits[NT];
            if constexpr (MASKED || RES) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int mm = RES ? 1 : ea.mask_mode(du_[k]);
                    const __amdgpu_buffer_rsrc_t rm = rsrc(RES ? ea.addsrc : (mm ? ea.mask(du_[k]) : ea.dst(du_[k])), k);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            mk[2 * i + h][2 * k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm ? wo[k][i][h] : OOB, 0, 0));
                            mk[2 * i + h][2 * k + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm ? wo2(k, i, h) : OOB, 0, 0));
                        }
                }
            }
            if constexpr (BITS) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int mm = ea.mask_mode(du_[k]);
                    const unsigned* bp = ea.bits_in(du_[k]);
                    const __amdgpu_buffer_rsrc_t rb = bits_rsrc(bp ? bp : ea.bin0, ea.nblk(du_[k]));
                    if constexpr (H2S_PREBITS) mbits[k] = mbits_pre[k];
                    else mbits[k] = __builtin_amdgcn_raw_buffer_load_b32(rb, (mm && bp && blk_[k]) ? bits_off(k, ea.nblk(du_[k])) : OOB, 0, 0);
                }
            }
            auto body = [&](auto act_tag) __attribute__((always_inline)) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(du_[k]), k);
                    const int mm = ea.mask_mode(du_[k]);
                    const float msl = mm == 1 ? 0.2f : (mm == 0 ? 1.f : 0.f);      // act'(x <= 0); a destination without a mask (its requests came back as zeros): 1
                    unsigned sb = 0u;
                    float hp[EK == EK_HEAD ? MT * 2 : 1][4];        // (EK_HEAD) this lane's partial head sums per pixel block
                    const bool keep = EK != EK_HEAD || ea.dst0 != nullptr;      // (EK_HEAD, eval forward: the 32-channel map is not stored)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 o0, o1;
                            if constexpr (FWDL) {                    // (backward-data has no bias: the launcher checks)
                                o0 = take_bias(2 * i + h, 2 * k, bias4[2 * k]); o1 = take_bias(2 * i + h, 2 * k + 1, bias4[2 * k + 1]);
#pragma unroll
                                for (int c = 0; c < 4; ++c) { float e = o0[c]; act_sign(e, sb, aslope, act_tag); o0[c] = e; }
#pragma unroll
                                for (int c = 0; c < 4; ++c) { float e = o1[c]; act_sign(e, sb, aslope, act_tag); o1[c] = e; }
                            } else if constexpr (RES) {
                                // (bias from LDS at its use: the 16 registers of bias4 are what this kernel does not have beside the 64 residual words)
                                o0 = take_bias(2 * i + h, 2 * k, *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * k + c4));
                                o1 = take_bias(2 * i + h, 2 * k + 1, *reinterpret_cast<const f32x4*>(bias_lds + n0 + 32 * k + 16 + c4));
                            } else if constexpr (BITS) {
                                const f32x4 v0 = take_raw(2 * i + h, 2 * k), v1 = take_raw(2 * i + h, 2 * k + 1);
                                const float fneg = dsc * msl;
#pragma unroll
                                for (int c = 0; c < 4; ++c) o0[c] = mask_scale(v0[c], mbits[k], dsc, fneg);
#pragma unroll
                                for (int c = 0; c < 4; ++c) o1[c] = mask_scale(v1[c], mbits[k], dsc, fneg);
                            } else {
                                o0 = take(2 * i + h, 2 * k); o1 = take(2 * i + h, 2 * k + 1);
                            }
                            if constexpr (MASKED) {
                                const f32x4 m1 = mk[2 * i + h][2 * k], m2 = mk[2 * i + h][2 * k + 1], mx = ror8(sel(lo8, m2, m1));
                                const f32x4 q0 = sel(lo8, m1, mx), q1 = sel(lo8, mx, m2);      // the masks of this lane's lower / upper block
                                const f32x4 t0 = o0 * msl, t1 = o1 * msl;
#pragma unroll
                                for (int c = 0; c < 4; ++c) { o0[c] = q0[c] > 0.f ? o0[c] : t0[c]; o1[c] = q1[c] > 0.f ? o1[c] : t1[c]; }
                            }
                            if constexpr (RES) {                       // trade first, then add what the two store instructions' addresses hold of the residual
                                trade(o0, o1);
                                o0 += mk[2 * i + h][2 * k]; o1 += mk[2 * i + h][2 * k + 1];
                                track(o0); track(o1);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rd, wo[k][i][h], 0, H2S_STORE_AUX);
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rd, wo2(k, i, h), 0, H2S_STORE_AUX);
                                continue;
                            }
                            track(o0); track(o1);
                            if constexpr (EK == EK_HEAD) {
#pragma unroll
                                for (int o = 0; o < 4; ++o) {
                                    float sacc = hw[2 * o].x * o0.x;
                                    sacc = __builtin_fmaf(hw[2 * o].y, o0.y, sacc); sacc = __builtin_fmaf(hw[2 * o].z, o0.z, sacc); sacc = __builtin_fmaf(hw[2 * o].w, o0.w, sacc);
                                    sacc = __builtin_fmaf(hw[2 * o + 1].x, o1.x, sacc); sacc = __builtin_fmaf(hw[2 * o + 1].y, o1.y, sacc);
                                    sacc = __builtin_fmaf(hw[2 * o + 1].z, o1.z, sacc); sacc = __builtin_fmaf(hw[2 * o + 1].w, o1.w, sacc);
                                    hp[i * 2 + h][o] = sacc;
                                }
                                if (!keep) continue;
                            }
                            if constexpr (!(H2S_ABL & 1)) trade(o0, o1);
                            if constexpr (H2S_ABL & 16) { asm volatile("" :: "v"(o0), "v"(o1)); continue; }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rd, wo[k][i][h], 0, H2S_STORE_AUX);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rd, wo2(k, i, h), 0, H2S_STORE_AUX);
                        }
                    if constexpr (EK == EK_HEAD) {
                        // the 4 lanes of a pixel (q = lane >> 4) add their partial sums: exchange with lane ^ 16 (each keeps two outputs), then with
                        // lane ^ 32 (each keeps one): lane q ends with output o = 2 (q & 1) + (q >> 1) of its pixel, summed in a fixed order
                        const int q = lane_e >> 4, a16 = (lane_e ^ 16) * 4, a32 = (lane_e ^ 32) * 4;
                        const bool q0 = q & 1, q1 = q >> 1;
                        const int oo = 2 * (q & 1) + (q >> 1);
                        const float hb = head_lds[128 + oo];
                        const int64_t plane = (int64_t)ea.OH * ea.OW;
                        float* outp = ea.head_out + ((int64_t)b * 4 + oo) * plane;
                        const float* resp = ea.head_res ? ea.head_res + ((int64_t)b * 4 + oo) * plane : nullptr;
#pragma unroll
                        for (int blk = 0; blk < MT * 2; ++blk) {
                            const float s0 = q0 ? hp[blk][0] : hp[blk][2], s1 = q0 ? hp[blk][1] : hp[blk][3];
                            const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, s0)));
                            const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a16, __builtin_bit_cast(int, s1)));
                            const float k0 = (q0 ? hp[blk][2] : hp[blk][0]) + r0, k1 = (q0 ? hp[blk][3] : hp[blk][1]) + r1;
                            const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a32, __builtin_bit_cast(int, q1 ? k0 : k1)));
                            float v = (q1 ? k1 : k0) + r2 + hb;
                            const int i = blk >> 1, h = blk & 1;
                            if (okp[i][h]) {
                                const int64_t off = (int64_t)(py0 + i) * ea.OW + px0 + 16 * h;
                                if (resp) v += resp[off];
                                outp[off] = v;
                            }
                        }
                    }
                    if constexpr (FWDL) {
                        const __amdgpu_buffer_rsrc_t rb = bits_rsrc(ea.bits_out, ea.nblk0);
                        if (keep) __builtin_amdgcn_raw_buffer_store_b32(sb, rb, (ea.bits_out && blk_[k] && !du_[k]) ? bits_off(k, ea.nblk0) : OOB, 0, 0);
                    }
                    track_done(du_[k]);
                }
            };
            // one wave-uniform branch per tile: with / without an activation (backward-data never has one: the launcher sends a masked layer
            // WITH an activation to the general kernel)
            if constexpr (!FWDL) body(std::false_type{});
            else if (ea.act != 0) body(std::true_type{});
            else body(std::false_type{});
            return;
        }
        // ---- the general case (residual, accumulation), branch-free as well: what a block does not use is requested out of range
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int du = du_[k], mm2 = ea.mask_mode(du), acc2 = ea.accum(du);
            const bool use_add2 = ea.addsrc && du == 0;
            const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(du), k);
            const __amdgpu_buffer_rsrc_t rm = rsrc(mm2 ? ea.mask(du) : ea.dst(du), k);
            const __amdgpu_buffer_rsrc_t rad = rsrc(use_add2 ? ea.addsrc : ea.dst(du), k);
            const float msl = mm2 == 1 ? 0.2f : (mm2 == 0 ? 1.f : 0.f);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                f32x4 m2[MT][2], ad2[MT][2], pr2[MT][2];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        m2[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm2 ? vo[k][i][h] : OOB, jj * 64, 0));
                        ad2[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rad, use_add2 ? vo[k][i][h] : OOB, jj * 64, 0));
                        pr2[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, acc2 ? vo[k][i][h] : OOB, jj * 64, 0));
                    }
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        f32x4 o = act4(take(2 * i + h, 2 * k + jj) + bias4[2 * k + jj] + ad2[i][h]);
                        const f32x4 t = o * msl;
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = m2[i][h][c] > 0.f ? o[c] : t[c];
                        o += pr2[i][h];
                        track(o);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[k][i][h], jj * 64, 0);
                    }
            }
            track_done(du);
        }
    };

    // ---- the consumers' loop: barrier, chunk, barrier, chunk, ...  (no vector-memory wait anywhere: the only operations a consumer has in
    // flight are its own epilogue's, and nothing here depends on them)
    int img = 0, st = 0;
#ifdef H2S_STAMPS
    long long t_mfma = 0, t_epi = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int nch = 0;
#endif
    H2S_BARRIER();                                                // barrier 0
    H2S_T(t_bar)
    if constexpr (EK == EK_HEAD) {
#pragma unroll
        for (int q = 0; q < 8; ++q) hw[q] = *reinterpret_cast<const f32x4*>(head_lds + (q >> 1) * 32 + (q & 1) * 16 + (lane >> 4) * 4);
    }
    for (;;) {
        const Ck n1 = chunk_at(K1{});
#ifdef H2S_STAMPS
        ++nch;
#endif
        if constexpr (EK == EK_BWDB && H2S_PREBITS) { if (g == nchunks - 1) prefetch_bits(cur); }
        mfma_chunk(st, img);
        H2S_T(t_mfma)
        // Experiment (H2S_TURNS, off): the two consumer waves of a SIMD (w, w + 4) TAKE TURNS with the epilogue of a tile -- waves 4-7 run it in
        // front of the chunk barrier as always, waves 0-3 pass the barrier first and run it behind it, while their SIMD partner already issues
        // the next tile's first MFMAs.  Measured (profiles/r5/ab_epilogue_turns.txt): forward layers +0.7 %, backward-data +2.9 %, step -0.9 %:
        // one wave alone feeds the matrix pipe at 2/3 of the rate of two, and the waiting partner's idle time replaces the overlap.
        const bool last_chunk = g == nchunks - 1;
        const bool barrier_first = H2S_TURNS && last_chunk && n1.ok && wave < NCW / 2;
        if (barrier_first) {
            H2S_BARRIER();
            H2S_T(t_bar)
        }
        if (last_chunk) epilogue(cur);
        H2S_T(t_epi)
        if (!n1.ok) break;
        if (!barrier_first) {
            H2S_BARRIER();
            H2S_T(t_bar)
        }
        if (last_chunk) next_tile(); else ++g;
        img ^= 1; st ^= 1;
    }
    // ---- max |stored value| of the wave per destination -> the amax slots (non-negative floats order like their bit patterns)
#pragma unroll
    for (int du = 0; du < 2; ++du) {
        if (!ha.amax_out[du]) continue;
        pnnp_amax_commit(du ? amx1 : amx0, ha.amax_out[du]);
    }
#ifdef H2S_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (lane == 0) {
        float* d = a.dst[0] + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
        d[0] = (float)t_mfma; d[1] = (float)t_epi; d[2] = (float)t_bar; d[3] = 0.f; d[4] = (float)(clock64() - tall); d[5] = (float)nch;
    }
#endif
}

// ---- split-K reduce: y = act(sum_s slab[s] + bias) in a fixed order (s = 0, 1, ...), one float4 per thread (coalesced), max |y| into the amax slot and -- a
// training forward -- the sign bits OR-ed into the tile-private words of csrc/h2.h (zeroed by the launcher; 8 threads share a word, no two the same bits)
__global__ void __launch_bounds__(256)
h2_splitk_reduce_kernel(const float* __restrict__ slab, int S, const float* __restrict__ bias, float* __restrict__ y, unsigned* __restrict__ bits,
                        unsigned* __restrict__ amax, int B, int H, int W, int N, int act) {
    const int tiles_x = (W + 31) >> 5, tiles_y = (H + TH - 1) / TH, nblk = N >> 5, nq = N >> 2;
    const int64_t img = (int64_t)B * H * W * N, total = img >> 2;     // floats per slab; float4 elements
    const float slope = act == 1 ? 0.2f : (act == 2 ? 0.f : 1.f);
    float am = 0.f;
    for (int64_t e4 = (int64_t)blockIdx.x * 256 + threadIdx.x; e4 < total; e4 += (int64_t)gridDim.x * 256) {
        const f32x4* p = reinterpret_cast<const f32x4*>(slab) + e4;
        const int64_t st4 = img >> 2;
        f32x4 o = p[0];
        int sidx = 1;
        for (; sidx + 3 < S; sidx += 4) {                           // (four independent loads in flight; the ORDER of the additions stays s = 0, 1, 2, ...)
            const f32x4 v0 = p[sidx * st4], v1 = p[(sidx + 1) * st4], v2 = p[(sidx + 2) * st4], v3 = p[(sidx + 3) * st4];
            o += v0; o += v1; o += v2; o += v3;
        }
        for (; sidx < S; ++sidx) o += p[sidx * st4];
        const int cq = (int)(e4 % nq), ch = cq * 4;
        if (bias) o += *reinterpret_cast<const f32x4*>(bias + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = o[c] > 0.f ? o[c] : o[c] * slope;
        reinterpret_cast<f32x4*>(y)[e4] = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        if (bits) {
            int64_t px = e4 / nq;
            const int xx = (int)(px % W); px /= W;
            const int yy = (int)(px % H); const int b = (int)(px / H);
            const int ty = yy / TH, wv = (yy % TH) >> 1, i = yy & 1, tx = xx >> 5, h = (xx >> 4) & 1, p16 = xx & 15;
            const int cb = ch >> 5, jj = (ch >> 4) & 1, q = (ch >> 2) & 3;
            const int64_t word = ((((int64_t)(b * tiles_y + ty) * tiles_x + tx) * nblk + cb) * NCW + wv) * 64 + q * 16 + p16;
            const unsigned nib = (o.x > 0.f ? 8u : 0u) | (o.y > 0.f ? 4u : 0u) | (o.z > 0.f ? 2u : 0u) | (o.w > 0.f ? 1u : 0u);
            atomicOr(bits + word, nib << (28 - ((i * 2 + h) * 2 + jj) * 4));
        }
    }
    if (amax) pnnp_amax_commit_block(am, amax);
}

template <int BN, int EK>
int launch_h2s(const H2Args& a, hipStream_t s) {
    using Cfg = SCfg<BN>;
    auto kern = igemm_h2s_kernel<BN, EK>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int tiles = ((a.g.DW + 31) / 32) * ((a.g.DH + TH - 1) / TH) * a.g.B * ((a.g.Ntot + BN - 1) / BN) * (EK == EK_GEN && a.ksplit > 1 ? a.ksplit : 1);
    if (tiles <= 0) return PNNP_OK;
    const int wgs = pnnp_persistent_grid(tiles);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

}  // namespace

// words of the tile-private sign-bit image of a [B][H][W][C] tensor (csrc/h2.h), C a multiple of 32
extern "C" int64_t pnnp_h2_bits_words(int B, int H, int W, int C) {
    return (int64_t)B * ((H + TH - 1) / TH) * ((W + 31) / 32) * ((C + 31) / 32) * NCW * 64;
}

// 64-column tiles unless the layer has fewer channels or they would leave CUs idle (csrc/conv_x3.hip); the pooled forward keeps 64 whenever it can
extern "C" int pnnp_h2_tile_columns(int B, int H, int W, int N, int pool) {
    if (N < 64) return 32;
    if (pool) return 64;
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t tiles64 = (int64_t)((W + 31) / 32) * ((H + TH - 1) / TH) * B * ((N + 63) / 64);
    return tiles64 * 4 >= (int64_t)cus * 3 ? 64 : 32;
}

// split-K policy: how many K slices a [B][H][W] layer with `chunks` 16-channel chunks of K and N channels written should be cut into so that the grid fills
// the chip (1 = no split): the smallest divisor of `chunks` that brings 32-column tiles x slices to 3/4 of the CUs, at least 2 chunks per slice
extern "C" int pnnp_h2_splitk(int B, int H, int W, int chunks, int N) {
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t tiles32 = (int64_t)((W + 31) / 32) * ((H + TH - 1) / TH) * B * ((N + 31) / 32);
    if (tiles32 * 4 >= (int64_t)cus * 3 || chunks < 4) return 1;
    int best = 1;
    for (int sp = 2; sp * 2 <= chunks; ++sp) {
        if (chunks % sp) continue;
        best = sp;
        if (tiles32 * sp * 4 >= (int64_t)cus * 3) break;
    }
    return best;
}
int pnnp_h2_splitk_reduce_launch(const float* slab, int S, const float* bias, float* y, unsigned* bits, unsigned* amax, int B, int H, int W, int N, int act, hipStream_t st) {
    const int64_t total = (int64_t)B * H * W * N / 4;
    if (total <= 0) return PNNP_OK;
    if (bits && hipMemsetAsync(bits, 0, (size_t)pnnp_h2_bits_words(B, H, W, N) * 4, st) != hipSuccess) return PNNP_E_LAUNCH;
    const int64_t blocks = (total + 255) / 256;
    // (at most one block per CU: every block ends with ONE atomicMax on the amax slot, and blocks that finish together serialise on it -- ~15 ns each)
    hipLaunchKernelGGL(h2_splitk_reduce_kernel, dim3((unsigned)(blocks > 256 ? 256 : blocks)), dim3(256), 0, st, slab, S, bias, y, bits, amax, B, H, W, N, act);
    return pnnp_launch_status();
}

// Validates like pnnp_igemm_x3_launch (csrc/conv_x3.hip); a.g.w: the h2 pack of csrc/pack_jobs.hip (kind 4).
int pnnp_igemm_h2s_launch(const H2Args& ha, int chan_per_seg, hipStream_t s) {
    const IgemmArgs& a = ha.g;
    if (a.nseg < 1 || a.nseg > 2 || chan_per_seg <= 0 || (chan_per_seg & 7) || (a.nseg > 1 && (chan_per_seg & 15)) || a.Ntot <= 0) return PNNP_E_INVALID;
    if (!ha.amax_in[0] || !ha.amax_w || (a.nseg > 1 && !ha.amax_in[1])) return PNNP_E_INVALID;
    if ((a.Ntot & 31) || a.Ntot > SCfg<64>::BIAS_MAX || a.in_mul != 1 || a.out_mul != 1 || a.n_sub || a.out_yoff || a.out_xoff) return PNNP_E_UNSUPPORTED;
    if (a.dst[1] && (a.n_split & 31)) return PNNP_E_UNSUPPORTED;
    if (a.addsrc && a.accum[0]) return PNNP_E_UNSUPPORTED;
    if ((a.dst_cs[0] & 3) || (a.dst[1] && (a.dst_cs[1] & 3))) return PNNP_E_UNSUPPORTED;
    if ((((uintptr_t)a.dst[0]) | ((uintptr_t)a.dst[1]) | ((uintptr_t)a.bias) | ((uintptr_t)a.mask[0]) | ((uintptr_t)a.mask[1]) |
         ((uintptr_t)a.addsrc) | ((uintptr_t)a.w)) & 15) return PNNP_E_INVALID;
    for (int i = 0; i < a.nseg; ++i) {
        if (a.seg[i].yoff || a.seg[i].xoff || (a.seg[i].cstride & 3) || (((uintptr_t)a.seg[i].ptr) & 15)) return PNNP_E_UNSUPPORTED;
        if (((int64_t)a.IH + 4) * a.IW * a.seg[i].cstride * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;     // 32-bit offsets inside one image
    }
    for (int d = 0; d < 2; ++d)
        if (a.dst[d] && (int64_t)a.OH * a.OW * a.dst_cs[d] * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    H2Args b = ha;
    b.g.chunks_per_seg = (chan_per_seg + 15) / 16;
    b.g.seg_channels = chan_per_seg;
    if (b.ksplit < 1) b.ksplit = 1;
    if (b.ksplit > 1) {
        // split-K: raw partial sums into a slab tensor of ksplit x B images -- nothing but the sums themselves (the reduce kernel adds bias, activation, amax)
        if ((b.g.nseg * b.g.chunks_per_seg) % b.ksplit || a.dst[1] || a.bias || a.act || a.mask_mode[0] || a.accum[0] || a.addsrc || a.pool_dst || a.pool_codes ||
            ha.bits_out || ha.bits_in[0] || ha.bits_in[1] || ha.head_out || ha.amax_out[0]) return PNNP_E_UNSUPPORTED;
        if ((int64_t)a.OH * a.OW * a.dst_cs[0] * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
        return pnnp_h2_tile_columns(a.B * b.ksplit, a.DH, a.DW, a.Ntot, 0) == 64 ? launch_h2s<64, EK_GEN>(b, s) : launch_h2s<32, EK_GEN>(b, s);
    }
    const int64_t wbytes = (int64_t)((a.Ntot + 31) / 32) * b.g.nseg * b.g.chunks_per_seg * WBLK;
    if (wbytes >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    const bool two = a.dst[1] != nullptr;
    for (int d = 0; d < 2; ++d) {                                  // the bit images must fit a buffer resource
        const unsigned* p = d ? ha.bits_in[1] : (ha.bits_in[0] ? ha.bits_in[0] : ha.bits_out);
        if (p && (ha.bits_nblk[d] <= 0 || pnnp_h2_bits_words(a.B, a.DH, a.DW, 32 * ha.bits_nblk[d]) * 4 >= (1ll << 31))) return PNNP_E_UNSUPPORTED;
    }
    if (ha.head_out) {
        // forward + 1x1 head (EK_HEAD): one 32-column block holds every channel of a pixel; plain single-destination forward layers only
        if (!ha.head_w || a.Ntot != 32 || two || a.mask_mode[0] || a.accum[0] || a.addsrc || a.pool_dst || a.pool_codes || ha.bits_in[0] || ha.bits_in[1] ||
            a.OH != a.DH || a.OW != a.DW || (((uintptr_t)ha.head_w | (uintptr_t)ha.head_out | (uintptr_t)ha.head_res | (uintptr_t)ha.head_b) & 3))
            return PNNP_E_UNSUPPORTED;
        if (!a.dst[0] && (ha.bits_out || ha.amax_out[0])) return PNNP_E_INVALID;     // (nothing stored: nothing to describe)
        return launch_h2s<32, EK_HEAD>(b, s);
    }
    if (!a.dst[0]) return PNNP_E_INVALID;
    if (a.pool_dst || a.pool_codes) {
        // fused MaxPool2d(2): plain forward layers only (one destination, no mask / residual / accumulate), even sizes
        if (!a.pool_dst || !a.pool_codes || two || a.mask_mode[0] || a.accum[0] || a.addsrc || (a.OH & 1) || (a.OW & 1) || a.OH != a.DH ||
            a.OW != a.DW || (a.pool_cs & 3) || a.pool_cs < a.Ntot || ((uintptr_t)a.pool_dst & 15) || ((uintptr_t)a.pool_codes & 3) || ha.bits_in[0] || ha.bits_in[1])
            return PNNP_E_UNSUPPORTED;
        if ((int64_t)(a.OH / 2) * (a.OW / 2) * a.pool_cs * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
        return pnnp_h2_tile_columns(a.B, a.DH, a.DW, a.Ntot, 1) == 64 ? launch_h2s<64, EK_POOL>(b, s) : launch_h2s<32, EK_POOL>(b, s);
    }
    const bool wide = pnnp_h2_tile_columns(a.B, a.DH, a.DW, a.Ntot, 0) == 64;
    const bool plain = !a.addsrc && !a.accum[0] && !(two && a.accum[1]);
    const bool m0 = a.mask_mode[0] != 0, m1 = two && a.mask_mode[1] != 0;
    const bool f0 = m0 && !ha.bits_in[0], f1 = m1 && !ha.bits_in[1];           // float32 masks
    const bool b0 = m0 && ha.bits_in[0], b1 = m1 && ha.bits_in[1];             // bit masks
    if ((f0 || f1) && (b0 || b1)) return PNNP_E_UNSUPPORTED;                   // one kind per launch
    if ((b0 || b1) && (!plain || a.act || a.bias)) return PNNP_E_UNSUPPORTED;  // bit masks: the masked backward-data epilogue only
    if (ha.bits_out && (m0 || m1 || !plain || two)) return PNNP_E_UNSUPPORTED; // sign bits: plain single-destination forward layers
    if (a.addsrc && !two && !a.accum[0] && !a.act && !m0 && !ha.bits_out) return wide ? launch_h2s<64, EK_RES>(b, s) : launch_h2s<32, EK_RES>(b, s);
    if (plain && !m0 && !m1) return wide ? launch_h2s<64, EK_FWD>(b, s) : launch_h2s<32, EK_FWD>(b, s);
    if (plain && !a.act && !a.bias) {
        if (b0 || b1) return wide ? launch_h2s<64, EK_BWDB>(b, s) : launch_h2s<32, EK_BWDB>(b, s);
        return wide ? launch_h2s<64, EK_BWD>(b, s) : launch_h2s<32, EK_BWD>(b, s);
    }
    return wide ? launch_h2s<64, EK_GEN>(b, s) : launch_h2s<32, EK_GEN>(b, s);
}
