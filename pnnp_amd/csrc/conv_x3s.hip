// 3x3 convolution (forward / backward-data) on the bf16 matrix cores, fp32 operands split into three bf16 pieces -- the scheme, the LDS
// images, the weight pack and the tile of csrc/conv_x3.hip -- with the workgroup's waves SPECIALISED (round 4):
//
//   8 CONSUMER waves (two per SIMD, 2 pixel rows x BN channels each): ds_read_b128 + v_mfma_f32_16x16x32_bf16 only, and the epilogue;
//   4 PRODUCER waves (one per SIMD): everything else -- the halo tile of the next chunk (fp32 NHWC global -> registers -> hi / mid / lo
//     split -> LDS image) and the weights of the next filter rows (LDS-DMA).
//
// Why (tools/ubench/mfma_valu_coissue.hip, profiles/r4/mfma_valu_coissue.txt): VALU instructions of a wave do NOT run under that wave's own
// MFMAs -- a wave that interleaves k vector instructions per MFMA needs (MFMA time + k x 4 cycles) per MFMA, exactly additive, and two
// such waves per SIMD only partly cover for each other (4 staging units per 8 MFMAs: 33 cycles per MFMA and SIMD instead of 13.5) -- but
// VALU instructions of ANOTHER wave of the SIMD do: two MFMA-only waves keep their 13.5 cycles per MFMA whatever a third, VALU-only wave
// does beside them.  conv_x3.hip's waves each stage 1/8 of the next halo between their own MFMAs (filter row 2 of every chunk: 146 vector
// instructions per 144 MFMAs, measured 5100 cycles against 2850 for the staging-free rows 0 and 1); moving the staging between the waves of
// a SIMD ("complementary pairing", X3_FILLMODE) changed nothing because every wave still paid for its own share.  Here the consumers never
// issue a vector-ALU or vector-memory instruction inside the K loop.
//
// Synchronisation: one s_barrier per work item (filter row of a 16-channel chunk), all 12 waves.  Between barrier i and i + 1 the consumers
// run item i (halo image c & 1 of chunk c, weight stage i % NSTAGE); the producers request the weights of item i + AHEAD into the stage
// item i - 1 has just left, split a share of chunk c + 1's halo into the other image, and wait (exact vmcnt) for the weights of item i + 1
// before they arrive at barrier i + 1.  Epilogue: straight from the accumulators (weights as the MFMA's first operand: a lane holds 4
// consecutive channels of one pixel), stores drain while the next tile starts -- the consumers have nothing else in flight to wait for.
#include "igemm.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int pnnp_igemm_x3s_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);

namespace {

constexpr int NCW = 8, NPW = 4, NTHR = 64 * (NCW + NPW);           // consumer / producer waves
constexpr int MT = 2, TH = NCW * MT, HR = TH + 2, HC = 34, NPIX = HR * HC;     // 16-row x 32-px tile, 612 halo pixels
// halo image in 16-byte words: [piece 3][k-octet 2][pixel, plane padded to a multiple of 16 words] (csrc/conv_x3.hip, X3_M16)
constexpr int NPIXP = (NPIX + 15) / 16 * 16;                       // 624
constexpr int XS_F4 = 3 * 2 * NPIXP, XS_PIECE_STRIDE = 2 * NPIXP, XS_BYTES = XS_F4 * 16;      // 59904
#define XS_PLANE(piece, oct) (((piece) * 2 + (oct)) * NPIXP)
constexpr int WBLK = 3 * 2 * 3 * 32 * 16;                          // one filter row of one 32-channel block: [tap 3][octet 2][piece 3][32][16 B] = 9216
constexpr int PTHR = 64 * NPW;                                     // producer threads
constexpr int NSLOT = (2 * NPIX + PTHR - 1) / PTHR;                // halo staging slots per producer thread: 1224 (pixel, octet) pairs / 256 -> 5
constexpr unsigned OOB = 0x80000000u;
#define X3S_VMCNT(N) (0x0f70 | ((N) & 15) | (((N) >> 4) << 14))    // s_waitcnt vmcnt(N) alone
// The item barrier as assembly (LDS operations of this wave done, then s_barrier; "memory": the compiler moves nothing across it).  Through
// __syncthreads() -- a workgroup fence + barrier -- the compiler waits for EVERY LDS-DMA a wave has in flight (vmcnt(0): it cannot know which
// stage the reads behind the barrier touch), i.e. for the weights requested a moment ago for the item after next.
#define X3S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int BN> struct SCfg {
    static constexpr int NT = BN / 32;
    static constexpr int WS_STAGE = NT * WBLK;                     // 9216 / 18432
    static constexpr int NDMA = WS_STAGE / 1024;                   // 1 KB LDS-DMA pieces per stage: 9 / 18
    static constexpr int DPW = (NDMA + NPW - 1) / NPW;             // LDS-DMA instructions per producer wave and item: 3 / 5
    // weight ring: BN = 64 two stages, one item ahead (an item is 72 MFMAs per consumer wave, > 2 us); BN = 32 three stages (filter row r
    // lives in stage r), two items ahead
    static constexpr int NSTAGE = BN == 32 ? 3 : 2, AHEAD = NSTAGE - 1;
    static constexpr int BIAS_MAX = 1024;                          // the layer's bias vector lives in LDS: at most this many output channels (the launcher checks)
    static constexpr int LDS_BYTES = 2 * XS_BYTES + NSTAGE * WS_STAGE + (BIAS_MAX + 64) * 4;      // 161024 / 151808
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

#ifndef X3S_SKEW
#define X3S_SKEW 0
#endif
#ifndef X3S_STORE_AUX
#define X3S_STORE_AUX 2              // cache-policy bits of the epilogue's full-resolution stores: 2 = nt (non-temporal: +0.3-0.5 % on the step, three
                                   // alternating same-box pairs, profiles/r4/ab_store_policy.txt); 0 = default, 1 = sc0, 3 = sc0 + nt measured too
#endif
#ifndef X3S_SKEW_PHASES
#define X3S_SKEW_PHASES 8
#endif
#ifdef X3S_STAMPS                 // debug build: cycle sums per wave, dumped into dst[0] (tools/x3s_stamps.py)
#define X3S_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
#else
#define X3S_T(v)
#endif
enum { EK_FWD = 0, EK_BWD = 1, EK_GEN = 2, EK_POOL = 3 };          // the epilogue a kernel carries (one straight-line path each): see `epilogue`

template <int BN, int EK>
__global__ void __launch_bounds__(NTHR, 1)
igemm_x3s_kernel(const IgemmArgs a) {
    constexpr bool POOL = EK == EK_POOL;
    using Cfg = SCfg<BN>;
    constexpr int NT = Cfg::NT, NSTAGE = Cfg::NSTAGE, AHEAD = Cfg::AHEAD, D = Cfg::DPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                     // two halo images
    char* wsb = smem + 2 * XS_BYTES;                                // the weight ring
    float* bias_lds = reinterpret_cast<float*>(smem + 2 * XS_BYTES + NSTAGE * Cfg::WS_STAGE);      // bias[0 .. Ntot) (zeros without a bias)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0 .. 7 consumers, 8 .. 11 producers

    // ---- the workgroup's tiles t, t + G, ...: decoded once, then stepped by mixed-radix addition (both roles walk the same sequence)
    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int G = gridDim.x;
    const int nchunks = a.nseg * a.chunks_per_seg;                  // 16-channel chunks of K
    struct Tile { int b, y0, x0, n0; };
    auto decode = [&](int t) {
        Tile o;
        const int nt_i = t % n_tiles;
        int m_i = t / n_tiles;
        const int tx = m_i % tiles_x; m_i /= tiles_x;
        o.x0 = tx * 32; o.y0 = (m_i % tiles_y) * TH; o.b = m_i / tiles_y; o.n0 = nt_i * BN;
        return o;
    };
    auto pick = [](bool c, const Tile& x, const Tile& y) { Tile o; o.b = c ? x.b : y.b; o.y0 = c ? x.y0 : y.y0; o.x0 = c ? x.x0 : y.x0; o.n0 = c ? x.n0 : y.n0; return o; };
    const Tile gstep = decode(G);
    auto advance = [&](Tile o) {
        o.n0 += gstep.n0; if (o.n0 >= n_tiles * BN) { o.n0 -= n_tiles * BN; o.x0 += 32; }
        o.x0 += gstep.x0; if (o.x0 >= tiles_x * 32) { o.x0 -= tiles_x * 32; o.y0 += TH; }
        o.y0 += gstep.y0; if (o.y0 >= tiles_y * TH) { o.y0 -= tiles_y * TH; o.b += 1; }
        o.b += gstep.b;
        return o;
    };
    int t = xcd_remap(blockIdx.x, G);
    if (t >= total) return;
    Tile cur = decode(t), nxt = pick(t + G < total, advance(cur), cur);
    Tile nxt2 = pick(t + 2 * G < total, advance(nxt), nxt);
    int g = 0;                                                       // chunk of the current tile
    // the k-th chunk after the current one, k = 1, 2: (tile, chunk, exists); past the end of this workgroup's work it falls back to the
    // current chunk (requests stay branch-free; weights are then requested with valid = false)
    struct Ck { Tile tile; int g; bool ok; };
    auto chunk_at = [&](int k) {
        int gk = g + k, hop = 0;
        if (gk >= nchunks) { gk -= nchunks; hop = 1; }
        if (gk >= nchunks) { gk -= nchunks; hop = 2; }
        Ck c;
        c.ok = t + hop * G < total;
        c.g = c.ok ? gk : g;
        c.tile = pick(!c.ok || hop == 0, cur, pick(hop == 1, nxt, nxt2));
        return c;
    };
    auto next_tile = [&]() { t += G; cur = nxt; nxt = nxt2; nxt2 = pick(t + 2 * G < total, advance(nxt), nxt); g = 0; };

    if (wave >= NCW) {
        // =============================================== PRODUCER ===============================================
#ifdef X3S_PPRIO
        __builtin_amdgcn_s_setprio(X3S_PPRIO);                        // experiment: the producers win the issue arbitration
#endif
        const int pw = wave - NCW, ptid = tid - 64 * NCW;            // 0 .. 3, 0 .. 255
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
            xdst[k] = XS_PLANE(0, oct) + pix;                       // + piece * XS_PIECE_STRIDE (+ image * XS_F4)
        }
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
        f32x4 ra[NSLOT][2];                                         // the halo of the NEXT chunk, 8 channels per slot
        // global loads of the halo tile of (tile, chunk gq) -> ra: hardware zero for pixels outside the image and channels past the segment
        auto load_halo = [&](const Tile& tl, int gq) {
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
                ra[k][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff, 0));
                ra[k][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, soff + 16, 0));
            }
        };
        // slot k of ra -> its three 16-byte words in halo image img
        auto stage_slot = [&](int k, int img) {
            u32x4 sh, sm, sl;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x4 v = ra[k][p >> 1];
                unsigned h, m, l;
                split2(v[(p & 1) * 2], v[(p & 1) * 2 + 1], h, m, l);
                sh[p] = h; sm[p] = m; sl[p] = l;
            }
            u32x4* d = xs + img * XS_F4 + xdst[k];
            d[0] = sh; d[XS_PIECE_STRIDE] = sm; d[2 * XS_PIECE_STRIDE] = sl;
        };
        // LDS-DMA of the weights of item (tile n0, chunk gq, filter row tr) into stage st: per 32-channel block 9216 contiguous bytes of the
        // pack, as 1 KB pieces dealt over the 4 producer waves; past the end a wave repeats the last piece (same bytes, same place)
        const int K16 = nchunks;
        auto dma_weights = [&](const Tile& tl, int gq, int tr, int st, bool valid) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                const int ins = min(pw + NPW * i, Cfg::NDMA - 1);
                const int j = ins / 9, r = ins - 9 * j;
                const int nb = (tl.n0 >> 5) + j;
                const bool ok = valid && nb * 32 < a.Ntot;        // (an invalid request still issues: the vmcnt counts below count instructions)
                const int soff = ok ? ((nb * K16 + gq) * 27648 + tr * WBLK + r * 1024) : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wsb + st * Cfg::WS_STAGE + ins * 1024),
                                                         16, ok ? (unsigned)lane * 16u : OOB, soff, 0, 0);
            }
        };
        // item number `it` counts filter rows over the whole run of the workgroup: item it lives in stage it % NSTAGE.  The item k rows
        // after row tr of the current chunk:
        auto dma_item_after = [&](int tr, int k, int st) {
            const int r2 = tr + k;                                   // 0 .. 2 + AHEAD
            const Ck c = chunk_at(r2 / 3);
            if (r2 < 3) dma_weights(cur, g, r2, st, true); else dma_weights(c.tile, c.g, r2 - 3, st, c.ok);
        };
#if X3S_SKEW > 0
        // Phase skew: workgroup (blockIdx >> 3) & 7 of its XCD starts X3S_SKEW cycles later per step.  All workgroups run tiles of the same
        // length from the same start, so without it every CU reaches its epilogue in the same moment: 256 x 128 KB of stores against
        // ~5.5 TB/s of write bandwidth -- 13 bytes per cycle and CU where a CU alone stores 85 (tools/ubench/store_rate.hip) -- and the
        // stores, though nobody waits for their completion, back up into the issue of the waves that carry them (cycle stamps: 8600 cycles
        // per forward tile, 18 000 per backward-data tile).  (The consumers wait at barrier 0 meanwhile.)
        {
            const long long t0 = clock64(), dly = (long long)((blockIdx.x >> 3) & (X3S_SKEW_PHASES - 1)) * X3S_SKEW;
            while (clock64() - t0 < dly) __builtin_amdgcn_s_sleep(8);
        }
#endif
        // ---- prologue: the bias vector, weights of items 0 .. AHEAD - 1, chunk 0's halo straight into image 0, chunk 1's halo into the registers
        for (int i = ptid; i < Cfg::BIAS_MAX + 64; i += PTHR) bias_lds[i] = (a.bias && i < a.Ntot) ? a.bias[i] : 0.f;
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) dma_item_after(0, k, k % NSTAGE);
        load_halo(cur, 0);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) stage_slot(k, 0);
        {
            const Ck n1 = chunk_at(1);
            load_halo(n1.tile, n1.g);                               // (past the end: the current chunk again, harmless)
        }
        __builtin_amdgcn_s_waitcnt(X3S_VMCNT(2 * NSLOT));           // the weights; chunk 1's halo stays in flight
        X3S_BARRIER();                                            // barrier 0: item 0 may start
        int img = 0, st = 0;                                        // image of the current chunk; stage of the current item
#ifdef X3S_STAMPS
        long long t_work = 0, t_wait = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int nch = 0;
#endif
        for (;;) {
            const Ck n1 = chunk_at(1), n2 = chunk_at(2);
#ifdef X3S_STAMPS
            ++nch;
#endif
            // One chunk = three items.  Per item: [weights of item + AHEAD] then a share of the staging of chunk + 1 (its halo has been in
            // flight for a whole chunk); behind the last share the registers are free and chunk + 2's halo is requested.  In front of every
            // barrier the weights of the NEXT item must have landed: vmcnt(what was issued behind them).
            // ---- item 0
            dma_item_after(0, AHEAD, (st + AHEAD) % NSTAGE);
            stage_slot(0, img ^ 1); stage_slot(1, img ^ 1);
            X3S_T(t_work)
            __builtin_amdgcn_s_waitcnt(X3S_VMCNT((AHEAD - 1) * D));
            X3S_T(t_wait)
            X3S_BARRIER();
            X3S_T(t_bar)
            // ---- item 1
            dma_item_after(1, AHEAD, (st + 1 + AHEAD) % NSTAGE);
            stage_slot(2, img ^ 1); stage_slot(3, img ^ 1);
            X3S_T(t_work)
            __builtin_amdgcn_s_waitcnt(X3S_VMCNT((AHEAD - 1) * D));
            X3S_T(t_wait)
            X3S_BARRIER();
            X3S_T(t_bar)
            // ---- item 2
            dma_item_after(2, AHEAD, (st + 2 + AHEAD) % NSTAGE);
#pragma unroll
            for (int k = 4; k < NSLOT; ++k) stage_slot(k, img ^ 1);
            load_halo(n2.tile, n2.g);
            X3S_T(t_work)
            __builtin_amdgcn_s_waitcnt(X3S_VMCNT((AHEAD - 1) * D + 2 * NSLOT));
            X3S_T(t_wait)
            if (!n1.ok) break;                                      // (the consumers' epilogue and exit need no barrier)
            X3S_BARRIER();
            X3S_T(t_bar)
            if (g == nchunks - 1) next_tile(); else ++g;
            img ^= 1; st = (st + 3) % NSTAGE;
        }
#ifdef X3S_STAMPS
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
    const int r16 = lane & 15, q16 = lane >> 4, oct16 = q16 & 1, ps16 = q16 >> 1;
    // operand forms (two pieces concatenated along K = 32): pixels  0 = [hi | mid], 1 = [hi | lo];  weights 0 = [hi' | hi'], 1 = [mid' | mid'],
    // 2 = [lo' | hi']:  P1 W2 = hi lo' + lo hi',  P0 W1 = hi mid' + mid mid',  P0 W0 = hi hi' + mid hi'  -- the six products of the scheme
    const int aoff0 = XS_PLANE(ps16 ? 1 : 0, oct16) + r16, aoff1 = XS_PLANE(ps16 ? 2 : 0, oct16) + r16;            // 16-byte words
    const int boff0 = ((oct16 * 3 + 0) * 32 + r16) * 16, boff1 = ((oct16 * 3 + 1) * 32 + r16) * 16,
              boff2 = ((oct16 * 3 + (ps16 ? 0 : 2)) * 32 + r16) * 16;                                            // bytes inside one tap
    // MFMAs of filter row tr (halo image img, weight stage st).  Order per tap: pass j (16 output channels) x pixel block mb x the three
    // products, smallest terms first.  The 8 pixel words of the tap stay in registers for all passes and are refreshed IN PLACE for the next
    // tap during the last pass; the 3 weight words of a pass are read one pass ahead into the other of two register sets.
    auto mfma_row = [&](int tr, int st, int img, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;         // first item of a tile: the first product of every block starts from zero (no zeroing in the epilogue)
        constexpr int GT = NB * MB * 3;                             // MFMAs per tap
        const char* wst = wsb + st * Cfg::WS_STAGE;
        const u32x4* xim = xs + img * XS_F4;
        u32x4 A[MB][2], Bv[2][3];
        auto a_read = [&](int tp, int mb, int f) {
            A[mb][f] = xim[(f ? aoff1 : aoff0) + (wave * MT + (mb >> 1) + tr) * HC + tp + 16 * (mb & 1)];
        };
        auto b_read = [&](int tp, int j, int f, int buf) {
            Bv[buf][f] = *reinterpret_cast<const u32x4*>(wst + (j >> 1) * WBLK + tp * 3072 + (f == 0 ? boff0 : (f == 1 ? boff1 : boff2)) + (j & 1) * 256);
        };
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { a_read(0, mb, 1); a_read(0, mb, 0); }
#pragma unroll
        for (int f = 0; f < 3; ++f) b_read(0, 0, 2 - f, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 3 * GT>([&](auto Gc) {
            constexpr int gi = decltype(Gc)::value;
            constexpr int tp = gi / GT, gt = gi % GT, j = gt / (MB * 3), w = gt % (MB * 3), mb = w / 3, sp = w % 3;
            constexpr int pass = tp * NB + j, buf = pass & 1;
#ifdef X3S_PRIOALT                 // experiment: the two consumer waves of a SIMD (w, w + 4) take turns as the arbitration winner, every X3S_PRIOALT MFMAs
            if constexpr (gi % X3S_PRIOALT == 0) {
                if ((((gi / X3S_PRIOALT) & 1) != 0) == (wave >= NCW / 2)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
#endif
#define X3S_MFMA(FA, FB) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Bv[buf][FB]), __builtin_bit_cast(bf16x8, A[mb][FA]), \
                                                                         (FIRST && tp == 0 && sp == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[mb][j], 0, 0, 0)
            if constexpr (sp == 0) X3S_MFMA(1, 2);                  // hi lo' + lo hi'
            else if constexpr (sp == 1) X3S_MFMA(0, 1);             // hi mid' + mid mid'
            else X3S_MFMA(0, 0);                                    // hi hi' + mid hi'
#undef X3S_MFMA
            if constexpr (w < 3 && pass + 1 < 3 * NB) b_read((pass + 1) / NB, (pass + 1) % NB, 2 - w, buf ^ 1);
            if constexpr (j == NB - 1 && tp < 2) {
                if constexpr (sp == 0) a_read(tp + 1, mb, 1);
                if constexpr (sp == 2) a_read(tp + 1, mb, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- the epilogue's kernel arguments, cached in ONE vector register (lane i = argument i) and fetched with v_readlane: the unrolled K loop
    // leaves the compiler no scalar registers for them, and it then RE-LOADS each from the kernel-argument segment where the epilogue uses it
    // (s_load + s_waitcnt lgkmcnt(0), 200-300 cycles a piece, ~25 per tile: cycle stamps of the first version put a forward epilogue at 9000
    // cycles per tile).  The layer's bias vector sits in LDS (written once by the producers) for the same reason: no memory latency here.
    enum { E_OH, E_OW, E_DH, E_DW, E_NTOT, E_NSPLIT, E_ACT, E_POOLCS, E_CS0, E_CS1, E_MM0, E_MM1, E_AC0, E_AC1,
           E_DST0, E_DST1 = E_DST0 + 2, E_MASK0 = E_DST1 + 2, E_MASK1 = E_MASK0 + 2, E_ADD = E_MASK1 + 2, E_BIAS = E_ADD + 2, E_PDST = E_BIAS + 2,
           E_PCODE = E_PDST + 2, E_COUNT = E_PCODE + 2 };
    static_assert(E_COUNT <= 64, "one lane per cached argument");
    unsigned argv = 0;
    {
        auto put = [&](int idx, unsigned v) { argv = lane == idx ? v : argv; };
        auto putp = [&](int idx, const void* q) { put(idx, (unsigned)(uintptr_t)q); put(idx + 1, (unsigned)((uintptr_t)q >> 32)); };
        put(E_OH, a.OH); put(E_OW, a.OW); put(E_DH, a.DH); put(E_DW, a.DW); put(E_NTOT, a.Ntot); put(E_NSPLIT, a.n_split); put(E_ACT, a.act);
        put(E_POOLCS, a.pool_cs); put(E_CS0, a.dst_cs[0]); put(E_CS1, a.dst_cs[1]); put(E_MM0, a.mask_mode[0]); put(E_MM1, a.mask_mode[1]);
        put(E_AC0, a.accum[0]); put(E_AC1, a.accum[1]);
        putp(E_DST0, a.dst[0]); putp(E_DST1, a.dst[1]); putp(E_MASK0, a.mask[0]); putp(E_MASK1, a.mask[1]); putp(E_ADD, a.addsrc); putp(E_BIAS, a.bias);
        putp(E_PDST, a.pool_dst); putp(E_PCODE, a.pool_codes);
    }
    struct EpiArgs {
        int OH, OW, DH, DW, Ntot, n_split, act, pool_cs, cs0, cs1, mm0, mm1, ac0, ac1;
        float *dst0, *dst1, *pool_dst; const float *mask0, *mask1, *addsrc, *bias; unsigned char* pool_codes;
        __device__ int dst_cs(int du) const { return du ? cs1 : cs0; }
        __device__ int mask_mode(int du) const { return du ? mm1 : mm0; }
        __device__ int accum(int du) const { return du ? ac1 : ac0; }
        __device__ float* dst(int du) const { return du ? dst1 : dst0; }
        __device__ const float* mask(int du) const { return du ? mask1 : mask0; }
    };
    auto epi_args = [&]() {
        auto rl = [&](int idx) { return (int)__builtin_amdgcn_readlane((int)argv, idx); };
        auto rp = [&](int idx) { return (uintptr_t)(unsigned)rl(idx) | ((uintptr_t)(unsigned)rl(idx + 1) << 32); };
        EpiArgs e;
        e.OH = rl(E_OH); e.OW = rl(E_OW); e.DH = rl(E_DH); e.DW = rl(E_DW); e.Ntot = rl(E_NTOT); e.n_split = rl(E_NSPLIT); e.act = rl(E_ACT);
        e.pool_cs = rl(E_POOLCS); e.cs0 = rl(E_CS0); e.cs1 = rl(E_CS1); e.mm0 = rl(E_MM0); e.mm1 = rl(E_MM1); e.ac0 = rl(E_AC0); e.ac1 = rl(E_AC1);
        e.dst0 = (float*)rp(E_DST0); e.dst1 = (float*)rp(E_DST1); e.mask0 = (const float*)rp(E_MASK0); e.mask1 = (const float*)rp(E_MASK1);
        e.addsrc = (const float*)rp(E_ADD); e.bias = (const float*)rp(E_BIAS); e.pool_dst = (float*)rp(E_PDST); e.pool_codes = (unsigned char*)rp(E_PCODE);
        return e;
    };


    // ---- epilogue of tile `tl`, straight from the accumulators: bias, activation, act' mask, residual and accumulation are float4 arithmetic
    // on the accumulator registers, every block goes out as one 16-byte store per lane (16 pixels x 64 bytes per instruction).  The fused
    // MaxPool2d(2) takes the other pixel of a pair from the neighbouring lane (DPP) and the other row from the wave's second accumulator row.
    auto epilogue = [&](const Tile& tl) __attribute__((always_inline)) {
        const EpiArgs ea = epi_args();
        const int b = tl.b, n0 = tl.n0;
        const int p16 = lane & 15, c4 = (lane >> 4) * 4;
        const int py0 = tl.y0 + wave * MT, px0 = tl.x0 + p16;
        int du_[NT], chw_[NT], cs_[NT]; bool blk_[NT];
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
            du_[k] = nwv >= ea.n_split ? 1 : 0; chw_[k] = nwv - (du_[k] ? ea.n_split : 0); cs_[k] = ea.dst_cs(du_[k]); blk_[k] = nwv < ea.Ntot;
        }
        // byte offset of this lane's pixel (row i, 16-pixel half h) and channel quad in the destination of 32-column block k, or out of range;
        // the 16-column block inside it (+ 64 bytes) goes through the instruction's scalar offset
        unsigned vo[NT][MT][2];
#pragma unroll
        for (int k = 0; k < NT; ++k)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = blk_[k] && py0 + i < ea.DH && px0 + 16 * h < ea.DW;
                    vo[k][i][h] = ok ? (unsigned)((((py0 + i) * ea.OW + px0 + 16 * h) * cs_[k] + chw_[k] + c4) * 4) : OOB;
#ifdef X3S_EPI_OOB                 // timing experiment only (wrong results): every store / mask request is dropped by the range check
                    vo[k][i][h] |= OOB;
#endif
                }
        auto rsrc = [&](const float* base, int k) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)b * ea.OH * ea.OW * cs_[k]), 0, ea.OH * ea.OW * cs_[k] * 4, 0x00020000);
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
        auto take = [&](int mb, int j) { const f32x4 v = acc[mb][j]; acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; return v; };
        // (starting the next tile's first products from a zero constant instead -- mfma_row's FIRST -- costs a second copy of filter row 0 whose
        //  accumulators the register allocator does not merge with the loop's: 44-65 spilled registers)
        // ---- full-line memory pattern (FWD / BWD / POOL: see the comment in front of the FWD / BWD block)
        const bool lo8 = p16 < 8;
        auto ror8 = [&](f32x4 v) {                               // (inline assembly: see the pool path about __builtin_amdgcn_update_dpp)
            float r0, r1, r2, r3;
            asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                         "v_mov_b32_dpp %2, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %7 row_ror:8 row_mask:0xf bank_mask:0xf"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            return f32x4{r0, r1, r2, r3};
        };
        auto sel = [&](bool c, f32x4 x, f32x4 y) { return f32x4{c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w}; };
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
            const int ph = ea.OH >> 1, pwd = ea.OW >> 1;
            const int64_t pimg = (int64_t)b * ph * pwd * ea.pool_cs;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_dst + pimg), 0, ph * pwd * ea.pool_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_codes + pimg), 0, ph * pwd * ea.pool_cs, 0x00020000);
#pragma unroll
            for (int k = 0; k < NT; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 wn[2][2];                                  // [16-column block of the pair][row]
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < 2; ++i) wn[jj][i] = act4(take(2 * i + h, 2 * k + jj) + bias4[2 * k + jj]);
#pragma unroll
                    for (int i = 0; i < 2; ++i) {                     // full resolution: whole lines (the halves of the block pair traded)
                        const f32x4 ox = ror8(sel(lo8, wn[1][i], wn[0][i]));
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, wn[0][i], ox)), rd, wo[k][i][h], 0, X3S_STORE_AUX);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, ox, wn[1][i])), rd, wo2(k, i, h), 0, X3S_STORE_AUX);
                    }
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * k + jj;
                        const f32x4 (&win)[2] = wn[jj];
                        f32x4 nbr[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            // the pixel to the right (even lanes) / left (odd lanes): quad_perm [1, 0, 3, 2].  As inline assembly (with the two wait
                            // states a DPP read needs behind the VALU write of its source): through __builtin_amdgcn_update_dpp the compiler's DPP
                            // combiner folded the four moves of a float4 into consumers reading element 0 (ROCm 7.2, caught by the pool parity test)
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
                        const bool ok2 = !(lane & 1) && blk_[k] && py0 < ea.DH && px < ea.DW;      // even sizes: the whole window is inside or outside
                        const unsigned po = (unsigned)(((py0 >> 1) * pwd + (px >> 1)) * ea.pool_cs + n0 + 16 * j + c4);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mx), rp, ok2 ? po * 4u : OOB, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(code, rc, ok2 ? po : OOB, 0, 0);
                    }
                }
            return;
        }
        // ---- FWD: no mask, no accumulation, no residual (every forward layer);  BWD: act' masks (a destination without one requests them out
        // of range: zeros come back, no memory traffic), nothing else.  All mask requests first, then add / max / select / store per block.
        if constexpr (EK == EK_FWD || EK == EK_BWD) {
            constexpr bool MASKED = EK == EK_BWD;
            // FULL-LINE memory pattern.  Straight from the accumulators a 16-byte store instruction covers 16 pixels x 64 bytes -- sixteen half
            // lines -- and a CU then stores 21 bytes per cycle where 8 pixels x 128 bytes run at 63 and 1 KB contiguous at 84
            // (tools/ubench/store_rate.hip, profiles/r4/store_rate.txt): 6000 of a 64-column tile's cycles.  So the two 16-column blocks of a
            // 32-column block trade halves first: lanes p < 8 of a 16-lane row send the UPPER block of their pixel to lane p + 8 and get the
            // LOWER block of pixel p + 8 back (one DPP row rotation by 8).  Instruction 1 then writes pixels 0-7 (lanes p < 8: their own lower
            // quads, lanes p >= 8: the upper quads of pixel p - 8), instruction 2 pixels 8-15: eight whole 128-byte lines each.  The act' masks
            // come in by the same pattern and are traded back.
            f32x4 mk[MASKED ? MB : 1][MASKED ? NB : 1];              // [.][2 k] = what instruction 1 fetched, [.][2 k + 1] = instruction 2
            if constexpr (MASKED) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int mm = ea.mask_mode(du_[k]);
                    const __amdgpu_buffer_rsrc_t rm = rsrc(mm ? ea.mask(du_[k]) : ea.dst(du_[k]), k);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            mk[2 * i + h][2 * k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm ? wo[k][i][h] : OOB, 0, 0));
                            mk[2 * i + h][2 * k + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, mm ? wo2(k, i, h) : OOB, 0, 0));
                        }
                }
            }
            auto body = [&](auto act_tag) __attribute__((always_inline)) {
                constexpr bool ACT = decltype(act_tag)::value;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(du_[k]), k);
                    const int mm = ea.mask_mode(du_[k]);
                    const float msl = mm == 1 ? 0.2f : (mm == 0 ? 1.f : 0.f);      // act'(x <= 0); a destination without a mask (its requests came back as zeros): 1
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 o0 = take(2 * i + h, 2 * k), o1 = take(2 * i + h, 2 * k + 1);
                            if constexpr (!MASKED) { o0 += bias4[2 * k]; o1 += bias4[2 * k + 1]; }      // (backward-data has no bias: the launcher checks)
                            if constexpr (ACT) { o0 = act4(o0); o1 = act4(o1); }
                            if constexpr (MASKED) {
                                const f32x4 m1 = mk[2 * i + h][2 * k], m2 = mk[2 * i + h][2 * k + 1], mx = ror8(sel(lo8, m2, m1));
                                const f32x4 q0 = sel(lo8, m1, mx), q1 = sel(lo8, mx, m2);      // the masks of this lane's lower / upper block
                                const f32x4 t0 = o0 * msl, t1 = o1 * msl;
#pragma unroll
                                for (int c = 0; c < 4; ++c) { o0[c] = q0[c] > 0.f ? o0[c] : t0[c]; o1[c] = q1[c] > 0.f ? o1[c] : t1[c]; }
                            }
                            const f32x4 ox = ror8(sel(lo8, o1, o0));
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, o0, ox)), rd, wo[k][i][h], 0, X3S_STORE_AUX);
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sel(lo8, ox, o1)), rd, wo2(k, i, h), 0, X3S_STORE_AUX);
                        }
                }
            };
            // one wave-uniform branch per tile: with / without an activation (backward-data never has one: the launcher sends a masked layer
            // WITH an activation to the general kernel)
            if constexpr (MASKED) body(std::false_type{});
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
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[k][i][h], jj * 64, 0);
                    }
            }
        }
    };

    // ---- the consumers' loop: barrier, item, barrier, item, ...  (no vector-memory wait anywhere: the only operations a consumer has in
    // flight are its own epilogue's, and nothing here depends on them)
    int img = 0, st = 0;
#ifdef X3S_STAMPS
    long long t_mfma = 0, t_epi = 0, t_bar = 0, tlast_ = clock64(), tall = tlast_; int nch = 0;
#endif
    X3S_BARRIER();                                                // barrier 0
    X3S_T(t_bar)
    for (;;) {
        const Ck n1 = chunk_at(1);
#ifdef X3S_STAMPS
        ++nch;
#endif
        mfma_row(0, st, img, std::false_type{});
        X3S_T(t_mfma)
        X3S_BARRIER();
        X3S_T(t_bar)
        mfma_row(1, (st + 1) % NSTAGE, img, std::false_type{});
        X3S_T(t_mfma)
        X3S_BARRIER();
        X3S_T(t_bar)
        mfma_row(2, (st + 2) % NSTAGE, img, std::false_type{});
        X3S_T(t_mfma)
        if (g == nchunks - 1) epilogue(cur);
        X3S_T(t_epi)
        if (!n1.ok) break;
        X3S_BARRIER();
        X3S_T(t_bar)
        if (g == nchunks - 1) next_tile(); else ++g;
        img ^= 1; st = (st + 3) % NSTAGE;
    }
#ifdef X3S_STAMPS
    __builtin_amdgcn_s_waitcnt(0x0f70);
    if (lane == 0) {
        float* d = a.dst[0] + ((int64_t)blockIdx.x * (NCW + NPW) + wave) * 8;
        d[0] = (float)t_mfma; d[1] = (float)t_epi; d[2] = (float)t_bar; d[3] = 0.f; d[4] = (float)(clock64() - tall); d[5] = (float)nch;
    }
#endif
}

template <int BN, int EK>
int launch_x3s(const IgemmArgs& a, hipStream_t s) {
    using Cfg = SCfg<BN>;
    auto kern = igemm_x3s_kernel<BN, EK>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * ((a.Ntot + BN - 1) / BN);
    if (tiles <= 0) return PNNP_OK;
    const int wgs = pnnp_persistent_grid(tiles);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

}  // namespace

// Same contract as pnnp_igemm_x3_launch (csrc/conv_x3.hip), which validates the arguments and forwards here.
int pnnp_igemm_x3s_launch(const IgemmArgs& b, int wide, hipStream_t s) {
    if (b.pool_dst) return wide ? launch_x3s<64, EK_POOL>(b, s) : launch_x3s<32, EK_POOL>(b, s);
    const bool two = b.dst[1] != nullptr;
    const bool plain = !b.addsrc && !b.accum[0] && !(two && b.accum[1]);
    const bool any_mask = b.mask_mode[0] || (two && b.mask_mode[1]);
    if (plain && !any_mask) return wide ? launch_x3s<64, EK_FWD>(b, s) : launch_x3s<32, EK_FWD>(b, s);
    if (plain && !b.act && !b.bias) return wide ? launch_x3s<64, EK_BWD>(b, s) : launch_x3s<32, EK_BWD>(b, s);
    return wide ? launch_x3s<64, EK_GEN>(b, s) : launch_x3s<32, EK_GEN>(b, s);
}
