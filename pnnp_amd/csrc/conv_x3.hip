// 3x3 convolution (forward / backward-data) with fp32 operands SPLIT INTO THREE bf16 PIECES, on the bf16 matrix cores.
//
// Why: on gfx950 the fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate -- 157 TFLOP/s, 1/16 of the bf16
// MFMA rate, and it blocks the SIMD's vector issue while it runs.  A float32 value a is EXACTLY
//      a = hi + mid + lo,   hi = bf16(a), mid = bf16(a - hi), lo = bf16(a - hi - mid)
// (3 x 8 significand bits + signs cover the 24-bit significand; both subtractions are exact in fp32), so a product of two
// float32 values is the sum of nine bf16 x bf16 products, each of which the matrix core forms exactly and adds into an
// fp32 accumulator.  Six of the nine are kept:
//      a*b ~= hi*hi' + (hi*mid' + mid*hi') + (hi*lo' + lo*hi' + mid*mid')          dropped: mid*lo', lo*mid', lo*lo' <= 2^-24 |ab|
// i.e. the truncation is one fp32 rounding per product -- the same order as the rounding of the fp32 MFMA's own multiply --
// while v_mfma_f32_32x32x16_bf16 does 8x the multiply-adds of the fp32 instruction in half its cycles: 6 passes cost
// 6/16 of the fp32 MFMA time (2.67x), and the VALU / LDS / global work of the loop now runs UNDER the matrix pipe instead
// of in front of it.  Results stay float32-accurate (tests: same tolerances as the fp32-MFMA kernels, and the 512x512
// gradient test measures the error against a float64 run of the reference next to the reference's own float32 error).
//
// Structure (reference ops: archs/Unet.py:16-52 Conv2d 3x3 pad 1 (+LeakyReLU), archs/modules.py:130-197):
//   M = output pixels: tile of 16 rows x 32 px; N = BN = 32 or 64 output channels; K = 16 channels x one filter ROW
//   (3 taps) per work item.  ONE persistent workgroup of 8 waves per CU (each wave 2 pixel rows x BN channels; two waves
//   per SIMD that meet at a barrier every item -- two independent 4-wave workgroups per CU ran unfairly: the older one
//   wins the matrix-pipe arbitration, finishes its tiles early and leaves the other alone at half occupancy) walks
//   (tile, channel chunk, filter row) items:
//     * activations: fp32 NHWC halo tile (18 x 34 px x 16 ch) global -> registers (buffer loads, hardware zero for the
//       halo outside the image), requested at the first filter row of the PREVIOUS chunk; split into hi/mid/lo and
//       written to the other of two LDS images xs[buf][k-octet][piece][pixel][8 bf16] in slices BETWEEN the MFMAs of that
//       chunk's last filter row (measured: done in one lump behind a barrier the split cost 8 % of a 64-channel layer
//       and 30-40 % of a 32-channel one -- every wave stages at the same time and the matrix pipe idles);
//     * weights: pre-split and pre-ordered per (32-channel block, chunk, tap) at pack time (csrc/pack_jobs.hip kind 2), so
//       an item's 9 / 18 KB come in by LDS-DMA (buffer_load ... lds, no registers, no VALU), double buffered, one item ahead;
//     * one ds_read_b128 = the 8 k-values a lane feeds to one v_mfma_f32_32x32x16_bf16; per filter tap a wave reads
//       (2 + BN/32) x 3 operands and issues 2 x BN/32 x 6 MFMAs;
//     * epilogue as in csrc/conv_igemm.hip (bias, activation, act' mask, residual, accumulate, split destinations,
//       16-byte stores through a wave-private LDS patch).
#include "igemm.h"
#include <type_traits>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

int pnnp_igemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);
int pnnp_igemm_x3s_launch(const IgemmArgs& a, int wide, hipStream_t s);      // csrc/conv_x3s.hip: the same tile with specialised waves
#ifndef X3_SPEC
#define X3_SPEC 1                  // 1: the 3x3 layers run on csrc/conv_x3s.hip (producer / consumer waves); 0: on the kernel below
#endif

namespace {

#ifndef X3_NWAVE
#define X3_NWAVE 8                 // waves per workgroup: 8 (two per SIMD, 2 pixel rows each) or 4 (one per SIMD, 4 rows each, 512 registers)
#endif
constexpr int NWAVE = X3_NWAVE, NTHR = 64 * NWAVE, MT = 16 / NWAVE;
constexpr int TH = NWAVE * MT, HR = TH + 2, HC = 34, NPIX = HR * HC;   // 16-row tile, 612 halo pixels
#ifndef X3_FILLMODE
#define X3_FILLMODE 0              // BN = 64: which waves stage the next chunk's halo in which filter row (see the main loop)
#endif
#define X3_VMCNT(N) (0x0f70 | ((N) & 15) | (((N) >> 4) << 14))      // s_waitcnt vmcnt(N) alone (N <= 63: the counter's high bits sit at 15:14)
#ifdef X3_NOBAR                    // timing experiment only (racy, wrong results): no per-item barriers -- what would ANY relaxation of them buy?
#define X3_SYNC()
#else
#define X3_SYNC() __syncthreads()
#endif
#ifndef X3_DEFER32
#define X3_DEFER32 0               // experiment (measured neutral, costs 40 registers): BN = 32: a tile's output stores go out between the next chunk's MFMAs (see `pend`)
#endif
#ifndef X3_M16
#define X3_M16 1                   // 1: v_mfma_f32_16x16x32_bf16 with two PIECES concatenated along K (see mfma_row16); 0: v_mfma_f32_32x32x16_bf16
#endif
#ifndef X3_DIRECT
#define X3_DIRECT 0                // 0: rounds 2-4's epilogue through an LDS patch (what this kernel -- now the fallback of csrc/conv_x3s.hip -- shipped and was
                                   //    tested with for three rounds); 1: stores straight from the accumulators, operands swapped in the MFMA (the first,
                                   //    neutral step towards conv_x3s: profiles/r4/ab_specialised_waves.txt)
#endif
#if X3_DIRECT && !X3_M16
#error "X3_DIRECT reads the 16x16 accumulator layout"
#endif
#if X3_M16
// halo image in 16-byte words: [piece 3][k-octet 2][pixel, plane padded to a multiple of 16 words].  A 16x16x32 operand read takes
// lanes 0-15 / 16-31 of a bank group from the two octet planes: with the plane stride a multiple of 256 bytes the ds_read_b128 is
// conflict-free (the lane groups of the instruction are {0-3,12-15,20-27} and {4-11,16-19,28-31}).
constexpr int NPIXP = (NPIX + 15) / 16 * 16;                       // 624
constexpr int XS_F4 = 3 * 2 * NPIXP;
#define XS_PLANE(piece, oct) (((piece) * 2 + (oct)) * NPIXP)
constexpr int XS_PIECE_STRIDE = 2 * NPIXP;
#else
constexpr int NPIXP = NPIX;
constexpr int XS_F4 = 2 * 3 * NPIX;                                // one halo image in 16-byte words: [k-octet 2][piece 3][pixel]
#define XS_PLANE(piece, oct) (((oct) * 3 + (piece)) * NPIX)
constexpr int XS_PIECE_STRIDE = NPIX;
#endif
constexpr int XS_BYTES = XS_F4 * 16;                               // 58752 (59904 with padded planes)
constexpr int WBLK = 3 * 2 * 3 * 32 * 16;                          // one filter row of one 32-channel block: [tap 3][octet 2][piece 3][32][16 B] = 9216
constexpr int NSLOT = (2 * NPIX + NTHR - 1) / NTHR;                 // halo staging slots per thread: 1224 (pixel, octet) pairs / 512 -> 3
constexpr int NSLICE = 4 * NSLOT;                                  // staging slices (two floats each) per chunk

template <int BN> struct X3Cfg {
    static constexpr int NT = BN / 32;
    static constexpr int WS_STAGE = NT * WBLK;                     // 9216 / 18432
    static constexpr int NDMA = WS_STAGE / 1024;                   // 1 KB LDS-DMA pieces per stage
    static constexpr int DPW = (NDMA + NWAVE - 1) / NWAVE;         // LDS-DMA instructions per wave and item: 2 / 3
    // Weight ring: BN = 64 requests an item's weights one item ahead (an item is 72 MFMAs per wave, ~2.3 us with two waves
    // per SIMD: more than the L2 round trip); a BN = 32 item is half as long, so its weights are requested TWO items ahead.
    static constexpr int NSTAGE = BN == 32 ? 3 : 2, AHEAD = NSTAGE - 1;
#if X3_M16
    // epilogue patch per wave: 16 pixels x 32 channels of floats; rows padded to 36 floats where the patch aliases a weight stage
    // (a 16x16 accumulator block puts pixel 4 q + r on lane group q: with 32-float rows the four groups write the same banks)
    static constexpr int EPS = BN == 64 ? 36 : 32;
#else
    static constexpr int EPS = 32;
#endif
    static constexpr int EPI = NWAVE * 16 * EPS * 4;               // per wave: (16 pixels x 32 channels) floats
    // BN = 64: the epilogue patches live in the weight stage the tile's last item has just consumed (a barrier in between)
    static constexpr bool EPI_ALIAS = WS_STAGE >= EPI;
#if X3_DIRECT
    static constexpr int LDS_BYTES = 2 * XS_BYTES + NSTAGE * WS_STAGE;                               // no epilogue patch: 156672 (BN = 64) / 147456 (BN = 32)
#else
    static constexpr int LDS_BYTES = 2 * XS_BYTES + NSTAGE * WS_STAGE + (EPI_ALIAS ? 0 : EPI);      // X3_M16 (padded planes): 156672 (BN = 64) / 163840 (BN = 32: ALL of the LDS)
#endif
    static_assert(LDS_BYTES <= 160 * 1024, "a workgroup's LDS: 160 KB on gfx950 (BN = 32 uses every byte: any growth must come out of something else)");
};

// which staging units (slice, step) of the next chunk's halo a filter row carries in its MFMA gaps: units LO .. HI - 1 of the 6 NSLICE
template <int LO, int HI> struct Fill { static constexpr int lo = LO, hi = HI; static constexpr bool value = HI > LO; };
using FillNone = Fill<0, 0>;
using FillAll = Fill<0, 6 * NSLICE>;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {                  // f(integral_constant<int, I>) ... for I .. N - 1: every index a constant
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // RNE, low half = a
    unsigned r; asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r;
}

// two floats -> one dword each of the hi / mid / lo words (a = hi + mid + lo exactly; both subtractions are exact)
__device__ __forceinline__ void split2(float a0, float a1, unsigned& h, unsigned& m, unsigned& l) {
    h = cvt_pk_bf16(a0, a1);
    const float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    m = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    l = cvt_pk_bf16(s0, s1);
}

// EK: which epilogue the kernel carries -- ONE straight-line path per instantiation, so that the waits behind it can count its stores (see
// `epilogue`): 0 forward (bias + activation), 1 backward-data (act' mask on every destination), 2 anything else (residual, accumulation,
// a mask on one destination only), 3 forward + the fused MaxPool2d(2)
enum { EK_FWD = 0, EK_BWD = 1, EK_GEN = 2, EK_POOL = 3 };
template <int BN, int EK>
__global__ void __launch_bounds__(NTHR, 1)
igemm_x3_kernel(const IgemmArgs a) {
    constexpr bool POOL = EK == EK_POOL;
    using Cfg = X3Cfg<BN>;
    constexpr int NT = Cfg::NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* xs = reinterpret_cast<u32x4*>(smem);                     // two halo images
    char* wsb = smem + 2 * XS_BYTES;                                // two weight stages
    [[maybe_unused]] float* epi_sep = reinterpret_cast<float*>(smem + 2 * XS_BYTES + Cfg::NSTAGE * Cfg::WS_STAGE);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform (the LDS-DMA pieces depend on it)
    [[maybe_unused]] const int l31 = lane & 31, half = lane >> 5;       // (the 32x32x16 build's operand geometry)

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int G = gridDim.x;
    const int nchunks = a.nseg * a.chunks_per_seg;                  // 16-channel chunks of K

    // ---- per-thread constants of the halo staging pattern: slot s = tid + 512 k -> (pixel s>>1, channel octet s&1)
    // (1224 slots over 3 x 512: a thread whose third slot would be past the end repeats its second one -- same address, same
    //  data, same thread -- so every slot is live and the staging code has no branches)
    constexpr unsigned OOB = 0x80000000u;
    int rk[NSLOT], qk[NSLOT]; unsigned pixk[NSLOT]; int xdst[NSLOT];
    const int oct = tid & 1;
#pragma unroll
    for (int k = 0; k < NSLOT; ++k) {
        int s = tid + NTHR * k;
        if (s >= 2 * NPIX) s -= NTHR;
        const int pix = s >> 1;
        const int r = pix / HC, q = pix - r * HC;
        rk[k] = r - 1;
        qk[k] = q - 1;
        pixk[k] = (unsigned)(r * a.IW + q);
        xdst[k] = XS_PLANE(0, oct) + pix;                           // + piece * XS_PIECE_STRIDE (+ image * XS_F4)
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
    // A workgroup's tiles are t, t + G, t + 2 G, ...: the step G is decoded ONCE and the next tile is a mixed-radix addition with carries
    // (a dozen scalar instructions) instead of three integer divisions by run-time values (cycle stamps of round 4: with decode() reachable
    // from the top of the chunk loop that top took 1100 cycles per chunk in the BN = 32 kernel, 8 % of a chunk).
    // (Tiles are picked FIELD BY FIELD: `c ? tileA : tileB` on whole structs becomes a select of ADDRESSES, which parks them in scratch memory.)
    auto pick = [](bool c, const Tile& x, const Tile& y) { Tile o; o.b = c ? x.b : y.b; o.y0 = c ? x.y0 : y.y0; o.x0 = c ? x.x0 : y.x0; o.n0 = c ? x.n0 : y.n0; return o; };
    const Tile gstep = decode(G);
    auto advance = [&](Tile o) {
        o.n0 += gstep.n0; if (o.n0 >= n_tiles * BN) { o.n0 -= n_tiles * BN; o.x0 += 32; }
        o.x0 += gstep.x0; if (o.x0 >= tiles_x * 32) { o.x0 -= tiles_x * 32; o.y0 += TH; }
        o.y0 += gstep.y0; if (o.y0 >= tiles_y * TH) { o.y0 -= tiles_y * TH; o.b += 1; }
        o.b += gstep.b;
        return o;
    };

#ifdef X3_STAMPS                  // debug build: where does an item's time go?  (cycle sums per wave, dumped into dst[0] at the end)
    long long tw = 0, tb = 0, tm = 0, tm1 = 0, tm2 = 0, te = 0, tea = 0, teb = 0, ter[5] = {0, 0, 0, 0, 0}, tall = clock64();
#define X3_T(v) { const long long now_ = clock64(); v += now_ - tlast_; tlast_ = now_; }
    long long tlast_ = clock64();
    auto dump_stamps = [&]() {
        __syncthreads();
        if (lane == 0) {
            float* d = a.dst[0] + ((int64_t)blockIdx.x * NWAVE + wave) * 16;
            d[0] = (float)tw; d[1] = (float)tb; d[2] = (float)tm; d[3] = (float)te; d[4] = (float)(clock64() - tall); d[5] = (float)tm1; d[6] = (float)tm2;
            d[7] = (float)tea; d[8] = (float)teb; for (int i = 0; i < 5; ++i) d[9 + i] = (float)ter[i];
        }
    };
#else
#define X3_T(v)
#endif
    f32x4 ra[NSLOT][2];                                             // halo registers of the NEXT chunk (8 channels per slot)
    u32x4 sh[NSLOT], sm[NSLOT], sl[NSLOT];                          // their hi / mid / lo words while the split is in progress

    // global loads of the halo tile of (tile, chunk g) -> ra (no wait): the scalar part, then one piece per staging slot
    __amdgpu_buffer_rsrc_t h_rs; int h_soff, h_rlo, h_rhi, h_qlo, h_qhi, h_cvalid; unsigned h_cs4;
    auto halo_prep = [&](const Tile& tl, int g) {
        const int si = g / a.chunks_per_seg, cc = g - si * a.chunks_per_seg;
        const IgemmSeg sg = a.seg[si];
        const int c0 = sg.coff + cc * 16;
        h_rlo = -tl.y0; h_rhi = a.IH - tl.y0; h_qlo = -tl.x0; h_qhi = a.IW - tl.x0;
        const int shift = (2 * a.IW + 2) * sg.cstride;             // the resource starts before the image: the scalar offset below stays >= 0
        h_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.ptr + ((int64_t)tl.b * a.IH * a.IW * sg.cstride - shift)), 0, 0x7fffffff, 0x00020000);
        h_soff = (((tl.y0 - 1) * a.IW + tl.x0 - 1) * sg.cstride + c0 + shift) * 4;
        h_cs4 = (unsigned)sg.cstride * 4u;
        h_cvalid = a.seg_channels - cc * 16 - oct * 8;             // > 0: this thread's octet exists (the last chunk of a segment may be half empty)
    };
    auto halo_slot = [&](int k) {
        // validity as ONE bitwise expression (a short-circuit && chain becomes per-lane branches around the loads, and the
        // register allocator then serialises the staging with vmcnt(0) waits)
        const int bad = (rk[k] - h_rlo) | (h_rhi - 1 - rk[k]) | (qk[k] - h_qlo) | (h_qhi - 1 - qk[k]) | (h_cvalid - 1);     // sign bit set <=> outside
        const unsigned vo = bad < 0 ? OOB : __umul24(pixk[k], h_cs4) + oct * 32;
        ra[k][0] = bload(h_rs, vo, h_soff);
        ra[k][1] = bload(h_rs, vo, h_soff + 16);
    };
    auto load_halo = [&](const Tile& tl, int g) {
        halo_prep(tl, g);
#pragma unroll
        for (int k = 0; k < NSLOT; ++k) halo_slot(k);
    };
    // One of the 12 slices of the halo staging: split two floats of slot q/4 (pair q%4); after a slot's fourth pair its three
    // 16-byte words go to halo image `img`.  Sliced so that it can sit between MFMA groups (~12 VALU + at most 3 LDS stores each).
    auto stage_slice = [&](int q, int img) {
        const int k = q >> 2, p = q & 3;
        const f32x4 v = ra[k][p >> 1];
        unsigned h, m, l;
        split2(v[(p & 1) * 2], v[(p & 1) * 2 + 1], h, m, l);
        sh[k][p] = h; sm[k][p] = m; sl[k][p] = l;
        if (p == 3) {
            u32x4* d = xs + img * XS_F4 + xdst[k];
            d[0] = sh[k]; d[XS_PIECE_STRIDE] = sm[k]; d[2 * XS_PIECE_STRIDE] = sl[k];
        }
    };
    // the same slice as five dependent pieces of 1-4 VALU instructions (step 0 .. 4) + the stores (step 5), one piece per MFMA gap
    float pa0[2], pa1[2];
    auto stage_piece = [&](int q, int u, int step, int img) {
#ifdef X3_SKIP_STORE              // timing experiment only (wrong results): no staging in the loop (see the prologue)
        return;
#endif
        const int k = q >> 2, p = q & 3;
        if (step == 0) {
            const f32x4 v = ra[k][p >> 1];
            pa0[u] = v[(p & 1) * 2]; pa1[u] = v[(p & 1) * 2 + 1];
            sh[k][p] = cvt_pk_bf16(pa0[u], pa1[u]);
        } else if (step == 1) {
            pa0[u] -= __uint_as_float(sh[k][p] << 16); pa1[u] -= __uint_as_float(sh[k][p] & 0xffff0000u);
        } else if (step == 2) {
            sm[k][p] = cvt_pk_bf16(pa0[u], pa1[u]);
        } else if (step == 3) {
            pa0[u] -= __uint_as_float(sm[k][p] << 16); pa1[u] -= __uint_as_float(sm[k][p] & 0xffff0000u);
        } else if (step == 4) {
            sl[k][p] = cvt_pk_bf16(pa0[u], pa1[u]);
        } else if (p == 3) {
            u32x4* d = xs + img * XS_F4 + xdst[k];
            d[0] = sh[k]; d[XS_PIECE_STRIDE] = sm[k]; d[2 * XS_PIECE_STRIDE] = sl[k];
        }
    };
    // LDS-DMA of the weights of item (tile n0, chunk g, filter row tr) into stage st: per 32-channel block 9216 contiguous
    // bytes of the pack, as 1 KB pieces dealt over the 8 waves
    const int K16 = nchunks;
    auto dma_piece = [&](const Tile& tl, int g, int tr, int st, bool valid, int i) {
#ifdef X3_SKIP_DMA                // timing experiment only (wrong results): weights are never refreshed
        if (g + tr > 0) valid = false;
#endif
        {
            // wave-uniform piece; past the end a wave repeats the last piece (same bytes to the same place) instead of branching
            const int ins = min(wave + NWAVE * i, Cfg::NDMA - 1);
            const int j = ins / 9, r = ins - 9 * j;
            const int nb = (tl.n0 >> 5) + j;
            const bool ok = valid && nb * 32 < a.Ntot;           // (an invalid request still issues: the vmcnt bookkeeping below counts instructions)
            const int soff = ok ? ((nb * K16 + g) * 27648 + tr * WBLK + r * 1024) : 0;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wsb + st * Cfg::WS_STAGE + ins * 1024),
                                                     16, ok ? (unsigned)lane * 16u : OOB, soff, 0, 0);
        }
    };
    auto dma_weights = [&](const Tile& tl, int g, int tr, int st, bool valid = true) {
#pragma unroll
        for (int i = 0; i < Cfg::DPW; ++i) dma_piece(tl, g, tr, st, valid, i);
    };

#if X3_M16
    // 16 x 16 accumulator blocks: acc[2 i + h][j] = pixel row i of the wave, 16-pixel half h, channels 16 j .. 16 j + 15;
    // lane l holds channel l & 15 of pixels 4 (l >> 4) + r, r = 0 .. 3
    constexpr int MB = 2 * MT, NB = BN / 16;
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r16 = lane & 15, q16 = lane >> 4, oct16 = q16 & 1, ps16 = q16 >> 1;
    // operand forms (two pieces concatenated along K = 32: k-blocks 0,1 = the 16 channels of the first piece, 2,3 = of the second):
    //   A form 0 = [hi | mid], form 1 = [hi | lo];   B form 0 = [hi' | hi'], 1 = [mid' | mid'], 2 = [lo' | hi']
    //   A1 B2 = hi lo' + lo hi',  A0 B1 = hi mid' + mid mid',  A0 B0 = hi hi' + mid hi'  -- the six products of the bf16x3 scheme
    const int aoff0 = XS_PLANE(ps16 ? 1 : 0, oct16) + r16, aoff1 = XS_PLANE(ps16 ? 2 : 0, oct16) + r16;            // 16-byte words
    const int boff0 = ((oct16 * 3 + 0) * 32 + r16) * 16, boff1 = ((oct16 * 3 + 1) * 32 + r16) * 16,
              boff2 = ((oct16 * 3 + (ps16 ? 0 : 2)) * 32 + r16) * 16;                                            // bytes inside one tap
    // the 8 accumulator values of a lane for (pixel row i, 32-column block k, 16-pixel half h2) -> the wave's patch [16 px][EPS]
    [[maybe_unused]] auto spill_half = [&](float* eb, int i, int k, int h2) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                eb[(4 * q16 + r) * Cfg::EPS + nb * 16 + r16] = acc[2 * i + h2][2 * k + nb][r];
                acc[2 * i + h2][2 * k + nb][r] = 0.f;
            }
    };
#else
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto spill_half = [&](float* eb, int i, int k, int h2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            eb[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[i][k][8 * h2 + r];
            acc[i][k][8 * h2 + r] = 0.f;
        }
    };
#endif

#if X3_M16
    // MFMA over the three taps of filter row tr on v_mfma_f32_16x16x32_bf16 (halo image img, weight stage st).
    // K = 32 of one instruction = the tap's 16 channels of one piece ++ the same 16 channels of another piece (the forms above), so
    // the six piece products of a (16 px x 16 ch) block are THREE instructions, and a chunk stays 16 channels -- no tap pairing,
    // no second halo image.  The 16x16x32 shape does the same multiply-adds per cycle as 32x32x16 at less energy per FLOP (the
    // chip is power-limited here: DESIGN section 5; bare loops on random data: +9-14 %).
    // Order per tap: pass j (16 output channels) x pixel block mb x the three products, smallest terms first.  Operands: the 8 A
    // words of the tap (4 pixel blocks x 2 forms) stay in registers for all passes and are refreshed IN PLACE for the next tap
    // during the last pass (form 1 right after its only use, form 0 after the block's third MFMA: >= 10 gaps ahead of their next
    // use); the 3 B words of a pass are read one pass ahead into the other of two register sets.  Everything that is not an MFMA
    // sits in the gaps between MFMAs, one or two instructions per gap, fenced (as in the 32x32x16 version below):
    //   reads as just described; FILL: the 72 (slice, step) units of the next chunk's halo staging (one per gap at BN = 32, every
    //   other gap at BN = 64); `requests`: the item's LDS-DMA / halo-load pieces in read-free gaps of pass 1 (BN = 64) or 0.
    auto mfma_row = [&](int tr, int st, int img, auto fill_tag, auto halo_tag, auto&& requests, auto&& hook) {
        constexpr bool FILL = decltype(fill_tag)::value;
        // halo requests of the row: 0 none; 1 all NSLOT + 1 pieces in tap 1 (BN = 64, filter row 0: the registers are free);
        // 2 LATE (BN = 32, filter row 2, whose gaps also carry the staging that empties those registers slot by slot): the scalar part and
        // slot 0 in tap 1 (slot 0's four slices were staged in tap 0), slot 1 in tap 2, slot 2 by the caller behind the row
        constexpr int HM = (int)decltype(halo_tag)::value;
        constexpr int NHP = HM == 1 ? NSLOT + 1 : (HM == 2 ? 2 : 0);
        static_assert(HM != 2 || (NSLOT == 3 && FILL && decltype(fill_tag)::lo == 0 && decltype(fill_tag)::hi == 6 * NSLICE), "late halo requests follow the staging of a three-slot tile");
        constexpr int GT = NB * MB * 3;                             // MFMAs (= gaps) per tap
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
        static_for<0, 3 * GT>([&](auto G) {
            constexpr int g = decltype(G)::value;
            // (block-major: the three products of a block back to back on its accumulator.  Product-major inside a pass -- one B word for MB
            //  consecutive MFMAs on independent accumulators -- measured 1.4 % slower on the step.)
            constexpr int tp = g / GT, gt = g % GT, j = gt / (MB * 3), w = gt % (MB * 3), mb = w / 3, sp = w % 3;
            constexpr int pass = tp * NB + j, buf = pass & 1;
#ifdef X3_PRIOALT                  // experiment (measured 2-4 % slower): the two waves of a SIMD take turns as the arbitration winner (1: per tap, 2: per 16-channel pass)
            if constexpr (X3_PRIOALT == 1 ? gt == 0 : w == 0) {
                constexpr int turn = X3_PRIOALT == 1 ? tp : pass;
                if (((turn + tr) & 1) == (wave >= NWAVE / 2 ? 1 : 0)) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
            }
#endif
#if X3_DIRECT                      // weights as the instruction's FIRST operand: D = channels x pixels, a lane holds 4 consecutive channels of ONE pixel (see `epilogue`)
#define X3_MFMA16(FA, FB) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, Bv[buf][FB]), __builtin_bit_cast(bf16x8, A[mb][FA]), acc[mb][j], 0, 0, 0)
#else
#define X3_MFMA16(FA, FB) acc[mb][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A[mb][FA]), __builtin_bit_cast(bf16x8, Bv[buf][FB]), acc[mb][j], 0, 0, 0)
#endif
            if constexpr (sp == 0) X3_MFMA16(1, 2);                 // hi lo' + lo hi'
            else if constexpr (sp == 1) X3_MFMA16(0, 1);            // hi mid' + mid mid'
            else X3_MFMA16(0, 0);                                   // hi hi' + mid hi'
#undef X3_MFMA16
            // the next pass's B words (first needed 12 - w gaps from here), smallest-term form first
            if constexpr (w < 3 && pass + 1 < 3 * NB) b_read((pass + 1) / NB, (pass + 1) % NB, 2 - w, buf ^ 1);
            // the next tap's A words, in place, during the tap's last pass
            if constexpr (j == NB - 1 && tp < 2) {
                if constexpr (sp == 0) a_read(tp + 1, mb, 1);
                if constexpr (sp == 2) a_read(tp + 1, mb, 0);
            }
            if constexpr (FILL) {
                // the row's U staging units (all 6 NSLICE of them, or the share the fill tag names) dealt evenly over its 3 GT gaps
                // (all of them: every gap at BN = 32, every other gap at BN = 64): its unit n sits in gap ceil((n + 1) 3 GT / U) - 1
                constexpr int ULO = decltype(fill_tag)::lo, U = decltype(fill_tag)::hi - ULO, GR = 3 * GT;
                static_assert(U <= GR, "at most one staging unit per gap");
                constexpr int n0 = (g * U + GR - 1) / GR, n1 = ((g + 1) * U + GR - 1) / GR;
                if constexpr (n1 > n0 && n0 < U) stage_piece((ULO + n0) / 6, 0, (ULO + n0) % 6, img ^ 1);
            }
            // requests: the even gaps w >= 4 of pass RP (no operand reads there; at BN = 64 no staging unit either)
            constexpr int RP = NB > 2 ? 1 : 0;
            static_assert(4 + 2 * ((Cfg::DPW > NHP ? Cfg::DPW : NHP) - 1) < MB * 3, "request slots of a pass");
            if constexpr (j == RP && w >= 4 && (w & 1) == 0) {
                constexpr int rp = (w - 4) / 2;
                if constexpr (tp == 0 && rp < Cfg::DPW) requests(rp);
                if constexpr (tp == 1 && rp < NHP) requests(Cfg::DPW + rp);
                if constexpr (HM == 2 && tp == 2 && rp == 0) requests(Cfg::DPW + 2);
            }
            hook(G);                                                 // (deferred output stores of the previous tile: BN = 32, see `pend`)
            __builtin_amdgcn_sched_barrier(0);
        });
    };
#else
    // MFMA over the three taps of filter row tr: halo image img, weight stage st.  FILL: the 12 staging slices of the next
    // chunk's halo (into image img ^ 1) are dealt over the 3 * MT * NT groups of six MFMAs.  `requests` (the LDS-DMA / halo
    // loads this item has to issue: ~200 scalar + vector instructions of address arithmetic) runs right after the first group
    // of MFMAs has been issued rather than in front of the item's first LDS reads.
    auto mfma_row = [&](int tr, int st, int img, auto fill_tag, auto halo_tag, auto&& requests, auto&&) {
        constexpr bool FILL = decltype(fill_tag)::value;
        const char* wst = wsb + st * Cfg::WS_STAGE;
        const u32x4* xim = xs + img * XS_F4;
        u32x4 av[2][MT][3], bv[2][NT][3];
        auto lds_load = [&](int tp, u32x4 (&ax)[MT][3], u32x4 (&bx)[NT][3]) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    ax[i][p] = xim[(half * 3 + p) * NPIX + (wave * MT + i + tr) * HC + tp + l31];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bx[j][p] = *reinterpret_cast<const u32x4*>(wst + j * WBLK + (((tp * 2 + half) * 3 + p) * 32 + l31) * 16);
        };
        lds_load(0, av[0], bv[0]);
#ifndef X3_LUMPS
        // Everything that is not an MFMA is cut into pieces of one LDS read or 1-4 VALU instructions and placed BETWEEN the MFMAs,
        // fenced so that it stays there: the six MFMAs of a group depend on each other through the accumulator, each leaves 32 cycles
        // of issue slots.  When the older wave of the SIMD has finished its item and waits at the barrier, the younger one runs alone:
        // whatever stands between its MFMA groups as a lump (12 operand reads in front of a tap, a staging slice behind a group) is
        // then matrix-pipe idle time.
        auto next_read = [&](int tp, int m, u32x4 (&ax)[MT][3], u32x4 (&bx)[NT][3]) {      // operand read m of tap tp (0 .. 3 (MT + NT) - 1)
            if (m < 3 * MT) {
                const int i = m / 3, pc = m % 3;
                ax[i][pc] = xim[(half * 3 + pc) * NPIX + (wave * MT + i + tr) * HC + tp + l31];
            } else {
                const int j = (m - 3 * MT) / 3, pc = (m - 3 * MT) % 3;
                bx[j][pc] = *reinterpret_cast<const u32x4*>(wst + j * WBLK + (((tp * 2 + half) * 3 + pc) * 32 + l31) * 16);
            }
        };
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            const u32x4 (&ax)[MT][3] = av[tp & 1];
            const u32x4 (&bx)[NT][3] = bv[tp & 1];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    constexpr int NGRP = 3 * MT * NT, SPG = NSLICE / NGRP;   // staging slices per group (1 or 2)
                    constexpr int NRD = 3 * (MT + NT), NGAP = 6 * MT * NT;     // operand reads of a tap; MFMA gaps of a tap
                    constexpr int NHP = decltype(halo_tag)::value ? NSLOT + 1 : 0;     // halo pieces: the scalar part + one per slot
                    constexpr int RSTEP = (NGAP - NRD) / (Cfg::DPW > NHP ? Cfg::DPW : NHP) > 0 ? (NGAP - NRD) / (Cfg::DPW > NHP ? Cfg::DPW : NHP) : 1;
                    static_assert(NRD + RSTEP * ((Cfg::DPW > NHP ? Cfg::DPW : NHP) - 1) < NGAP, "request pieces must fit behind the reads");
                    static_assert(NSLICE % NGRP == 0, "slices per group");
                    const int gi = i * NT + j, grp = tp * MT * NT + gi;
#define X3_MFMA(PA, PB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax[i][PA]), __builtin_bit_cast(bf16x8, bx[j][PB]), acc[i][j], 0, 0, 0)
                    // the item's requests, one piece per gap, in the gaps behind the ones that carry the next tap's operand reads (an
                    // LDS-DMA instruction issued among 12 ds_read_b128 costs 100-185 cycles of issue, in a read-free gap 25-60):
                    // tap 0: the DPW weight pieces from gap NRD on, every RSTEP-th gap; tap 1: the halo pieces likewise
#define X3_GAP(STEP) { const int m = gi * 6 + STEP;                                                                                  \
                       if (tp + 1 < 3 && m < NRD) next_read(tp + 1, m, av[(tp + 1) & 1], bv[(tp + 1) & 1]);                          \
                       if constexpr (FILL) { _Pragma("unroll") for (int u = 0; u < SPG; ++u) stage_piece(grp * SPG + u, u, STEP, img ^ 1); } \
                       if (m >= NRD && (m - NRD) % RSTEP == 0) {                                                                     \
                           const int rp = (m - NRD) / RSTEP;                                                                         \
                           if (tp == 0 && rp < Cfg::DPW) requests(rp);                                                              \
                           if (tp == 1 && rp < NHP) requests(Cfg::DPW + rp);                                                        \
                       }                                                                                                             \
                       __builtin_amdgcn_sched_barrier(0); }
                    // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
                    X3_MFMA(0, 2); X3_GAP(0)
                    X3_MFMA(2, 0); X3_GAP(1)
                    X3_MFMA(1, 1); X3_GAP(2)
                    X3_MFMA(0, 1); X3_GAP(3)
                    X3_MFMA(1, 0); X3_GAP(4)
                    X3_MFMA(0, 0); X3_GAP(5)
#undef X3_GAP
#undef X3_MFMA
                }
        }
#else
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            if (tp + 1 < 3) lds_load(tp + 1, av[(tp + 1) & 1], bv[(tp + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 (&ax)[MT][3] = av[tp & 1];
            const u32x4 (&bx)[NT][3] = bv[tp & 1];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
#define X3_MFMA(PA, PB) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ax[i][PA]), __builtin_bit_cast(bf16x8, bx[j][PB]), acc[i][j], 0, 0, 0)
                    X3_MFMA(0, 2); X3_MFMA(2, 0); X3_MFMA(1, 1); X3_MFMA(0, 1); X3_MFMA(1, 0); X3_MFMA(0, 0);
#undef X3_MFMA
                    if (tp == 0 && i == 0 && j == 0) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int rp = 0; rp < Cfg::DPW + (decltype(halo_tag)::value ? NSLOT + 1 : 0); ++rp) requests(rp);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (FILL) {
                        constexpr int NGRP = 3 * MT * NT;                       // the NSLICE slices are dealt evenly over the groups
                        const int grp = (tp * MT + i) * NT + j;
#pragma unroll
                        for (int q = grp * NSLICE / NGRP; q < (grp + 1) * NSLICE / NGRP; ++q) stage_slice(q, img ^ 1);
                        __builtin_amdgcn_sched_barrier(0);                    // keep the slice behind ITS group of MFMAs
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
    };

#endif      // X3_M16

    // ---- the epilogue's kernel arguments, cached in ONE vector register (lane i = argument i) and fetched with v_readlane.
    // Round 4 (ISA + cycle stamps): the main loop leaves no scalar registers for them, so the compiler RE-LOADED them from the kernel-argument
    // segment wherever the epilogue used one -- ~25 `s_load` + `s_waitcnt lgkmcnt(0)` chains of 150-300 cycles each per tile, with the matrix
    // pipe idle: 3400 cycles of a BN = 32 tile's 29 000, 7000 of a BN = 64 tile.  A value that comes out of a v_readlane cannot be
    // re-materialised from memory; if it has to leave its scalar register it goes into a spill lane (a v_readlane again).
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

#if X3_DIRECT
    // ---- epilogue of tile `tl`, straight from the accumulators (round 4).  With the WEIGHTS as the first operand of the 16x16x32 instruction the
    // accumulator block is channels x pixels: lane l holds channels 4 (l >> 4) .. + 3 of pixel l & 15 -- sixteen contiguous bytes of the NHWC
    // destination.  So bias, activation, act' mask, residual and accumulation are plain float4 arithmetic on the accumulator registers and every
    // block goes out as ONE 16-byte store per lane (16 pixels x 64 bytes per instruction; a CU's store path takes ~12 cycles per store
    // instruction whatever its width: tools/ubench/store_rate.hip): no LDS patch, no transposition through it (rounds 2-4: 64 ds_write_b32 +
    // 16 ds_read_b128 per wave and tile, each round a dependent chain behind `s_waitcnt lgkmcnt`), no barrier in front of the epilogue, and the
    // stores drain while the next tile's first filter row runs (the waits behind an epilogue count them: `run` / the 32-column loop below).
    // The fused MaxPool2d(2) takes the other pixel of a pair from the neighbouring lane (DPP quad_perm) and the other row from the wave's
    // second accumulator row.
    constexpr int NST = POOL ? NB * 2 * 4 : MB * NB;                // vector-memory stores of one epilogue per wave (the vmcnt units behind it)
    static_assert(NST + 2 * NSLOT + 2 * Cfg::DPW <= 63, "the waits behind an epilogue count its stores");
    auto nohook = [](auto) {};
    auto epilogue = [&](const Tile& tl, float*) __attribute__((always_inline)) {
        X3_T(te)
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
                }
        auto rsrc = [&](const float* base, int k) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (int64_t)b * ea.OH * ea.OW * cs_[k]), 0, ea.OH * ea.OW * cs_[k] * 4, 0x00020000);
        };
        const float aslope = ea.act == 1 ? 0.2f : (ea.act == 2 ? 0.f : 1.f);
        f32x4 bias4[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ea.bias && blk_[j >> 1]) bias4[j] = *reinterpret_cast<const f32x4*>(ea.bias + n0 + 16 * j + c4);
        }
        auto act4 = [&](f32x4 o) {
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], aslope * o[c]);
            return o;
        };
        auto take = [&](int mb, int j) { const f32x4 v = acc[mb][j]; acc[mb][j] = f32x4{0.f, 0.f, 0.f, 0.f}; return v; };
        if constexpr (POOL) {
            // Forward layer in front of MaxPool2d(2) (archs/Unet.py:35,41,47,53): single destination, bias + activation only.  A wave owns rows
            // 2w, 2w + 1 of its 32 columns: a lane's two accumulator rows + the same two of lane ^ 1 are one 2x2 window of 4 channels; the even
            // lane writes the pooled float4 and the four codes (bits 0-1 first maximum in the order (0,0) (0,1) (1,0) (1,1), bits 2-5 the signs)
            // of csrc/misc.hip maxpool_fwd_codes_kernel.
            static_assert(MT == 2, "a wave owns one row pair");
            const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(0), 0);
            const int ph = ea.OH >> 1, pw = ea.OW >> 1;
            const int64_t pimg = (int64_t)b * ph * pw * ea.pool_cs;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_dst + pimg), 0, ph * pw * ea.pool_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_codes + pimg), 0, ph * pw * ea.pool_cs, 0x00020000);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 win[2], nbr[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        win[i] = act4(take(2 * i + h, j) + bias4[j]);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, win[i]), rd, vo[j >> 1][i][h], (j & 1) * 64, 0);
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
                    const bool ok2 = !(lane & 1) && blk_[j >> 1] && py0 < ea.DH && px < ea.DW;     // even sizes: the whole window is inside or outside
                    const unsigned po = (unsigned)(((py0 >> 1) * pw + (px >> 1)) * ea.pool_cs + n0 + 16 * j + c4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mx), rp, ok2 ? po * 4u : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(code, rc, ok2 ? po : OOB, 0, 0);
                }
            X3_T(teb)
            return;
        }
        // ---- FWD: no mask, no accumulation, no residual (every forward layer);  BWD: act' mask on every destination, nothing else.
        // All mask requests first, then one add / max / select / store per block.
        if constexpr (EK == EK_FWD || EK == EK_BWD) {
            constexpr bool MASKED = EK == EK_BWD;
            f32x4 mk[MASKED ? MB : 1][MASKED ? NB : 1];
            if constexpr (MASKED) {
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const __amdgpu_buffer_rsrc_t rm = rsrc(ea.mask(du_[k]), k);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int h = 0; h < 2; ++h)
                                mk[2 * i + h][2 * k + jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, vo[k][i][h], jj * 64, 0));
                }
            }
            X3_T(tea)
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(du_[k]), k);
                const float msl = ea.mask_mode(du_[k]) == 1 ? 0.2f : 0.f;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 o = act4(take(2 * i + h, 2 * k + jj) + bias4[2 * k + jj]);
                            if constexpr (MASKED) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[c] *= (mk[2 * i + h][2 * k + jj][c] > 0.f) ? 1.f : msl;
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[k][i][h], jj * 64, 0);
                        }
            }
            X3_T(teb)
            return;
        }
        // ---- the general case (residual, accumulation, a mask on one destination only), branch-free as well: what a block does not use is
        // requested out of range (no memory traffic, zeros come back), so the number of vector-memory operations does not depend on the flags
        X3_T(tea)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int du = du_[k], mm2 = ea.mask_mode(du), acc2 = ea.accum(du);
            const bool use_add2 = ea.addsrc && du == 0;
            const __amdgpu_buffer_rsrc_t rd = rsrc(ea.dst(du), k);
            const __amdgpu_buffer_rsrc_t rm = rsrc(mm2 ? ea.mask(du) : ea.dst(du), k);
            const __amdgpu_buffer_rsrc_t rad = rsrc(use_add2 ? ea.addsrc : ea.dst(du), k);
            const float msl = mm2 == 1 ? 0.2f : 0.f;
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
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] *= (m2[i][h][c] > 0.f || !mm2) ? 1.f : msl;
                        o += pr2[i][h];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[k][i][h], jj * 64, 0);
                    }
            }
        }
        X3_T(teb)
    };
#else
    // ---- deferred output stores of the 32-column kernel (X3_DEFER32).  A CU's vector-memory STORE path takes ~16-19 bytes per cycle (round 4:
    // eight 1 KB stores per wave = 64 KB per CU in 3450 cycles, whatever surrounds them), which at the end of a K = 32 tile is 12 % of the
    // tile with the matrix pipe idle.  The epilogue therefore leaves the tile's eight full-resolution stores in registers (value + offset) and
    // they go out ONE at a time, ~12 MFMAs apart, in gaps of the NEXT chunk's filter rows 0 and 1 (pass 1: no requests there); the staging
    // registers are idle in those rows, so the 40 registers fit.  In front of barriers instead (tried) the stores block all waves together.
    constexpr bool DEFER = X3_DEFER32 && BN == 32 && X3_M16;
    constexpr int NST = NT * MT * 2 * 2;                            // full-resolution 16-byte stores per wave and tile
    [[maybe_unused]] f32x4 pend_o[NST];
    [[maybe_unused]] unsigned pend_ok = 0, pend_lo = 0;             // per lane: bit s = store s is inside the map; its lane offset (pixel pr, channel quad)
    [[maybe_unused]] int pend_b = 0, pend_du = 0, pend_y = 0, pend_x = 0, pend_c = 0;     // the tile: image, destination, first row / column of this wave, channel base
    [[maybe_unused]] bool pend = false;
    auto nohook = [](auto) {};
    auto pend_rsrc = [&]() {
        const EpiArgs ea = epi_args();
        const int cs2 = ea.dst_cs(pend_du);
        return __builtin_amdgcn_make_buffer_rsrc((void*)(ea.dst(pend_du) + (int64_t)pend_b * ea.OH * ea.OW * cs2), 0, ea.OH * ea.OW * cs2 * 4, 0x00020000);
    };
    // store s = (row i, half h2, pixel group e): scalar offset of its first pixel + the lane offset (or out of range)
    auto pend_store = [&](int sidx, const __amdgpu_buffer_rsrc_t& prd) {
        const int so_ow = (int)__builtin_amdgcn_readlane((int)argv, E_OW), so_cs = (int)__builtin_amdgcn_readlane((int)argv, pend_du ? E_CS1 : E_CS0);
        const int e = sidx & 1, h2 = (sidx >> 1) & 1, i = sidx >> 2;
        const int so = (((pend_y + i) * so_ow + pend_x + 16 * h2 + (POOL ? e : 8 * e)) * so_cs + pend_c) * 4;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pend_o[sidx]), prd, (pend_ok >> sidx) & 1 ? pend_lo : OOB, so, 0);
    };
    // store hook of filter row R (0 / 1): gaps (tap tp, pass 1, w = 3 or 9) -> store number 6 R + 2 tp + (w == 9)
    auto store_hook = [&](auto row_tag, const __amdgpu_buffer_rsrc_t& prd) {
        return [&, prd](auto G) {
            constexpr int R = decltype(row_tag)::value, g = decltype(G)::value;
            constexpr int GT = (BN / 16) * (2 * MT) * 3, tp = g / GT, gt = g % GT, j = gt / (2 * MT * 3), w = gt % (2 * MT * 3);
            if constexpr (j == 1 && (w == 3 || w == 9)) {
                constexpr int sidx = 6 * R + 2 * tp + (w == 9 ? 1 : 0);
                if constexpr (sidx < NST) pend_store(sidx, prd);
            }
        };
    };

    // ---- epilogue of tile `tl` (csrc/conv_igemm.hip's fast path: n_split / n_sub are multiples of 32, so destination, mask and
    // channel base are wave-uniform per 32-column block); half a 32x32 tile (16 pixels) at a time through a 2 KB patch
    auto epilogue = [&](const Tile& tl, float* epi) __attribute__((always_inline)) {
        X3_T(te)
        const EpiArgs ea = epi_args();
        const int b = tl.b, x0 = tl.x0, y0 = tl.y0, n0 = tl.n0;
        float* eb = epi + wave * (16 * Cfg::EPS);
        const int q4 = (lane & 7) * 4, pr = lane >> 3;
        if constexpr (POOL) {
            // Forward layer in front of MaxPool2d(2) (archs/Unet.py:35,41,47,53): single destination, bias + activation only.
            // A wave owns the two rows 2j, 2j+1 of its 32 columns; a lane takes the pixel PAIR (2 pr, 2 pr + 1) of a 16-pixel
            // half from the patch, so after both rows it holds a whole 2x2 window of 4 channels: it writes the pooled float4
            // and the four codes (bits 0-1 first maximum in the order (0,0) (0,1) (1,0) (1,1), bits 2-5 the signs) of
            // csrc/misc.hip maxpool_fwd_codes_kernel -- the pool kernel and its re-read of the full-resolution map go away.
            const int cs2 = ea.dst_cs(0);
            const int64_t imgo = (int64_t)b * ea.OH * ea.OW * cs2;
            const int ibytes = ea.OH * ea.OW * cs2 * 4;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.dst(0) + imgo), 0, ibytes, 0x00020000);
            const int ph = ea.OH >> 1, pw = ea.OW >> 1;
            const int64_t pimg = (int64_t)b * ph * pw * ea.pool_cs;
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_dst + pimg), 0, ph * pw * ea.pool_cs * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.pool_codes + pimg), 0, ph * pw * ea.pool_cs, 0x00020000);
            const float aslope = ea.act == 1 ? 0.2f : (ea.act == 2 ? 0.f : 1.f);
            const int py0 = y0 + wave * MT;
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
                const bool n_ok = nwv + q4 < ea.Ntot;
                f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
                if (ea.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(ea.bias + nwv + q4);
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    f32x4 win[2][2];
                    const int px = x0 + 16 * h2 + 2 * pr;
                    const bool ok2 = py0 < ea.DH && px < ea.DW && n_ok;            // even sizes: the whole window is inside or outside
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        spill_half(eb, i, k, h2);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            f32x4 o = *reinterpret_cast<const f32x4*>(eb + (2 * pr + e) * Cfg::EPS + q4) + bias4;
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], aslope * o[c]);
                            win[i][e] = o;
                            const unsigned vo = ok2 ? (unsigned)((((py0 + i) * ea.OW + px + e) * cs2 + nwv + q4) * 4) : OOB;
                            if constexpr (DEFER) { pend_o[(i * 2 + h2) * 2 + e] = o; pend_ok |= (ok2 ? 1u : 0u) << ((i * 2 + h2) * 2 + e); }
                            else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo, 0, 0);
                        }
                    }
                    f32x4 mx;
                    unsigned code = 0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float w0 = win[0][0][c], w1 = win[0][1][c], w2 = win[1][0][c], w3 = win[1][1][c];
                        unsigned arg = 0; float best = w0;
                        if (w1 > best) { best = w1; arg = 1; }                  // first maximum wins
                        if (w2 > best) { best = w2; arg = 2; }
                        if (w3 > best) { best = w3; arg = 3; }
                        const unsigned cj = arg | (w0 > 0.f ? 4u : 0u) | (w1 > 0.f ? 8u : 0u) | (w2 > 0.f ? 16u : 0u) | (w3 > 0.f ? 32u : 0u);
                        mx[c] = fmaxf(fmaxf(w0, w1), fmaxf(w2, w3));
                        code |= cj << (8 * c);
                    }
                    const unsigned po = (unsigned)(((py0 >> 1) * pw + (px >> 1)) * ea.pool_cs + nwv + q4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mx), rp, ok2 ? po * 4u : OOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(code, rc, ok2 ? po : OOB, 0, 0);
                }
            }
            if constexpr (DEFER) {
                pend = true; pend_b = b; pend_du = 0; pend_y = py0; pend_x = x0; pend_c = n0;
                pend_lo = __umul24(2 * pr, cs2 * 4) + q4 * 4;           // the pixel pair (2 pr, 2 pr + 1): + e through the scalar offset
            }
            return;
        }
        // ---- the two common cases as ONE straight-line block (round 4).  Per-round cycle stamps put every 16-pixel round of the general code
        // below at 800-900 cycles for ~80 instructions: the rounds are separated by the (wave-uniform) branches on mask / accumulate /
        // residual, so the compiler cannot interleave them, and a wave that runs its epilogue ALONE on its SIMD (its partner waits at the
        // barrier) executes one dependent chain -- LDS write -> read -> bias -> activation -> store -- after the other: 3400 cycles per
        // BN = 32 tile, 7000 per BN = 64 tile, with the matrix pipe idle.  Without the branches all patch round trips are issued back to
        // back (LDS operations of a wave execute in order: the ONE patch is rewritten right behind the reads of the previous round) and
        // the rounds' arithmetic overlaps.
        //   FWD: no mask, no accumulation, no residual (every forward layer)        BWD: act' mask on every destination, nothing else
        {
            const bool two = ea.dst1 != nullptr;
            const bool plain = !ea.addsrc && !ea.ac0 && !(two && ea.ac1);
            const bool is_fwd = plain && !ea.mm0 && !(two && ea.mm1), is_bwd = plain && ea.mm0 && (!two || ea.mm1);
            auto fast = [&](auto masked_tag) __attribute__((always_inline)) {
                constexpr bool MASKED = decltype(masked_tag)::value;
                int du_[NT], chw_[NT], cs_[NT]; bool blk_[NT]; float msl_[NT];
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
                    du_[k] = nwv >= ea.n_split ? 1 : 0; chw_[k] = nwv - (du_[k] ? ea.n_split : 0); cs_[k] = ea.dst_cs(du_[k]);
                    blk_[k] = nwv < ea.Ntot; msl_[k] = ea.mask_mode(du_[k]) == 1 ? 0.2f : 0.f;
                }
                auto voff = [&](int k, int i, int h2, int e) {
                    const bool ok = blk_[k] && y0 + wave * MT + i < ea.DH && x0 + 16 * h2 + pr + 8 * e < ea.DW;
                    unsigned v = ok ? (unsigned)((((y0 + wave * MT + i) * ea.OW + x0 + 16 * h2 + pr + 8 * e) * cs_[k] + chw_[k] + q4) * 4) : OOB;
#ifdef X3_EPI_OOB
                    v |= OOB;
#endif
                    return v;
                };
                f32x4 mk[MASKED ? NT : 1][MT][2][2], pv[NT][MT][2][2], bias4[NT];
                if constexpr (MASKED) {
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.mask(du_[k]) + (int64_t)b * ea.OH * ea.OW * cs_[k]), 0,
                                                                                             ea.OH * ea.OW * cs_[k] * 4, 0x00020000);
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                                for (int e = 0; e < 2; ++e)
                                    mk[k][i][h2][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, voff(k, i, h2, e), 0, 0));
                    }
                }
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    bias4[k] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (ea.bias && blk_[k]) bias4[k] = *reinterpret_cast<const f32x4*>(ea.bias + n0 + k * 32 + q4);
                }
#pragma unroll
                for (int k = 0; k < NT; ++k)
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2) {
                            spill_half(eb, i, k, h2);
#pragma unroll
                            for (int e = 0; e < 2; ++e) pv[k][i][h2][e] = *reinterpret_cast<const f32x4*>(eb + (pr + 8 * e) * Cfg::EPS + q4);
                        }
                X3_T(tea)
                const float aslope = ea.act == 1 ? 0.2f : (ea.act == 2 ? 0.f : 1.f);
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.dst(du_[k]) + (int64_t)b * ea.OH * ea.OW * cs_[k]), 0,
                                                                                         ea.OH * ea.OW * cs_[k] * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                f32x4 o = pv[k][i][h2][e] + bias4[k];
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], aslope * o[c]);
                                if constexpr (MASKED) {
#pragma unroll
                                    for (int c = 0; c < 4; ++c) o[c] *= (mk[k][i][h2][e][c] > 0.f) ? 1.f : msl_[k];
                                }
                                const unsigned vo = voff(k, i, h2, e);
                                if constexpr (DEFER) { pend_o[(i * 2 + h2) * 2 + e] = o; pend_ok |= (vo != OOB ? 1u : 0u) << ((i * 2 + h2) * 2 + e); }
                                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo, 0, 0);
                            }
                }
                if constexpr (DEFER) {
                    pend = true; pend_b = b; pend_y = y0 + wave * MT; pend_x = x0; pend_du = du_[0]; pend_c = chw_[0]; pend_lo = __umul24(pr, cs_[0] * 4) + q4 * 4;
                }
                X3_T(teb)
            };
            if (is_fwd) { fast(std::false_type{}); return; }
            if (is_bwd) { fast(std::true_type{}); return; }
        }
        // act' masks of the whole tile requested up front (backward-data): one HBM round trip per tile instead of one per 16-pixel half
        // (2 MT NT of them, each waited for right after its request -- the epilogue is not overlapped with MFMAs, so that latency was
        // all exposed: the masked backward-data layers ran 5-20 % behind their forward twins)
        f32x4 mpre[NT][MT][2][2];
        if (ea.mask_mode(0) | ea.mask_mode(1)) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
                const int du = nwv >= ea.n_split ? 1 : 0;
                const int chw = nwv - (du ? ea.n_split : 0);
                const int cs2 = ea.dst_cs(du);
                if (ea.mask_mode(du)) {
                    const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void*)(ea.mask(du) + (int64_t)b * ea.OH * ea.OW * cs2), 0,
                                                                                         ea.OH * ea.OW * cs2 * 4, 0x00020000);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const int py = y0 + wave * MT + i, px = x0 + 16 * h2 + pr + 8 * e;
                                const bool ok2 = py < ea.DH && px < ea.DW && nwv + q4 < ea.Ntot;
                                mpre[k][i][h2][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                    rm, ok2 ? (unsigned)(((py * ea.OW + px) * cs2 + chw + q4) * 4) : OOB, 0, 0));
                            }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        X3_T(tea)
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int nwv = __builtin_amdgcn_readfirstlane(n0 + k * 32);
            const bool n_ok = nwv + q4 < ea.Ntot;
            const int du = nwv >= ea.n_split ? 1 : 0;
            const int chw = nwv - (du ? ea.n_split : 0);
            const int cs2 = ea.dst_cs(du), mm2 = ea.mask_mode(du), acc2 = ea.accum(du);
            const int64_t imgo = (int64_t)b * ea.OH * ea.OW * cs2;
            const int ibytes = ea.OH * ea.OW * cs2 * 4;
            float* dstb = ea.dst(du) + imgo;
            const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dstb, 0, ibytes, 0x00020000);
            const bool use_add2 = ea.addsrc && du == 0;
            const __amdgpu_buffer_rsrc_t rad = __builtin_amdgcn_make_buffer_rsrc((void*)(use_add2 ? ea.addsrc + imgo : dstb), 0, ibytes, 0x00020000);
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (ea.bias && n_ok) bias4 = *reinterpret_cast<const f32x4*>(ea.bias + nwv + q4);
            const float aslope = ea.act == 1 ? 0.2f : (ea.act == 2 ? 0.f : 1.f), mslope = mm2 == 1 ? 0.2f : 0.f;
#ifdef X3_STAMPS
            if (k == 0) X3_T(ter[0])
#endif
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int py = y0 + wave * MT + i;
                const bool rowok = py < ea.DH;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    spill_half(eb, i, k, h2);
                    unsigned vo[2];
                    f32x4 v2[2], m2[2], ad2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int p = pr + 8 * e, px = x0 + 16 * h2 + p;
                        const bool ok2 = rowok && px < ea.DW && n_ok;
                        vo[e] = ok2 ? (unsigned)(((py * ea.OW + px) * cs2 + chw + q4) * 4) : OOB;
#ifdef X3_EPI_OOB                  // timing experiment only (wrong results): every store is dropped by the range check -- same instructions, no write traffic
                        vo[e] |= OOB;
#endif
                        v2[e] = *reinterpret_cast<const f32x4*>(eb + p * Cfg::EPS + q4);
                    }
                    if (mm2) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) m2[e] = mpre[k][i][h2][e];
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
                        if constexpr (DEFER) {
                            pend_o[(i * 2 + h2) * 2 + e] = o; pend_ok |= (vo[e] != OOB ? 1u : 0u) << ((i * 2 + h2) * 2 + e);
                            pend_du = du; pend_c = chw; pend_lo = __umul24(pr, cs2 * 4) + q4 * 4;
                        } else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[e], 0, 0);
                    }
#ifdef X3_STAMPS
                    if (k == 0) X3_T(ter[1 + i * 2 + h2])
#endif
                }
            }
        }
        if constexpr (DEFER) { pend = true; pend_b = b; pend_y = y0 + wave * MT; pend_x = x0; }
        X3_T(teb)
    };

#endif      // X3_DIRECT

    // ---- main loop over (tile, 16-channel chunk).  A chunk's three filter rows are straight-line code, so every s_waitcnt
    // below is exact: vector-memory operations complete in issue order.  Items are numbered it = 3 * chunk + row; item it
    // reads weight stage it % NSTAGE and, right after its barrier, requests the weights of item it + AHEAD.
    int t = xcd_remap(blockIdx.x, G);
    if (t >= total) return;
    Tile cur = decode(t), nxt = pick(t + G < total, advance(cur), cur);  // nxt: the tile this workgroup takes after cur
    Tile nxt2 = pick(t + 2 * G < total, advance(nxt), nxt);              // ... and the one after that (a chunk two ahead may belong to it when K is one chunk)
    int g = 0, img = 0;
    constexpr int D = Cfg::DPW, HL = 2 * NSLOT;                          // vmcnt units: weight requests of one item, halo loads of one chunk
    // the k-th chunk after the current one, k = 1, 2: (tile, chunk, exists); past the end of this workgroup's work it falls back to the
    // current chunk (requests stay branch-free and the instruction counts exact; weights are then requested with valid = false)
    struct Ck { Tile tile; int g; bool ok; };
    auto chunk_at = [&](int k) {
        int gk = g + k, hop = 0;
        if (gk >= nchunks) { gk -= nchunks; hop = 1; }
        if (gk >= nchunks) { gk -= nchunks; hop = 2; }                   // (k <= 2: at most two tile changes, and only when nchunks == 1)
        Ck c;
        c.ok = t + hop * G < total;
        c.g = c.ok ? gk : g;
        c.tile = pick(!c.ok || hop == 0, cur, pick(hop == 1, nxt, nxt2));
        return c;
    };
    auto next_tile = [&]() { t += G; cur = nxt; nxt = nxt2; nxt2 = pick(t + 2 * G < total, advance(nxt), nxt); g = 0; };
    dma_weights(cur, 0, 0, 0);
    load_halo(cur, 0);
    __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
    for (int q = 0; q < NSLICE; ++q) stage_slice(q, 0);                 // the first chunk's halo: nothing to hide it behind yet
#ifdef X3_SKIP_STORE              // timing experiment only (wrong results): no staging in the loop; BOTH images hold the first chunk's (real, non-zero)
#pragma unroll                    // data, so that the matrix cores switch as much as on real operands (all-zero operands alone run 24 % faster)
    for (int q = 0; q < NSLICE; ++q) stage_slice(q, 1);
#endif
    if constexpr (BN == 64) {
        // two weight stages, one item ahead.  Issue order per chunk:
        //   row 0: [weights row 1][halo of the NEXT chunk]   row 1: [weights row 2]   row 2: [weights of the next chunk's row 0]
        int st = 0;
        // Where the next chunk's halo staging (146 VALU + 9 LDS stores per wave) sits -- X3_FILLMODE:
        //   0  every wave in filter row 2 (rounds 2-3);
        //   1  waves 0-3 in row 1, waves 4-7 in row 2: waves w and w + 4 share a SIMD, so in each of the two rows one wave of a SIMD
        //      is VALU-dense while its partner issues MFMAs + operand reads only (complementary pairing, split by wave >= 4);
        //   2  the same with the halves swapped;   3  every wave, half of the units in row 1 and half in row 2.
        // A wave that stages in row 1 needs its halo loads (requested in row 0, tap 1) there already: vmcnt(0) instead of vmcnt(HL).
        // The loop is instantiated per (units in row 1, units in row 2): with modes 1 / 2 the two wave halves run two straight-line bodies.
        auto run = [&](auto f1, auto f2) __attribute__((always_inline)) {
#if X3_DIRECT
        __builtin_amdgcn_s_waitcnt(0x0f70);                             // (the wait + barrier in front of filter row 0 stand at the END of the loop body)
        X3_SYNC();
#endif
        for (;;) {
            const Ck n1 = chunk_at(1);
            // ---- filter row 0: its weights have landed; the barrier publishes them and halo image img (written during the
            // previous chunk's rows 1 / 2), and says every wave is done with the other image and stage
#if !X3_DIRECT
            __builtin_amdgcn_s_waitcnt(0x0f70);                         // vmcnt(0)
            X3_T(tw)
            X3_SYNC();
            X3_T(tb)
#endif
            mfma_row(0, st, img, FillNone{}, std::true_type{}, [&](int rp) {
                if (rp < D) dma_piece(cur, g, 1, st ^ 1, true, rp); else if (rp == D) halo_prep(n1.tile, n1.g); else halo_slot(rp - D - 1); }, nohook);
            X3_T(tm)
            // ---- filter row 1
            if constexpr (decltype(f1)::value) __builtin_amdgcn_s_waitcnt(0x0f70);
            else __builtin_amdgcn_s_waitcnt(0x0f70 | HL);               // the weights of row 1; the halo loads stay in flight
            X3_T(tw)
            X3_SYNC();
            X3_T(tb)
            mfma_row(1, st ^ 1, img, f1, std::false_type{}, [&](int rp) { dma_piece(cur, g, 2, st, true, rp); }, nohook);
            X3_T(tm1)
            // ---- filter row 2
            __builtin_amdgcn_s_waitcnt(0x0f70);                         // [halo loads][weights of row 2]: wait for all
            X3_T(tw)
            X3_SYNC();
            X3_T(tb)
            mfma_row(2, st, img, f2, std::false_type{}, [&](int rp) { dma_piece(n1.tile, n1.g, 0, st ^ 1, n1.ok, rp); }, nohook);
            X3_T(tm2)
#if X3_DIRECT
            // The next row 0's wait and barrier, per path: outstanding are [weights of the next row 0] and, behind an epilogue, its NST stores --
            // which may stay in flight through that row (row 1's wait covers them).  Both stand INSIDE the paths: behind a merge the compiler,
            // which waits for every LDS-DMA in front of a barrier by its own count, could no longer count the stores and would wait for them.
            if (g == nchunks - 1) {
                epilogue(cur, nullptr);
                if (!n1.ok) break;
                next_tile();
                st ^= 1; img ^= 1;
                __builtin_amdgcn_s_waitcnt(X3_VMCNT(NST));
                X3_T(tw)
                X3_SYNC();
                X3_T(tb)
            } else {
                ++g;
                st ^= 1; img ^= 1;
                __builtin_amdgcn_s_waitcnt(0x0f70);
                X3_T(tw)
                X3_SYNC();
                X3_T(tb)
            }
#else
            if (g == nchunks - 1) {
                __syncthreads();                                        // every wave has finished reading stage st: it holds the epilogue patches now
                epilogue(cur, reinterpret_cast<float*>(wsb + st * Cfg::WS_STAGE));
                X3_T(te)
            }
            if (!n1.ok) break;
            if (g == nchunks - 1) next_tile(); else ++g;
            st ^= 1; img ^= 1;
#endif
        }
        };
        constexpr int FM = X3_FILLMODE;
        if constexpr (FM == 0) run(FillNone{}, FillAll{});
        else if constexpr (FM == 3) run(Fill<0, 3 * NSLICE>{}, Fill<3 * NSLICE, 6 * NSLICE>{});
        else if ((FM == 1) == (wave < NWAVE / 2)) run(FillAll{}, FillNone{});
        else run(FillNone{}, FillAll{});
#ifdef X3_STAMPS
        dump_stamps();
#endif
    } else {
        // three weight stages, two items ahead: with three items per chunk, filter row r always lives in stage r.  The halo of
        // chunk c+2 is requested at the END of chunk c's row 2 (as soon as the registers are free): three items of flight
        // time.  Issue order per chunk:
        //   row 0: [weights row 2]   row 1: [weights next row 0]   row 2: [weights next row 1] [halo of the chunk after next]
        {
            const Ck n1 = chunk_at(1);                                  // what the steady state requested one chunk earlier
            dma_weights(cur, 0, 1, 1);
            load_halo(n1.tile, n1.g);
        }
#if X3_DIRECT
        __builtin_amdgcn_s_waitcnt(X3_VMCNT(D + HL));
        __syncthreads();
#endif
        for (;;) {
            const Ck n1 = chunk_at(1), n2 = chunk_at(2);
            // ---- row 0: outstanding [w row 0][w row 1][halo next]
            X3_T(te)
            auto req0 = [&](int rp) { dma_piece(cur, g, 2, 2, true, rp); };
            auto req1 = [&](int rp) { dma_piece(n1.tile, n1.g, 0, 0, n1.ok, rp); };
#if X3_DIRECT
            // (The wait + barrier in front of row 0 stand at the END of the loop body, once per path: behind an epilogue its NST stores are
            //  the youngest operations and stay in flight.  Row 1's wait then leaves the newest HL + D operations out whichever path came
            //  before -- behind an epilogue that is the weights of row 2 and the last stores, i.e. it only asks for the first ones, issued a
            //  filter row earlier -- and row 2's wait covers them all.)
            mfma_row(0, 0, img, FillNone{}, std::false_type{}, req0, nohook);
            X3_T(tm)
            __builtin_amdgcn_s_waitcnt(X3_VMCNT(HL + D));                  // [w row 1] | [halo next][(stores)][w row 2]
            X3_T(tw)
            __syncthreads();
            X3_T(tb)
            mfma_row(1, 1, img, FillNone{}, std::false_type{}, req1, nohook);
            X3_T(tm1)
            __builtin_amdgcn_s_waitcnt(0x0f70 | D);                        // [halo next][(stores)][w row 2] | [w next row 0]
            X3_T(tw)
            __syncthreads();
#else
            __builtin_amdgcn_s_waitcnt(0x0f70 | (D + HL));
            X3_T(tw)
            __syncthreads();
            X3_T(tb)
            if (DEFER && pend) {
                // The previous tile's stores ride in this chunk's rows 0 (six) and 1 (two).  Waits and barriers stand INSIDE this path: the
                // compiler waits for every LDS-DMA issued before a barrier and can count the younger operations (here: the stores) only on
                // straight-line code -- behind a merge with the store-free path it would fall back to vmcnt(0), i.e. wait for the stores.
                const __amdgpu_buffer_rsrc_t prd = pend_rsrc();
                mfma_row(0, 0, img, FillNone{}, std::false_type{}, req0, store_hook(std::integral_constant<int, 0>{}, prd));
                X3_T(tm)
                __builtin_amdgcn_s_waitcnt(0x0f70 | ((HL + D + 6) & 15) | (((HL + D + 6) >> 4) << 14));      // [w row 1][halo next][w row 2][6 stores]
                X3_T(tw)
                __syncthreads();
                X3_T(tb)
                mfma_row(1, 1, img, FillNone{}, std::false_type{}, req1, store_hook(std::integral_constant<int, 1>{}, prd));
                X3_T(tm1)
                __builtin_amdgcn_s_waitcnt(0x0f70 | (D + NST - 6));                                          // ... [w next row 0][2 stores]
                X3_T(tw)
                __syncthreads();
                pend = false; pend_ok = 0;
            } else {
                mfma_row(0, 0, img, FillNone{}, std::false_type{}, req0, nohook);
                // ---- row 1: outstanding [w row 1][halo next][w row 2]
                X3_T(tm)
                __builtin_amdgcn_s_waitcnt(0x0f70 | (HL + D));
                X3_T(tw)
                __syncthreads();
                X3_T(tb)
                mfma_row(1, 1, img, FillNone{}, std::false_type{}, req1, nohook);
                // ---- row 2: outstanding [halo next][w row 2][w next row 0]: the halo registers and row 2's weights
                X3_T(tm1)
                __builtin_amdgcn_s_waitcnt(0x0f70 | D);
                X3_T(tw)
                __syncthreads();
            }
#endif
            X3_T(tb)
#if X3_M16
            // the halo of the chunk after next: its requests ride in this row's gaps as the staging frees the registers (LATE, see mfma_row);
            // same issue order as a lump behind the row -- [weights of the next row 1][halo] -- so the vmcnt counts above hold
            mfma_row(2, 2, img, FillAll{}, std::integral_constant<int, 2>{}, [&](int rp) {
                if (rp < D) dma_piece(n1.tile, n1.g, 1, 1, n1.ok, rp); else if (rp == D) halo_prep(n2.tile, n2.g); else halo_slot(rp - D - 1); }, nohook);
            halo_slot(2);
#else
            mfma_row(2, 2, img, FillAll{}, std::false_type{}, [&](int rp) { dma_piece(n1.tile, n1.g, 1, 1, n1.ok, rp); }, nohook);
            load_halo(n2.tile, n2.g);
#endif
            X3_T(tm2)
#if X3_DIRECT
            if (g == nchunks - 1) {
                epilogue(cur, nullptr);
                if (!n1.ok) break;
                next_tile();
                img ^= 1;
                __builtin_amdgcn_s_waitcnt(X3_VMCNT(D + HL + NST));        // [w row 0] | [w row 1][halo next][NST stores]
                X3_T(tw)
                __syncthreads();
            } else {
                ++g;
                img ^= 1;
                __builtin_amdgcn_s_waitcnt(X3_VMCNT(D + HL));              // [w row 0] | [w row 1][halo next]
                X3_T(tw)
                __syncthreads();
            }
        }
#else
            if (g == nchunks - 1) epilogue(cur, epi_sep);
            if (!n1.ok) break;
            if (g == nchunks - 1) next_tile(); else ++g;
            img ^= 1;
        }
#endif
#if !X3_DIRECT
        if (DEFER && pend) {                                            // the last tile's stores
            const __amdgpu_buffer_rsrc_t prd = pend_rsrc();
#pragma unroll
            for (int i = 0; i < NST; ++i) pend_store(i, prd);
        }
#endif
#ifdef X3_STAMPS
        X3_T(te)
        dump_stamps();
#endif
    }
}

template <int BN, int EK>
int launch_x3(const IgemmArgs& a, hipStream_t s) {
    using Cfg = X3Cfg<BN>;
    auto kern = igemm_x3_kernel<BN, EK>;
    static PnnpPerDevice lds_once;
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    // ~150 KB of LDS: one 8-wave workgroup per CU resident; pnnp_set_persistent_split(n) launches n per CU with 1/n share each
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * ((a.Ntot + BN - 1) / BN);
    if (tiles <= 0) return PNNP_OK;
    const int wgs = pnnp_persistent_grid(tiles);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHR), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

}  // namespace

// a.w: the x3 pack of csrc/pack_jobs.hip (kind 2).  chan_per_seg: channels each K segment contributes (multiple of 8;
// of 16 when there are several segments).  Only what the 3x3 / stride-1 layers need: in_mul = out_mul = 1, no sub-pixel N.
int pnnp_igemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 2 || chan_per_seg <= 0 || (chan_per_seg & 7) || (a.nseg > 1 && (chan_per_seg & 15)) || a.Ntot <= 0) return PNNP_E_INVALID;
    if ((a.Ntot & 31) || a.in_mul != 1 || a.out_mul != 1 || a.n_sub || a.out_yoff || a.out_xoff) return PNNP_E_UNSUPPORTED;
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
    IgemmArgs b = a;
    b.chunks_per_seg = (chan_per_seg + 15) / 16;
    b.seg_channels = chan_per_seg;
    const int64_t wbytes = (int64_t)((a.Ntot + 31) / 32) * b.nseg * b.chunks_per_seg * 27648;
    if (wbytes >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    if (a.pool_dst || a.pool_codes) {
        // fused MaxPool2d(2): plain forward layers only (one destination, no mask / residual / accumulate), even sizes
        if (!a.pool_dst || !a.pool_codes || a.dst[1] || a.mask_mode[0] || a.accum[0] || a.addsrc || (a.OH & 1) || (a.OW & 1) || a.OH != a.DH ||
            a.OW != a.DW || (a.pool_cs & 3) || a.pool_cs < a.Ntot || ((uintptr_t)a.pool_dst & 15) || ((uintptr_t)a.pool_codes & 3))
            return PNNP_E_UNSUPPORTED;
        if ((int64_t)(a.OH / 2) * (a.OW / 2) * a.pool_cs * 4 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
        if (X3_SPEC && a.Ntot <= 1024) return pnnp_igemm_x3s_launch(b, a.Ntot >= 64, s);      // (conv_x3s keeps the bias vector in LDS: up to 1024 columns)
        return a.Ntot >= 64 ? launch_x3<64, EK_POOL>(b, s) : launch_x3<32, EK_POOL>(b, s);
    }
    // 64-column tiles unless they leave CUs idle: a layer with fewer (16 x 32 px x 64 ch) tiles than CUs (conv5_1 backward-data at
    // B = 16: 128; everything in a single-crop forward) runs on 32-column tiles, twice as many
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t tiles64 = (int64_t)((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * ((a.Ntot + 63) / 64);
    const bool wide = a.Ntot >= 64 && tiles64 * 4 >= (int64_t)cus * 3;
    if (X3_SPEC && a.Ntot <= 1024) return pnnp_igemm_x3s_launch(b, wide, s);
    // which epilogue (see the kernel template): forward, masked backward-data, or the general one
    const bool two = a.dst[1] != nullptr;
    const bool plain = !a.addsrc && !a.accum[0] && !(two && a.accum[1]);
    const bool is_fwd = plain && !a.mask_mode[0] && !(two && a.mask_mode[1]), is_bwd = plain && a.mask_mode[0] && (!two || a.mask_mode[1]);
    if (is_fwd) return wide ? launch_x3<64, EK_FWD>(b, s) : launch_x3<32, EK_FWD>(b, s);
    if (is_bwd) return wide ? launch_x3<64, EK_BWD>(b, s) : launch_x3<32, EK_BWD>(b, s);
    return wide ? launch_x3<64, EK_GEN>(b, s) : launch_x3<32, EK_GEN>(b, s);
}
