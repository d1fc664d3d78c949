// Implicit-GEMM convolution kernels (forward / backward-data of conv3x3, ConvTranspose2d k2s2
// forward and backward-data) -- see igemm.h for the data layout.
//   reference ops: archs/Unet.py:16-51,54-94 (Conv2d 3x3 pad 1, ConvTranspose2d 2x2 s2,
//   LeakyReLU(0.2)), archs/ResUnet.py:15-44.
//
// Structure: PERSISTENT workgroups (grid = CUs x resident workgroups per CU) walk a flattened
// list of (output tile, K chunk) work items.  While the MFMAs of item i run out of LDS, the
// global loads of item i+1 (input halo tile + weight tile) are already in flight into registers
// and are written to LDS after the barrier that retires item i (issue-early / write-late); the
// epilogue stores of a finished tile overlap the next tile's loads the same way.
#include "igemm.h"

namespace {

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
struct IgemmCfg {
    static constexpr int P = (TAPS == 9) ? 1 : 0;
    static constexpr int TH = WM * MT;
    static constexpr int HR = TH + 2 * P, HC = 32 + 2 * P, NPIX = HR * HC;
    static constexpr int KQ = KC / 4;
    static constexpr int XS_F4 = KQ * NPIX;            // float4 count of the input tile
    static constexpr int WS_F4 = TAPS * KQ * BN;       // float4 count of the weight tile
    static constexpr int NA = (XS_F4 + 255) / 256;     // float4 per thread, input tile
    static constexpr int NB = (WS_F4 + 255) / 256;     // float4 per thread, weight tile
    static constexpr int LDS_BYTES = (XS_F4 + WS_F4) * 16 + 4 * 4096;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(WN * NT * 32 == BN, "N tiling");
    static_assert(KC % 8 == 0, "KC multiple of 8");
};

__device__ __forceinline__ int xcd_remap(int id, int n) {
    // blocks are dealt round-robin over the 8 XCDs: give each XCD a contiguous run of logical
    // ids so neighbouring tiles (shared halo rows / shared input tile) hit the same L2.  Bijective.
    const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
__global__ void __launch_bounds__(256, 2)
igemm_kernel(const IgemmArgs a) {
    using Cfg = IgemmCfg<TAPS, KC, BN, MT, NT, WM, WN>;
    constexpr int P = Cfg::P, TH = Cfg::TH, HC = Cfg::HC, NPIX = Cfg::NPIX, KQ = Cfg::KQ;
    constexpr int NA = Cfg::NA, NB = Cfg::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4* xs = reinterpret_cast<f32x4*>(smem);       // ext-vector type: see the note at lds_load
    f32x4* ws = xs + Cfg::XS_F4;
    float* epi = reinterpret_cast<float*>(ws + Cfg::WS_F4);     // 4 waves x 32x32 floats (epilogue transpose)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    const int tiles_x = (a.DW + 31) >> 5, tiles_y = (a.DH + TH - 1) / TH;
    const int n_tiles = (a.Ntot + BN - 1) / BN;
    const int total = tiles_x * tiles_y * a.B * n_tiles;
    const int G = gridDim.x;
    const int nchunks = a.nseg * a.chunks_per_seg;
    const int K4 = nchunks * KQ;                       // Ktot / 4

    f32x4 ra[NA], rb[NB];                             // staging registers of the NEXT work item

    // Per-thread constants of the staging pattern (which halo pixel / channel quad / weight slot each of this thread's
    // float4s is), computed once.  The loads go through buffer resources: lane offset = a loop-invariant pixel index times
    // the segment's channel stride, tile / segment / chunk offset = one SGPR, and a halo pixel outside the image (four
    // compares against per-item scalar bounds) or an unused slot gets an offset beyond num_records, for which the hardware
    // returns zeros.  (Was: per-load 64-bit addresses behind divergent bounds branches -- ~180 VALU instructions per chunk,
    // which on this chip come straight out of the matrix pipe's time; now ~45.)
    constexpr unsigned OOB = 0x80000000u;              // host guarantees image and weight offsets < 2^31 bytes
    int rk[NA], qk[NA]; unsigned pixk[NA], cq16[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int i = tid + 256 * k;
        const int cq = i % KQ, pix = i / KQ;
        const int r = pix / HC, q = pix - r * HC;       // halo-tile coordinates, 0-based
        const bool used = i < Cfg::XS_F4;
        rk[k] = used ? r - P : (1 << 20);               // an unused slot fails every row test
        qk[k] = q - P;
        pixk[k] = (unsigned)((r * a.IW + q) * a.in_mul);
        cq16[k] = cq * 16;
    }
    unsigned wvoff[NB]; int nk[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int i = tid + 256 * k;
        const int n = i % BN, rest = i / BN;
        const int cq = rest % KQ, tt = rest / KQ;
        nk[k] = n;
        wvoff[k] = (i < Cfg::WS_F4) ? (unsigned)(((tt * K4 + cq) * a.Ntot + n) * 16) : OOB;     // byte offset at chunk 0, n0 = 0
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, 0x7fffffff, 0x00020000);
    auto bload = [](__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };

    struct Tile { int b, y0, x0, n0; };
    auto decode = [&](int t) {                          // scalar divisions: once per tile, not per chunk
        Tile o;
        const int nt_i = t % n_tiles;
        int m_i = t / n_tiles;
        const int tx = m_i % tiles_x; m_i /= tiles_x;
        o.x0 = tx * 32; o.y0 = (m_i % tiles_y) * TH; o.b = m_i / tiles_y; o.n0 = nt_i * BN;
        return o;
    };

    // issue the global loads of (tile, segment si, chunk-in-segment cc, global chunk g) into ra / rb (no wait)
    auto prefetch = [&](const Tile& tl, int si, int cc, int g) {
        const IgemmSeg sg = a.seg[si];
        const int c0 = sg.coff + cc * KC;
        const int mul = a.in_mul, sh = mul - 1;         // in_mul is 1 or 2: ceil(n / mul) == (n + sh) >> sh
        // input pixel of halo coordinate (r, q) (relative to the tile, -P based) is ((y0 + r) * mul + yoff, (x0 + q) * mul + xoff)
        const int rlo = ((-sg.yoff + sh) >> sh) - tl.y0, rhi = ((a.IH - sg.yoff + sh) >> sh) - tl.y0;
        const int qlo = ((-sg.xoff + sh) >> sh) - tl.x0, qhi = ((a.IW - sg.xoff + sh) >> sh) - tl.x0;
        // the resource starts `shift` elements before the image so that the scalar offset below is never negative
        const int shift = ((P * mul + 1) * a.IW + P * mul + 1) * sg.cstride;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(sg.ptr + ((int64_t)tl.b * a.IH * a.IW * sg.cstride - shift)), 0, 0x7fffffff, 0x00020000);
        const int soff = ((((tl.y0 - P) * mul + sg.yoff) * a.IW + (tl.x0 - P) * mul + sg.xoff) * sg.cstride + c0 + shift) * 4;
        const unsigned cs4 = (unsigned)sg.cstride * 4u;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const bool ok = rk[k] >= rlo && rk[k] < rhi && qk[k] >= qlo && qk[k] < qhi;
            ra[k] = bload(rs, ok ? __umul24(pixk[k], cs4) + cq16[k] : OOB, soff);
        }
        const int wso = (g * KQ * a.Ntot + tl.n0) * 16;
        const int nlim = a.Ntot - tl.n0;
#pragma unroll
        for (int k = 0; k < NB; ++k) rb[k] = bload(rsw, nk[k] < nlim ? wvoff[k] : OOB, wso);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int t = xcd_remap(blockIdx.x, G);
    int g = 0, si = 0, cc = 0;                          // global chunk, segment, chunk within segment
    Tile cur = decode(t < total ? t : 0), nxt = cur;
    if (t < total) prefetch(cur, 0, 0, 0);
    bool first = true;
    while (t < total) {
        if (!first) __syncthreads();                   // every wave is done reading the previous item
        first = false;
        // ---- write the staged registers into the LDS images
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int i = tid + 256 * k;
            if (i < Cfg::XS_F4) xs[(i % KQ) * NPIX + i / KQ] = ra[k];
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            const int i = tid + 256 * k;
            if (i < Cfg::WS_F4) ws[i] = rb[k];
        }
        __syncthreads();
        // ---- next work item: its loads fly while this item's MFMAs run
        int ng = g + 1, nt = t, nsi = si, ncc = cc + 1;
        if (ncc == a.chunks_per_seg) { ncc = 0; ++nsi; }
        if (ng == nchunks) {
            ng = 0; nsi = 0; ncc = 0; nt = t + G;
            if (nt < total) nxt = decode(nt);
        }
        if (nt < total) prefetch(nxt, nsi, ncc, ng);
        // ---- MFMA over taps x channel octets.  The LDS reads of step s+1 are issued BEFORE the MFMAs of
        // step s (register double buffer): left to itself hipcc issues each step's ds_reads right in front
        // of its MFMAs with s_waitcnt lgkmcnt(0), exposing the LDS latency once per 8-16 MFMAs.  The operands must be
        // ext_vector_type(4) values: loads of HIP's float4 STRUCT are four scalar loads that the SLP vectoriser re-emits
        // next to their first use, which silently undoes the prefetch.
        {
            constexpr int NSTEP = TAPS * (KC / 8);
            f32x4 av[2][MT], bv[2][NT];
            auto lds_load = [&](int step, f32x4 (&ax)[MT], f32x4 (&bx)[NT]) {
                const int tp = step / (KC / 8), j = step % (KC / 8);
                const int dy = (TAPS == 9) ? tp / 3 : 0, dx = (TAPS == 9) ? tp % 3 : 0;
#pragma unroll
                for (int i = 0; i < MT; ++i) ax[i] = xs[(2 * j + half) * NPIX + (wm * MT + i + dy) * HC + dx + l31];
#pragma unroll
                for (int i = 0; i < NT; ++i) bx[i] = ws[(tp * KQ + 2 * j + half) * BN + (wn * NT + i) * 32 + l31];
            };
            lds_load(0, av[0], bv[0]);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                if (step + 1 < NSTEP) {
                    lds_load(step + 1, av[(step + 1) & 1], bv[(step + 1) & 1]);      // the next step's DS reads first ...
                }
                __builtin_amdgcn_sched_barrier(0);     // (a fence, not a sched_group_barrier: a group of "n DS reads" is
                                                       //  satisfied by THIS step's reads just as well)
                const f32x4 (&ax)[MT] = av[step & 1];
                const f32x4 (&bx)[NT] = bv[step & 1];
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i].x, bx[k].x, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i].y, bx[k].y, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i].z, bx[k].z, acc[i][k], 0, 0, 0);
                        acc[i][k] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i].w, bx[k].w, acc[i][k], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);                                    // ... then this step's MFMAs
            }
        }
        if (g == nchunks - 1) {
            // ---- epilogue.  The accumulator has the channel on the lane and 16 pixels in registers
            // (col = lane&31, row = (r&3) + 8(r>>2) + 4(lane>>5)): storing it directly is 16 dword
            // stores per tile and is store-ISSUE bound.  Each wave transposes its 32x32 tile through
            // a private 4 KB LDS patch instead, so a lane owns 4 consecutive channels of a pixel and
            // all epilogue traffic (mask / residual / accumulate loads, the store) is 16 bytes wide:
            // 4 instructions per tile, each covering 8 whole 128-byte pixel rows.
            const int b = cur.b, x0 = cur.x0, y0 = cur.y0, n0 = cur.n0;
            float* eb = epi + wave * 1024;
            const bool any_mask = (a.mask_mode[0] | a.mask_mode[1]) != 0, any_accum = (a.accum[0] | a.accum[1]) != 0;
            const float* mask_any = a.mask_mode[0] ? a.mask[0] : a.mask[1];
            const float* accum_any = a.accum[0] ? a.dst[0] : a.dst[1];
            const int q4 = (lane & 7) * 4, pr = lane >> 3;
            const bool uni = (a.n_split & 31) == 0 && (a.n_sub & 31) == 0;      // kernel-uniform
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const int n = n0 + (wn * NT + k) * 32 + q4;             // first of this lane's 4 channels
                const bool n_ok = n < a.Ntot;
                const int d = (n >= a.n_split) ? 1 : 0;
                // sub-pixel mode (ConvTranspose2d forward): column n = sub*n_sub + channel, the four
                // sub-pixels (a,c) = (sub>>1, sub&1) of a 2x2 output block are one GEMM
                const int sub = a.n_sub ? n / a.n_sub : 0;
                const int nn = n - sub * a.n_sub;
                const int yoff = a.out_yoff + (sub >> 1), xoff = a.out_xoff + (sub & 1);
                const int ch = nn - (d ? a.n_split : 0);
                float* dst = a.dst[d];
                const float* msk = a.mask[d];
                const int cs = a.dst_cs[d], mmode = a.mask_mode[d], accum = a.accum[d];
                float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.bias && n_ok) bias = *reinterpret_cast<const float4*>(a.bias + nn);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int py = y0 + wm * MT + i;
                    const int oy = py * a.out_mul + yoff;
                    const int rowo = (int)(((int64_t)b * a.OH + oy) * a.OW) * cs + ch;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        eb[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[i][k][r];
                        acc[i][k][r] = 0.f;
                    }
                    if (uni) {
                        // Fast path (n_split / n_sub multiples of 32: destination, sub-pixel and channel base are the same for the
                        // whole wave): buffer-resource loads and stores on image b of the destination -- a lane outside the tile
                        // domain / image gets an out-of-range offset (loads return zeros, stores are dropped), so there are no
                        // per-store branches and no 64-bit addresses -- and a branch-free activation.
                        const int nwv = __builtin_amdgcn_readfirstlane(n0 + (wn * NT + k) * 32);
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
                        const __amdgpu_buffer_rsrc_t rad = __builtin_amdgcn_make_buffer_rsrc((void*)((a.addsrc && du == 0) ? a.addsrc + imgo : dstb), 0, ibytes, 0x00020000);
                        const bool use_add2 = a.addsrc && du == 0;
                        unsigned vo[4];
                        f32x4 v2[4], m2[4], ad2[4];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int p = pr + 8 * it, px = x0 + p;
                            const int oxp = px * a.out_mul + xo2;
                            const bool ok2 = rowok && px < a.DW && n_ok && oxp >= 0 && oxp < a.OW;
                            vo[it] = ok2 ? (unsigned)(((oy2 * a.OW + oxp) * cs2 + chw + q4) * 4) : 0x80000000u;
                            v2[it] = *reinterpret_cast<const f32x4*>(eb + p * 32 + q4);
                        }
                        if (mm2) {
#pragma unroll
                            for (int it = 0; it < 4; ++it) m2[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rm, vo[it], 0, 0));
                        }
                        if (use_add2) {
#pragma unroll
                            for (int it = 0; it < 4; ++it) ad2[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rad, vo[it], 0, 0));
                        }
                        if (acc2) {
#pragma unroll
                            for (int it = 0; it < 4; ++it) ad2[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, vo[it], 0, 0));
                        }
                        const f32x4 bias4 = {bias.x, bias.y, bias.z, bias.w};
                        const float aslope = a.act == 1 ? 0.2f : (a.act == 2 ? 0.f : 1.f), mslope = mm2 == 1 ? 0.2f : 0.f;
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            f32x4 o = v2[it] + bias4;
                            if (use_add2) o += ad2[it];
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], aslope * o[c]);
                            if (mm2) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[c] *= (m2[it][c] > 0.f) ? 1.f : mslope;
                            }
                            if (acc2) o += ad2[it];
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rd, vo[it], 0, 0);
                        }
                    } else if (py < a.DH && oy >= 0 && oy < a.OH) {
                        // Epilogue loads are issued under WAVE-UNIFORM conditions (kernel arguments) and from
                        // always-valid addresses (offset 0 for lanes that do not take part), never under a
                        // per-lane branch: a per-element "load or not" makes hipcc branch around every load and
                        // wait vmcnt(0) after each one.
                        float4 v[4], mv[4], add[4];
                        int idx[4]; bool ok[4];
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            const int p = pr + 8 * it, px = x0 + p;
                            const int oxp = px * a.out_mul + xoff;
                            ok[it] = px < a.DW && n_ok && oxp >= 0 && oxp < a.OW;
                            idx[it] = rowo + oxp * cs;
                            v[it] = *reinterpret_cast<const float4*>(eb + p * 32 + q4);
                            mv[it] = make_float4(1.f, 1.f, 1.f, 1.f);
                            add[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        if (any_mask) {
                            const float* mb = mmode ? msk : mask_any;
#pragma unroll
                            for (int it = 0; it < 4; ++it) mv[it] = *reinterpret_cast<const float4*>(mb + ((mmode && ok[it]) ? idx[it] : 0));
                        }
                        if (a.addsrc) {
#pragma unroll
                            for (int it = 0; it < 4; ++it) add[it] = *reinterpret_cast<const float4*>(a.addsrc + ((d == 0 && ok[it]) ? idx[it] : 0));
                        }
                        if (any_accum) {
                            const float* ab = accum ? dst : accum_any;
#pragma unroll
                            for (int it = 0; it < 4; ++it) add[it] = *reinterpret_cast<const float4*>(ab + ((accum && ok[it]) ? idx[it] : 0));
                        }
                        const bool use_add = (a.addsrc && d == 0), use_acc = accum != 0;
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            float o[4] = {v[it].x + bias.x, v[it].y + bias.y, v[it].z + bias.z, v[it].w + bias.w};
                            const float ad[4] = {add[it].x, add[it].y, add[it].z, add[it].w};
                            const float mk[4] = {mv[it].x, mv[it].y, mv[it].z, mv[it].w};
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                if (use_add) o[c] += ad[c];
                                if (a.act == 1) o[c] = o[c] > 0.f ? o[c] : 0.2f * o[c];
                                else if (a.act == 2) o[c] = fmaxf(o[c], 0.f);
                                if (mmode) o[c] *= (mk[c] > 0.f) ? 1.f : (mmode == 1 ? 0.2f : 0.f);
                                if (use_acc) o[c] += ad[c];
                            }
                            if (ok[it]) *reinterpret_cast<float4*>(dst + idx[it]) = make_float4(o[0], o[1], o[2], o[3]);
                        }
                    }
                }
            }
        }
        t = nt; g = ng; si = nsi; cc = ncc;
        if (g == 0) cur = nxt;
    }
}

template <int TAPS, int KC, int BN, int MT, int NT, int WM, int WN>
int launch_cfg(const IgemmArgs& a, hipStream_t s) {
    using Cfg = IgemmCfg<TAPS, KC, BN, MT, NT, WM, WN>;
    auto kern = igemm_kernel<TAPS, KC, BN, MT, NT, WM, WN>;
    static PnnpPerDevice lds_once, occ;       // per-device: LDS limit raised, resident workgroups per CU
    if (pnnp_allow_lds(lds_once, kern, Cfg::LDS_BYTES) != PNNP_OK) return PNNP_E_LAUNCH;
    const int per_cu = occ.get([&] {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kern), 256, Cfg::LDS_BYTES) != hipSuccess || n < 1) n = 1;
        return n;
    });
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int tiles = ((a.DW + 31) / 32) * ((a.DH + Cfg::TH - 1) / Cfg::TH) * a.B * ((a.Ntot + BN - 1) / BN);
    if (tiles <= 0) return PNNP_OK;
    int wgs = per_cu * cus;
    if (wgs > tiles) wgs = tiles;
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), Cfg::LDS_BYTES, s, a);
    return pnnp_launch_status();
}

int pick_bn(int ntot) { return ntot >= 128 ? 128 : (ntot >= 64 ? 64 : 32); }

int pick_kc(int bn, int chan, int taps = 9) {
    if (taps == 1) {                 // 1x1: the weight tile is 9x smaller, so a chunk can carry 4x the channels
        int kc = 16;                 // (16 MFMAs per chunk and wave at KC = 8 drown in barriers and staging)
        while (kc > 8 && (chan % kc)) kc >>= 1;
        return kc;
    }
    // weight tile <= 36 KB and at most 16 channels per chunk: ~40 KB of LDS per workgroup keeps
    // 3-4 workgroups resident per CU
    int kc = 1024 / bn;
    if (kc > 16) kc = 16;
    while (kc > 8 && (chan % kc)) kc >>= 1;
    return kc;
}

// (Measured and rejected: twice-as-tall tiles (MT=4) -- the extra accumulators push the kernel past 256
// VGPRs, it spills, and the 32/64-channel layers lose 5-20 %.)
template <int TAPS>
int launch_taps(const IgemmArgs& a, int kc_chan, hipStream_t s) {
    const int bn = pick_bn(a.Ntot);
    const int kc = pick_kc(bn, kc_chan, TAPS);
    if (kc_chan % kc) return PNNP_E_UNSUPPORTED;
    if constexpr (TAPS == 1) {
        if (kc == 32) {
            if (bn == 32) return launch_cfg<TAPS, 32, 32, 2, 1, 4, 1>(a, s);
            if (bn == 64) return launch_cfg<TAPS, 32, 64, 2, 2, 4, 1>(a, s);
            return launch_cfg<TAPS, 32, 128, 2, 2, 2, 2>(a, s);
        }
        if (kc == 16 && bn == 128) return launch_cfg<TAPS, 16, 128, 2, 2, 2, 2>(a, s);
    }
    if (bn == 32) {
        if (kc == 16) return launch_cfg<TAPS, 16, 32, 2, 1, 4, 1>(a, s);
        return launch_cfg<TAPS, 8, 32, 2, 1, 4, 1>(a, s);
    }
    if (bn == 64) {
        if (kc == 16) return launch_cfg<TAPS, 16, 64, 2, 2, 4, 1>(a, s);
        return launch_cfg<TAPS, 8, 64, 2, 2, 4, 1>(a, s);
    }
    return launch_cfg<TAPS, 8, 128, 2, 2, 2, 2>(a, s);
}

}  // namespace

int pnnp_igemm_launch(const IgemmArgs& a, int taps, int chan_per_seg, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 9 || chan_per_seg <= 0 || (chan_per_seg & 7) || a.Ntot <= 0) return PNNP_E_INVALID;
    if (a.addsrc && a.accum[0]) return PNNP_E_UNSUPPORTED;   // the epilogue shares one register set for both
    // 16-byte epilogue accesses: channel counts / splits in multiples of 4, 16-byte aligned bases
    if ((a.Ntot & 3) || (a.dst_cs[0] & 3) || (a.dst[1] && ((a.dst_cs[1] & 3) || (a.n_split & 3)))) return PNNP_E_UNSUPPORTED;
    if ((((uintptr_t)a.dst[0]) | ((uintptr_t)a.dst[1]) | ((uintptr_t)a.bias) | ((uintptr_t)a.mask[0]) | ((uintptr_t)a.mask[1]) |
         ((uintptr_t)a.addsrc)) & 15) return PNNP_E_INVALID;
    for (int i = 0; i < a.nseg; ++i) {                       // 32-bit element offsets when staging
        if ((int64_t)a.B * a.IH * a.IW * a.seg[i].cstride >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
        // buffer-resource addressing: byte offsets inside one image (+ a (P*mul+1)-row shift) must stay below 2^31
        if (((int64_t)a.IH + 4) * a.IW * a.seg[i].cstride * 4 >= (1ll << 31) || a.in_mul < 1 || a.in_mul > 2) return PNNP_E_UNSUPPORTED;
        if (a.seg[i].yoff < -1 || a.seg[i].xoff < -1) return PNNP_E_UNSUPPORTED;
    }
    for (int d = 0; d < 2; ++d)                              // 32-bit element offsets in the epilogue
        if (a.dst[d] && (int64_t)a.B * a.OH * a.OW * a.dst_cs[d] >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    IgemmArgs b = a;
    const int kc = pick_kc(pick_bn(a.Ntot), chan_per_seg, taps);
    b.chunks_per_seg = chan_per_seg / kc;
    if (taps == 9) return launch_taps<9>(b, chan_per_seg, s);
    if (taps == 1) return launch_taps<1>(b, chan_per_seg, s);
    return PNNP_E_UNSUPPORTED;
}
