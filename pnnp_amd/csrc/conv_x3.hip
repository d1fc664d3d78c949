// 3x3 convolution (forward / backward-data) on the bf16 matrix cores: argument validation and dispatch.
//
// The kernel itself is csrc/conv_x3s.hip (round 4: 8 MFMA-only consumer waves + 4 producer waves).  Rounds 2-3's kernel -- every wave staging
// its own share of the next halo between its MFMAs -- lived in this file until round 5 (git history: igemm_x3_kernel); its scheme, LDS images,
// weight pack and tile are what conv_x3s.hip keeps, and DESIGN Appendix A.1 tells what sixteen experiments on it taught.  It was reachable
// only for layers with more than 1024 output channels (conv_x3s keeps the bias vector in LDS); those are refused now
// (pnnp_x3_supported), and the engines run them on the fp32-MFMA kernels.
#include "igemm.h"

int pnnp_igemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);
int pnnp_igemm_x3s_launch(const IgemmArgs& a, int wide, hipStream_t s);      // csrc/conv_x3s.hip

namespace { constexpr int TH = 16; }                                          // rows of a tile (csrc/conv_x3s.hip)

// a.w: the x3 pack of csrc/pack_jobs.hip (kind 2).  chan_per_seg: channels each K segment contributes (multiple of 8;
// of 16 when there are several segments).  Only what the 3x3 / stride-1 layers need: in_mul = out_mul = 1, no sub-pixel N.
int pnnp_igemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s) {
    if (a.nseg < 1 || a.nseg > 2 || chan_per_seg <= 0 || (chan_per_seg & 7) || (a.nseg > 1 && (chan_per_seg & 15)) || a.Ntot <= 0) return PNNP_E_INVALID;
    if (a.Ntot > 1024) return PNNP_E_UNSUPPORTED;                     // (conv_x3s keeps the bias vector in LDS: up to 1024 columns)
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
        return pnnp_igemm_x3s_launch(b, a.Ntot >= 64, s);
    }
    // 64-column tiles unless they leave CUs idle: a layer with fewer (16 x 32 px x 64 ch) tiles than CUs (conv5_1 backward-data at
    // B = 16: 128; everything in a single-crop forward) runs on 32-column tiles, twice as many
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t tiles64 = (int64_t)((a.DW + 31) / 32) * ((a.DH + TH - 1) / TH) * a.B * ((a.Ntot + 63) / 64);
    const bool wide = a.Ntot >= 64 && tiles64 * 4 >= (int64_t)cus * 3;
    return pnnp_igemm_x3s_launch(b, wide, s);
}
