// Pointwise GEMMs (ConvTranspose2d k2 s2, Conv2d 1x1, stride-2 3x3: one tap per K segment) on the bf16 matrix cores: argument validation
// and dispatch.  The kernel is csrc/gemm_x3s.hip (round 4: producer and consumer waves).  Round 3's kernel (32-channel items, every wave
// staging between its own MFMAs) lived in this file until round 5 (git history: gemm_x3_kernel; DESIGN Appendix A.1).
#include "igemm.h"

int pnnp_gemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);
int pnnp_gemm_x3_check(const IgemmArgs& a, int chan_per_seg);        // the validation alone (csrc/conv_api.hip: the fp16x2 entries share it)
int pnnp_gemm_x3s_launch(const IgemmArgs& a, hipStream_t s);        // csrc/gemm_x3s.hip

namespace { constexpr int WBLK = 2 * 3 * 32 * 16; }                 // one 16-channel k-step of one 32-column block: [octet 2][piece 3][32][16 B] = 3072

// a.w: x3 pack with ONE tap: [N/32][K/16][octet 2][piece 3][32][8 bf16].  chan_per_seg: channels per K segment, a multiple of 32.
int pnnp_gemm_x3_check(const IgemmArgs& a, int chan_per_seg) {
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
    return PNNP_OK;
}
int pnnp_gemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s) {
    const int rc = pnnp_gemm_x3_check(a, chan_per_seg);
    if (rc != PNNP_OK) return rc;
    IgemmArgs b = a;
    b.seg_channels = chan_per_seg;
    const int64_t wbytes = (int64_t)(a.Ntot / 32) * b.nseg * (chan_per_seg / 16) * WBLK;
    if (wbytes >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    b.chunks_per_seg = chan_per_seg / 16;                           // csrc/gemm_x3s.hip walks K in 16-channel items
    return pnnp_gemm_x3s_launch(b, s);
}
