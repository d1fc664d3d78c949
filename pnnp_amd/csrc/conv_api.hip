// C-ABI entry points of the convolution stack: weight re-packing and the named
// forward / backward-data operations built on the implicit-GEMM kernel (conv_igemm.hip).
#include "igemm.h"
#include "h2.h"

int pnnp_igemm_launch(const IgemmArgs& a, int taps, int chan_per_seg, hipStream_t s);
int pnnp_igemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);       // csrc/conv_x3.hip (3x3, bf16x3 split)
int pnnp_gemm_x3_launch(const IgemmArgs& a, int chan_per_seg, hipStream_t s);        // csrc/gemm_x3.hip (one tap per K segment, bf16x3 split)
int pnnp_gemm_x3_check(const IgemmArgs& a, int chan_per_seg);                        // its argument validation alone
int pnnp_gemm_h2s_launch(const H2Args& a, hipStream_t s);                             // csrc/gemm_h2s.hip (the same GEMMs on the fp16x2 scheme)

namespace {

// dst[((t*K/4 + k/4)*N + n)*4 + k%4] = src[off + k*sk + n*sn + tap(t)*st],  tap(t) = flip ? T-1-t : t
// rows k >= Kvalid are zero (channel padding of the 4-channel network input / output).
int pack_launch(const float* src, float* dst, int T, int K, int N, int64_t sk, int64_t sn, int64_t st, int64_t off,
                int flip, void* stream, int Kvalid = 1 << 30, int Ndst = 0, int n_off = 0) {
    if (!src || !dst || T <= 0 || K <= 0 || N <= 0 || (K & 3)) return PNNP_E_INVALID;
    PnnpPackJob j{};
    j.src = src; j.dst = dst; j.kind = 0; j.T = T; j.K = K; j.N = N; j.sk = sk; j.sn = sn; j.st = st; j.off = off;
    j.flip = flip; j.Kvalid = Kvalid; j.Ndst = Ndst ? Ndst : N; j.n_off = n_off;
    return pnnp_pack_jobs_f32(&j, 1, stream);           // one job through the batched kernel (csrc/pack_jobs.hip)
}

void base_args(IgemmArgs& a) {
    a = IgemmArgs{};
    a.in_mul = 1; a.out_mul = 1;
    a.n_split = 1 << 30;
}

// argument builders shared by the fp32-MFMA entry points and their bf16x3 twins
void fwd_args(IgemmArgs& a, const float* x1, int C1, const float* x2, int C2, const void* w_packed, const float* bias,
              const float* residual, float* y, int B, int H, int W, int Cout, int act) {
    base_args(a);
    a.seg[0] = IgemmSeg{x1, C1, 0, 0, 0};
    a.nseg = 1;
    if (x2) { a.seg[1] = IgemmSeg{x2, C2, 0, 0, 0}; a.nseg = 2; }
    a.IH = H; a.IW = W; a.B = B; a.DH = H; a.DW = W; a.OH = H; a.OW = W;
    a.w = reinterpret_cast<const float*>(w_packed); a.Ntot = Cout;
    a.dst[0] = y; a.dst_cs[0] = Cout;
    a.bias = bias; a.act = act; a.addsrc = residual;
}

void bwd_args(IgemmArgs& a, const float* g, int Cout, const void* w_dgrad, float* dx1, int C1, const float* mask1, int mode1, int accum1,
              float* dx2, int C2, const float* mask2, int mode2, int accum2, int B, int H, int W) {
    base_args(a);
    a.seg[0] = IgemmSeg{g, Cout, 0, 0, 0};
    a.nseg = 1;
    a.IH = H; a.IW = W; a.B = B; a.DH = H; a.DW = W; a.OH = H; a.OW = W;
    a.w = reinterpret_cast<const float*>(w_dgrad); a.Ntot = C1 + (dx2 ? C2 : 0);
    a.dst[0] = dx1; a.dst_cs[0] = C1; a.mask[0] = mask1; a.mask_mode[0] = mask1 ? mode1 : 0; a.accum[0] = accum1;
    if (dx2) {
        a.n_split = C1;                              // destination is chosen per output column (lane)
        a.dst[1] = dx2; a.dst_cs[1] = C2; a.mask[1] = mask2; a.mask_mode[1] = mask2 ? mode2 : 0; a.accum[1] = accum2;
    }
}

void bwd_res_args(IgemmArgs& a, const float* g, int Cout, const void* w_dgrad, float* dx, int C1, const float* addsrc, const float* mask,
                  int mode, int B, int H, int W) {
    base_args(a);
    a.seg[0] = IgemmSeg{g, Cout, 0, 0, 0};
    a.nseg = 1;
    a.IH = H; a.IW = W; a.B = B; a.DH = H; a.DW = W; a.OH = H; a.OW = W;
    a.w = reinterpret_cast<const float*>(w_dgrad); a.Ntot = C1;
    a.dst[0] = dx; a.dst_cs[0] = C1; a.addsrc = addsrc; a.mask[0] = mask; a.mask_mode[0] = mask ? mode : 0;
}

void convt_fwd_args(IgemmArgs& a, const float* x, int Cin, const void* w_packed, const float* bias, float* y, int B, int H, int W, int Cout) {
    base_args(a);
    a.seg[0] = IgemmSeg{x, Cin, 0, 0, 0};
    a.nseg = 1;
    a.IH = H; a.IW = W; a.B = B; a.DH = H; a.DW = W;
    a.OH = 2 * H; a.OW = 2 * W; a.out_mul = 2;
    a.w = reinterpret_cast<const float*>(w_packed); a.Ntot = 4 * Cout; a.n_sub = Cout;       // the four sub-pixel GEMMs in one launch
    a.dst[0] = y; a.dst_cs[0] = Cout; a.bias = bias;
}

void convt_bwd_args(IgemmArgs& a, const float* g, int Cout, const void* w_dgrad, float* dx, int Cin, const float* mask, int mode, int B, int H, int W) {
    base_args(a);
    for (int s = 0; s < 4; ++s) a.seg[s] = IgemmSeg{g, Cout, 0, s >> 1, s & 1};
    a.nseg = 4; a.in_mul = 2;
    a.IH = 2 * H; a.IW = 2 * W; a.B = B; a.DH = H; a.DW = W; a.OH = H; a.OW = W;
    a.w = reinterpret_cast<const float*>(w_dgrad); a.Ntot = Cin;
    a.dst[0] = dx; a.dst_cs[0] = Cin; a.mask[0] = mask; a.mask_mode[0] = mask ? mode : 0;
}

void s2_fwd_args(IgemmArgs& a, const float* x, int Cin, const void* w_packed, const float* bias, float* y, int B, int H, int W, int Cout, int act) {
    base_args(a);
    for (int t = 0; t < 9; ++t) a.seg[t] = IgemmSeg{x, Cin, 0, t / 3 - 1, t % 3 - 1};
    a.nseg = 9; a.in_mul = 2;
    a.IH = H; a.IW = W; a.B = B; a.DH = H / 2; a.DW = W / 2; a.OH = H / 2; a.OW = W / 2;
    a.w = reinterpret_cast<const float*>(w_packed); a.Ntot = Cout;
    a.dst[0] = y; a.dst_cs[0] = Cout; a.bias = bias; a.act = act;
}

// class cls = (py, px) of the input pixel parity; returns the number of taps (K segments) that reach it
int s2_bwd_args(IgemmArgs& a, int cls, const float* g, int Cout, float* dx, int Cin, const float* mask, int mode, int accum, int B, int H, int W) {
    const int py = cls >> 1, px = cls & 1;
    base_args(a);
    int ns = 0;
    for (int iy = 0; iy < (py ? 2 : 1); ++iy)
        for (int ix = 0; ix < (px ? 2 : 1); ++ix) {
            const int dy = py ? 2 * iy : 1, dxk = px ? 2 * ix : 1;      // tap (dy, dxk)
            a.seg[ns++] = IgemmSeg{g, Cout, 0, dy == 0 ? 1 : 0, dxk == 0 ? 1 : 0};
        }
    a.nseg = ns;
    a.IH = H / 2; a.IW = W / 2; a.B = B; a.DH = H / 2; a.DW = W / 2; a.OH = H; a.OW = W;
    a.out_mul = 2; a.out_yoff = py; a.out_xoff = px;
    a.Ntot = Cin;
    a.dst[0] = dx; a.dst_cs[0] = Cin; a.mask[0] = mask; a.mask_mode[0] = mask ? mode : 0; a.accum[0] = accum;
    return ns;
}

}  // namespace

extern "C" {

// Conv2d weight [Cout][Cin][kh][kw] (kh*kw = taps: 9 or 1)
//   forward pack  [tap][Cin/4][Cout][4]                      (K = Cin, N = Cout)
//   dgrad pack    [tap'][Cout/4][Cin][4], tap' = taps-1-tap  (K = Cout, N = Cin; flipped kernel)
//   Cin_pad / Cout_pad (>= Cin / Cout, multiples of 4): K rows beyond the real channel count are
//   zero, so the 4-channel network input / output can travel as 8-channel NHWC tensors.
int pnnp_pack_conv_weight_f32(const float* w, float* fwd, float* dgrad, int Cout, int Cin, int taps,
                              int Cin_pad, int Cout_pad, void* stream) {
    if (Cin_pad < Cin || Cout_pad < Cout) return PNNP_E_INVALID;
    int rc = PNNP_OK;
    if (fwd) rc = pack_launch(w, fwd, taps, Cin_pad, Cout, taps, (int64_t)Cin * taps, 1, 0, 0, stream, Cin);
    if (rc == PNNP_OK && dgrad) rc = pack_launch(w, dgrad, taps, Cout_pad, Cin, (int64_t)Cin * taps, taps, 1, 0, 1, stream, Cout);
    return rc;
}

// ConvTranspose2d weight [Cin][Cout][2][2]
//   forward pack  [Cin/4][4*Cout][4], column n = s*Cout + co, s = (a,c)   (K = Cin, N = 4*Cout)
//   dgrad pack    [(s*Cout + co)/4][Cin][4]                          (K = 4*Cout, N = Cin)
int pnnp_pack_convt_weight_f32(const float* w, float* fwd, float* dgrad, int Cin, int Cout, void* stream) {
    int rc = PNNP_OK;
    for (int s = 0; s < 4 && rc == PNNP_OK; ++s) {
        if (fwd) rc = pack_launch(w, fwd, 1, Cin, Cout, (int64_t)Cout * 4, 4, 0, s, 0, stream, 1 << 30, 4 * Cout, s * Cout);
        if (rc == PNNP_OK && dgrad)
            rc = pack_launch(w, dgrad + (int64_t)s * Cout * Cin, 1, Cout, Cin, 4, (int64_t)Cout * 4, 0, s, 0, stream);
    }
    return rc;
}

// y = act(conv3x3(cat[x1, x2]) + bias)    archs/Unet.py:55-92 ; NHWC fp32, pad 1, stride 1.
//   x2 may be null (C2 = 0).  taps = 9 (3x3) or 1 (1x1).  act: 0 none, 1 LeakyReLU(0.2), 2 ReLU.
//   residual (optional, NHWC [B][H][W][Cout]) is added before the activation.
int pnnp_conv_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* w_packed, const float* bias,
                      const float* residual, float* y, int B, int H, int W, int Cout, int taps, int act, void* stream) {
    if (!x1 || !w_packed || !y || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    fwd_args(a, x1, C1, x2, C2, w_packed, bias, residual, y, B, H, W, Cout, act);
    return pnnp_igemm_launch(a, taps, C1, as_stream(stream));
}

// backward-data of the layer above: g = dL/d(pre-activation output) [B][H][W][Cout] ->
//   dx1 [..][C1] (and dx2 [..][C2] for a concat layer).  Each destination may be multiplied by the
//   activation derivative of the tensor that produced it (mask = that saved activation,
//   mode 1 LeakyReLU', 2 ReLU') and may accumulate (+=).  w_dgrad from pnnp_pack_conv_weight_f32.
int pnnp_conv_bwd_data_f32(const float* g, int Cout, const float* w_dgrad,
                           float* dx1, int C1, const float* mask1, int mode1, int accum1,
                           float* dx2, int C2, const float* mask2, int mode2, int accum2,
                           int B, int H, int W, int taps, void* stream) {
    if (!g || !w_dgrad || !dx1 || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    bwd_args(a, g, Cout, w_dgrad, dx1, C1, mask1, mode1, accum1, dx2, C2, mask2, mode2, accum2, B, H, W);
    return pnnp_igemm_launch(a, taps, Cout, as_stream(stream));
}

// backward-data through a block with an identity shortcut (ResidualBlock, archs/modules.py:193-197):
//   dx = (conv_bwd_data(g) + addsrc) * act'(mask)      addsrc, dx: [B][H][W][C1]
int pnnp_conv_bwd_data_res_f32(const float* g, int Cout, const float* w_dgrad, float* dx, int C1,
                               const float* addsrc, const float* mask, int mode, int B, int H, int W, int taps, void* stream) {
    if (!g || !w_dgrad || !dx || !addsrc || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    bwd_res_args(a, g, Cout, w_dgrad, dx, C1, addsrc, mask, mode, B, H, W);
    return pnnp_igemm_launch(a, taps, Cout, as_stream(stream));
}

// ---------------------------------------------------------------- 3x3 / stride 1 / pad 1 on the bf16 matrix cores (bf16x3 split)
// Same contracts as pnnp_conv_fwd_f32 / pnnp_conv_bwd_data_f32 / pnnp_conv_bwd_data_res_f32 with taps = 9; the weights are the x3
// packs of pnnp_pack_jobs_add_x3 (C1 < 16 allowed: the pack's reduction length is padded to 16).
int pnnp_x3_supported(int K, int N) { return (K > 0 && (K % 8) == 0 && N > 0 && (N % 32) == 0 && N <= 1024) ? 1 : 0; }      // (<= 1024 channels written: the kernels keep the bias vector in LDS)
// The bf16x3 forward / backward-data / pointwise launchers address ONE image of a map through a 32-bit byte offset (bit 31 = "outside"):
// does a [H][W][cstride] fp32 image fit?  (the limit pnnp_igemm_x3_launch / pnnp_gemm_x3_launch enforce with PNNP_E_UNSUPPORTED;
// callers choose the kernel family with this BEFORE packing weights)
int pnnp_x3_image_fits(int H, int W, int cstride) {
    return (H > 0 && W > 0 && cstride > 0 && ((int64_t)H + 4) * W * cstride * 4 < (1ll << 31)) ? 1 : 0;
}

int pnnp_conv3x3_x3_fwd_f32(const float* x1, int C1, const float* x2, int C2, const void* w_x3, const float* bias,
                            const float* residual, float* y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_x3 || !y || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    fwd_args(a, x1, C1, x2, C2, w_x3, bias, residual, y, B, H, W, Cout, act);
    return pnnp_igemm_x3_launch(a, C1, as_stream(stream));
}

// pnnp_conv3x3_x3_fwd_f32 + MaxPool2d(2) of its (activated) output in the same kernel: y as before, pooled [B][H/2][W/2][Cout]
// and the codes of pnnp_maxpool2_fwd_codes_f32 (bit-identical to running that kernel on y).  H, W even; no residual.
int pnnp_conv3x3_x3_fwd_pool_f32(const float* x1, int C1, const float* x2, int C2, const void* w_x3, const float* bias, float* y,
                                 float* pooled, unsigned char* codes, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_x3 || !y || !pooled || !codes || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1) || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    fwd_args(a, x1, C1, x2, C2, w_x3, bias, nullptr, y, B, H, W, Cout, act);
    a.pool_dst = pooled; a.pool_codes = codes; a.pool_cs = Cout;
    return pnnp_igemm_x3_launch(a, C1, as_stream(stream));
}

int pnnp_conv3x3_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad,
                                 float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                 float* dx2, int C2, const float* mask2, int mode2, int accum2,
                                 int B, int H, int W, void* stream) {
    if (!g || !w_x3_dgrad || !dx1 || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    bwd_args(a, g, Cout, w_x3_dgrad, dx1, C1, mask1, mode1, accum1, dx2, C2, mask2, mode2, accum2, B, H, W);
    return pnnp_igemm_x3_launch(a, Cout, as_stream(stream));
}

int pnnp_conv3x3_x3_bwd_data_res_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int C1,
                                     const float* addsrc, const float* mask, int mode, int B, int H, int W, void* stream) {
    if (!g || !w_x3_dgrad || !dx || !addsrc || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    bwd_res_args(a, g, Cout, w_x3_dgrad, dx, C1, addsrc, mask, mode, B, H, W);
    return pnnp_igemm_x3_launch(a, Cout, as_stream(stream));
}

// ---------------------------------------------------------------- 3x3 / stride 1 / pad 1 on the fp16 matrix cores (fp16x2 split, csrc/conv_h2s.hip)
// Same layer contracts as the pnnp_conv3x3_x3_* entries; in addition every tensor that is split on the fly comes with its amax slot
// (csrc/h2.h), every destination may report max |stored value| into a slot, a forward layer may write the sign bits of its activated output
// (pnnp_h2_bits_words words) and backward-data may take its act' masks as those bits instead of the float32 activation.
int pnnp_h2_supported(int K, int N) { return (K > 0 && (K % 8) == 0 && N > 0 && (N % 32) == 0 && N <= 1024) ? 1 : 0; }

int pnnp_conv3x3_h2_fwd_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2, int C2, const unsigned* amax_x2,
                            const void* w_h2, const unsigned* amax_w, const float* bias, const float* residual, float* y,
                            unsigned* amax_y, unsigned* bits_y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_h2 || !y || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args a{};
    fwd_args(a.g, x1, C1, x2, C2, w_h2, bias, residual, y, B, H, W, Cout, act);
    a.amax_in[0] = amax_x1; a.amax_in[1] = x2 ? amax_x2 : nullptr; a.amax_w = amax_w; a.amax_out[0] = amax_y;
    a.bits_out = bits_y; a.bits_nblk[0] = (Cout + 31) / 32;
    return pnnp_igemm_h2s_launch(a, C1, as_stream(stream));
}

// small grids: K cut into `ksplit` slices (pnnp_h2_splitk), raw partial sums into ws [ksplit][B][H][W][Cout], then one reduce kernel (bias, activation,
// amax, sign bits); ksplit <= 1: the ordinary launch
int pnnp_conv3x3_h2_fwd_splitk_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2, int C2, const unsigned* amax_x2,
                                   const void* w_h2, const unsigned* amax_w, const float* bias, float* y, unsigned* amax_y, unsigned* bits_y,
                                   int B, int H, int W, int Cout, int act, int ksplit, float* ws, int64_t ws_floats, void* stream) {
    if (ksplit <= 1) return pnnp_conv3x3_h2_fwd_f32(x1, C1, amax_x1, x2, C2, amax_x2, w_h2, amax_w, bias, nullptr, y, amax_y, bits_y, B, H, W, Cout, act, stream);
    if (!x1 || !w_h2 || !y || !ws || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1) || (Cout & 31)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    if (ws_floats < (int64_t)ksplit * B * H * W * Cout) return PNNP_E_WORKSPACE;
    if ((int64_t)ksplit * B >= (1 << 20)) return PNNP_E_UNSUPPORTED;
    H2Args a{};
    fwd_args(a.g, x1, C1, x2, C2, w_h2, nullptr, nullptr, ws, B, H, W, Cout, 0);
    a.amax_in[0] = amax_x1; a.amax_in[1] = x2 ? amax_x2 : nullptr; a.amax_w = amax_w;
    a.ksplit = ksplit;
    const int rc = pnnp_igemm_h2s_launch(a, C1, as_stream(stream));
    if (rc != PNNP_OK) return rc;
    return pnnp_h2_splitk_reduce_launch(ws, ksplit, bias, y, bits_y, amax_y, B, H, W, Cout, act, as_stream(stream));
}

// conv3x3 + activation + the network's 1x1 head (archs/Unet.py:93-94: conv9_2, lrelu, conv10_1) in one kernel; y null: the 32-channel map is not stored
int pnnp_conv3x3_h2_fwd_head_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2, int C2, const unsigned* amax_x2,
                                 const void* w_h2, const unsigned* amax_w, const float* bias, float* y, unsigned* amax_y, unsigned* bits_y,
                                 const float* head_w, const float* head_b, const float* head_res, float* head_out,
                                 int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_h2 || !head_w || !head_out || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (Cout != 32) return PNNP_E_UNSUPPORTED;
    if (B == 0) return PNNP_OK;
    H2Args a{};
    fwd_args(a.g, x1, C1, x2, C2, w_h2, bias, nullptr, y, B, H, W, Cout, act);
    a.amax_in[0] = amax_x1; a.amax_in[1] = x2 ? amax_x2 : nullptr; a.amax_w = amax_w; a.amax_out[0] = y ? amax_y : nullptr;
    a.bits_out = y ? bits_y : nullptr; a.bits_nblk[0] = 1;
    a.head_w = head_w; a.head_b = head_b; a.head_res = head_res; a.head_out = head_out;
    return pnnp_igemm_h2s_launch(a, C1, as_stream(stream));
}

int pnnp_conv3x3_h2_fwd_pool_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2, int C2, const unsigned* amax_x2,
                                 const void* w_h2, const unsigned* amax_w, const float* bias, float* y, float* pooled, unsigned char* codes,
                                 unsigned* amax_y, unsigned* bits_y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_h2 || !y || !pooled || !codes || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1) || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args a{};
    fwd_args(a.g, x1, C1, x2, C2, w_h2, bias, nullptr, y, B, H, W, Cout, act);
    a.g.pool_dst = pooled; a.g.pool_codes = codes; a.g.pool_cs = Cout;
    a.amax_in[0] = amax_x1; a.amax_in[1] = x2 ? amax_x2 : nullptr; a.amax_w = amax_w; a.amax_out[0] = amax_y;      // (the pooled map is a subset of y)
    a.bits_out = bits_y; a.bits_nblk[0] = (Cout + 31) / 32;
    return pnnp_igemm_h2s_launch(a, C1, as_stream(stream));
}

// mask1 / mask2: the float32 activation, or bits1 / bits2: its sign bits from the forward kernel (then the float32 pointer is not read)
int pnnp_conv3x3_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                 float* dx1, int C1, const float* mask1, const unsigned* bits1, int mode1, int accum1, unsigned* amax_dx1,
                                 float* dx2, int C2, const float* mask2, const unsigned* bits2, int mode2, int accum2, unsigned* amax_dx2,
                                 int B, int H, int W, void* stream) {
    if (!g || !w_h2_dgrad || !dx1 || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args a{};
    bwd_args(a.g, g, Cout, w_h2_dgrad, dx1, C1, mask1, mode1, accum1, dx2, C2, mask2, mode2, accum2, B, H, W);
    if (bits1) a.g.mask_mode[0] = mode1;
    if (bits2 && dx2) a.g.mask_mode[1] = mode2;
    a.amax_in[0] = amax_g; a.amax_w = amax_w; a.amax_out[0] = amax_dx1; a.amax_out[1] = dx2 ? amax_dx2 : nullptr;
    a.bits_in[0] = bits1; a.bits_in[1] = dx2 ? bits2 : nullptr; a.bits_nblk[0] = (C1 + 31) / 32; a.bits_nblk[1] = (C2 + 31) / 32;
    return pnnp_igemm_h2s_launch(a, Cout, as_stream(stream));
}

int pnnp_conv3x3_h2_bwd_data_res_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                     float* dx, int C1, const float* addsrc, const float* mask, int mode, unsigned* amax_dx,
                                     int B, int H, int W, void* stream) {
    if (!g || !w_h2_dgrad || !dx || !addsrc || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args a{};
    bwd_res_args(a.g, g, Cout, w_h2_dgrad, dx, C1, addsrc, mask, mode, B, H, W);
    a.amax_in[0] = amax_g; a.amax_w = amax_w; a.amax_out[0] = amax_dx;
    return pnnp_igemm_h2s_launch(a, Cout, as_stream(stream));
}

// ---------------------------------------------------------------- pointwise layers on the bf16 matrix cores (csrc/gemm_x3.hip)
// Same contracts as their fp32-MFMA twins below / above; weights are the x3 packs of pnnp_pack_jobs_add_x3_convt / _1x1 / _s2;
// channel counts in multiples of 32.
int pnnp_gemm_x3_supported(int K, int N) { return (K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0) ? 1 : 0; }

int pnnp_convt2x2_x3_fwd_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, int B, int H, int W, int Cout, void* stream) {
    return pnnp_convt2x2_x3_fwd_amax_f32(x, Cin, w_x3, bias, y, nullptr, B, H, W, Cout, stream);
}
// ... and max |y| into an amax slot of the fp16x2 family (csrc/h2.h: y is what the decoder's first 3x3 layer splits)
int pnnp_convt2x2_x3_fwd_amax_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, unsigned* amax_y,
                                  int B, int H, int W, int Cout, void* stream) {
    if (!x || !w_x3 || !y || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    convt_fwd_args(a, x, Cin, w_x3, bias, y, B, H, W, Cout);
    a.amax_out[0] = amax_y;
    return pnnp_gemm_x3_launch(a, Cin, as_stream(stream));
}

int pnnp_convt2x2_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int Cin, const float* mask, int mode,
                                  int B, int H, int W, void* stream) {
    return pnnp_convt2x2_x3_bwd_data_amax_f32(g, Cout, w_x3_dgrad, dx, Cin, mask, mode, nullptr, B, H, W, stream);
}
int pnnp_convt2x2_x3_bwd_data_amax_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int Cin, const float* mask, int mode,
                                       unsigned* amax_dx, int B, int H, int W, void* stream) {
    if (!g || !w_x3_dgrad || !dx || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    convt_bwd_args(a, g, Cout, w_x3_dgrad, dx, Cin, mask, mode, B, H, W);
    a.amax_out[0] = amax_dx;
    return pnnp_gemm_x3_launch(a, Cout, as_stream(stream));
}

int pnnp_conv1x1_x3_fwd_f32(const float* x1, int C1, const float* x2, int C2, const void* w_x3, const float* bias, const float* residual,
                            float* y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_x3 || !y || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    fwd_args(a, x1, C1, x2, C2, w_x3, bias, residual, y, B, H, W, Cout, act);
    return pnnp_gemm_x3_launch(a, C1, as_stream(stream));
}

int pnnp_conv1x1_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                 float* dx2, int C2, const float* mask2, int mode2, int accum2, int B, int H, int W, void* stream) {
    if (!g || !w_x3_dgrad || !dx1 || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    bwd_args(a, g, Cout, w_x3_dgrad, dx1, C1, mask1, mode1, accum1, dx2, C2, mask2, mode2, accum2, B, H, W);
    return pnnp_gemm_x3_launch(a, Cout, as_stream(stream));
}

int pnnp_conv3x3s2_x3_fwd_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, int B, int H, int W, int Cout, int act, void* stream) {
    return pnnp_conv3x3s2_x3_fwd_amax_f32(x, Cin, w_x3, bias, y, nullptr, B, H, W, Cout, act, stream);
}
// ... with max |y| raised into an amax slot of the fp16x2 family (csrc/h2.h)
int pnnp_conv3x3s2_x3_fwd_amax_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, unsigned* amax_y,
                                   int B, int H, int W, int Cout, int act, void* stream) {
    if (!x || !w_x3 || !y || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    s2_fwd_args(a, x, Cin, w_x3, bias, y, B, H, W, Cout, act);
    a.amax_out[0] = amax_y;
    return pnnp_gemm_x3_launch(a, Cin, as_stream(stream));
}

int pnnp_conv3x3s2_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_s2dgrad, float* dx, int Cin, const float* mask, int mode, int accum,
                                   int B, int H, int W, void* stream) {
    return pnnp_conv3x3s2_x3_bwd_data_amax_f32(g, Cout, w_x3_s2dgrad, dx, Cin, mask, mode, accum, nullptr, B, H, W, stream);
}
// ... with max |dx| (of the sums stored, when accumulating) raised into an amax slot: the four parity-class launches share it
int pnnp_conv3x3s2_x3_bwd_data_amax_f32(const float* g, int Cout, const void* w_x3_s2dgrad, float* dx, int Cin, const float* mask, int mode, int accum,
                                        unsigned* amax_dx, int B, int H, int W, void* stream) {
    if (!g || !w_x3_s2dgrad || !dx || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    int slice = 0;
    for (int cls = 0; cls < 4; ++cls) {
        IgemmArgs a;
        const int ns = s2_bwd_args(a, cls, g, Cout, dx, Cin, mask, mode, accum, B, H, W);
        a.amax_out[0] = amax_dx;
        a.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w_x3_s2dgrad) + (int64_t)slice * Cout * Cin * 6);
        const int rc = pnnp_gemm_x3_launch(a, Cout, as_stream(stream));
        if (rc != PNNP_OK) return rc;
        slice += ns;
    }
    return PNNP_OK;
}

// ---------------------------------------------------------------- pointwise layers on the fp16 matrix cores (csrc/gemm_h2s.hip, csrc/h2.h)
// Contracts of the _x3_ entries above + the amax slots of the fp16x2 family; weights: the kind-6 packs of pnnp_pack_jobs_add_h2_convt / _1x1.
// K (channels of a segment) in multiples of 32, N (GEMM columns: 4 Cout for ConvTranspose2d forward) in multiples of 64.
int pnnp_gemm_h2_supported(int K, int N) { return (K > 0 && N > 0 && K % 32 == 0 && N % 32 == 0) ? 1 : 0; }      // (round 6: 32-column tiles for N = 32 mod 64)
namespace {
int gemm_h2_go(H2Args& h, int chan_per_seg, hipStream_t st) {
    const int rc = pnnp_gemm_x3_check(h.g, chan_per_seg);
    if (rc != PNNP_OK) return rc;
    if (!pnnp_gemm_h2_supported(chan_per_seg, h.g.Ntot)) return PNNP_E_UNSUPPORTED;
    h.g.seg_channels = chan_per_seg;
    h.g.chunks_per_seg = chan_per_seg / 32;                         // csrc/gemm_h2s.hip walks K in 32-channel items
    if ((int64_t)(h.g.Ntot / 32) * h.g.nseg * h.g.chunks_per_seg * 4096 >= (1ll << 31)) return PNNP_E_UNSUPPORTED;
    return pnnp_gemm_h2s_launch(h, st);
}
}  // namespace
int pnnp_convt2x2_h2_fwd_f32(const float* x, int Cin, const unsigned* amax_x, const void* w_h2, const unsigned* amax_w, const float* bias, float* y,
                             unsigned* amax_y, int B, int H, int W, int Cout, void* stream) {
    if (!x || !w_h2 || !y || !amax_x || !amax_w || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    convt_fwd_args(h.g, x, Cin, w_h2, bias, y, B, H, W, Cout);
    h.g.amax_out[0] = amax_y; h.amax_in[0] = amax_x; h.amax_w = amax_w;
    return gemm_h2_go(h, Cin, as_stream(stream));
}
int pnnp_convt2x2_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w, float* dx, int Cin,
                                  const float* mask, int mode, unsigned* amax_dx, int B, int H, int W, void* stream) {
    if (!g || !w_h2_dgrad || !dx || !amax_g || !amax_w || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    convt_bwd_args(h.g, g, Cout, w_h2_dgrad, dx, Cin, mask, mode, B, H, W);
    h.g.amax_out[0] = amax_dx; h.amax_in[0] = amax_g; h.amax_w = amax_w;
    return gemm_h2_go(h, Cout, as_stream(stream));
}
// ... with the LeakyReLU' / ReLU' mask as the sign bits the 3x3 forward kernel stored for the layer's INPUT map (bits: pnnp_h2_bits_words(B, H, W, Cin) words)
int pnnp_convt2x2_h2_bwd_data_bits_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w, float* dx, int Cin,
                                       const unsigned* bits, int mode, unsigned* amax_dx, int B, int H, int W, void* stream) {
    if (!g || !w_h2_dgrad || !dx || !amax_g || !amax_w || !bits || !mode || B < 0 || H <= 0 || W <= 0 || (Cin & 31)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    convt_bwd_args(h.g, g, Cout, w_h2_dgrad, dx, Cin, nullptr, 0, B, H, W);
    h.g.mask_mode[0] = mode;
    h.g.amax_out[0] = amax_dx; h.amax_in[0] = amax_g; h.amax_w = amax_w;
    h.bits_in[0] = bits; h.bits_nblk[0] = Cin / 32;
    return gemm_h2_go(h, Cout, as_stream(stream));
}
int pnnp_conv1x1_h2_fwd_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2, int C2, const unsigned* amax_x2, const void* w_h2,
                            const unsigned* amax_w, const float* bias, const float* residual, float* y, unsigned* amax_y,
                            int B, int H, int W, int Cout, int act, void* stream) {
    if (!x1 || !w_h2 || !y || !amax_x1 || !amax_w || (x2 && !amax_x2) || B < 0 || H <= 0 || W <= 0 || C1 <= 0 || (x2 && C2 != C1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    fwd_args(h.g, x1, C1, x2, C2, w_h2, bias, residual, y, B, H, W, Cout, act);
    h.g.amax_out[0] = amax_y; h.amax_in[0] = amax_x1; h.amax_in[1] = x2 ? amax_x2 : nullptr; h.amax_w = amax_w;
    return gemm_h2_go(h, C1, as_stream(stream));
}
int pnnp_conv1x1_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                 float* dx1, int C1, const float* mask1, int mode1, int accum1, unsigned* amax_dx1,
                                 float* dx2, int C2, const float* mask2, int mode2, int accum2, int B, int H, int W, void* stream) {
    if (!g || !w_h2_dgrad || !dx1 || !amax_g || !amax_w || B < 0 || H <= 0 || W <= 0 || Cout <= 0 || C1 <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    bwd_args(h.g, g, Cout, w_h2_dgrad, dx1, C1, mask1, mode1, accum1, dx2, C2, mask2, mode2, accum2, B, H, W);
    h.g.amax_out[0] = amax_dx1; h.amax_in[0] = amax_g; h.amax_w = amax_w;
    return gemm_h2_go(h, Cout, as_stream(stream));
}

int pnnp_conv3x3s2_h2_fwd_f32(const float* x, int Cin, const unsigned* amax_x, const void* w_h2, const unsigned* amax_w, const float* bias, float* y,
                              unsigned* amax_y, int B, int H, int W, int Cout, int act, void* stream) {
    if (!x || !w_h2 || !y || !amax_x || !amax_w || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    H2Args h{};
    s2_fwd_args(h.g, x, Cin, w_h2, bias, y, B, H, W, Cout, act);
    h.g.amax_out[0] = amax_y; h.amax_in[0] = amax_x; h.amax_w = amax_w;
    return gemm_h2_go(h, Cin, as_stream(stream));
}
int pnnp_conv3x3s2_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_s2dgrad, const unsigned* amax_w, float* dx, int Cin,
                                   const float* mask, int mode, int accum, unsigned* amax_dx, int B, int H, int W, void* stream) {
    if (!g || !w_h2_s2dgrad || !dx || !amax_g || !amax_w || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    int slice = 0;
    for (int cls = 0; cls < 4; ++cls) {                              // one launch per input-pixel parity class (pnnp_conv3x3s2_x3_bwd_data_amax_f32)
        H2Args h{};
        const int ns = s2_bwd_args(h.g, cls, g, Cout, dx, Cin, mask, mode, accum, B, H, W);
        h.g.amax_out[0] = amax_dx; h.amax_in[0] = amax_g; h.amax_w = amax_w;
        h.g.w = reinterpret_cast<const float*>(reinterpret_cast<const char*>(w_h2_s2dgrad) + (int64_t)slice * Cout * Cin * 4);
        const int rc = gemm_h2_go(h, Cout, as_stream(stream));
        if (rc != PNNP_OK) return rc;
        slice += ns;
    }
    return PNNP_OK;
}

// ConvTranspose2d(Cin, Cout, 2, stride=2) forward   archs/Unet.py:35-47
//   x [B][H][W][Cin] -> y [B][2H][2W][Cout];  one GEMM with N = 4*Cout (the 4 output sub-pixels).
int pnnp_convt2x2_fwd_f32(const float* x, int Cin, const float* w_packed, const float* bias, float* y,
                          int B, int H, int W, int Cout, void* stream) {
    if (!x || !w_packed || !y || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    convt_fwd_args(a, x, Cin, w_packed, bias, y, B, H, W, Cout);
    return pnnp_igemm_launch(a, 1, Cin, as_stream(stream));
}

// backward-data of ConvTranspose2d: g [B][2H][2W][Cout] -> dx [B][H][W][Cin] (x act'(mask))
int pnnp_convt2x2_bwd_data_f32(const float* g, int Cout, const float* w_dgrad, float* dx, int Cin,
                               const float* mask, int mode, int B, int H, int W, void* stream) {
    if (!g || !w_dgrad || !dx || B < 0 || H <= 0 || W <= 0) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    convt_bwd_args(a, g, Cout, w_dgrad, dx, Cin, mask, mode, B, H, W);
    return pnnp_igemm_launch(a, 1, Cout, as_stream(stream));
}

// ---------------------------------------------------------------- Conv2d 3x3, stride 2, pad 1
// (ResUnet down-sampling: archs/modules.py:130-138 `conv3x3`, archs/ResUnet.py:18-27)
// forward: 9 strided 1x1 taps as 9 K segments of one GEMM; w_packed is the ordinary forward pack
// [9][Cin/4][Cout][4] (tap-major == segment-major).  x [B][H][W][Cin] -> y [B][H/2][W/2][Cout].
int pnnp_conv3x3s2_fwd_f32(const float* x, int Cin, const float* w_packed, const float* bias, float* y,
                           int B, int H, int W, int Cout, int act, void* stream) {
    if (!x || !w_packed || !y || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    IgemmArgs a;
    s2_fwd_args(a, x, Cin, w_packed, bias, y, B, H, W, Cout, act);
    return pnnp_igemm_launch(a, 1, Cin, as_stream(stream));
}

// backward-data weights of the stride-2 conv: 9 slices [Cout/4][Cin][4] = W[co][ci][t] ordered by the
// parity class of the input pixel they reach: (even,even) t4 | (even,odd) t3,t5 | (odd,even) t1,t7 |
// (odd,odd) t0,t2,t6,t8.   dst: 9*Cout*Cin floats.
static const int S2_TAP_ORDER[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
int pnnp_pack_conv3x3s2_dgrad_f32(const float* w, float* dst, int Cout, int Cin, void* stream) {
    int rc = PNNP_OK;
    for (int i = 0; i < 9 && rc == PNNP_OK; ++i)
        rc = pack_launch(w, dst + (int64_t)i * Cout * Cin, 1, Cout, Cin, (int64_t)Cin * 9, 9, 0, S2_TAP_ORDER[i], 0, stream);
    return rc;
}

// backward-data: g [B][H/2][W/2][Cout] -> dx [B][H][W][Cin].  Every input pixel parity class
// (Y&1, X&1) is its own GEMM over the 1 / 2 / 2 / 4 taps that reach it, so each dx element is written
// exactly once (no read-modify-write over the 9 taps).
int pnnp_conv3x3s2_bwd_data_f32(const float* g, int Cout, const float* w_s2dgrad, float* dx, int Cin,
                                const float* mask, int mode, int accum, int B, int H, int W, void* stream) {
    if (!g || !w_s2dgrad || !dx || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    int slice = 0;
    for (int cls = 0; cls < 4; ++cls) {
        IgemmArgs a;
        const int ns = s2_bwd_args(a, cls, g, Cout, dx, Cin, mask, mode, accum, B, H, W);
        a.w = w_s2dgrad + (int64_t)slice * Cout * Cin;
        const int rc = pnnp_igemm_launch(a, 1, Cout, as_stream(stream));
        if (rc != PNNP_OK) return rc;
        slice += ns;
    }
    return PNNP_OK;
}

}  // extern "C"
