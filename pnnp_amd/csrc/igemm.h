// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
//   M = output pixels (tile of TH rows x 32 px), N = output channels, K = taps x channels.
//   A (im2col) is never materialised: a halo tile of the input lives in LDS and each MFMA
//   A-operand is read from it at the tap's offset.  B is the layer's weights, pre-packed to
//   [tap][K/4][N][4] so a chunk is a few contiguous 16-byte runs.
//
//   LDS images (both read with conflict-free ds_read_b128):
//     xs[KC/4][halo pixel][4 ch]   lane l reads pixel (l&31), channel quad 2j+(l>>5)
//     ws[tap][KC/4][BN][4 ch]      lane l reads column (l&31), channel quad 2j+(l>>5)
//   One b128 read feeds 4 MFMAs: step s multiplies channel 8j+s (lanes 0-31) and 8j+4+s
//   (lanes 32-63) -- the two k-slices of the 32x32x2 instruction.
//
// The kernel is generic over "segments": K is the concatenation of up to 9 sources, each a
// tensor + channel range + pixel offset.  That expresses
//   conv3x3 on torch.cat([up, skip], 1) without materialising the cat   (2 segments)
//   ConvTranspose2d(k2,s2) backward-data as a 1x1 over the 4 sub-pixel sources (4 segments)
// and the output side can scatter with a stride/offset (ConvTranspose2d forward) and split N
// across two destinations (backward-data of a concat layer).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct IgemmSeg {
    const float* ptr;   // NHWC tensor [B][IH][IW][cstride]
    int cstride;        // channels per pixel of that tensor
    int coff;           // first channel used
    int yoff, xoff;     // input pixel = tile pixel * in_mul + (yoff, xoff)
};

struct IgemmArgs {
    IgemmSeg seg[9];               // up to 9 K segments (a stride-2 3x3 conv is 9 strided 1x1 taps)
    int nseg, chunks_per_seg;      // K = nseg * chunks_per_seg * KC channels (x TAPS)
    int seg_channels;              // (csrc/conv_x3.hip) channels a segment contributes; its last 16-channel chunk may be half empty
    int in_mul, IH, IW;
    int B, DH, DW;                 // tile domain (what M iterates over)
    const float* w;                // packed [TAPS][Ktot/4][Ntot][4]
    int Ntot;
    float* dst[2];                 // n < n_split -> dst[0], else dst[1]
    int dst_cs[2];
    int n_split;
    int n_sub;                     // > 0: column n = sub*n_sub + channel, sub-pixel (sub>>1, sub&1) added to the out offset
    int out_mul, out_yoff, out_xoff, OH, OW;   // out pixel = tile pixel * out_mul + off
    const float* bias;             // [Ntot] or null
    int act;                       // 0 none, 1 LeakyReLU(0.2), 2 ReLU
    const float* mask[2];          // saved activation at the dst position (or null)
    int mask_mode[2];              // 0 none, 1 x lrelu'(mask), 2 x relu'(mask)
    int accum[2];                  // dst += result
    const float* addsrc;           // optional residual tensor with dst[0] geometry, added before act (or null)
    // (csrc/conv_x3.hip, forward only) MaxPool2d(2) of the activated output fused into the epilogue: the pooled map
    // [B][OH/2][OW/2][pool_cs] and the one-byte argmax + sign codes of pnnp_maxpool2_fwd_codes_f32 (or both null)
    float* pool_dst;
    unsigned char* pool_codes;
    int pool_cs;
    // (csrc/gemm_x3s.hip) max |stored value| per destination into an amax slot of the fp16x2 family (csrc/h2.h), or null
    unsigned* amax_out[2];
};
