// The fp16x2 ("h2") family: float32 operands on the fp16 matrix cores (round 5).
//
//   a float32 value a, multiplied by a power of two s that brings its TENSOR's largest magnitude into [2^14, 2^15), splits into
//       hi = f16(s a),   lo = f16(s a - hi)            (round to nearest even; v_fma_mixlo/hi_f16: one instruction each)
//   with |s a - hi - lo| <= 2^-22 |s a| for |a| >= 2^-18 max|a| and an ABSOLUTE error <= 2^-40 max|a| below that (lo becomes an fp16
//   subnormal, which the gfx950 matrix core keeps: tools/ubench/h2_probe.hip).  A product a b is hi hi' + hi lo' + lo hi' -- THREE fp16 x fp16
//   products (exact in the fp32 accumulator) where the bf16x3 family needs six; the dropped lo lo' is below 2^-22 |a b|.  Against float64
//   the scheme is at the fp32-MFMA kernels' error or below for every reduction length the networks have (profiles/r5/h2_probe.txt).
//
// Both pieces are FLOATING-point numbers -- hi the element's top 11 significand bits, lo its next 11 -- so the scale only matters where lo leaves fp16's
// normal range: for every element within 2^-18 of the maximum the split, the products and the result are the same numbers times a power of two
// WHATEVER scale was used (a crop beside batch-mates x 100 brighter: its few pixels below 2^-18 of THEIR maximum move its output by 1e-10 relative, the
// bright crops' outputs are bit-identical: tests/test_gpu_dp2.py).
//
// The scale needs max|a| of the tensor BEFORE the kernel that splits it starts: every kernel that writes a tensor the h2 kernels read also
// writes max|.| of what it stored into a 4-byte slot (atomicMax on the float's bit pattern: non-negative floats order like unsigned
// integers).  Slots are zeroed by the caller once per pass.  A slot that is too LARGE only costs range at the small end; one that is
// too small would saturate -- so every producer's bound is a true upper bound of what it wrote (accumulating producers report the sum).
#pragma once
#include "igemm.h"

// scale exponent se of a tensor from its amax slot: s = 2^se brings amax into [2^14, 2^15).  amax = 0 / subnormal (an all-zero tensor) and
// inf / NaN (a diverged one: the fp16 conversion then carries inf / NaN into the products) take s = 1.
__host__ __device__ inline int pnnp_h2_scale_exp(unsigned amax_bits) {
    const int E = (int)((amax_bits >> 23) & 0xffu);
    if (E == 0 || E == 255) return 0;
    const int se = 141 - E;                  // 14 - (E - 127)
    return se > 127 ? 127 : se;              // (amax < 2^-113: the scale stays a normal float32; the tensor just sits lower in fp16's range)
}

struct H2Args {
    IgemmArgs g;                             // geometry, sources, destinations, fp32 masks: as for the bf16x3 kernels
    const unsigned* amax_in[2];              // amax slot of K segment 0 / 1 (the tensors that are split on the fly); [1] null without a second segment
    const unsigned* amax_w;                  // amax slot of the weight tensor (the pack was scaled with it: csrc/pack_jobs.hip kind 4)
    unsigned* amax_out[2];                   // max|stored value| per destination (or null)
    unsigned* bits_out;                      // forward: sign bits of the stored (activated) output, tile-private layout (or null)
    const unsigned* bits_in[2];              // backward-data: act' mask of destination 0 / 1 as bits written by the forward kernel (or null)
    int bits_nblk[2];                        // 32-channel blocks of the tensor behind bits_out ([0]) / bits_in[du]
    // forward layer + the network's 1x1 head in ONE kernel (csrc/conv_h2s.hip EK_HEAD; Ntot == 32): head_out NCHW [B][4][OH][OW] = head_w [4][32] . act(y) + head_b
    // (+ head_res, NCHW like head_out); g.dst[0] may then be null (the 32-channel map is not stored).  head_out null: an ordinary launch.
    const float* head_w; const float* head_b; const float* head_res; float* head_out;
    // split-K for small grids (csrc/conv_h2s.hip): ksplit > 1 workgroups share an output tile, each walking 1 / ksplit of the K chunks and writing raw partial
    // sums to image ks B + b of g.dst[0] = a [ksplit][B][OH][OW][cs] slab tensor (general epilogue, no bias / activation / mask / bits); 0 or 1: off
    int ksplit;
};
// Tile-private bit layout: the 16-row x 32-px x 32-channel block (image b, tile row ty, tile column tx, channel block cb) of a tensor with
// nblk 32-channel blocks is 512 words; word 64 w + l belongs to lane l of consumer wave w, bit 31 - (((i 2 + h) 2 + jj) 4 + c) = element
// (row 2 w + i, pixel 16 h + (l & 15), channel 16 jj + 4 (l >> 4) + c) > 0.  Forward and backward-data tiles of the same tensor coincide,
// so a lane reads back exactly the word the same lane position wrote: 4 bytes per lane instead of 8 x 16.
int pnnp_igemm_h2s_launch(const H2Args& a, int chan_per_seg, hipStream_t s);
int pnnp_h2_splitk_reduce_launch(const float* slab, int S, const float* bias, float* y, unsigned* bits, unsigned* amax, int B, int H, int W, int N, int act, hipStream_t st);
