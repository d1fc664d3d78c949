/*
 * pnnp_hip.h -- C ABI of libpnnp_hip.so, the MI355X (gfx950) implementation of the
 * PNNP data-parallel hot path.
 *
 * The reference (fenghansen/PNNP) is pure Python: it has no FFI of its own.  Its
 * "operator interface" for this path is a set of Python callables; each entry point
 * below is what a ctypes binding of that callable binds to (INTEGRATION.md shows the
 * reference-side stub).  Citations are file:line under the reference tree.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.
 *   - every pointer is a DEVICE pointer owned by the caller unless marked [host].
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it; the
 *     library never synchronises, allocates or frees device memory.
 *   - return value: 0 = ok, negative = error (pnnp_error_string()).
 *   - activations between layers are NHWC fp32 ("pixel-major": [B][H][W][C]);
 *     network input/output at the boundary are NCHW fp32 like the reference's tensors.
 */
#ifndef PNNP_HIP_H
#define PNNP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PNNP_OK 0
#define PNNP_E_INVALID (-1)     /* bad argument (shape, alignment, null)              */
#define PNNP_E_UNSUPPORTED (-2) /* valid request this build has no kernel for          */
#define PNNP_E_LAUNCH (-3)      /* hipLaunchKernel reported an error                   */
#define PNNP_E_WORKSPACE (-4)   /* caller-provided workspace too small                 */

int pnnp_version(void);
const char* pnnp_error_string(int code);
/* ABI version: bumped whenever a struct of this header changes its layout or an entry point its meaning (history: INTEGRATION.md, "Upgrade
 * notes").  A binding written against this header compares pnnp_abi_version() with the PNNP_ABI_VERSION it was compiled / written with and
 * pnnp_pack_job_bytes() with its own sizeof(PnnpPackJob) before it passes a job table (pnnp_amd/ops.py does, on first use).
 *   6 (round 6): PnnpPackJob carries a trailing `amax` pointer since round 5 (an older caller's job ARRAY would be read with the wrong stride);
 *                pnnp_x3_supported refuses more than 1024 output channels (PNNP_E_UNSUPPORTED from the pnnp_conv3x3_x3_* entries beyond it). */
#define PNNP_ABI_VERSION 6
int pnnp_abi_version(void);
int pnnp_pack_job_bytes(void);
/* Number of compute units etc. of the current device (0 on failure). */
int pnnp_device_cus(void);
/* Workgroups per CU of the persistent forward / backward-data convolution kernels (default 1: one per CU with an equal static share of
 * the tiles).  n > 1 launches n per CU with 1/n share each; the surplus waits in the hardware dispatcher and goes to whichever CU frees up
 * first, so a kernel resident on some CUs beside the convolution (an RCCL collective overlapping the backward pass: replaces what
 * nn.DataParallel's reduce did behind autograd, base_trainer.py:115-118) stretches a layer by ~CUs / (CUs - k) instead of doubling it.
 * Process-wide setting, 1 <= n <= 16: the ONE deliberate exception to "no global mutable state"  -- it changes grid sizes, never a
 * result bit (tests/test_gpu_overlap.py), and a per-launch argument would have to thread through every convolution entry point.  Callers
 * scope it: HipTrainStep sets it around the backward pass of a step whose all-reduce overlaps and restores the previous value. */
void pnnp_set_persistent_split(int n);
int pnnp_get_persistent_split(void);

/* ---------------------------------------------------------------- Bayer pack / unpack
 * raw2bayer  utils/isp_ops.py:84-96     u16|f32 [B][H][W] -> f32 [B][4][H/2][W/2]
 *   plane order R,G1,B,G2 = Bayer offsets (0,0),(0,1),(1,1),(1,0).
 *   norm: (x - black[c]) / (wp - black[c]) evaluated in float64 and rounded to
 *   float32 once (numpy promotion of the reference); clip: clamp to [0,1].
 *   black [host] = bias[c] + bl.  Bit-exact with the reference.
 */
int pnnp_pack_bayer_u16(const uint16_t* src, int B, int H, int W, int64_t src_row_stride,
                        int64_t src_batch_stride, float* dst, const double* black4 /*[host]*/,
                        double wp, int norm, int clip, void* stream);
int pnnp_pack_bayer_f32(const float* src, int B, int H, int W, int64_t src_row_stride,
                        int64_t src_batch_stride, float* dst, const double* black4 /*[host]*/,
                        double wp, int norm, int clip, void* stream);
/* pack_raw_bayer  data_process/process.py:40-64: CFA-pattern-aware pack.  pos4 [host] = Bayer offset
 *   (dy<<1|dx) of the R, G1, B, G2 sites; black4 [host] per-channel black level; float32 arithmetic. */
int pnnp_pack_bayer_pattern(const void* src, int is_f32, int B, int H, int W, int64_t src_row_stride,
                            int64_t src_batch_stride, float* dst, const double* black4 /*[host]*/,
                            double wp, int clip, const int* pos4 /*[host]*/, void* stream);
/* bayer2raw  utils/isp_ops.py:98-112    f32 [B][4][h][w] -> u16 [B][2h][2w]
 *   clamp(x,0,1) * (wp-bl) + bl in float32 (two roundings), C-cast truncation.        */
int pnnp_unpack_bayer_u16(const float* src, int B, int h, int w, uint16_t* dst,
                          int wp, int bl, void* stream);
/* bayer2rggb / rggb2bayer  utils/isp_ops.py:57-63 ; bayer2rows / rows2bayer :65-81
 *   pure index moves on elements of `elem_bytes` (2, 4 or 8).                         */
int pnnp_bayer_to_rggb(const void* src, void* dst, int H, int W, int elem_bytes, void* stream);
int pnnp_rggb_to_bayer(const void* src, void* dst, int h, int w, int elem_bytes, void* stream);
int pnnp_bayer_to_rows(const void* src, void* dst, int H, int W, int elem_bytes, void* stream);
int pnnp_rows_to_bayer(const void* src, void* dst, int h, int W, int elem_bytes, void* stream);

/* ---------------------------------------------------------------- noise sampler
 * generate_noisy_obs   data_process/process.py:591-631  (PNNP_NOISE_MODE_OBS)
 * generate_noisy_torch data_process/process.py:634-673  (PNNP_NOISE_MODE_TORCH)
 *   y, out: f32 [B][C][H][W];  params: f32 [B][PNNP_NPARAM] (device), one row per crop.
 *   Counter-based RNG: Philox4x32-10, key = seed, counter = (element, crop_base+b,
 *   draw slot, offset) -- results do not depend on B, grid or GPU count.
 *   Specification of the sampler = oracle/pnnp_oracle.c (pnnp_oracle_noise_sample).
 */
enum {
    PNNP_P_K = 0, PNNP_P_SIGGS, PNNP_P_SIGTL, PNNP_P_LAM, PNNP_P_SIGR, PNNP_P_Q,
    PNNP_P_RATIO, PNNP_P_WP, PNNP_P_BL, PNNP_P_BIAS0, PNNP_P_BIAS1, PNNP_P_BIAS2,
    PNNP_P_BIAS3, PNNP_NPARAM = 16
};
#define PNNP_NOISE_P 0x01u      /* 'p' Poisson shot noise                             */
#define PNNP_NOISE_G 0x02u      /* 'g' Tukey-lambda read noise (else Gaussian)        */
#define PNNP_NOISE_R 0x04u      /* 'r' row noise, one draw per (channel,row)          */
#define PNNP_NOISE_Q 0x08u      /* 'q' quantisation noise                             */
#define PNNP_NOISE_D 0x10u      /* 'd' dark bias per channel                          */
#define PNNP_NOISE_B 0x20u      /* 'b' black frame: no read noise                     */
#define PNNP_NOISE_ORI 0x100u   /* ori=True: do not multiply by ratio                 */
#define PNNP_NOISE_CLIP 0x200u  /* clip=True: clamp to [0,1] instead of [-bl/wp,1]    */
#define PNNP_NOISE_MODE_TORCH 0x1000u /* quirks of generate_noisy_torch (else _obs)   */
#define PNNP_NOISE_POST_MAX1 0x2000u  /* then min(z,1): Trainer.preprocess clamp, trainer_SID.py:481-485 */
#define PNNP_NOISE_POST_MIN0 0x4000u  /* then max(z,0) (clip other than HALF_CLIP)     */
#define PNNP_NOISE_TORCH_TUKEY 0x8000u /* extension: allow 'g' in TORCH mode with the Tukey-lambda
                                          read noise of generate_noisy_obs (process.py:611) instead of the
                                          NotImplementedError of process.py:654 */
int pnnp_noise_sample_f32(const float* y, float* out, int B, int C, int H, int W,
                          const float* params, unsigned flags, float mfm /* sqrt(MultiFrameMean) */,
                          uint64_t seed, uint64_t offset, uint32_t crop_base, void* stream);

/* ---------------------------------------------------------------- denoiser layers (NHWC fp32)
 * What the archs/ modules of the reference do through torch.nn (archs/Unet.py:16-99,
 * archs/ResUnet.py:15-88, archs/modules.py:130-197).  Arithmetic: fp32 operands and fp32
 * accumulation on the matrix cores (v_mfma_f32_32x32x2_f32 == an fmaf chain), so results
 * differ from the reference only by summation order.
 *
 * Weights are consumed in a packed, kernel-friendly order produced on the device from the
 * parameter tensors (which keep the reference's state_dict layout):
 *   Conv2d  [Cout][Cin][k][k] -> fwd   [taps][Cin/4][Cout][4]   (Cin*taps*Cout floats)
 *                             -> dgrad [taps][Cout/4][Cin][4]   (flipped taps)
 *   ConvTranspose2d [Cin][Cout][2][2] -> fwd 4 x [Cin/4][Cout][4], dgrad [(4*Cout)/4][Cin][4]
 * Channel counts must be multiples of 4 (weights, gradients) / 8 (activations read as K).
 */
int pnnp_pack_conv_weight_f32(const float* w, float* fwd /*or null*/, float* dgrad /*or null*/,
                              int Cout, int Cin, int taps, int Cin_pad, int Cout_pad, void* stream);
int pnnp_pack_convt_weight_f32(const float* w, float* fwd /*or null*/, float* dgrad /*or null*/,
                               int Cin, int Cout, void* stream);

/* y = act(conv(cat[x1,x2]) + bias (+ residual)); taps 9 (3x3, pad 1) or 1 (1x1); x2 null when
 * there is no concat (the cat of archs/Unet.py:75,80,85,90 is never materialised).
 * act: 0 none, 1 LeakyReLU(0.2), 2 ReLU. */
int pnnp_conv_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* w_packed,
                      const float* bias, const float* residual, float* y, int B, int H, int W,
                      int Cout, int taps, int act, void* stream);
/* backward-data: g = dL/d(pre-activation output) -> dx1 (and dx2 for a concat layer).
 * mask_i/mode_i: multiply by the activation derivative of the tensor that produced x_i
 * (mask = that saved activation; mode 1 LeakyReLU', 2 ReLU', 0/null none); accum_i: dx_i +=. */
int pnnp_conv_bwd_data_f32(const float* g, int Cout, const float* w_dgrad,
                           float* dx1, int C1, const float* mask1, int mode1, int accum1,
                           float* dx2, int C2, const float* mask2, int mode2, int accum2,
                           int B, int H, int W, int taps, void* stream);
/* The same two operators for 3x3 / stride 1 / pad 1 layers through Winograd F(2x2,3x3) (2.25x fewer MFMA passes;
 * results differ from the direct form by fp32 rounding of the transforms, ~1e-6 relative).  Weights are packed by
 * pnnp_pack_conv_weight_wino_f32 into U = G g G^T, 16*Cout*Cin floats each; supported when the channels read are a
 * multiple of 8 and the channels written a multiple of 64 (pnnp_wino_supported). */
int64_t pnnp_wino_weight_floats(int Cout, int Cin);
int pnnp_wino_supported(int K_read, int N_written);
int pnnp_pack_conv_weight_wino_f32(const float* w, float* fwd /*or null*/, float* dgrad /*or null*/, int Cout, int Cin, void* stream);
int pnnp_conv3x3_wino_fwd_f32(const float* x1, int C1, const float* x2, int C2, const float* u_fwd, const float* bias,
                              const float* residual /*or null*/, float* y, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3_wino_bwd_data_res_f32(const float* g, int Cout, const float* u_dgrad, float* dx, int C1,
                                       const float* addsrc, const float* mask, int mode, int B, int H, int W, void* stream);
int pnnp_conv3x3_wino_bwd_data_f32(const float* g, int Cout, const float* u_dgrad,
                                   float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                   float* dx2, int C2, const float* mask2, int mode2, int accum2,
                                   int B, int H, int W, void* stream);
/* Winograd backward-weight (dg = G^T [sum_tiles (A dY A^T) (.) (B^T d B)] G): same contract as
 * pnnp_conv_bwd_weight_f32 with taps = 9; needs H % 4 == 0, W % 8 == 0 and Cout, C1, C2 multiples of 64. */
/* ---------------------------------------------------------------- weight re-packing, batched (csrc/pack_jobs.hip)
 * The optimiser moves every weight every step (trainer_SID.py:101), so every layer's packed images are rebuilt every step.  A job
 * is one such rebuild; pnnp_pack_jobs_f32 runs a HOST array of jobs in ceil(n / 32) launches.  The builders append the jobs of one
 * layer (the same ones pnnp_pack_conv_weight_f32 / pnnp_pack_convt_weight_f32 / pnnp_pack_conv_weight_wino_f32 run): build the
 * table once, run it once per step.  kind 0 = strided gather, kind 1 = Winograd filter transform (K = Cout, N = Cin, T = dgrad). */
typedef struct PnnpPackJob {
    const float* src; float* dst;
    int kind, T, K, N;
    int64_t sk, sn, st, off;
    int flip, Kvalid, Ndst, n_off;
    const unsigned* amax;            /* kind 4 (fp16x2 pack): the weight tensor's amax slot */
} PnnpPackJob;
int pnnp_pack_jobs_f32(const PnnpPackJob* jobs /*[host]*/, int n, void* stream);
int pnnp_pack_jobs_add_conv(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd /*or null*/, float* dgrad /*or null*/,
                            int Cout, int Cin, int taps, int Cin_pad, int Cout_pad);
int pnnp_pack_jobs_add_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd /*or null*/, float* dgrad /*or null*/,
                             int Cin, int Cout);
int pnnp_pack_jobs_add_wino(PnnpPackJob* jobs, int* n, int cap, const float* w, float* fwd /*or null*/, float* dgrad /*or null*/,
                            int Cout, int Cin);
int pnnp_pack_jobs_add_conv3x3s2_dgrad(PnnpPackJob* jobs, int* n, int cap, const float* w, float* dst, int Cout, int Cin);

/* ---- Conv2d 3x3 / stride 1 / pad 1 on the bf16 matrix cores with float32 operands split into three bf16 pieces
 * (csrc/conv_x3.hip: a = hi + mid + lo exactly, six of the nine piece products kept, fp32 accumulation -- float32-accurate
 * results at 6/16 of the fp32-MFMA time).  Same contracts as pnnp_conv_fwd_f32 / pnnp_conv_bwd_data_f32 /
 * pnnp_conv_bwd_data_res_f32 with taps = 9 (archs/Unet.py:16-52,54-92; archs/modules.py:176-197); the weights are x3 packs:
 * kind-2 jobs of the pack table, pnnp_x3_weight_bytes(K, N) bytes each (K = channels reduced over, N = channels written). */
int pnnp_x3_supported(int K, int N);
/* size limit of the convolution kernels (bf16x3 AND fp32-MFMA families: all address one image through a buffer resource): one
 * image [H][W][cstride] of every map they touch must fit a 32-bit byte offset ((H + 4) W cstride 4 < 2^31); beyond it the launchers
 * return PNNP_E_UNSUPPORTED -- ask first and tile the frame (the Python engines refuse such a frame before launching anything).
 * bf16x3 dynamic range: a piece is a bf16, which has float32's exponent range, so for 2^-110 (7.7e-34) <= |a| <= float32 max all
 * three pieces are normal numbers and hi + mid + lo == a exactly; below that the lo (then mid) piece becomes a bf16 subnormal and
 * may be flushed by the matrix core: accuracy degrades gracefully to 16 (8) significand bits, never to garbage
 * (tests/test_gpu_x3.py::test_x3_dynamic_range, ::test_x3_below_the_supported_range_degrades_gracefully).  Products and sums are
 * float32 and obey float32's own range. */
int pnnp_x3_image_fits(int H, int W, int cstride);
int64_t pnnp_x3_weight_bytes(int K, int N);
int pnnp_pack_jobs_add_x3(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/,
                          int Cout, int Cin, int Cin_pad);
int pnnp_conv3x3_x3_fwd_f32(const float* x1, int C1, const float* x2, int C2, const void* w_x3, const float* bias,
                            const float* residual, float* y, int B, int H, int W, int Cout, int act, void* stream);
/* the same layer with the MaxPool2d(2) behind it (archs/Unet.py:35,41,47,53) fused into its epilogue: y as above, pooled
 * [B][H/2][W/2][Cout] and codes exactly as pnnp_maxpool2_fwd_codes_f32 would produce from y (H, W even). */
int pnnp_conv3x3_x3_fwd_pool_f32(const float* x1, int C1, const float* x2, int C2, const void* w_x3, const float* bias, float* y,
                                 float* pooled, unsigned char* codes, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad,
                                 float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                 float* dx2 /*or null*/, int C2, const float* mask2, int mode2, int accum2,
                                 int B, int H, int W, void* stream);
int pnnp_conv3x3_x3_bwd_data_res_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int C1,
                                     const float* addsrc, const float* mask, int mode, int B, int H, int W, void* stream);
/* ---- the same layers on the fp16 matrix cores with float32 operands split into TWO scaled fp16 pieces ("h2": csrc/conv_h2s.hip, csrc/h2.h):
 * hi = f16(s a), lo = f16(s a - hi), a b = hi hi' + hi lo' + lo hi' -- three fp16 products per multiply where bf16x3 needs six, fp32
 * accumulation; s = the power of two that brings the TENSOR's largest magnitude into [2^14, 2^15).  That maximum travels in a 4-byte "amax
 * slot" beside every tensor the kernels split: the bit pattern of max |element| (non-negative floats order like unsigned integers), zeroed by
 * the caller once per pass and raised with atomicMax by whichever kernel writes the tensor (amax_y / amax_dx arguments here;
 * pnnp_amax_f32 for a tensor from elsewhere).  A slot may over-estimate (costs range at the small end), never under-estimate.
 * Supported range: every float32 tensor whose elements of interest lie within 2^-18 of its maximum keeps 22 significand bits per operand;
 * smaller elements carry an absolute error of 2^-40 of the maximum (tests/test_gpu_h2.py runs the float64 yardsticks of the bf16x3 family).
 * Sign bits: a forward layer can store (activated output > 0) as one bit per element (bits_y: pnnp_h2_bits_words(B, H, W, Cout) words, a
 * tile-private order that only pnnp_conv3x3_h2_bwd_data_f32 reads back) -- the LeakyReLU' / ReLU' mask of the backward pass at 1/32 of the
 * float32 activation's traffic (archs/Unet.py:52,57-69).  Weights: kind-4 packs of pnnp_h2_weight_bytes(K, N) bytes, scaled with the
 * weight tensor's own slot (pnnp_pack_jobs_add_amax in an earlier launch of the pack table, then pnnp_pack_jobs_add_h2). */
int pnnp_h2_supported(int K, int N);
/* GEMM columns (output channels) per workgroup tile that pnnp_conv3x3_h2_* picks for a [B][H][W] map with N channels written: 64, or 32 when N < 64
 * or 64-column tiles would leave CUs idle (the rule the launcher itself uses).  The 32-column tiles are the HBM-bound instantiations of the
 * 512 x 512 level (bench.py reports them against the HBM roofline).  pool != 0: the forward + MaxPool2d(2) entry. */
int pnnp_h2_tile_columns(int B, int H, int W, int N, int pool);
int64_t pnnp_h2_weight_bytes(int K, int N);
int64_t pnnp_h2_bits_words(int B, int H, int W, int C);
int pnnp_amax_f32(const float* x, int64_t count, unsigned* slot, void* stream);
int pnnp_pack_jobs_add_amax(PnnpPackJob* jobs, int* n, int cap, const float* x, int64_t count, unsigned* slot);
int pnnp_pack_jobs_add_h2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/,
                          int Cout, int Cin, int Cin_pad, const unsigned* amax_w);
int pnnp_conv3x3_h2_fwd_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2 /*or null*/, int C2, const unsigned* amax_x2,
                            const void* w_h2, const unsigned* amax_w, const float* bias, const float* residual, float* y,
                            unsigned* amax_y /*or null*/, unsigned* bits_y /*or null*/, int B, int H, int W, int Cout, int act, void* stream);
/* Split-K for small grids (round 6; an eval forward on ONE 512 x 512 crop gives the deep layers 16-32 output tiles for 256 CUs): pnnp_h2_splitk returns the
 * number of K slices for a layer with `chunks` = segments x ceil(C / 16) chunks of K (1 = launch as usual); with ksplit > 1 the kernel writes raw partial
 * sums into ws ([ksplit][B][H][W][Cout] floats) and a reduce kernel adds them in a fixed order (deterministic), then bias, activation, amax_y and the sign
 * bits exactly as pnnp_conv3x3_h2_fwd_f32 would have.  Another partition of the K sum: results differ from the unsplit launch by float32 rounding only. */
int pnnp_h2_splitk(int B, int H, int W, int chunks, int N);
int pnnp_conv3x3_h2_fwd_splitk_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2 /*or null*/, int C2, const unsigned* amax_x2,
                                   const void* w_h2, const unsigned* amax_w, const float* bias, float* y, unsigned* amax_y, unsigned* bits_y,
                                   int B, int H, int W, int Cout, int act, int ksplit, float* ws, int64_t ws_floats, void* stream);
/* The last 3x3 layer + LeakyReLU + the 1x1 head conv10_1 (archs/Unet.py:93-94; ResUnet: archs/ResUnet.py conv_out) in ONE kernel (round 6): Cout == 32
 * (one workgroup tile holds every channel of a pixel), head_w [4][32] / head_b [4] are the parameters as they are, head_out NCHW [B][4][H][W] float32
 * (+ head_res, NCHW like head_out: the `res` networks' input, or null).  y (the 32-channel map, with amax_y / bits_y as in pnnp_conv3x3_h2_fwd_f32) may be
 * NULL: an eval forward neither writes nor re-reads it; a training forward passes it (backward needs it: pnnp_head_bwd_f32).  The head's sums run in
 * float32 on the vector ALUs over the ACTIVATED float32 accumulators (the same arithmetic as pnnp_head_fwd_f32, another summation order). */
int pnnp_conv3x3_h2_fwd_head_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2 /*or null*/, int C2, const unsigned* amax_x2,
                                 const void* w_h2, const unsigned* amax_w, const float* bias, float* y /*or null*/, unsigned* amax_y, unsigned* bits_y,
                                 const float* head_w, const float* head_b /*or null*/, const float* head_res /*or null*/, float* head_out,
                                 int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3_h2_fwd_pool_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2 /*or null*/, int C2, const unsigned* amax_x2,
                                 const void* w_h2, const unsigned* amax_w, const float* bias, float* y, float* pooled, unsigned char* codes,
                                 unsigned* amax_y /*or null*/, unsigned* bits_y /*or null*/, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                 float* dx1, int C1, const float* mask1, const unsigned* bits1, int mode1, int accum1, unsigned* amax_dx1,
                                 float* dx2 /*or null*/, int C2, const float* mask2, const unsigned* bits2, int mode2, int accum2, unsigned* amax_dx2,
                                 int B, int H, int W, void* stream);
/* backward-weight: both operands are split on the fly; same workspace (pnnp_x3_wgrad_workspace_floats), supported shapes
 * (pnnp_x3_wgrad_supported, pnnp_x3_wgrad_fits) and contract as pnnp_conv3x3_x3_bwd_weight_f32 */
int pnnp_conv3x3_h2_bwd_weight_f32(const float* g, int g_cs, int Cout, const unsigned* amax_g, const float* x1, int x1_cs, int C1, const unsigned* amax_x1,
                                   const float* x2 /*or null*/, int x2_cs, int C2, const unsigned* amax_x2, float* dW, float* dbias /*or null*/,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_conv3x3_h2_bwd_data_res_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                     float* dx, int C1, const float* addsrc, const float* mask, int mode, unsigned* amax_dx,
                                     int B, int H, int W, void* stream);
/* The same pointwise layers on the fp16 matrix cores (csrc/gemm_h2s.hip; round 5): the fp16x2 scheme of the 3x3 kernels (csrc/h2.h) -- per-tensor
 * power-of-two scale from 4-byte amax slots, two fp16 pieces per operand, three products per multiply instead of bf16x3's six.  Contracts of the
 * _x3_ entries below + the slots: amax_x / amax_g of the tensor that is split on the fly, amax_w of the weight tensor (the kind-6 packs of
 * pnnp_pack_jobs_add_h2_convt / _1x1 / _s2 were scaled with it; pnnp_h2mat_bytes(K, N) bytes), amax_y / amax_dx (or null) raised to max |stored|.
 * K (channels per segment) % 32 == 0, N (GEMM columns: 4 Cout for ConvTranspose2d forward) % 32 == 0 (round 6; was 64): pnnp_gemm_h2_supported. */
int pnnp_gemm_h2_supported(int K, int N);
int64_t pnnp_h2mat_bytes(int K, int N);
int pnnp_pack_jobs_add_h2_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/, int Cin, int Cout, const unsigned* amax);
int pnnp_pack_jobs_add_h2_1x1(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/, int Cout, int Cin, const unsigned* amax);
int pnnp_pack_jobs_add_h2_s2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null: 9 x pnnp_h2mat_bytes(Cout, Cin)*/, int Cout, int Cin, const unsigned* amax);
int pnnp_conv3x3s2_h2_fwd_f32(const float* x, int Cin, const unsigned* amax_x, const void* w_h2, const unsigned* amax_w, const float* bias /*or null*/, float* y,
                              unsigned* amax_y /*or null*/, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3s2_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_s2dgrad, const unsigned* amax_w, float* dx, int Cin,
                                   const float* mask /*or null*/, int mode, int accum, unsigned* amax_dx /*or null*/, int B, int H, int W, void* stream);
/* ... and their backward-weights (csrc/wgrad_h2g.hip): contracts of the _x3_ entries + the amax slots of the two tensors that are split; shapes and
 * workspace: pnnp_h2g_wgrad_supported / _workspace_floats (kind as pnnp_x3g_wgrad_supported: everything it takes, and stride-2 with Cout % 64 == 0) */
int pnnp_h2g_wgrad_supported(int kind, int M, int N);
int64_t pnnp_h2g_wgrad_workspace_floats(int kind, int B, int UH, int UW, int M, int N);
int pnnp_convt2x2_h2_bwd_weight_f32(const float* x, int Cin, const unsigned* amax_x, const float* g, int Cout, const unsigned* amax_g, float* dW, float* dbias /*or null*/,
                                    int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_conv3x3s2_h2_bwd_weight_f32(const float* g, int Cout, const unsigned* amax_g, const float* x, int Cin, const unsigned* amax_x, float* dW, float* dbias /*or null*/,
                                     int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_conv1x1_h2_bwd_weight_f32(const float* g, int g_cs, int Cout, const unsigned* amax_g, const float* x1, int x1_cs, int C1, const unsigned* amax_x1,
                                   const float* x2 /*or null*/, int x2_cs, int C2, const unsigned* amax_x2, float* dW, float* dbias /*or null*/,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_convt2x2_h2_fwd_f32(const float* x, int Cin, const unsigned* amax_x, const void* w_h2, const unsigned* amax_w, const float* bias /*or null*/, float* y,
                             unsigned* amax_y /*or null*/, int B, int H, int W, int Cout, void* stream);
int pnnp_convt2x2_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w, float* dx, int Cin,
                                  const float* mask /*or null*/, int mode, unsigned* amax_dx /*or null*/, int B, int H, int W, void* stream);
/* ... with the act' mask as the SIGN BITS pnnp_conv3x3_h2_fwd_f32 / _fwd_pool_f32 stored for the layer's input map (round 6: the float32 activation is not read:
 * 503 MB per UNet step); bits: pnnp_h2_bits_words(B, H, W, Cin) words, Cin % 32 == 0, mode 1 LeakyReLU(0.2)' / 2 ReLU'.  Bit-identical to the float32-mask entry. */
int pnnp_convt2x2_h2_bwd_data_bits_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w, float* dx, int Cin,
                                       const unsigned* bits, int mode, unsigned* amax_dx, int B, int H, int W, void* stream);
int pnnp_conv1x1_h2_fwd_f32(const float* x1, int C1, const unsigned* amax_x1, const float* x2 /*or null*/, int C2, const unsigned* amax_x2, const void* w_h2,
                            const unsigned* amax_w, const float* bias /*or null*/, const float* residual /*or null*/, float* y, unsigned* amax_y /*or null*/,
                            int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv1x1_h2_bwd_data_f32(const float* g, int Cout, const unsigned* amax_g, const void* w_h2_dgrad, const unsigned* amax_w,
                                 float* dx1, int C1, const float* mask1, int mode1, int accum1, unsigned* amax_dx1 /*or null*/,
                                 float* dx2 /*or null*/, int C2, const float* mask2, int mode2, int accum2, int B, int H, int W, void* stream);
/* pointwise layers on the bf16 matrix cores (csrc/gemm_x3.hip): ConvTranspose2d k2 s2 (archs/Unet.py:35-47), Conv2d 1x1 (ResidualBlock
 * shortcuts, archs/modules.py:176-197), Conv2d 3x3 stride 2 (archs/modules.py:130-138); same contracts as pnnp_convt2x2_* /
 * pnnp_conv_fwd_f32 + pnnp_conv_bwd_data_f32 with taps = 1 / pnnp_conv3x3s2_*; channel counts in multiples of 32; the weights are
 * kind-3 packs of pnnp_x3mat_bytes(K, N) bytes (stride-2 backward-data: 9 x pnnp_x3mat_bytes(Cout, Cin)). */
int pnnp_gemm_x3_supported(int K, int N);
int64_t pnnp_x3mat_bytes(int K, int N);
int pnnp_pack_jobs_add_x3_convt(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/, int Cin, int Cout);
int pnnp_pack_jobs_add_x3_1x1(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/, int Cout, int Cin);
int pnnp_pack_jobs_add_x3_s2(PnnpPackJob* jobs, int* n, int cap, const float* w, void* fwd /*or null*/, void* dgrad /*or null*/, int Cout, int Cin);
int pnnp_convt2x2_x3_fwd_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, int B, int H, int W, int Cout, void* stream);
int pnnp_convt2x2_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int Cin, const float* mask, int mode,
                                  int B, int H, int W, void* stream);
/* ... the same two with max |output| raised into an amax slot of the fp16x2 family (see pnnp_conv3x3_h2_fwd_f32) */
int pnnp_convt2x2_x3_fwd_amax_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, unsigned* amax_y /*or null*/,
                                  int B, int H, int W, int Cout, void* stream);
int pnnp_convt2x2_x3_bwd_data_amax_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx, int Cin, const float* mask, int mode,
                                       unsigned* amax_dx /*or null*/, int B, int H, int W, void* stream);
int pnnp_conv1x1_x3_fwd_f32(const float* x1, int C1, const float* x2 /*or null*/, int C2, const void* w_x3, const float* bias, const float* residual,
                            float* y, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv1x1_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_dgrad, float* dx1, int C1, const float* mask1, int mode1, int accum1,
                                 float* dx2 /*or null*/, int C2, const float* mask2, int mode2, int accum2, int B, int H, int W, void* stream);
int pnnp_conv3x3s2_x3_fwd_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3s2_x3_bwd_data_f32(const float* g, int Cout, const void* w_x3_s2dgrad, float* dx, int Cin, const float* mask, int mode, int accum,
                                   int B, int H, int W, void* stream);
int pnnp_conv3x3s2_x3_fwd_amax_f32(const float* x, int Cin, const void* w_x3, const float* bias, float* y, unsigned* amax_y /*or null*/,
                                   int B, int H, int W, int Cout, int act, void* stream);
int pnnp_conv3x3s2_x3_bwd_data_amax_f32(const float* g, int Cout, const void* w_x3_s2dgrad, float* dx, int Cin, const float* mask, int mode, int accum,
                                        unsigned* amax_dx /*or null*/, int B, int H, int W, void* stream);
/* backward-weight of the same layers (csrc/wgrad_x3.hip; pixel-major LDS images read with ds_read_b64_tr_b16): same contract
 * as pnnp_conv_bwd_weight_f32 with taps = 9; channel counts in multiples of 32; workspace from the query. */
int pnnp_x3_wgrad_supported(int H, int W, int Cout, int C1, int C2);
/* its size limit: the WHOLE batch of a map [B][H][W][cstride] must fit a 32-bit byte offset ((B H + 2) W cstride 4 < 2^31) */
int pnnp_x3_wgrad_fits(int B, int H, int W, int cstride);
int64_t pnnp_x3_wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin);
int pnnp_conv3x3_x3_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                                   const float* x2 /*or null*/, int x2_cs, int C2, float* dW, float* dbias /*or null*/,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);

/* backward-weight of the POINTWISE / STRIDED layers on the same scheme (csrc/wgrad_x3g.hip, round 4; they ran on the fp32 matrix cores
 * before): kind 0 = Conv2d 1x1 (M = Cout, N = C1 + C2; replaces pnnp_conv_bwd_weight_f32 with taps = 1, archs/modules.py:184-187),
 * 1 = ConvTranspose2d 2x2 stride 2 (M = Cin, N = Cout; replaces pnnp_convt2x2_bwd_weight_f32, archs/Unet.py:35-47),
 * 2 = Conv2d 3x3 stride 2 (M = Cout, N = Cin; replaces pnnp_conv3x3s2_bwd_weight_f32, archs/ResUnet.py:18-27).  Same contracts as
 * the entries they replace; `supported` says whether (M, N) has a tile configuration (otherwise use the fp32 entry); the workspace
 * query takes the LOW-resolution map (the layer input of the ConvTranspose2d, the output of the stride-2 convolution). */
int pnnp_x3g_wgrad_supported(int kind, int M, int N);
int64_t pnnp_x3g_wgrad_workspace_floats(int kind, int B, int UH, int UW, int M, int N);
int pnnp_convt2x2_x3_bwd_weight_f32(const float* x, int Cin, const float* g, int Cout, float* dW, float* dbias /*or null*/,
                                    int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_conv3x3s2_x3_bwd_weight_f32(const float* g, int Cout, const float* x, int Cin, float* dW, float* dbias /*or null*/,
                                     int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);
int pnnp_conv1x1_x3_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                                   const float* x2 /*or null*/, int x2_cs, int C2, float* dW, float* dbias /*or null*/,
                                   int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats, void* stream);

/* (The Winograd kernel's cycle-stamp hook `pnnp_wino_set_debug` exists only in profiling builds, -DPNNP_WINO_DEBUG=1: the shipped
 * library exports no debug hook, reads no environment variable and keeps no state between calls besides idempotent per-device
 * caches of device facts and the persistent-split setting above.) */
int pnnp_wino_wgrad_supported(int H, int W, int Cout, int C1, int C2);
int64_t pnnp_wino_wgrad_workspace_floats(int B, int H, int W, int Cout, int Cin);
int pnnp_conv3x3_wino_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                                     const float* x2, int x2_cs, int C2, float* dW, float* dbias /*or null*/,
                                     int B, int H, int W, int accumulate, float* workspace, int64_t workspace_floats,
                                     void* stream);
/* backward-weight: dW [Cout][C1+C2][taps] (+ dbias [Cout]); workspace from the query below. */
int64_t pnnp_wgrad_workspace_floats(int B, int H, int W, int M, int N, int taps);
int pnnp_wgrad_splits(int B, int H, int W, int M, int N, int taps);
/* backward-data through an identity shortcut: dx = (conv_bwd_data(g) + addsrc) * act'(mask)
 * (ResidualBlock, archs/modules.py:193-197). */
int pnnp_conv_bwd_data_res_f32(const float* g, int Cout, const float* w_dgrad, float* dx, int C1,
                               const float* addsrc, const float* mask, int mode, int B, int H, int W,
                               int taps, void* stream);
/* *_cs = channels per pixel of that tensor (>= the channels used: the 4-channel boundary
 * tensors travel zero-padded to 8 channels). */
int pnnp_conv_bwd_weight_f32(const float* g, int g_cs, int Cout, const float* x1, int x1_cs, int C1,
                             const float* x2, int x2_cs, int C2,
                             float* dW, float* dbias /*or null*/, int B, int H, int W, int taps,
                             int accumulate, float* workspace, int64_t workspace_floats, void* stream);
/* ConvTranspose2d(Cin, Cout, 2, stride=2): x [B][H][W][Cin] <-> y [B][2H][2W][Cout]. */
int pnnp_convt2x2_fwd_f32(const float* x, int Cin, const float* w_packed, const float* bias, float* y,
                          int B, int H, int W, int Cout, void* stream);
int pnnp_convt2x2_bwd_data_f32(const float* g, int Cout, const float* w_dgrad, float* dx, int Cin,
                               const float* mask, int mode, int B, int H, int W, void* stream);
int pnnp_convt2x2_bwd_weight_f32(const float* x, int Cin, const float* g, int Cout, float* dW,
                                 float* dbias /*[Cout] = channel sums of g, or null*/, int B, int H, int W, int accumulate,
                                 float* workspace, int64_t workspace_floats, void* stream);
/* Conv2d 3x3 stride 2 pad 1 (ResUnet down-sampling `conv3x3`, archs/modules.py:130-138,
 * archs/ResUnet.py:18-27): x [B][H][W][Cin] <-> y [B][H/2][W/2][Cout].  forward takes the ordinary
 * forward pack; backward-data its own pack (9*Cout*Cin floats); backward-weight workspace is
 * pnnp_wgrad_workspace_floats(B, H/2, W/2, Cout, Cin, 18). */
int pnnp_conv3x3s2_fwd_f32(const float* x, int Cin, const float* w_packed, const float* bias, float* y,
                           int B, int H, int W, int Cout, int act, void* stream);
int pnnp_pack_conv3x3s2_dgrad_f32(const float* w, float* dst, int Cout, int Cin, void* stream);
int pnnp_conv3x3s2_bwd_data_f32(const float* g, int Cout, const float* w_s2dgrad, float* dx, int Cin,
                                const float* mask, int mode, int accum, int B, int H, int W, void* stream);
int pnnp_conv3x3s2_bwd_weight_f32(const float* g, int Cout, const float* x, int Cin, float* dW,
                                  float* dbias /*or null*/, int B, int H, int W, int accumulate,
                                  float* workspace, int64_t workspace_floats, void* stream);
/* MaxPool2d(2) (archs/Unet.py:57-69); backward routes to the first maximum of each window,
 * multiplies by act'(x) and optionally accumulates into gx (skip-connection gradient). */
int pnnp_maxpool2_fwd_f32(const float* x, float* y, int B, int H, int W, int C, void* stream);
int pnnp_maxpool2_bwd_f32(const float* x, const float* gy, float* gx, int B, int H, int W, int C,
                          int act_mode, int accumulate, void* stream);
/* The same pair with a one-byte code per pooled element (bits 0-1: first maximum of the window, bits 2-5: sign of its four
 * elements) written by the forward pass, so that the backward pass needs gy and the codes only (no second read of x). */
int pnnp_maxpool2_fwd_codes_f32(const float* x, float* y, unsigned char* codes /*[B][H/2][W/2][C]*/, int B, int H, int W, int C, void* stream);
int pnnp_maxpool2_bwd_codes_f32(const unsigned char* codes, const float* gy, float* gx, int B, int H, int W, int C, int act_mode,
                                int accumulate, void* stream);
/* ... and max |gx| (of the sums it stored, when accumulating) into an amax slot of the fp16x2 family (see pnnp_conv3x3_h2_fwd_f32) */
int pnnp_maxpool2_bwd_codes_amax_f32(const unsigned char* codes, const float* gy, float* gx, int B, int H, int W, int C, int act_mode,
                                     int accumulate, unsigned* amax_gx /*or null*/, void* stream);
/* ---------------------------------------------------------------- the thin ends of the networks (csrc/thin.hip)
 * The 1x1 head conv10_1 (archs/Unet.py:80,93: nf -> out_nc, no activation) and the first 3x3 convolution's weight gradient
 * (archs/Unet.py:31 conv1_1, in_nc -> nf).  With 4 channels on one side these are HBM streams over the full-resolution
 * nf-channel map, done in float32 on the vector ALUs; weights are read in the torch layout (no packed copy).
 *   head forward:  out NCHW [B][cout][H][W] = bias + x W^T (+ residual NCHW)   -- replaces the 1x1 GEMM + pnnp_nhwc_to_nchw_f32
 *   head backward: gx = (g W) * act'(x) [mode 0 none / 1 LeakyReLU(0.2) / 2 ReLU; x is the activation output],
 *                  dW [cout][cin] and dbias [cout] (+)= in the same pass over x and g (g: first cout of gcs >= 4 channels)
 *   first forward: y [B][H][W][ycs] = act(conv3x3(x) + bias), the 4 input channels NOT padded to a GEMM chunk
 *   first backward-weight: dW [cout][cin][3][3], dbias (+)=; x [B][H][W][xcs] must be ZERO in channels cin .. 3.
 * *_supported() say whether these kernels take the layer (else use pnnp_conv_* with taps 1 / 9); ws >= *_workspace_floats(). */
int pnnp_head_supported(int cin, int cout, int64_t npix);
int64_t pnnp_head_bwd_workspace_floats(int cin);
int pnnp_head_fwd_f32(const float* x, int xcs, int cin, const float* w /*[cout][cin]*/, const float* bias, const float* residual /*or null*/,
                      float* out, int B, int H, int W, int cout, void* stream);
int pnnp_head_bwd_f32(const float* g, int gcs, const float* x, int xcs, int cin, const float* w, float* gx, int gxcs, int mode,
                      float* dW, float* dbias /*or null*/, int B, int H, int W, int cout, int accumulate, float* ws, int64_t ws_floats,
                      void* stream);
int pnnp_head_bwd_amax_f32(const float* g, int gcs, const float* x, int xcs, int cin, const float* w, float* gx, int gxcs, int mode,
                           float* dW, float* dbias /*or null*/, int B, int H, int W, int cout, int accumulate, float* ws, int64_t ws_floats,
                           unsigned* amax_gx /*or null: max |gx| into an amax slot of the fp16x2 family*/, void* stream);
int pnnp_first_wgrad_supported(int cin, int cout, int H, int W);
int64_t pnnp_first_wgrad_workspace_floats(int cout);
int pnnp_first_fwd_f32(const float* x, int xcs, int cin, const float* w /*[cout][cin][3][3]*/, const float* bias /*or null*/, float* y, int ycs,
                       int B, int H, int W, int cout, int act /*0 none, 1 LeakyReLU(0.2), 2 ReLU*/, void* stream);
int pnnp_first_fwd_amax_f32(const float* x, int xcs, int cin, const float* w, const float* bias /*or null*/, float* y, int ycs,
                            int B, int H, int W, int cout, int act, unsigned* amax_y /*or null: max |y| into an amax slot*/, void* stream);
int pnnp_first_bwd_weight_f32(const float* g, int gcs, int cout, const float* x, int xcs, int cin, float* dW, float* dbias /*or null*/,
                              int B, int H, int W, int accumulate, float* ws, int64_t ws_floats, void* stream);
/* boundary layout changes: NCHW [B][C][H][W] <-> NHWC [B][H][W][Cp] (Cp >= C, zero padded);
 * the NCHW result can add a residual (arch 'res' flag, archs/Unet.py:95-98). */
int pnnp_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream);
/* the same into a frame that is reflect-padded by `pad` pixels on every side, dst [B][H+2pad][W+2pad][Cp]: the eval loop's
 * F.pad(imgs_lr, (4,4,4,4), mode='reflect') for frames whose width is not a multiple of 16 (trainer_SID.py:221-226), folded into the
 * layout change the network input goes through anyway */
int pnnp_nchw_to_nhwc_reflect_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, int pad, void* stream);
/* ... with max |element| raised into an amax slot of the fp16x2 family (the network input's scale: conv1_1 on pnnp_conv3x3_h2_fwd_f32) */
int pnnp_nchw_to_nhwc_reflect_amax_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, int pad, unsigned* amax /*or null*/, void* stream);
int pnnp_nhwc_to_nchw_f32(const float* src, const float* residual /*or null*/, float* dst,
                          int B, int C, int H, int W, int Cp, void* stream);
/* out[c] (+)= sum over pixels (bias gradient of a ConvTranspose2d); workspace >= 1024*C floats */
int pnnp_channel_sum_f32(const float* x, float* out, int64_t npix, int C, int accumulate,
                         float* workspace, void* stream);
/* loss = mean|clamp(pred,0,1) - hr| (trainer_SID.py:99, losses/base_loss.py:92-107) on NCHW
 * tensors; loss_out[0] = loss, loss_out[1+b] = sum_b (clamp(pred)-clamp(hr))^2 for PSNR_Loss
 * (losses/__init__.py:4-15); grad_nhwc (or null) = dL/dpred as [B][H][W][Cp].
 * workspace >= 128*B floats. */
int pnnp_l1_clamp_loss_f32(const float* pred, const float* hr, float* grad_nhwc, float* loss_out,
                           int B, int C, int H, int W, int Cp, float* workspace, void* stream);
/* the same with the `ori` branch of the train loop (trainer_SID.py:97-99: pred = pred * ratio before the loss):
 * scale [B] (device, or null = 1) multiplies crop b's prediction first; loss, SSE and dL/dpred (x scale[b]) follow. */
/* pnnp_l1_clamp_loss_scaled_f32 with the TARGET clamped to [0,1] inside the kernel when clamp_target != 0 (preprocess's
 * `imgs_hr.clamp(0, 1)` under dst.clip, trainer_SID.py:485, without an elementwise pass of its own) */
int pnnp_l1_clamp_loss_tc_f32(const float* pred, const float* hr, const float* scale /*[B] or null*/, float* grad_nhwc, float* loss_out,
                              int B, int C, int H, int W, int Cp, float* workspace, int clamp_target, void* stream);
int pnnp_l1_clamp_loss_scaled_f32(const float* pred, const float* hr, const float* scale /*[B] or null*/, float* grad_nhwc,
                                  float* loss_out, int B, int C, int H, int W, int Cp, float* workspace, void* stream);
/* pnnp_l1_clamp_loss_tc_f32 with dL/dpred multiplied by grad_weight: the weight B_local world / B_global of a rank's mean gradient when one
 * global batch is split unevenly over the data-parallel ranks (base_trainer.py:115-118 scatters the batch the same way); loss and SSE unweighted. */
int pnnp_l1_clamp_loss_w_f32(const float* pred, const float* hr, const float* scale /*[B] or null*/, float* grad_nhwc, float* loss_out,
                             int B, int C, int H, int W, int Cp, float* workspace, int clamp_target, float grad_weight, void* stream);
/* torch.optim.Adam step (trainer_SID.py:44,101) over a flat parameter buffer; step is 1-based;
 * grad_scale is applied to g first (1/world_size after a sum all-reduce). */
int pnnp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                       float beta2, float eps, int step, float grad_scale, void* stream);

/* ---------------------------------------------------------------- NoiseFlow.sample
 * archs/noise_flow.py:173-188: z ~ N(0,I) pushed through the reversed bijector chain
 * 8 x [AffineCoupling^-1, Conv2d1x1^-1] (+ GainISO^-1 after the 4th pair, SignalDependantISO^-1
 * at the end).  NCHW fp32 [B][4][H][W].  `step` [host] = 317 floats (struct NfStep in csrc/nf.hip):
 * conv2d_1 w[4][2][9] b[4], BN1 scale[4] offset[4], conv2d_2 w[4][4] b[4], BN2 scale[4] offset[4],
 * conv2d_3 w[4][5][9] b[4], exp(3*logs)[4], scale, W^-1[4][4] (x ISO gain where it applies). */
int pnnp_normal_fill_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
int pnnp_nf_step_f32(const float* x, float* y, int B, int H, int W, const float* step /*[host]*/,
                     const float* clean /*or null*/, float sdn_a, float sdn_b, float out_mul, void* stream);
/* the same step with the trainer's preprocess around sample() fused in (trainer_SID.py:463-472,481-485; trainer_LRID.py:419-427):
 * the clean crop is divided by clean_div[b] (or clean_div_s) where it enters the signal-dependent scale (`clean = imgs_hr / ratio`);
 * with mix_base the result is clamp(mix_base + (sample * mix_mul[b] or mix_mul_s), clamp_lo, clamp_hi) (`imgs_lr = imgs_hr + noise *
 * ratio; imgs_lr.clamp(lb, 1)`); flag (device int, or null): bit 0 is set when the scale a*clean + b is negative anywhere -- the
 * reference's `assert scale >= 0` (signal_dependant.py:50) without a host round trip per step. */
int pnnp_nf_step_mix_f32(const float* x, float* y, int B, int H, int W, const float* step /*[host]*/,
                         const float* clean /*or null*/, float sdn_a, float sdn_b, float out_mul,
                         const float* clean_div /*[B] or null*/, float clean_div_s, const float* mix_base /*or null*/,
                         const float* mix_mul /*[B] or null*/, float mix_mul_s, float clamp_lo, float clamp_hi, int* flag /*or null*/,
                         const float* bn_stats /*device [24] or null*/, void* stream);
/* training-mode sampling (trainer_LRID.py:34-39,420-427: the proxy is never put in eval mode): with bn_stats = the batch statistics
 * pnnp_nf_train_stats_f32 left on the device, the step's BatchNorm slots hold weight / bias and scale = gamma rstd, offset =
 * beta - (mean + conv bias) scale are formed in the kernel; pnnp_nf_bn_update_f32 moves the running buffers of the coupling's two
 * BatchNorm2d layers as nn.BatchNorm2d does (momentum 0.1, unbiased variance, num_batches_tracked += 1; n = B H W).  No host round trip. */
int pnnp_nf_bn_update_f32(const float* bn_stats, const float* conv_bias1, const float* conv_bias2, float* running_mean1,
                          float* running_var1, float* running_mean2, float* running_var2, long long* batches1 /*or null*/,
                          long long* batches2 /*or null*/, double n, void* stream);
/* One [Conv2d1x1, AffineCoupling] pair of the forward (density) chain, NoiseFlow.forward (archs/noise_flow.py:113-130);
 * partial [B][ceil(H/32)*ceil(W/32)] receives the per-workgroup sums of the pixel-wise log-det terms. */
int pnnp_nf_fwd_step_f32(const float* x, float* y, float* partial, int B, int H, int W, const float* step /*[host]*/,
                         const float* clean, float sdn_a, float sdn_b, void* stream);

/* ---------------------------------------------------------------- NoiseFlow NLL fitting (csrc/nf_train.hip)
 * net.train(); nll, _ = net.loss(noise=, clean=, iso=); nll.backward()  (trainer_NF_SID.py:102,116-126; archs/noise_flow.py:113-165):
 * one [SignalDependantISO | GainISO, Conv2d1x1, AffineCoupling] pair of the forward chain with BatchNorm in training mode
 * (batch statistics, affine_coupling.py:257-264), and its backward.  All pointers are DEVICE pointers (parameters are read
 * from the device so that a step needs no host round trip):
 *   wm [4][4]   the Conv2d1x1 matrix W = P L U (conv2d1x1.py:58-65), divided by the GainISO scale where that layer precedes it
 *   ab {a, b}   SignalDependantISO scale sqrt(a*clean + b), a = beta1/gain, b = beta2 (signal_dependant.py:37-51); with clean, first pair
 *   prm [301]   W1[4][2][9] B1[4] G1[4] BE1[4] W2[4][4] B2[4] G2[4] BE2[4] W3[4][5][9] B3[4] LOGS[4] SCALE  (G/BE = BatchNorm weight/bias)
 *   bn [24]     out: mean1[4] rstd1[4] var1[4] mean2[4] rstd2[4] var2[4] (biased variances; eps 1e-5; means of the BIAS-FREE conv outputs)
 *   h1, h2, out3 [B][4][H][W]  saved conv2d_1 / conv2d_2 outputs (pre-BatchNorm, without their biases, which a batch-statistics
 *               BatchNorm cancels exactly) and conv2d_3 output * exp(3 logs)
 *   ldpart [tiles][2]  per-workgroup (sum of the pixel log-det terms, sum z^2), tiles = pnnp_nf_train_tiles(B,H,W), crop-major
 *   part        scratch: max(tiles*197, pblocks*28) floats covers both calls (pblocks = pnnp_nf_train_pblocks) */
int pnnp_nf_train_tiles(int B, int H, int W);
int pnnp_nf_train_pblocks(int B, int H, int W);
int pnnp_nf_train_fwd_pair_f32(const float* x, const float* clean /*or null*/, const float* ab, const float* wm, const float* prm,
                               float* bn, float* h1, float* h2, float* out3, float* z, float* ldpart, float* part, int B, int H, int W,
                               void* stream);
/* Backward of the pair: dz = gradient of its output (times dzmul; the last pair passes z and dzmul = -dL/dF / B for the N(0,I)
 * prior), cobj = dL/d(objective) per crop.  dx: gradient of its input.  sums [319] (deterministic reductions):
 *   dW3[180] dB3[4] dLOGS[4] dSCALE dBE2[4] dG2[4] | dW2[16] dB2[4] dBE1[4] dG1[4] | dW1[72] dB1[4] dWm[16] da db
 * dy2, dy1 [B][4][H][W] and dv23 [B][2][H][W] are scratch. */
int pnnp_nf_train_bwd_pair_f32(const float* x, const float* clean /*or null*/, const float* ab, const float* wm, const float* prm,
                               const float* bn, const float* h1, const float* h2, const float* out3, const float* dz, float dzmul,
                               float cobj, float* dx, float* sums, float* dy2, float* dy1, float* dv23, float* part, int B, int H,
                               int W, void* stream);
/* batch statistics of a coupling's two BatchNorm layers for training-mode SAMPLING (trainer_LRID.py:34-39,420-427: the proxy is
 * sampled without .eval()): u [B][4][H][W] feeds the coupling network with its first two planes; ident = device 4x4 identity;
 * bn [24] as above; h1, h2 scratch [B][4][H][W]; part scratch max(tiles, pblocks) * 8 floats. */
int pnnp_nf_train_stats_f32(const float* u, const float* ident, const float* prm, float* bn, float* h1, float* h2, float* part,
                            int B, int H, int W, void* stream);

/* SNA_torch (data_process/process.py:562-588): shot-noise augmentation under a white-balance gain change.
 * gt [C][H][W] -> dn (the extra Poisson noise, / (wp-bl), x ratio unless ori) and dy (the signal change); aug_wb4 is a
 * HOST array of the four plane gains.  Counter-based RNG as in pnnp_noise_sample_f32. */
int pnnp_sna_f32(const float* gt, float* dn, float* dy, int C, int H, int W, const float* aug_wb4, float K, float wp, float bl,
                 float ratio, int black_lr, int ori, uint64_t seed, uint64_t offset, uint32_t crop, void* stream);

/* HighBitRecovery.map (data_process/process.py:718-751): pixels whose rounded value x lies in [low, high) are re-drawn
 * inside their quantisation bin, d' = ppf(cdf[x-low] + u*range[x-low]) + (d - x); d = data*in_mul; the result is
 * divided by out_div (norm=True) or, if out_div == 0, shifted by out_add (norm=False).  cdf/range: float64 device LUTs of
 * the host-side HB2LB_LUT; rand: optional float64 uniforms (else the counter RNG). */
int pnnp_hbr_map_f32(const float* data, float* out, int64_t n, const double* cdf, const double* range, int low, int high,
                     int dist /* 0 normal, 1 Tukey-lambda */, double loc, double scale, double lam, const double* rand,
                     float in_mul, float out_div, float out_add, int keep_delta, uint64_t seed, uint64_t offset, void* stream);

/* ---------------------------------------------------------------- dataset-side crop / augment
 * init_random_crop_point + random_crop + data_aug (data_process/syn_datasets.py:69-107,162-173; the 4-way
 * variant real_datasets.py:98-137), fused behind raw2bayer (utils/isp_ops.py:84-96), the linear dark-shading
 * subtraction (real_datasets.py:360-368) and the random_gains white-balance augmentation
 * (syn_datasets.py:313-322).  desc [device, n x 4 int32] = {h0, w0, rot90 k, flip}; gains [device, n x 3 f64]
 * = {rgb, red, blue} or NULL; dark [device, H x W] f32/f64 or NULL. */
int pnnp_crop_pack_bayer_u16(const uint16_t* frame, int H, int W, const void* dark, int dark_is_f64, double dark_add,
                             float* dst /* [n][4][ps][ps] */, int n, int ps, const int* desc, const double* gains,
                             const double* black4, double wp, int norm, int clip, int post_clip, void* stream);
int pnnp_crop_aug_f32(const float* img /* [C][h][w] */, int C, int h, int w, float* dst /* [n][C][ps][ps] */,
                      int n, int ps, const int* desc, const double* gains, int post_clip, void* stream);

/* ---------------------------------------------------------------- eval epilogue
 * IlluminanceCorrect.correct (data_process/__init__.py:165-175) and the raw-domain PSNR / SSIM of
 * quality_assess(tensor2im(.), tensor2im(.), data_range=255) (utils/visualization.py:9-31). */
int pnnp_illuminance_correct_f32(const float* pred, const float* src, float* out, int64_t n,
                                 double* workspace /* >= 512 doubles */, void* stream);
/* The elementwise tail of one eval iteration in one pass (trainer_SID.py:226-235: crop the reflect-padded output back, the input residual
 * of a `res` network, `ori`: x ratio, both frames clamped to [0,1]): dn = clamp((net_out[crop] (+ lr_in)) * ratio, 0, 1),
 * lr_out (or null) = clamp(lr_in * ratio, 0, 1).  net_out [C][HP][WP] holds the frame at (pad, pad); the others are [C][H][W]. */
int pnnp_eval_post_f32(const float* net_out, const float* lr_in, float* dn, float* lr_out /*or null*/, int C, int H, int W, int HP, int WP,
                       int pad, float ratio, const float* ratio_dev /*device scalar overriding ratio, or null*/, int add_residual, void* stream);
int pnnp_psnr_ssim_f32(const float* a, const float* b, float* out2 /* {psnr, ssim} */, int C, int H, int W,
                       double* workspace /* >= 2*C*ceil(H/32)*ceil(W/32) doubles */, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNNP_HIP_H */
