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
/* Number of compute units etc. of the current device (0 on failure). */
int pnnp_device_cus(void);

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
int pnnp_noise_sample_f32(const float* y, float* out, int B, int C, int H, int W,
                          const float* params, unsigned flags, float mfm /* sqrt(MultiFrameMean) */,
                          uint64_t seed, uint64_t offset, uint32_t crop_base, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PNNP_HIP_H */
