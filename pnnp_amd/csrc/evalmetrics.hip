// Eval-side epilogue on the device: what trainer_SID.py:230-248 does per image after
// the network: clamp -> IlluminanceCorrect (data_process/__init__.py:144-175) -> raw-domain PSNR / SSIM on
// x255 floats (utils/visualization.py:9-31 -> skimage.metrics.peak_signal_noise_ratio /
// structural_similarity(data_range=255, channel_axis=-1): 7x7 uniform window, K1=.01, K2=.03, sample
// covariance, mean over the window-valid region, mean over channels).  scikit-image is not installed in
// the build image, so the SSIM restatement is pinned to its published definition only (oracle/metrics_np.py).
// All reductions are two-stage with a fixed order (bitwise reproducible).
#include "common.h"

namespace {

constexpr int NB = 256;      // partial blocks of the 1-D reductions

// partial[b] = { sum pc*src, sum pc*pc } over elements with src != 1,  pc = clamp(pred, 0, 1)
__global__ void __launch_bounds__(256)
illum_partial_kernel(const float* __restrict__ pred, const float* __restrict__ src, double* __restrict__ partial, int64_t n) {
    double num = 0.0, den = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = pnnp_clampf(pred[i], 0.f, 1.f), s = src[i];
        if (s != 1.f) { num += (double)p * s; den += (double)p * p; }
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = num; r2[threadIdx.x] = den;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { r1[threadIdx.x] += r1[threadIdx.x + k]; r2[threadIdx.x] += r2[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = r1[0]; partial[2 * blockIdx.x + 1] = r2[0]; }
}

__global__ void __launch_bounds__(256)
illum_apply_kernel(const float* __restrict__ pred, float* __restrict__ out, const double* __restrict__ partial, int64_t n) {
    __shared__ float scale;
    if (threadIdx.x == 0) {
        double num = 0.0, den = 0.0;
        for (int b = 0; b < NB; ++b) { num += partial[2 * b]; den += partial[2 * b + 1]; }
        scale = (float)num / (float)den;          // torch: float32 dots, num / den
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = scale * pnnp_clampf(pred[i], 0.f, 1.f);
}

// torch.clamp(0, 1): a NaN stays a NaN (fminf / fmaxf return the OTHER operand for a NaN, which would turn a diverged network's
// output into 0 and its PSNR into a finite number); comparisons with a NaN are false, so it falls through both selects.
__device__ __forceinline__ float clamp01_nan(float v) { return pnnp_clampf(v, 0.f, 1.f); }

// The elementwise tail of the eval iteration in ONE pass (trainer_SID.py:226-235): crop the padded network output back, add the input
// residual of a `res` network (left out of the padded forward: (f(pad x) + pad x)[crop] = f(pad x)[crop] + x), `ori`: x ratio on both
// frames, clamp both to [0, 1].  Plane c of the network output is [HP][WP], the frame sits at (pad, pad).
__global__ void __launch_bounds__(256)
eval_post_kernel(const float* __restrict__ net_out, const float* __restrict__ lr_in, float* __restrict__ dn, float* __restrict__ lr_out,
                 int C, int H, int W, int HP, int WP, int pad, float ratio_host, const float* __restrict__ ratio_dev, int add_residual) {
    const int64_t n = (int64_t)C * H * W;
    const float ratio = ratio_dev ? ratio_dev[0] : ratio_host;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W); const int64_t q = i / W; const int y = (int)(q % H), c = (int)(q / H);
        const float l = lr_in[i];
        float d = net_out[((int64_t)c * HP + y + pad) * WP + x + pad];
        if (add_residual) d += l;
        dn[i] = clamp01_nan(d * ratio);                         // (imgs_dn * ratio).clamp(0, 1): one rounding for the product, like the tensor op
        if (lr_out) lr_out[i] = clamp01_nan(l * ratio);
    }
}

// One 32x32 tile of one channel: squared error over the tile and SSIM map over the window-valid pixels.  The 7x7 window sums
// of x, y, x^2, y^2, xy are separable: 7-tap row sums of the 38x38 halo tile into LDS (5 x 38 x 32), then 7-tap column sums
// per pixel -- 14 + 35 LDS reads per pixel instead of 98, 77 additions instead of 245.
__global__ void __launch_bounds__(256)
psnr_ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ partial,
                         int H, int W, float c1, float c2) {
    __shared__ float xs[38][39], ys[38][39];
    __shared__ float hs[5][38][33];
    const int c = blockIdx.z, ty0 = blockIdx.y * 32, tx0 = blockIdx.x * 32;
    const float* ac = a + (int64_t)c * H * W;
    const float* bc = b + (int64_t)c * H * W;
    for (int i = threadIdx.x; i < 38 * 38; i += 256) {
        const int r = i / 38, q = i % 38;
        const int gy = ty0 + r - 3, gx = tx0 + q - 3;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        xs[r][q] = in ? pnnp_clampf(ac[(int64_t)gy * W + gx] * 255.f, 0.f, 255.f) : 0.f;    // tensor2im: x255, clip
        ys[r][q] = in ? pnnp_clampf(bc[(int64_t)gy * W + gx] * 255.f, 0.f, 255.f) : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 38 * 32; i += 256) {                 // row sums: hs[.][r][q] = sum_v f(r, q + v)
        const int r = i >> 5, q = i & 31;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int v = 0; v < 7; ++v) {
            const float x = xs[r][q + v], y = ys[r][q + v];
            sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
        }
        hs[0][r][q] = sx; hs[1][r][q] = sy; hs[2][r][q] = sxx; hs[3][r][q] = syy; hs[4][r][q] = sxy;
    }
    __syncthreads();
    double se = 0.0, ss = 0.0;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int r = i >> 5, q = i & 31;
        const int gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
        const float d = xs[r + 3][q + 3] - ys[r + 3][q + 3];
        se += (double)d * d;
        if (gy >= 3 && gy < H - 3 && gx >= 3 && gx < W - 3) {          // crop(S, (win-1)//2)
            float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                sx += hs[0][r + u][q]; sy += hs[1][r + u][q]; sxx += hs[2][r + u][q]; syy += hs[3][r + u][q]; sxy += hs[4][r + u][q];
            }
            const float inv = 1.f / 49.f, covn = 49.f / 48.f;              // use_sample_covariance=True
            const float ux = sx * inv, uy = sy * inv;
            const float vx = covn * (sxx * inv - ux * ux), vy = covn * (syy * inv - uy * uy), vxy = covn * (sxy * inv - ux * uy);
            const float A1 = 2.f * ux * uy + c1, A2 = 2.f * vxy + c2, B1 = ux * ux + uy * uy + c1, B2 = vx + vy + c2;
            ss += (double)((A1 * A2) / (B1 * B2));
        }
    }
    __shared__ double r1[256], r2[256];
    r1[threadIdx.x] = se; r2[threadIdx.x] = ss;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { r1[threadIdx.x] += r1[threadIdx.x + k]; r2[threadIdx.x] += r2[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t blk = ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partial[2 * blk] = r1[0]; partial[2 * blk + 1] = r2[0];
    }
}

// one workgroup: deterministic tree sums of the per-tile partials (a single thread walking 12 k partials took 1 ms)
__global__ void __launch_bounds__(256)
psnr_ssim_finish_kernel(const double* __restrict__ partial, float* __restrict__ out, int nblk_per_chan, int C, int H, int W) {
    __shared__ double r1[256], r2[256];
    __shared__ double se_tot, ssim_tot;
    if (threadIdx.x == 0) { se_tot = 0.0; ssim_tot = 0.0; }
    const double valid = (double)(H - 6) * (W - 6);
    for (int c = 0; c < C; ++c) {
        double se = 0.0, s = 0.0;
        for (int k = threadIdx.x; k < nblk_per_chan; k += 256) {
            se += partial[2 * ((int64_t)c * nblk_per_chan + k)]; s += partial[2 * ((int64_t)c * nblk_per_chan + k) + 1];
        }
        r1[threadIdx.x] = se; r2[threadIdx.x] = s;
        __syncthreads();
        for (int k = 128; k > 0; k >>= 1) {
            if (threadIdx.x < k) { r1[threadIdx.x] += r1[threadIdx.x + k]; r2[threadIdx.x] += r2[threadIdx.x + k]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { se_tot += r1[0]; ssim_tot += r2[0] / valid; }       // per-channel mean SSIM ...
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mse = se_tot / ((double)C * H * W);
        out[0] = (float)(10.0 * log10(255.0 * 255.0 / mse));     // peak_signal_noise_ratio(data_range=255)
        out[1] = (float)(ssim_tot / C);                          // ... averaged over channels
    }
}

}  // namespace

extern "C" {

// out = <pc,src>/<pc,pc> * pc, pc = clamp(pred,0,1), dots over elements with src != 1
// (IlluminanceCorrect.correct, data_process/__init__.py:165-175).  workspace >= 512 doubles.
int pnnp_illuminance_correct_f32(const float* pred, const float* src, float* out, int64_t n, double* workspace, void* stream) {
    if (!pred || !src || !out || !workspace || n <= 0) return PNNP_E_INVALID;
    hipLaunchKernelGGL(illum_partial_kernel, dim3(NB), dim3(256), 0, as_stream(stream), pred, src, workspace, n);
    int64_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(illum_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), pred, out, workspace, n);
    return pnnp_launch_status();
}

// out[0] = PSNR, out[1] = SSIM of clip(a*255,0,255) vs clip(b*255,0,255), a/b [C][H][W] fp32
// (tensor2im + quality_assess, utils/visualization.py:9-31).  workspace >= 2*C*ceil(H/32)*ceil(W/32) doubles.
int pnnp_psnr_ssim_f32(const float* a, const float* b, float* out, int C, int H, int W, double* workspace, void* stream) {
    if (!a || !b || !out || !workspace || C <= 0 || H < 7 || W < 7) return PNNP_E_INVALID;
    const dim3 grid((W + 31) / 32, (H + 31) / 32, C);
    const float c1 = (0.01f * 255.f) * (0.01f * 255.f), c2 = (0.03f * 255.f) * (0.03f * 255.f);
    hipLaunchKernelGGL(psnr_ssim_partial_kernel, grid, dim3(256), 0, as_stream(stream), a, b, workspace, H, W, c1, c2);
    hipLaunchKernelGGL(psnr_ssim_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), workspace, out, (int)(grid.x * grid.y), C, H, W);
    return pnnp_launch_status();
}

// dn = clamp((net_out[crop] (+ lr_in)) * ratio, 0, 1), lr_out (or null) = clamp(lr_in * ratio, 0, 1): the elementwise tail of one eval
// iteration (trainer_SID.py:226-235).  net_out: [C][HP][WP] with the frame at (pad, pad) (pad = 0, HP = H, WP = W: no crop);
// lr_in, dn, lr_out: [C][H][W].  ratio_dev (device scalar) overrides ratio when not null.  NaN stays NaN like torch.clamp (clamp01_nan: selects, not fminf / fmaxf).
int pnnp_eval_post_f32(const float* net_out, const float* lr_in, float* dn, float* lr_out, int C, int H, int W, int HP, int WP, int pad,
                       float ratio, const float* ratio_dev, int add_residual, void* stream) {
    if (!net_out || !lr_in || !dn || C <= 0 || H <= 0 || W <= 0 || pad < 0 || HP < H + 2 * pad || WP < W + 2 * pad) return PNNP_E_INVALID;
    const int64_t n = (int64_t)C * H * W;
    hipLaunchKernelGGL(eval_post_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       net_out, lr_in, dn, lr_out, C, H, W, HP, WP, pad, ratio, ratio_dev, add_residual);
    return pnnp_launch_status();
}

}  // extern "C"
