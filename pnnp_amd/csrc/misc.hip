// HBM-bound companions of the convolution stack: layout changes at the network boundary,
// 2x2 max-pool (forward / backward fused with the LeakyReLU derivative), the clamp + L1 loss
// with its gradient, channel sums (bias gradients) and a fused Adam step.
//   reference: archs/Unet.py:57-69 (MaxPool2d(2)), trainer_SID.py:99 (L1 on pred.clamp(0,1)),
//   losses/__init__.py:4-15 (PSNR_Loss), trainer_SID.py:44,101 (Adam).
#include "common.h"

namespace {

int grid1d(int64_t n, int cap = 256 * 8) {
    int64_t b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// [B][C][H][W] -> [B][H][W][Cp], channels >= C zero-filled (Cp multiple of 4).
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W, int Cp, int pad, unsigned* __restrict__ amax) {
    float amx = 0.f;                                                 // max |element| (csrc/h2.h: the network input's scale when conv1_1 runs on the fp16x2 kernel)
    // dst is [B][H + 2 pad][W + 2 pad][Cp]: pad > 0 reflects the frame (F.pad(..., mode='reflect'): index -k -> k, H-1+k -> H-1-k)
    const int Ho = H + 2 * pad, Wo = W + 2 * pad;
    const int64_t hw = (int64_t)H * W, hwo = (int64_t)Ho * Wo, total = (int64_t)B * hwo * (Cp / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cq = (int)(i % (Cp / 4));
        const int64_t p = i / (Cp / 4);
        const int64_t b = p / hwo;
        int64_t s = p % hwo;
        if (pad) {
            int y = (int)(s / Wo) - pad, x = (int)(s % Wo) - pad;
            y = y < 0 ? -y : (y >= H ? 2 * H - 2 - y : y);
            x = x < 0 ? -x : (x >= W ? 2 * W - 2 - x : x);
            s = (int64_t)y * W + x;
        }
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * cq + k;
            v[k] = c < C ? src[(b * C + c) * hw + s] : 0.f;
        }
        *reinterpret_cast<float4*>(dst + p * Cp + 4 * cq) = make_float4(v[0], v[1], v[2], v[3]);
        amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (amax) pnnp_amax_commit_block(amx, amax);
}

// [B][H][W][Cp] -> [B][C][H][W] (+ residual NCHW, archs/Unet.py:95-98)
__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(const float* __restrict__ src, const float* __restrict__ residual, float* __restrict__ dst,
                    int B, int C, int H, int W, int Cp) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * C * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = i % hw, bc = i / hw;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        float v = src[(b * hw + s) * Cp + c];
        if (residual) v += residual[i];
        dst[i] = v;
    }
}

// NHWC 2x2 max pool, one thread per 4 channels of one output pixel.
__global__ void __launch_bounds__(256)
maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int H, int W, int C) {
    const int h = H / 2, w = W / 2, cq = C / 4;
    const int64_t total = (int64_t)B * h * w * cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        int64_t p = i / cq;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int64_t b = p / h;
        const float* s = x + ((b * H + 2 * oy) * W + 2 * ox) * C + c;
        const float4 a0 = *reinterpret_cast<const float4*>(s), a1 = *reinterpret_cast<const float4*>(s + C);
        const float4 a2 = *reinterpret_cast<const float4*>(s + (int64_t)W * C), a3 = *reinterpret_cast<const float4*>(s + (int64_t)W * C + C);
        float4 m;
        m.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x)); m.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
        m.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z)); m.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
        *reinterpret_cast<float4*>(y + ((b * h + oy) * w + ox) * C + c) = m;
    }
}

// Backward of y = maxpool2(x) where x = act(pre):  gx[pos] (+)= (pos is the window's first max ?
// gy : 0) * act'(x[pos]).  act_mode 0 none, 1 LeakyReLU(0.2)', 2 ReLU'.  accumulate: the skip
// connection's gradient is already in gx (it was written by the decoder's backward-data).
__global__ void __launch_bounds__(256)
maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx,
                   int B, int H, int W, int C, int act_mode, int accumulate) {
    const int h = H / 2, w = W / 2, cq = C / 4;
    const int64_t total = (int64_t)B * h * w * cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        int64_t p = i / cq;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int64_t b = p / h;
        const int64_t base = ((b * H + 2 * oy) * W + 2 * ox) * C + c;
        const int64_t off[4] = {0, (int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
        float xv[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(x + base + off[k]);
            xv[k][0] = t.x; xv[k][1] = t.y; xv[k][2] = t.z; xv[k][3] = t.w;
        }
        const float4 g4 = *reinterpret_cast<const float4*>(gy + ((b * h + oy) * w + ox) * C + c);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int arg = 0; float best = xv[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (xv[k][j] > best) { best = xv[k][j]; arg = k; }   // first max wins
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float d = 1.f;
                if (act_mode == 1) d = xv[k][j] > 0.f ? 1.f : 0.2f;
                else if (act_mode == 2) d = xv[k][j] > 0.f ? 1.f : 0.f;
                o[k][j] = (k == arg) ? g[j] * d : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4* d = reinterpret_cast<float4*>(gx + base + off[k]);
            float4 v = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
            if (accumulate) { const float4 t = *d; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            *d = v;
        }
    }
}

// The same pair with a one-byte code per pooled element written by the forward pass -- bits 0-1: position of the window's
// first maximum, bits 2-5: sign (> 0) of the four window elements -- so that the backward pass does not have to read the
// full-resolution activation again (it is 1/3 of that kernel's HBM traffic): gx is fully determined by gy and the code.
__global__ void __launch_bounds__(256)
maxpool_fwd_codes_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ codes, int B, int H, int W, int C) {
    const int h = H / 2, w = W / 2, cq = C / 4;
    const int64_t total = (int64_t)B * h * w * cq;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        int64_t p = i / cq;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int64_t b = p / h;
        const float* s = x + ((b * H + 2 * oy) * W + 2 * ox) * C + c;
        const int64_t off[4] = {0, (int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
        float xv[4][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(s + off[k]);
            xv[k][0] = t.x; xv[k][1] = t.y; xv[k][2] = t.z; xv[k][3] = t.w;
        }
        float m[4]; unsigned code = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int arg = 0; float best = xv[0][j];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (xv[k][j] > best) { best = xv[k][j]; arg = k; }   // first max wins (as in the backward kernel above)
            unsigned cj = (unsigned)arg;
#pragma unroll
            for (int k = 0; k < 4; ++k) cj |= (xv[k][j] > 0.f ? 1u : 0u) << (2 + k);
            m[j] = fmaxf(fmaxf(xv[0][j], xv[1][j]), fmaxf(xv[2][j], xv[3][j]));
            code |= cj << (8 * j);
        }
        const int64_t o = ((b * h + oy) * w + ox) * C + c;
        *reinterpret_cast<float4*>(y + o) = make_float4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<unsigned*>(codes + o) = code;
    }
}

__global__ void __launch_bounds__(256)
maxpool_bwd_codes_kernel(const unsigned char* __restrict__ codes, const float* __restrict__ gy, float* __restrict__ gx,
                         int B, int H, int W, int C, int act_mode, int accumulate, unsigned* __restrict__ amax) {
    float amx = 0.f;                                                 // max |stored value| (the fp16x2 family's scale of gx: csrc/h2.h)
    const int h = H / 2, w = W / 2, cq = C / 4;
    const int64_t total = (int64_t)B * h * w * cq;
    const float slope = act_mode == 1 ? 0.2f : (act_mode == 2 ? 0.f : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cq) * 4;
        int64_t p = i / cq;
        const int ox = (int)(p % w); p /= w;
        const int oy = (int)(p % h);
        const int64_t b = p / h;
        const int64_t base = ((b * H + 2 * oy) * W + 2 * ox) * C + c, po = ((b * h + oy) * w + ox) * C + c;
        const int64_t off[4] = {0, (int64_t)C, (int64_t)W * C, (int64_t)W * C + C};
        const unsigned code = *reinterpret_cast<const unsigned*>(codes + po);
        const float4 g4 = *reinterpret_cast<const float4*>(gy + po);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        float o[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned cj = (code >> (8 * j)) & 0xffu, arg = cj & 3u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = ((cj >> (2 + k)) & 1u) ? 1.f : slope;
                o[k][j] = ((unsigned)k == arg) ? g[j] * d : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float4* d = reinterpret_cast<float4*>(gx + base + off[k]);
            float4 v = make_float4(o[k][0], o[k][1], o[k][2], o[k][3]);
            if (accumulate) { const float4 t = *d; v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
            *d = v;
            amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    }
    if (amax) pnnp_amax_commit_block(amx, amax);
}

// per-channel sum over pixels of an NHWC tensor: partial[blockIdx][C] then a fixed-order finish
__global__ void __launch_bounds__(256)
channel_sum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, int64_t npix, int C) {
    // thread t owns the channel quad (t % (C/4)) of every (256/(C/4))-th pixel: float4 loads,
    // a wave reads whole pixels (fully coalesced); C/4 must divide 256 (C = 4 .. 1024, power of 2)
    const int cq = C >> 2;
    const int q = threadIdx.x % cq, lane_pix = threadIdx.x / cq, pix_per_iter = 256 / cq;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t p = (int64_t)blockIdx.x * pix_per_iter + lane_pix; p < npix; p += (int64_t)gridDim.x * pix_per_iter) {
        const float4 v = *reinterpret_cast<const float4*>(x + p * C + 4 * q);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    __shared__ float4 red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < cq) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < pix_per_iter; ++k) {
            const float4 v = red[k * cq + threadIdx.x];
            t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
        }
        *reinterpret_cast<float4*>(partial + (int64_t)blockIdx.x * C + 4 * threadIdx.x) = t;
    }
}

__global__ void __launch_bounds__(256)
rows_sum_kernel(const float* __restrict__ partial, float* __restrict__ out, int rows, int C, int accumulate) {
    // 32 channels x 8 row-phases per block (fixed summation order)
    __shared__ float red[8][32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    float s = 0.f;
    if (c < C)
        for (int r = ty; r < rows; r += 8) s += partial[(int64_t)r * C + c];
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][tx];
        out[c] = accumulate ? out[c] + t : t;
    }
}

// L1 on clamp(pred,0,1) vs hr, NCHW in; gradient written NHWC with Cp channels (padding zero).
// partial[block] = {sum |d|, then per-crop SSE is accumulated in sse_partial[block]} ; a block
// never straddles two crops (grid = B x blocks_per_crop).
__global__ void __launch_bounds__(256)
l1_clamp_kernel(const float* __restrict__ pred, const float* __restrict__ hr, const float* __restrict__ scale,
                float* __restrict__ grad_nhwc, float* __restrict__ partial, int C, int64_t hw, int Cp, float inv_n, int blocks_per_crop,
                int clamp_target, float grad_weight) {
    const int b = blockIdx.x / blocks_per_crop, blk = blockIdx.x % blocks_per_crop;
    const float sc = scale ? scale[b] : 1.f;            // `ori`: pred * ratio before the loss (trainer_SID.py:97-98)
    const float gsc = inv_n * sc * grad_weight;     // grad_weight: this rank's share of a global batch (uneven data-parallel shards), 1 otherwise
    float l1 = 0.f, sse = 0.f;
    for (int64_t s = (int64_t)blk * 256 + threadIdx.x; s < hw; s += (int64_t)blocks_per_crop * 256) {
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = 0.f;
        for (int c = 0; c < C; ++c) {
            const int64_t i = ((int64_t)b * C + c) * hw + s;
            const float p = pred[i] * sc, t0 = hr[i];
            const float t = clamp_target ? pnnp_clampf(t0, 0.f, 1.f) : t0;      // preprocess: imgs_hr.clamp(0, 1) when dst.clip (trainer_SID.py:485)
            const float pc = pnnp_clampf(p, 0.f, 1.f);             // (a NaN prediction gives a NaN loss, as pred.clamp(0, 1) does: trainer_SID.py:99)
            const float d = pc - t;
            l1 += fabsf(d);
            const float tc = pnnp_clampf(t, 0.f, 1.f);            // PSNR uses clamped hr (trainer_SID.py:112-114)
            sse += (pc - tc) * (pc - tc);
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            if (c < 8) g[c] = (p >= 0.f && p <= 1.f) ? sgn * gsc : 0.f;     // clamp passes grad on [0,1]
        }
        if (grad_nhwc) {
            float* d = grad_nhwc + ((int64_t)b * hw + s) * Cp;
            *reinterpret_cast<float4*>(d) = make_float4(g[0], g[1], g[2], g[3]);
            if (Cp >= 8) *reinterpret_cast<float4*>(d + 4) = make_float4(g[4], g[5], g[6], g[7]);
        }
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = l1; r2[threadIdx.x] = sse;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) { r1[threadIdx.x] += r1[threadIdx.x + k]; r2[threadIdx.x] += r2[threadIdx.x + k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = r1[0]; partial[2 * blockIdx.x + 1] = r2[0]; }
}

// out[0] = mean L1 over the batch; out[1 + b] = SSE of crop b
__global__ void __launch_bounds__(64)
l1_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int B, int blocks_per_crop, float inv_n) {
    const int lane = threadIdx.x;               // one wave; fixed summation order (lane-strided, then a shuffle tree)
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        float l = 0.f, s = 0.f;
        for (int k = lane; k < blocks_per_crop; k += 64) { l += partial[2 * (b * blocks_per_crop + k)]; s += partial[2 * (b * blocks_per_crop + k) + 1]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { l += __shfl_xor(l, o); s += __shfl_xor(s, o); }
        tot += l;
        if (lane == 0) out[1 + b] = s;
    }
    if (lane == 0) out[0] = tot * inv_n;
}

// torch.optim.Adam (defaults, no weight decay / amsgrad), one flat launch over all parameters.
// grad_scale multiplies g first (1/world_size after a sum all-reduce).
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float grad_scale) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<const float4*>(g)[i];
        float4 M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        float* pp = &P.x; float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * grad_scale;
            mm[k] = mm[k] + (gk - mm[k]) * (1.f - b1);                // exp_avg.lerp_(grad, 1-beta1)
            vv[k] = vv[k] * b2 + (1.f - b2) * gk * gk;               // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] = pp[k] - (lr / bc1) * (mm[k] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(m)[i] = M; reinterpret_cast<float4*>(v)[i] = V;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gk = g[i] * grad_scale;
        m[i] = m[i] + (gk - m[i]) * (1.f - b1);
        v[i] = v[i] * b2 + (1.f - b2) * gk * gk;
        p[i] = p[i] - (lr / bc1) * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
    }
}

}  // namespace

extern "C" {

int pnnp_nchw_to_nhwc_reflect_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, int pad, void* stream) {
    return pnnp_nchw_to_nhwc_reflect_amax_f32(src, dst, B, C, H, W, Cp, pad, nullptr, stream);
}
// ... with max |element| raised into an amax slot of the fp16x2 family
int pnnp_nchw_to_nhwc_reflect_amax_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, int pad, unsigned* amax, void* stream) {
    if (!src || !dst || B < 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || (Cp & 3) || pad < 0 || pad >= H || pad >= W) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid1d((int64_t)B * (H + 2 * pad) * (W + 2 * pad) * (Cp / 4), (amax && (int64_t)B * (H + 2 * pad) * (W + 2 * pad) * (Cp / 4) < (1 << 22)) ? 256 : 256 * 8)), dim3(256), 0, as_stream(stream),
                       src, dst, B, C, H, W, Cp, pad, amax);
    return pnnp_launch_status();
}

int pnnp_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
    return pnnp_nchw_to_nhwc_reflect_f32(src, dst, B, C, H, W, Cp, 0, stream);
}

int pnnp_nhwc_to_nchw_f32(const float* src, const float* residual, float* dst, int B, int C, int H, int W, int Cp, void* stream) {
    if (!src || !dst || B < 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid1d((int64_t)B * C * H * W)), dim3(256), 0, as_stream(stream), src, residual, dst, B, C, H, W, Cp);
    return pnnp_launch_status();
}

int pnnp_maxpool2_fwd_f32(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if (!x || !y || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 3)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid1d((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, as_stream(stream), x, y, B, H, W, C);
    return pnnp_launch_status();
}

int pnnp_maxpool2_bwd_f32(const float* x, const float* gy, float* gx, int B, int H, int W, int C, int act_mode,
                          int accumulate, void* stream) {
    if (!x || !gy || !gx || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 3)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid1d((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, as_stream(stream),
                       x, gy, gx, B, H, W, C, act_mode, accumulate);
    return pnnp_launch_status();
}

// MaxPool2d(2) forward that also writes the one-byte codes [B][H/2][W/2][C] for pnnp_maxpool2_bwd_codes_f32.
int pnnp_maxpool2_fwd_codes_f32(const float* x, float* y, unsigned char* codes, int B, int H, int W, int C, void* stream) {
    if (!x || !y || !codes || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 3)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(maxpool_fwd_codes_kernel, dim3(grid1d((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, as_stream(stream), x, y, codes, B, H, W, C);
    return pnnp_launch_status();
}

// Backward of the pool (+ the activation in front of it) from the codes: same result as pnnp_maxpool2_bwd_f32 without reading x.
int pnnp_maxpool2_bwd_codes_amax_f32(const unsigned char* codes, const float* gy, float* gx, int B, int H, int W, int C, int act_mode,
                                     int accumulate, unsigned* amax_gx, void* stream) {
    if (!codes || !gy || !gx || B < 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || (C & 3)) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    hipLaunchKernelGGL(maxpool_bwd_codes_kernel, dim3(grid1d((int64_t)B * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0, as_stream(stream),
                       codes, gy, gx, B, H, W, C, act_mode, accumulate, amax_gx);
    return pnnp_launch_status();
}
int pnnp_maxpool2_bwd_codes_f32(const unsigned char* codes, const float* gy, float* gx, int B, int H, int W, int C, int act_mode,
                                int accumulate, void* stream) {
    return pnnp_maxpool2_bwd_codes_amax_f32(codes, gy, gx, B, H, W, C, act_mode, accumulate, nullptr, stream);
}

// out[c] (+)= sum over pixels of x[pix][c];  workspace >= 1024*C floats
int pnnp_channel_sum_f32(const float* x, float* out, int64_t npix, int C, int accumulate, float* workspace, void* stream) {
    if (!x || !out || !workspace || npix <= 0 || C < 4 || (C & 3) || C > 1024 || (256 % (C / 4))) return PNNP_E_INVALID;
    const int blocks = 1024;       // workspace holds blocks x C partial sums (deterministic two-stage sum)
    hipLaunchKernelGGL(channel_sum_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, workspace, npix, C);
    hipLaunchKernelGGL(rows_sum_kernel, dim3((C + 31) / 32), dim3(256), 0, as_stream(stream), workspace, out, blocks, C, accumulate);
    return pnnp_launch_status();
}

// loss_out[0] = mean |clamp(pred,0,1) - hr| (trainer_SID.py:99); loss_out[1+b] = sum_b (clamp(pred)-clamp(hr))^2
// (PSNR_b = -10 log10(SSE_b / (C*H*W)), losses/__init__.py:4-15).  grad_nhwc (optional): dL/dpred laid out
// [B][H][W][Cp] for the backward pass.  workspace >= 2 * B * 64 floats.
int pnnp_l1_clamp_loss_w_f32(const float* pred, const float* hr, const float* scale, float* grad_nhwc, float* loss_out,
                             int B, int C, int H, int W, int Cp, float* workspace, int clamp_target, float grad_weight, void* stream);

int pnnp_l1_clamp_loss_scaled_f32(const float* pred, const float* hr, const float* scale, float* grad_nhwc, float* loss_out,
                                  int B, int C, int H, int W, int Cp, float* workspace, void* stream) {
    return pnnp_l1_clamp_loss_tc_f32(pred, hr, scale, grad_nhwc, loss_out, B, C, H, W, Cp, workspace, 0, stream);
}

// the same with the target clamped to [0,1] inside the kernel (the trainer's `imgs_hr.clamp(0, 1)` of preprocess, trainer_SID.py:485,
// without a separate elementwise pass over the clean crops)
int pnnp_l1_clamp_loss_tc_f32(const float* pred, const float* hr, const float* scale, float* grad_nhwc, float* loss_out,
                              int B, int C, int H, int W, int Cp, float* workspace, int clamp_target, void* stream) {
    return pnnp_l1_clamp_loss_w_f32(pred, hr, scale, grad_nhwc, loss_out, B, C, H, W, Cp, workspace, clamp_target, 1.0f, stream);
}

// ... and dL/dpred multiplied by grad_weight (a rank's mean-gradient weight B_local world / B_global when a global batch is split
// unevenly over data-parallel ranks; the loss and the SSE are not weighted): no elementwise pass over the gradient map
int pnnp_l1_clamp_loss_w_f32(const float* pred, const float* hr, const float* scale, float* grad_nhwc, float* loss_out,
                             int B, int C, int H, int W, int Cp, float* workspace, int clamp_target, float grad_weight, void* stream) {
    if (!pred || !hr || !loss_out || !workspace || B <= 0 || C <= 0 || C > 8 || (grad_nhwc && (Cp < C || (Cp != 4 && Cp != 8)))) return PNNP_E_INVALID;
    const int bpc = 64;
    const float inv_n = 1.0f / ((float)B * C * H * W);
    hipLaunchKernelGGL(l1_clamp_kernel, dim3(B * bpc), dim3(256), 0, as_stream(stream), pred, hr, scale, grad_nhwc, workspace, C,
                       (int64_t)H * W, Cp, inv_n, bpc, clamp_target, grad_weight);
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(64), 0, as_stream(stream), workspace, loss_out, B, bpc, inv_n);
    return pnnp_launch_status();
}

int pnnp_l1_clamp_loss_f32(const float* pred, const float* hr, float* grad_nhwc, float* loss_out, int B, int C, int H, int W,
                           int Cp, float* workspace, void* stream) {
    return pnnp_l1_clamp_loss_scaled_f32(pred, hr, nullptr, grad_nhwc, loss_out, B, C, H, W, Cp, workspace, stream);
}

int pnnp_adam_step_f32(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                       int step, float grad_scale, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return PNNP_E_INVALID;
    if ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return PNNP_E_INVALID;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));         // python-float bias corrections
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n >> 2)), dim3(256), 0, as_stream(stream), p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
    return pnnp_launch_status();
}

}  // extern "C"
