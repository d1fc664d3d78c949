// Physics-based raw noise sampler: one fused HBM-streaming kernel per batch of crops.
//   generate_noisy_obs   data_process/process.py:591-631   (mode OBS)
//   generate_noisy_torch data_process/process.py:634-673   (mode TORCH)
//
//   z = clip((shot + read + row + quant + bias) / (wp-bl), lo, 1) [* ratio]
//   shot  = Poisson(mfm*y'/K) * K / mfm        y' = y*(wp-bl)/ratio         ('p')
//         = y' + N(0,1)*sqrt(max(y'/K,1e-10))*K/mfm                         (OBS without 'p')
//   read  = N(0,1)*sigGs/mfm   or Tukey-lambda(lam)*sigTL/mfm ('g', OBS only)
//   row   = N(0,1)*sigR/mfm, ONE draw per (packed channel, row), broadcast along W
//   quant = U(-.5,.5) DN (OBS)  or  (U(0,1)-.5)*q*(wp-bl) (TORCH)
//
// RNG: Philox4x32-10, key = (seed_lo, seed_hi ^ offset_hi),
//      counter = (element-in-crop, crop_base+b, slot, offset_lo).  Slot 0 feeds the first
//      Poisson attempt (lanes x,y) and the read-noise Box-Muller pair (lanes z,w); slot 1
//      the quantisation uniform; slots 2.. further PTRS rejection rounds; slot 0x40000000
//      with element = c*H+h is the row draw.  The sample therefore depends only on
//      (seed, offset, global crop index, element) -- not on batch size, grid or GPU count.
// Poisson: lam < 10 sequential inversion; lam >= 10 Hoermann's PTRS transformed rejection
// (exact, no Gaussian approximation).  Specification: oracle/pnnp_oracle.c.
//
// Thread layout: one thread owns 4 consecutive pixels of a row (float4 load/store, fully
// coalesced); the row-noise normal is drawn once by the first lane of each run of lanes that
// share a row and broadcast inside the wavefront (ballot + shuffle), never per pixel.
#include "common.h"

namespace {

struct Philox {
    uint32_t k0, k1;
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    // one 64-bit product per multiplier and round (v_mad_u64_u32 runs at the full VALU rate on gfx950: tools/ubench/intmul_rate.hip;
    // written as __umulhi + * the compiler emitted v_mul_hi_u32 and v_mul_lo_u32 separately: 40 multiplies per call instead of 20)
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        c0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0; c1 = (uint32_t)p1; c2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1; c3 = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// Hardware transcendentals (v_log_f32, v_exp_f32, v_sqrt_f32, v_rcp_f32: ~1 ulp) where the specification (oracle/pnnp_oracle.c, libm)
// leaves room: a sample may differ from the oracle's by ~2e-7 relative, a Poisson count only when a uniform falls within ~1e-6 of a
// CDF / acceptance boundary (tier A of the tests: >= 99.9 % of pixels equal to 1e-5).  cos() stays libm-accurate: its absolute
// error is scaled by sigma * ratio into the output.  libm's logf / expf / sqrtf / IEEE division were ~60 % of the kernel's instructions.
__device__ __forceinline__ float fast_log(float x) { return __logf(x); }
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// 23 random bits + 1/2 ulp: exact in fp32, in [2^-24, 1-2^-24]
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

__device__ __forceinline__ float box_muller(uint32_t a, uint32_t b) {
    const float u1 = u01(a), u2 = u01(b);
    return fast_sqrt(-2.0f * fast_log(u1)) * cosf(6.28318530717958647692f * u2);
}

struct Ctx {
    uint32_t k0, k1, crop, off;
};

// ln(k!) for integral k >= 0: table below 10, Stirling series above (truncation < 5e-9 at k = 10; libm's lgammaf is several hundred
// instructions behind branches and sat in the rejection path of nearly every wave).  Same operations as oracle/pnnp_oracle.c.
__device__ __forceinline__ float log_factorial(float k) {
    if (k < 10.f) {
        const float LF[10] = {0.f, 0.f, 0.69314718f, 1.79175947f, 3.17805383f, 4.78749174f, 6.57925121f, 8.52516136f, 10.60460290f, 12.80182748f};
        float v = 0.f;
#pragma unroll
        for (int i = 2; i < 10; ++i) v = (k == (float)i) ? LF[i] : v;
        return v;
    }
    const float x = k + 1.f;
    const float r = __builtin_amdgcn_rcpf(x);
    return (x - 0.5f) * fast_log(x) - x + 0.91893853f + 0.083333333f * r - 0.0027777778f * (r * r * r);
}

__device__ float poisson_f32(float lam, uint32_t elem, const Ctx& c, uint32_t r0, uint32_t r1) {
    if (!(lam > 0.f)) return 0.f;
    if (lam < 10.f) {
        const float u = u01(r0);
        float p = fast_exp(-lam), s = p, k = 0.f;
        // the rounded CDF can saturate below the largest uniform (1 - 2^-24): stop when a term no longer moves the sum
        while (u > s) { k += 1.f; p *= fast_div(lam, k); const float s2 = s + p; if (s2 == s) break; s = s2; }
        return k;
    }
    const float slam = fast_sqrt(lam), loglam = fast_log(lam);
    const float b = 0.931f + 2.53f * slam;
    const float a = -0.059f + 0.02483f * b;
    const float inv_alpha = 1.1239f + fast_div(1.1328f, b - 3.4f);
    const float vr = 0.9277f - fast_div(3.6224f, b - 2.f);
    uint32_t x0 = r0, x1 = r1;
    uint4 extra = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < 64; ++it) {
        if (it > 0) {
            if (it & 1) { extra = philox4x32_10(elem, c.crop, 1u + ((it + 1) >> 1), c.off, c.k0, c.k1); x0 = extra.x; x1 = extra.y; }
            else { x0 = extra.z; x1 = extra.w; }
        }
        const float U = u01(x0) - 0.5f, V = u01(x1);
        const float us = 0.5f - fabsf(U);
        const float k = floorf((fast_div(2.f * a, us) + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return k;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (fast_log(V) + fast_log(inv_alpha) - fast_log(fast_div(a, us * us) + b) <= -lam + k * loglam - log_factorial(k)) return k;
    }
    return floorf(lam + 0.5f);
}

__device__ __forceinline__ float tukey_lambda(float u, float lam) {
    // (u^lam - (1-u)^lam)/lam, evaluated through expm1 so the cancellation for small |lam|
    // (the calibrated cameras have |lam| < 0.3) does not amplify rounding
    const float lu = logf(u), l1u = log1pf(-u);
    if (lam == 0.f) return lu - l1u;
    return (expm1f(lam * lu) - expm1f(lam * l1u)) / lam;
}

__global__ void __launch_bounds__(256)
noise_sample_kernel(const float* __restrict__ y, float* __restrict__ out, int B, int C, int H, int W,
                    const float* __restrict__ params, unsigned flags, float mfm,
                    uint32_t k0, uint32_t k1, uint32_t off, uint32_t crop_base) {
    const int wq = (W + 3) >> 2;
    const int64_t rows = (int64_t)B * C * H;
    const int64_t total = rows * wq;
    const bool torch_mode = flags & PNNP_NOISE_MODE_TORCH;
    const bool use_p = flags & PNNP_NOISE_P, use_g = flags & PNNP_NOISE_G, use_r = flags & PNNP_NOISE_R;
    const bool use_q = flags & PNNP_NOISE_Q, use_d = flags & PNNP_NOISE_D, use_b = flags & PNNP_NOISE_B;
    const bool extras = torch_mode || !use_b;        // OBS: 'b' removes read, row, quant and bias
    const int lane = threadIdx.x & 63;
    // grid-stride with the SAME trip count for all lanes of a wave (shuffles below)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t iters = (total + stride - 1) / stride;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t t = t0 + it * stride;
        const bool live = t < total;
        const int64_t tt = live ? t : total - 1;
        const int xq = (int)(tt % wq);
        const int64_t row = tt / wq;                     // (b*C + c)*H + h
        const int h = (int)(row % H);
        const int c = (int)((row / H) % C);
        const int b = (int)(row / ((int64_t)H * C));
        const float* P = params + (int64_t)b * PNNP_NPARAM;
        const float K = P[PNNP_P_K], ratio = P[PNNP_P_RATIO], wp = P[PNNP_P_WP], bl = P[PNNP_P_BL];
        const float span = wp - bl;
        Ctx ctx{k0, k1, crop_base + (uint32_t)b, off};

        // ---- row noise: one draw per run of lanes sharing `row`, broadcast in-wave
        float row_noise = 0.f;
        if (use_r && extras) {
            const int64_t prev = __shfl_up(row, 1);
            const bool leader = (lane == 0) || (prev != row);
            float n = 0.f;
            if (leader) {
                const uint4 r = philox4x32_10((uint32_t)(c * H + h), ctx.crop, 0x40000000u, off, k0, k1);
                n = box_muller(r.x, r.y);
            }
            const unsigned long long leaders = __ballot(leader);
            const unsigned long long upto = leaders & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            const int src = 63 - __clzll(upto);
            n = __shfl(n, src);
            row_noise = __fdiv_rn(__fmul_rn(n, P[PNNP_P_SIGR]), mfm);
        }
        if (!live) continue;

        const int nx = min(4, W - 4 * xq);
        const int64_t base = row * W + 4 * xq;
        float v[4];
        if (nx == 4 && ((((uintptr_t)(y + base)) & 15) == 0)) {
            const float4 q = *reinterpret_cast<const float4*>(y + base);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = i < nx ? y[base + i] : 0.f;
        }
        const float sig_read = __fdiv_rn(use_g ? P[PNNP_P_SIGTL] : P[PNNP_P_SIGGS], mfm);
        const float qscale = torch_mode ? __fmul_rn(P[PNNP_P_Q], span) : 1.0f;
        const float bias = (use_d && extras) ? P[PNNP_P_BIAS0 + (c & 3)] : 0.f;
        const float lo = (flags & PNNP_NOISE_CLIP) ? 0.f : -__fdiv_rn(bl, wp);
        const uint32_t e0 = (uint32_t)((c * H + h) * (int64_t)W + 4 * xq);
        float o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t elem = e0 + i;
            const uint4 r = philox4x32_10(elem, ctx.crop, 0u, off, k0, k1);
            float yy = __fmul_rn(v[i], span);
            yy = fast_div(yy, ratio);
            float shot;
            if (use_p) {
                const float lam = fast_div(__fmul_rn(mfm, yy), K);
                shot = fast_div(__fmul_rn(poisson_f32(lam, elem, ctx, r.x, r.y), K), mfm);
            } else {
                const float n = box_muller(r.x, r.y);
                const float s = fast_sqrt(fmaxf(fast_div(yy, K), 1e-10f));
                shot = __fadd_rn(yy, fast_div(__fmul_rn(__fmul_rn(n, s), K), mfm));
            }
            float acc = shot;
            if (!use_b) {
                const float rd = use_g ? tukey_lambda(u01(r.z), P[PNNP_P_LAM]) : box_muller(r.z, r.w);
                acc = __fadd_rn(acc, __fmul_rn(rd, sig_read));
            }
            if (extras) {
                if (use_r) acc = __fadd_rn(acc, row_noise);
                if (use_q) {
                    const uint4 rq = philox4x32_10(elem, ctx.crop, 1u, off, k0, k1);
                    acc = __fadd_rn(acc, __fmul_rn(u01(rq.x) - 0.5f, qscale));
                }
                if (use_d) acc = __fadd_rn(acc, bias);
            }
            float z = fast_div(acc, span);
            z = fminf(fmaxf(z, lo), 1.f);
            if (!(flags & PNNP_NOISE_ORI)) z = __fmul_rn(z, ratio);
            // the trainer's clamp of the noisy input (trainer_SID.py:481-485), fused into the store
            if (flags & PNNP_NOISE_POST_MIN0) z = fmaxf(z, 0.f);
            if (flags & PNNP_NOISE_POST_MAX1) z = fminf(z, 1.f);
            o[i] = z;
        }
        if (nx == 4 && ((((uintptr_t)(out + base)) & 15) == 0)) {
            *reinterpret_cast<float4*>(out + base) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
            for (int i = 0; i < nx; ++i) out[base + i] = o[i];
        }
    }
}

}  // namespace

// SNA_torch (data_process/process.py:562-588): shot-noise augmentation of a paired crop under a white-balance gain
// change.  Per pixel of plane c:  g = gt*(wp-bl)/ratio;  dy = g*aug[c];  dn = Poisson(dy/K)*K;  if black_lr dy -= g;
// dy = dy*ratio/(wp-bl);  dn = dn/(wp-bl);  if !ori dn *= ratio.   gt, dn, dy: [C][H][W] (C = 4 planes R,G1,B,G2).
__global__ void __launch_bounds__(256)
sna_kernel(const float* __restrict__ gt, float* __restrict__ dn, float* __restrict__ dy, int C, int64_t plane,
           float a0, float a1, float a2, float a3, float K, float span, float ratio, int black_lr, int ori,
           uint32_t k0, uint32_t k1, uint32_t off, uint32_t crop) {
    const int64_t total = (int64_t)C * plane;
    const Ctx ctx{k0, k1, crop, off};
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t / plane) & 3;
        const float aug = c == 0 ? a0 : (c == 1 ? a1 : (c == 2 ? a2 : a3));
        const uint32_t elem = (uint32_t)t;
        const uint4 r = philox4x32_10(elem, crop, 0x20000000u, off, k0, k1);
        const float g = __fdiv_rn(__fmul_rn(gt[t], span), ratio);
        float y = __fmul_rn(g, aug);
        float n = __fmul_rn(poisson_f32(__fdiv_rn(y, K), elem, ctx, r.x, r.y), K);
        if (black_lr) y = __fsub_rn(y, g);
        y = __fdiv_rn(__fmul_rn(y, ratio), span);
        n = __fdiv_rn(n, span);
        if (!ori) n = __fmul_rn(n, ratio);
        dn[t] = n; dy[t] = y;
    }
}

// HighBitRecovery.map (data_process/process.py:718-751): re-draw every integer-valued pixel inside its quantisation
// bin according to the read-noise distribution:  x = round(d);  if low <= x < high:  d' = ppf(cdf[x] + u * range[x]) + (d - x).
// cdf/range: per-integer LUT built on the host with scipy (HB2LB_LUT, :697-716); dist 0 = normal(loc, scale),
// 1 = Tukey-lambda(lam, loc, scale).  The quantile is evaluated in float64 (the bins 6 sigma out have ranges ~1e-9).
__global__ void __launch_bounds__(256)
hbr_map_kernel(const float* __restrict__ data, float* __restrict__ out, int64_t n, const double* __restrict__ cdf,
               const double* __restrict__ range, int low, int high, int dist, double loc, double scale, double lam,
               const double* __restrict__ rand, float in_mul, float out_div, float out_add, int keep_delta,
               uint32_t k0, uint32_t k1, uint32_t off) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float df = __fmul_rn(data[i], in_mul);
        float x = rintf(df);                                   // np.round: half to even
        const float delta = keep_delta ? __fsub_rn(df, x) : 0.f;
        const int xi = (int)x;
        if (xi >= low && xi < high) {
            double u;
            if (rand) u = rand[i];
            else {
                const uint4 r = philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0x48425221u, off, k0, k1);
                u = ((double)r.x * 4294967296.0 + (double)r.y + 0.5) * (1.0 / 18446744073709551616.0);
            }
            const double p = cdf[xi - low] + u * range[xi - low];
            double q;
            if (dist == 0) q = -1.4142135623730951 * erfcinv(2.0 * p);                      // norm.ppf
            else q = (lam == 0.0) ? log(p / (1.0 - p)) : (pow(p, lam) - pow(1.0 - p, lam)) / lam;   // tukeylambda.ppf
            x = (float)(loc + scale * q);
        }
        float v = __fadd_rn(x, delta);
        v = out_div != 0.f ? __fdiv_rn(v, out_div) : __fadd_rn(v, out_add);
        out[i] = v;
    }
}

extern "C" int pnnp_hbr_map_f32(const float* data, float* out, int64_t n, const double* cdf, const double* range, int low, int high,
                                int dist, double loc, double scale, double lam, const double* rand, float in_mul, float out_div,
                                float out_add, int keep_delta, uint64_t seed, uint64_t offset, void* stream) {
    if (n < 0 || (n && (!data || !out)) || !cdf || !range || high < low || !(scale > 0.0)) return PNNP_E_INVALID;
    if (n == 0) return PNNP_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(hbr_map_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), data, out, n, cdf, range, low, high, dist,
                       loc, scale, lam, rand, in_mul, out_div, out_add, keep_delta, (uint32_t)seed,
                       (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32), (uint32_t)offset);
    return pnnp_launch_status();
}

extern "C" int pnnp_sna_f32(const float* gt, float* dn, float* dy, int C, int H, int W, const float* aug_wb4 /* host */,
                            float K, float wp, float bl, float ratio, int black_lr, int ori, uint64_t seed, uint64_t offset,
                            uint32_t crop, void* stream) {
    if (!gt || !dn || !dy || !aug_wb4 || C < 0 || H < 0 || W < 0 || !(K > 0.f) || !(ratio > 0.f)) return PNNP_E_INVALID;
    const int64_t plane = (int64_t)H * W, total = (int64_t)C * plane;
    if (total == 0) return PNNP_OK;
    if (total >= (1ll << 32)) return PNNP_E_INVALID;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
    hipLaunchKernelGGL(sna_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), gt, dn, dy, C, plane,
                       aug_wb4[0], aug_wb4[1], aug_wb4[2], aug_wb4[3], K, wp - bl, ratio, black_lr, ori, k0, k1, (uint32_t)offset, crop);
    return pnnp_launch_status();
}

extern "C" int pnnp_noise_sample_f32(const float* y, float* out, int B, int C, int H, int W,
                                     const float* params, unsigned flags, float mfm, uint64_t seed,
                                     uint64_t offset, uint32_t crop_base, void* stream) {
    if (!y || !out || !params || B < 0 || C < 0 || H < 0 || W < 0 || !(mfm > 0.f)) return PNNP_E_INVALID;
    if ((flags & PNNP_NOISE_MODE_TORCH) && (flags & PNNP_NOISE_G) && !(flags & PNNP_NOISE_TORCH_TUKEY)) return PNNP_E_UNSUPPORTED;   // process.py:654
    if ((flags & PNNP_NOISE_MODE_TORCH) && !(flags & PNNP_NOISE_P)) return PNNP_E_UNSUPPORTED;  // process.py:651
    if ((int64_t)C * H * W >= (1ll << 32)) return PNNP_E_INVALID;
    const int64_t total = (int64_t)B * C * H * ((W + 3) / 4);
    if (total == 0) return PNNP_OK;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
    hipLaunchKernelGGL(noise_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                       y, out, B, C, H, W, params, flags, mfm, k0, k1, (uint32_t)offset, crop_base);
    return pnnp_launch_status();
}
