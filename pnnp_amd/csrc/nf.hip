// NoiseFlow.sample on the device (reference: archs/noise_flow.py:173-188 and the bijector
// inverses it chains: flow_layers/affine_coupling.py:27-34,245-295, conv2d1x1.py:47-92,
// gain.py:79-93, signal_dependant.py:37-57).
//
// The reversed chain is 8 x [AffineCoupling^-1, Conv2d1x1^-1] with the ISO gain after the 4th pair
// and the signal-dependent scale at the end.  One kernel = one pair (+ the optional final scale):
// an HBM-streaming pass (read 4 planes, write 4 planes) whose coupling network
//   conv3x3(2->4)+BN+ReLU -> conv1x1(4->4)+BN+ReLU -> [pad + border-ones channel] -> conv3x3(5->4)
// is evaluated from an LDS tile (z0 with a 2-pixel halo, hidden map with a 1-pixel halo); the scalar
// ISO gain is folded into the 4x4 inverse matrix on the host.  NCHW fp32, like the reference.
#include "common.h"
#include <string.h>
#include <type_traits>

struct NfStep {              // 317 floats, passed by value as a kernel argument
    float w1[4][2][9], b1[4], s1[4], o1[4];   // conv2d_1, eval-mode BatchNorm folded to y = s*x + o
    float w2[4][4], b2[4], s2[4], o2[4];      // conv2d_2 (1x1) + BatchNorm
    float w3[4][5][9], b3[4], e3[4];          // conv2d_3 (input channel 4 = border-ones), exp(3*logs)
    float scale;                              // log_scale = scale * tanh(.)
    float winv[4][4];                         // the Conv2d1x1 inverse that follows (x gain where it applies)
};

// Optional fusion of the trainer's preprocess around NoiseFlow.sample (trainer_SID.py:463-472,481-485; trainer_LRID.py:419-427):
//   clean = imgs_hr / ratio                      -> clean_div: the kernel divides the clean crop it reads for the signal-dependent scale
//   imgs_lr = imgs_hr + sample(...) * ratio       -> mix_base = imgs_hr, mix_mul / mix_mul_s = ratio (per crop / one scalar)
//   imgs_lr.clamp(lb, 1)                         -> lo, hi
//   assert scale >= 0 (signal_dependant.py:50)   -> flag: bit 0 set on the device when a * clean + b < 0 anywhere (read it later)
struct NfMix {
    const float* clean_div;   // [B] or null
    float clean_div_s;        // used when clean_div is null (1 = none)
    const float* mix_base;    // [B][4][H][W] or null: y = clamp(mix_base + (sample * mix_mul), lo, hi)
    const float* mix_mul;     // [B] or null
    float mix_mul_s;
    float lo, hi;
    int* flag;                // or null
    // training-mode BatchNorm (trainer_LRID.py:34-39 never calls .eval() on the proxy it samples from): device [24] = the batch
    // statistics [mean1, rstd1, var1, mean2, rstd2, var2] x 4 of pnnp_nf_train_stats_f32 (bias-free means), or null.  With it the
    // step's s1 / o1 / s2 / o2 slots hold the BatchNorm weight / bias and scale, offset are formed here:
    //   scale = gamma * rstd,  offset = beta - (mean + conv bias) * scale        (no host round trip for the statistics)
    const float* bn_stats;
};

namespace {

constexpr int TS = 32;                        // output tile
constexpr int ZW = TS + 4, HW_ = TS + 2;      // z0 tile with halo 2, hidden tile with halo 1

__device__ __forceinline__ uint4 philox_nf(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float u01_nf(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

// base + r * m with TWO roundings, like the tensor ops `noise * ratio` and `imgs_lr += noise` (HIP's __fmul_rn / __fadd_rn are plain
// operators: without the pragma the compiler contracts them into one fma)
__device__ __forceinline__ float mul_then_add(float base, float r, float m) {
#pragma clang fp contract(off)
    const float t = r * m;
    return base + t;
}

// z ~ N(0,1): 4 values per Philox block (two Box-Muller pairs, both outputs used)
__global__ void __launch_bounds__(256)
normal_fill_kernel(float* __restrict__ out, int64_t n, uint32_t k0, uint32_t k1, uint32_t off) {
    const int64_t n4 = (n + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 r = philox_nf((uint32_t)i, (uint32_t)(i >> 32), 0x4E46u, off, k0, k1);
        const float ra = sqrtf(-2.f * logf(u01_nf(r.x))), rb = sqrtf(-2.f * logf(u01_nf(r.z)));
        const float ta = 6.28318530717958647692f * u01_nf(r.y), tb = 6.28318530717958647692f * u01_nf(r.w);
        const float v[4] = {ra * cosf(ta), ra * sinf(ta), rb * cosf(tb), rb * sinf(tb)};
        for (int k = 0; k < 4; ++k)
            if (4 * i + k < n) out[4 * i + k] = v[k];
    }
}

// The launcher's re-arrangement of NfStep for nf_step_kernel: OUTPUT channel innermost, so that the weights of two output channels are an
// aligned register pair and one v_pk_fma_f32 advances both (the vector fp32 peak is the packed rate; the plain v_fma is half).
struct NfStepDev {
    float w1[2][9][4], b1[4], s1[4], o1[4];
    float w2[4][4], b2[4], s2[4], o2[4];      // w2[c][o]
    float w3[5][9][4], b3[4], e3[4];
    float winv[4][4];                         // winv[c][o]
    float scale;                              // (last: every group of four above is 16-byte aligned)
};
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 ld2(const float* q) { return f2{q[0], q[1]}; }
__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
// tanh and exp on the hardware transcendentals (v_exp_f32, v_rcp_f32: ~1 ulp each).  tanh(v) = 1 - 2 / (1 + e^{2v}): absolute error
// ~1e-7, which exp(-scale tanh) turns into a RELATIVE 1e-7 of the sample (the sampling tests' bar is 2e-4; the density direction,
// whose log-likelihood is compared at 2e-5, keeps libm: nf_fwd_step_kernel).  libm's tanhf + expf were ~200 instructions per pixel.
__device__ __forceinline__ float fast_tanh(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }

__global__ void __launch_bounds__(256)
nf_step_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, const NfStepDev pk,
               const float* __restrict__ clean, float sdn_a, float sdn_b, float out_mul, const NfMix mx) {
    __shared__ f2 z0s[ZW][ZW + 1];              // (channel 0, channel 1) of a position: one 8-byte read per tap
    __shared__ float4 hs[HW_][HW_ + 1];         // the four hidden channels of a position: one 16-byte read per tap
    // The coupling network's 317 parameters live in LDS and are read as broadcast 8 / 16-byte loads next to their use.  As scalar
    // registers (a by-value kernel argument) they do not fit -- a phase needs 112 / 212 of them, a wave has 102 -- and the compiler
    // parked the overflow in vector-register lanes: 700 v_readlane_b32 in the per-pixel loops, more than the arithmetic.
    __shared__ __attribute__((aligned(16))) float wsh[(sizeof(NfStepDev) / 4 + 3) & ~3];
    const NfStepDev& p = *reinterpret_cast<const NfStepDev*>(wsh);
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
    const int64_t plane = (int64_t)H * W;
    const float* xb = x + (int64_t)b * 4 * plane;
    for (int i = threadIdx.x; i < (int)(sizeof(NfStepDev) / 4); i += 256) wsh[i] = reinterpret_cast<const float*>(&pk)[i];
    __syncthreads();
    // BatchNorm as y = s x + o: folded on the host (eval) or, here, from the batch statistics (then s1 / o1 / s2 / o2 hold gamma / beta)
    if (mx.bn_stats && threadIdx.x < 4) {
        NfStepDev& pw = *reinterpret_cast<NfStepDev*>(wsh);
        const int o = threadIdx.x;
        const float sa = p.s1[o] * mx.bn_stats[4 + o], sb = p.s2[o] * mx.bn_stats[16 + o];
        pw.o1[o] = p.o1[o] - (mx.bn_stats[o] + p.b1[o]) * sa; pw.s1[o] = sa;
        pw.o2[o] = p.o2[o] - (mx.bn_stats[12 + o] + p.b2[o]) * sb; pw.s2[o] = sb;
    }
    // z0 tile (channels 0,1) with halo 2; zero outside the image (conv2d_1 pads with zeros)
    for (int i = threadIdx.x; i < ZW * ZW; i += 256) {
        const int r = i / ZW, q = i % ZW;
        const int gy = ty0 + r - 2, gx = tx0 + q - 2;
        const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
        const int64_t pix = (int64_t)gy * W + gx;
        z0s[r][q] = in ? f2{xb[pix], xb[plane + pix]} : f2{0.f, 0.f};
    }
    __syncthreads();
    // hidden map h2 (4 channels) on the tile + halo 1; positions outside the image are the zero pad.  NP positions per thread AT ONCE
    // with the weight loops outside the position loop: each weight pair is loaded into scalar registers once and used NP times.  (With
    // the position loop outside, the 112 / 212 weights a phase needs were all live across it -- more than the 102 scalar registers --
    // and the compiler parked them in vector-register lanes: 700 v_readlane_b32 in the loop bodies, more than the arithmetic.)
    auto hidden = [&](auto np_tag, int i0) {
        constexpr int NP = decltype(np_tag)::value;
        int opq = 0;
        asm volatile("" : "+v"(opq));              // (an offset the compiler cannot see through: the parameter reads stay inside this call
        const NfStepDev& p = *reinterpret_cast<const NfStepDev*>(wsh + 4 * opq);   //  instead of being hoisted -- 112 registers -- above the loop around it)
        int rr[NP], qq[NP];
        bool in[NP];
        f2 a01[NP], a23[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int i = min(i0 + 256 * k, HW_ * HW_ - 1);
            rr[k] = i / HW_; qq[k] = i % HW_;
            const int gy = ty0 + rr[k] - 1, gx = tx0 + qq[k] - 1;
            in[k] = gy >= 0 && gy < H && gx >= 0 && gx < W;
            a01[k] = ld2(p.b1); a23[k] = ld2(p.b1 + 2);
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            f2 z[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) z[k] = z0s[rr[k] + t / 3][qq[k] + t % 3];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f2 wa = ld2(p.w1[c][t]), wb = ld2(p.w1[c][t] + 2);
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const f2 zz = splat(c ? z[k].y : z[k].x);
                    a01[k] = pk_fma(wa, zz, a01[k]); a23[k] = pk_fma(wb, zz, a23[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);          // (keeps the scheduler from hoisting every tap's LDS reads to the top: 186 registers)
        }
        f2 g01[NP], g23[NP];
        float h1[NP][4];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            a01[k] = pk_fma(ld2(p.s1), a01[k], ld2(p.o1));
            a23[k] = pk_fma(ld2(p.s1 + 2), a23[k], ld2(p.o1 + 2));
            h1[k][0] = fmaxf(a01[k].x, 0.f); h1[k][1] = fmaxf(a01[k].y, 0.f); h1[k][2] = fmaxf(a23[k].x, 0.f); h1[k][3] = fmaxf(a23[k].y, 0.f);
            g01[k] = ld2(p.b2); g23[k] = ld2(p.b2 + 2);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f2 wa = ld2(p.w2[c]), wb = ld2(p.w2[c] + 2);
#pragma unroll
            for (int k = 0; k < NP; ++k) { g01[k] = pk_fma(wa, splat(h1[k][c]), g01[k]); g23[k] = pk_fma(wb, splat(h1[k][c]), g23[k]); }
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            g01[k] = pk_fma(ld2(p.s2), g01[k], ld2(p.o2));
            g23[k] = pk_fma(ld2(p.s2 + 2), g23[k], ld2(p.o2 + 2));
            const float4 h2 = make_float4(fmaxf(g01[k].x, 0.f), fmaxf(g01[k].y, 0.f), fmaxf(g23[k].x, 0.f), fmaxf(g23[k].y, 0.f));
            hs[rr[k]][qq[k]] = in[k] ? h2 : make_float4(0.f, 0.f, 0.f, 0.f);      // (lanes past the end redo the LAST position: same value, and no branch for the arithmetic to sink into)
        }
    };
    static_assert(HW_ * HW_ > 4 * 256 && HW_ * HW_ <= 5 * 256, "hidden tile: four full rounds and a tail");
#pragma unroll 1
    for (int half = 0; half < 2; ++half) hidden(std::integral_constant<int, 2>{}, threadIdx.x + 512 * half);
    if ((threadIdx.x & ~63) + 1024 < HW_ * HW_) hidden(std::integral_constant<int, 1>{}, threadIdx.x + 1024);      // (wave-uniform: waves 0 .. 2)
    __syncthreads();
    // outputs: 2 + 2 pixels per thread (same column, rows 8 apart), for the same reason
    constexpr int NPX = 2;
    static_assert(TS * TS == 2 * NPX * 256, "two rounds of NPX pixels");
    const int q = threadIdx.x % TS;
    const float cdiv = mx.clean_div ? mx.clean_div[b] : mx.clean_div_s;
    const float mmul = mx.mix_mul ? mx.mix_mul[b] : mx.mix_mul_s;
    bool neg = false;
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
    const int r0 = threadIdx.x / TS + 16 * half;
    int opq = 0;
    asm volatile("" : "+v"(opq));
    const NfStepDev& p = *reinterpret_cast<const NfStepDev*>(wsh + 4 * opq);
    const int gx = tx0 + q;
    f2 s01[NPX], s23[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) { s01[k] = ld2(p.b3); s23[k] = ld2(p.b3 + 2); }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float4 h[NPX];
#pragma unroll
        for (int k = 0; k < NPX; ++k) h[k] = hs[r0 + 8 * k + t / 3][q + t % 3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f2 wa = ld2(p.w3[c][t]), wb = ld2(p.w3[c][t] + 2);
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
                const float hv = c == 0 ? h[k].x : (c == 1 ? h[k].y : (c == 2 ? h[k].z : h[k].w));
                s01[k] = pk_fma(wa, splat(hv), s01[k]); s23[k] = pk_fma(wb, splat(hv), s23[k]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // channel 4: ones on the border ring of the padded map, i.e. exactly the out-of-image taps (only image-border pixels have one)
    {
        bool edge = gx == 0 || gx == W - 1;
#pragma unroll
        for (int k = 0; k < NPX; ++k) { const int gy = ty0 + r0 + 8 * k; edge |= gy == 0 || gy == H - 1; }
        if (edge) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f2 wa = ld2(p.w3[4][t]), wb = ld2(p.w3[4][t] + 2);
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    const int yy = ty0 + r0 + 8 * k + t / 3 - 1, xx = gx + t % 3 - 1;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W) { s01[k] += wa; s23[k] += wb; }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int r = r0 + 8 * k, gy = ty0 + r;
        if (gy >= H || gx >= W) continue;
        const f2 o01 = s01[k] * ld2(p.e3), o23 = s23[k] * ld2(p.e3 + 2);                   // o3[0..3]
        const int64_t pix = (int64_t)gy * W + gx;
        const f2 z0 = z0s[r + 2][q + 2];
        const float z1a = xb[2 * plane + pix], z1b = xb[3 * plane + pix];
        const float v[4] = {z0.x, z0.y,
                            (z1a - o01.x) * __expf(-(p.scale * fast_tanh(o23.x))),
                            (z1b - o01.y) * __expf(-(p.scale * fast_tanh(o23.y)))};
        f2 r01 = {0.f, 0.f}, r23 = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r01 = pk_fma(ld2(p.winv[c]), splat(v[c]), r01);
            r23 = pk_fma(ld2(p.winv[c] + 2), splat(v[c]), r23);
        }
        const float rs[4] = {r01.x, r01.y, r23.x, r23.y};
        float post = out_mul;
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (clean) {
                float cl = clean[((int64_t)b * 4 + o) * plane + pix];
                if (cdiv != 1.f) cl = __fdiv_rn(cl, cdiv);               // imgs_hr / ratio, rounded like the tensor op
                const float sc2 = sdn_a * cl + sdn_b;
                neg = neg || sc2 < 0.f;
                post = out_mul * sqrtf(sc2);
            }
            float rr = rs[o] * post;
            if (mx.mix_base) {                                           // imgs_hr + noise * ratio, two roundings like the tensor ops, then the clamp
                rr = mul_then_add(mx.mix_base[((int64_t)b * 4 + o) * plane + pix], rr, mmul);
                if (rr == rr) rr = fminf(fmaxf(rr, mx.lo), mx.hi);          // (a NaN stays a NaN, as in Tensor.clamp)
            }
            y[((int64_t)b * 4 + o) * plane + pix] = rr;
        }
    }
    }
    if (neg && mx.flag) atomicOr(mx.flag, 1);
}

// nn.BatchNorm2d's buffer update for the two BatchNorm layers of a coupling whose batch statistics pnnp_nf_train_stats_f32 left on the
// device (momentum 0.1, unbiased variance; the kernels' means are bias-free, BatchNorm sees conv output + bias)
__global__ void nf_bn_update_kernel(const float* __restrict__ bn, const float* __restrict__ cb1, const float* __restrict__ cb2,
                                    float* rm1, float* rv1, float* rm2, float* rv2, long long* nb1, long long* nb2, float unbias) {
    const int t = threadIdx.x;
    if (t < 4) {
        rm1[t] = rm1[t] * 0.9f + 0.1f * (bn[t] + cb1[t]);
        rv1[t] = rv1[t] * 0.9f + 0.1f * bn[8 + t] * unbias;
    } else if (t < 8) {
        const int o = t - 4;
        rm2[o] = rm2[o] * 0.9f + 0.1f * (bn[12 + o] + cb2[o]);
        rv2[o] = rv2[o] * 0.9f + 0.1f * bn[20 + o] * unbias;
    } else if (t == 8) {
        if (nb1) *nb1 += 1;
        if (nb2) *nb2 += 1;
    }
}

// Density direction (NoiseFlow.forward, archs/noise_flow.py:113-130): one [Conv2d1x1, AffineCoupling] pair of the forward
// chain on x [B][4][H][W] -> y, and the per-workgroup partial sums of its pixel-wise log-det terms.
//   x' = x / sqrt(sdn_a*clean + sdn_b)   (SignalDependantISO forward, first pair only; its log-det is -sum log scale)
//   v  = W x'                            (p.winv holds W = P L U here, with 1/gain folded in after GainISO)
//   z0 = v[0:2];  z1 = v[2:4] * exp(log_scale(z0)) + shift(z0)        (affine_coupling.py:36-53)
//   partial[b][block] = sum over the block's pixels of (log_scale_a + log_scale_b - sum_c log scale_c)
__global__ void __launch_bounds__(256)
nf_fwd_step_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ partial, int H, int W, const NfStep p,
                   const float* __restrict__ clean, float sdn_a, float sdn_b) {
    __shared__ float z0s[2][ZW][ZW + 1];
    __shared__ float hs[4][HW_][HW_ + 1];
    __shared__ float red[256];
    const int b = blockIdx.z, ty0 = blockIdx.y * TS, tx0 = blockIdx.x * TS;
    const int64_t plane = (int64_t)H * W;
    const float* xb = x + (int64_t)b * 4 * plane;
    const float* cb = clean ? clean + (int64_t)b * 4 * plane : nullptr;
    // v0, v1 = rows 0,1 of W applied to x' on the tile + halo 2; zero outside the image (conv2d_1 pads with zeros)
    for (int i = threadIdx.x; i < ZW * ZW; i += 256) {
        const int r = i / ZW, q = i % ZW;
        const int gy = ty0 + r - 2, gx = tx0 + q - 2;
        float v0 = 0.f, v1 = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const int64_t pix = (int64_t)gy * W + gx;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float xv = xb[c * plane + pix];
                if (cb) xv = xv / sqrtf(sdn_a * cb[c * plane + pix] + sdn_b);
                v0 += p.winv[0][c] * xv; v1 += p.winv[1][c] * xv;
            }
        }
        z0s[0][r][q] = v0; z0s[1][r][q] = v1;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < HW_ * HW_; i += 256) {
        const int r = i / HW_, q = i % HW_;
        const int gy = ty0 + r - 1, gx = tx0 + q - 1;
        float h2[4] = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            float h1[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float s = p.b1[o];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int t = 0; t < 9; ++t) s += p.w1[o][c][t] * z0s[c][r + t / 3][q + t % 3];
                h1[o] = fmaxf(p.s1[o] * s + p.o1[o], 0.f);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float s = p.b2[o];
#pragma unroll
                for (int c = 0; c < 4; ++c) s += p.w2[o][c] * h1[c];
                h2[o] = fmaxf(p.s2[o] * s + p.o2[o], 0.f);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) hs[o][r][q] = h2[o];
    }
    __syncthreads();
    float ld = 0.f;
    for (int i = threadIdx.x; i < TS * TS; i += 256) {
        const int r = i / TS, q = i % TS;
        const int gy = ty0 + r, gx = tx0 + q;
        if (gy >= H || gx >= W) continue;
        float o3[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float s = p.b3[o];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = gy + t / 3 - 1, xx = gx + t % 3 - 1;
#pragma unroll
                for (int c = 0; c < 4; ++c) s += p.w3[o][c][t] * hs[c][r + t / 3][q + t % 3];
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) s += p.w3[o][4][t];
            }
            o3[o] = s * p.e3[o];
        }
        const int64_t pix = (int64_t)gy * W + gx;
        float v2 = 0.f, v3 = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float xv = xb[c * plane + pix];
            if (cb) {
                const float sc = sqrtf(sdn_a * cb[c * plane + pix] + sdn_b);
                xv = xv / sc;
                ld -= logf(sc);
            }
            v2 += p.winv[2][c] * xv; v3 += p.winv[3][c] * xv;
        }
        const float lsa = p.scale * tanhf(o3[2]), lsb = p.scale * tanhf(o3[3]);
        ld += lsa + lsb;
        float* yb = y + (int64_t)b * 4 * plane + pix;
        yb[0] = z0s[0][r + 2][q + 2]; yb[plane] = z0s[1][r + 2][q + 2];
        yb[2 * plane] = v2 * expf(lsa) + o3[0]; yb[3 * plane] = v3 * expf(lsb) + o3[1];
    }
    red[threadIdx.x] = ld;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[((int64_t)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = red[0];
}

}  // namespace

extern "C" {

// out[i] ~ N(0,1), counter-based (Philox4x32-10, Box-Muller); the prior draw of NoiseFlow.sample
// (archs/noise_flow.py:208-221 draws it with torch.normal).
int pnnp_normal_fill_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
    if (n < 0 || (n && !out)) return PNNP_E_INVALID;
    if (n == 0) return PNNP_OK;
    int64_t blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(normal_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, n,
                       (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32), (uint32_t)offset);
    return pnnp_launch_status();
}

// One [AffineCoupling^-1, Conv2d1x1^-1] pair of the reversed chain on x [B][4][H][W] -> y.
// step [host]: 317 floats laid out as struct NfStep.  clean (optional, [B][4][H][W]): multiply the
// result by sqrt(sdn_a*clean + sdn_b) (SignalDependantISO^-1, last pair); out_mul: scalar factor.
int pnnp_nf_bn_update_f32(const float* bn_stats, const float* conv_bias1, const float* conv_bias2, float* running_mean1,
                          float* running_var1, float* running_mean2, float* running_var2, long long* batches1, long long* batches2,
                          double n, void* stream) {
    if (!bn_stats || !conv_bias1 || !conv_bias2 || !running_mean1 || !running_var1 || !running_mean2 || !running_var2 || n < 1) return PNNP_E_INVALID;
    hipLaunchKernelGGL(nf_bn_update_kernel, dim3(1), dim3(64), 0, as_stream(stream), bn_stats, conv_bias1, conv_bias2, running_mean1, running_var1,
                       running_mean2, running_var2, batches1, batches2, (float)(n / (n > 1 ? n - 1 : 1)));
    return pnnp_launch_status();
}

int pnnp_nf_step_mix_f32(const float* x, float* y, int B, int H, int W, const float* step /*[host]*/,
                         const float* clean, float sdn_a, float sdn_b, float out_mul,
                         const float* clean_div, float clean_div_s, const float* mix_base, const float* mix_mul, float mix_mul_s,
                         float clamp_lo, float clamp_hi, int* flag, const float* bn_stats, void* stream) {
    if (!x || !y || !step || B < 0 || H <= 0 || W <= 0 || x == y || y == mix_base) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    NfStep hp;
    static_assert(sizeof(NfStep) == 317 * sizeof(float) && sizeof(NfStepDev) == sizeof(NfStep), "NfStep layout");
    memcpy(&hp, step, sizeof hp);
    NfStepDev p;
    for (int o = 0; o < 4; ++o) {
        for (int c = 0; c < 2; ++c) for (int t = 0; t < 9; ++t) p.w1[c][t][o] = hp.w1[o][c][t];
        for (int c = 0; c < 5; ++c) for (int t = 0; t < 9; ++t) p.w3[c][t][o] = hp.w3[o][c][t];
        for (int c = 0; c < 4; ++c) { p.w2[c][o] = hp.w2[o][c]; p.winv[c][o] = hp.winv[o][c]; }
        p.b1[o] = hp.b1[o]; p.s1[o] = hp.s1[o]; p.o1[o] = hp.o1[o]; p.b2[o] = hp.b2[o]; p.s2[o] = hp.s2[o]; p.o2[o] = hp.o2[o];
        p.b3[o] = hp.b3[o]; p.e3[o] = hp.e3[o];
    }
    p.scale = hp.scale;
    NfMix mx{clean_div, clean_div_s, mix_base, mix_mul, mix_mul_s, clamp_lo, clamp_hi, flag, bn_stats};
    hipLaunchKernelGGL(nf_step_kernel, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B), dim3(256), 0, as_stream(stream),
                       x, y, H, W, p, clean, sdn_a, sdn_b, out_mul, mx);
    return pnnp_launch_status();
}

int pnnp_nf_step_f32(const float* x, float* y, int B, int H, int W, const float* step /*[host]*/,
                     const float* clean, float sdn_a, float sdn_b, float out_mul, void* stream) {
    return pnnp_nf_step_mix_f32(x, y, B, H, W, step, clean, sdn_a, sdn_b, out_mul, nullptr, 1.f, nullptr, nullptr, 1.f, 0.f, 0.f, nullptr, nullptr, stream);
}

// One [Conv2d1x1, AffineCoupling] pair of the FORWARD (density) chain, NoiseFlow.forward (archs/noise_flow.py:113-130).
// step [host]: struct NfStep with W (not its inverse) in the matrix slot.  partial [B][ceil(H/32)*ceil(W/32)]: the
// pixel-wise log-det terms summed per workgroup (the caller adds them and the scalar terms).  clean: SDN forward.
int pnnp_nf_fwd_step_f32(const float* x, float* y, float* partial, int B, int H, int W, const float* step /*[host]*/,
                         const float* clean, float sdn_a, float sdn_b, void* stream) {
    if (!x || !y || !partial || !step || B < 0 || H <= 0 || W <= 0 || x == y) return PNNP_E_INVALID;
    if (B == 0) return PNNP_OK;
    NfStep p;
    memcpy(&p, step, sizeof p);
    hipLaunchKernelGGL(nf_fwd_step_kernel, dim3((W + TS - 1) / TS, (H + TS - 1) / TS, B), dim3(256), 0, as_stream(stream),
                       x, y, partial, H, W, p, clean, sdn_a, sdn_b);
    return pnnp_launch_status();
}

}  // extern "C"
