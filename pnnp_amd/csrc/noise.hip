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
//      counter = (element-in-crop, crop_base+b, slot, offset_lo).  A QUAD of 4 consecutive pixels of a row
//      (= one thread) shares blocks keyed by its first element: slot 0 = the four shot uniforms, slot 1 = the
//      read noise (two Box-Muller pairs, cos -> even pixel, sin -> odd pixel; Tukey-lambda: one uniform each),
//      slot 2 = the second uniform of the first PTRS round (only computed when a pixel of the thread has
//      lam >= 10), slot 3 = quantisation.  Later PTRS rounds of a pixel: blocks keyed by the pixel's own
//      element, slots 9..; slot 0x40000000 with element = c*H+h is the row draw.  (Round 1 drew one block per
//      pixel and a second one for 'q': twice the Philox work and twice the log / sqrt / cos.)  The sample
//      depends only on (seed, offset, global crop index, element) -- not on batch size, grid or GPU count.
// Poisson: lam < 10 sequential inversion; lam >= 10 Hoermann's PTRS transformed rejection
// (exact, no Gaussian approximation).  Specification: oracle/pnnp_oracle.c.
//
// Thread layout: one thread owns 4 consecutive pixels of a row (float4 load/store, fully coalesced).  The kernel is bound by VECTOR
// INSTRUCTION ISSUE, not by HBM (profiles/r5/noise_sampler_pmc.txt: ~100 % of the SIMDs' issue slots, 40 - 50 of 64 lanes alive), so its
// shape follows from keeping lanes alive:
//   * straight-line work (Philox blocks, Box-Muller pairs, PTRS's quick acceptance) is done for the 4 pixels of every lane;
//   * data-dependent work -- the inversion loop, PTRS's logarithm test and its retries -- goes through wave-private LDS queues and is run
//     on dense groups of 64 pixels (see the kernel);
//   * the row-noise normal -- one draw per (crop, channel, row) -- is drawn for all rows of a block side by side, not by one lane per wave;
//   * index arithmetic is 32-bit with multiply-shift division (the 64-bit / and % are software routines).
#include "common.h"

namespace {

// Division of a 31-bit index by a run-time constant (Granlund-Montgomery, round-up variant): q = (mulhi(t, m) + t) >> l with
// l = ceil(log2 d), m = ceil(2^(32+l) / d) - 2^32.  Exact for t < 2^31 (the sum cannot wrap).  The 64-bit / and % the index
// arithmetic used before are software routines on this hardware: four of them were ~600 instructions per quad, as much as the sampling.
struct FastDiv {
    uint32_t m, l, d;
};
__device__ __forceinline__ uint32_t fd_div(uint32_t t, const FastDiv f) { return (__umulhi(t, f.m) + t) >> f.l; }

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

// cos(2 pi u) and sin(2 pi u) for u in (0, 1), absolute error < 1.5e-7 (the oracle's cosf / sinf of the ROUNDED product
// 6.2831853f * u are no closer to the true values: that rounding alone moves the angle by up to 3.7e-7 rad).  Octant reduction on u
// itself -- exact, u is a 23-bit fraction -- then the Taylor polynomials on [-pi/4, pi/4]: ~22 VALU for both.
__device__ __forceinline__ void sincos_2pi(float u, float& sn, float& cs) {
    const float t = u * 4.f;                                   // quarter turns
    const float q = rintf(t);                                  // nearest quarter turn, 0 .. 4
    const float r = (t - q) * 1.57079632679489661923f;         // |r| <= pi/4 (t - q is exact)
    const float r2 = r * r;
    const float c = fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.4801587e-5f, -1.3888889e-3f), 4.1666668e-2f), -0.5f), 1.f);
    const float s = r * fmaf(r2, fmaf(r2, fmaf(r2, fmaf(r2, 2.7557319e-6f, -1.9841270e-4f), 8.3333338e-3f), -1.6666667e-1f), 1.f);
    const int qi = (int)q & 3;                                 // angle = qi * pi/2 + r
    const float cq = (qi & 1) ? s : c, sq = (qi & 1) ? c : s;  // qi=0: (c, s)  1: (-s, c)  2: (-c, -s)  3: (s, -c)
    cs = (qi == 1 || qi == 2) ? -cq : cq;
    sn = (qi >= 2) ? -sq : sq;
}
// the two normals of one Box-Muller pair: n0 = r cos, n1 = r sin
__device__ __forceinline__ void box_muller2(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float r = fast_sqrt(-2.0f * fast_log(u01(a)));
    float sn, cs;
    sincos_2pi(u01(b), sn, cs);
    n0 = r * cs; n1 = r * sn;
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

// Sequential inversion for lam < 10 (lanes with `take` false idle along): the oracle's loop
//     while (u > s) { k += 1; p *= lam / k; s2 = s + p; if (s2 == s) break; s = s2; }
// with the step counter WAVE-UNIFORM -- every lane that is still searching is at the same k.  The first 24 steps are unrolled, so
// 1/k is a literal (a per-lane v_rcp_f32 is a quarter-rate instruction: 4 of the loop's 11 issue slots) and a step is 6 VALU
// instructions + one ballot on the scalar unit; beyond 24 (P < 5e-5 per pixel at lam = 10) a plain loop takes over.  The dark
// regime spends most of its time here: a wave iterates to the largest count among its 64 lanes, for each of its 4 pixels.
__device__ __forceinline__ float poisson_small(float lam, float u_in, bool take) {
    float p = fast_exp(-lam), s = p, k = 0.f;
    float u = take ? u_in : -1.f;                                  // a lane is searching while u > s; done / idle lanes carry u = -1
#define PS_STEP(FJ, RJ) {                                                                                        \
        k = (u > s) ? (FJ) : k;                                                                                   \
        p *= lam * (RJ);                                                                                          \
        const float s2 = s + p;                                                                                   \
        /* the rounded CDF can saturate below the largest uniform (1 - 2^-24): stop when a term no longer moves the sum */ \
        u = (s2 == s) ? -1.f : u;                                                                                 \
        s = s2; }
    bool more = true;
#pragma unroll
    for (int j = 1; j <= 24; ++j) {
        if (more) {
            more = __builtin_amdgcn_ballot_w64(u > s) != 0;
            if (more) PS_STEP((float)j, 1.0f / (float)j)
        }
    }
    if (more)
        for (int j = 25; j < 64; ++j) {
            if (__builtin_amdgcn_ballot_w64(u > s) == 0) break;
            PS_STEP((float)j, __builtin_amdgcn_rcpf((float)j))
        }
#undef PS_STEP
    return k;
}

// Hoermann's PTRS for lam >= 10, all rounds (the oracle's loop).  Round 0 uses (r0, r1); later rounds draw their own blocks.
__device__ float poisson_ptrs(float lam, uint32_t elem, const Ctx& c, uint32_t r0, uint32_t r1) {
    const float slam = fast_sqrt(lam), loglam = fast_log(lam);
    const float b = 0.931f + 2.53f * slam;
    const float a = -0.059f + 0.02483f * b;
    const float inv_alpha = 1.1239f + fast_div(1.1328f, b - 3.4f);
    const float vr = 0.9277f - fast_div(3.6224f, b - 2.f);
    uint32_t x0 = r0, x1 = r1;
    uint4 extra = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < 64; ++it) {
        if (it > 0) {
            if (it & 1) { extra = philox4x32_10(elem, c.crop, 8u + ((it + 1) >> 1), c.off, c.k0, c.k1); x0 = extra.x; x1 = extra.y; }
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
// PTRS round 0, the quick acceptance only (86 % of the pixels at lam >= 10 leave here): no logarithm, no retry.  A pixel this does not
// accept is finished by poisson_ptrs, which repeats the round -- the same operations on the same inputs -- and goes on from there.
__device__ __forceinline__ bool ptrs_quick(float lam, uint32_t r0, uint32_t r1, float& k) {
    const float b = 0.931f + 2.53f * fast_sqrt(lam);
    const float a = -0.059f + 0.02483f * b;
    const float vr = 0.9277f - fast_div(3.6224f, b - 2.f);
    const float U = u01(r0) - 0.5f, V = u01(r1);
    const float us = 0.5f - fabsf(U);
    k = floorf((fast_div(2.f * a, us) + b) * U + lam + 0.43f);
    return us >= 0.07f && V <= vr;
}

// one pixel on its own (sna_kernel): every lane of the wave must come through here (poisson_small is wave-convergent)
__device__ float poisson_f32(float lam, uint32_t elem, const Ctx& c, uint32_t r0, uint32_t r1) {
    const bool small = lam > 0.f && lam < 10.f;
    const float ks = poisson_small(lam, u01(r0), small);
    if (!(lam > 0.f)) return 0.f;
    if (small) return ks;
    return poisson_ptrs(lam, elem, c, r0, r1);
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
                    uint32_t k0, uint32_t k1, uint32_t off, uint32_t crop_base, const FastDiv dwq, const FastDiv dH, const FastDiv dC) {
    const uint32_t wq = (uint32_t)(W + 3) >> 2;
    const uint32_t total = (uint32_t)B * C * H * wq;              // < 2^31: the launcher splits larger batches
    const bool torch_mode = flags & PNNP_NOISE_MODE_TORCH;
    const bool use_p = flags & PNNP_NOISE_P, use_g = flags & PNNP_NOISE_G, use_r = flags & PNNP_NOISE_R;
    const bool use_q = flags & PNNP_NOISE_Q, use_d = flags & PNNP_NOISE_D, use_b = flags & PNNP_NOISE_B;
    const bool extras = torch_mode || !use_b;        // OBS: 'b' removes read, row, quant and bias
    const int lane = threadIdx.x & 63;
    // Wave-private queues (the header's "Poisson" paragraph): entry = (lam, shot uniform, first-round V, element), tag = slot | crop << 8.
    // PTRS pixels fill q_ent from the front, inversion pixels from the back; a wave has 256 pixels in flight, so they never meet.
    __shared__ uint4 q_ent_all[4][256];
    __shared__ uint32_t q_tag_all[4][256];
    __shared__ float q_res_all[4][256];
    __shared__ float row_n[256];                 // the row-noise normals of the rows this block touches
    uint4* const q_ent = q_ent_all[threadIdx.x >> 6];
    uint32_t* const q_tag = q_tag_all[threadIdx.x >> 6];
    float* const q_res = q_res_all[threadIdx.x >> 6];
    // grid-stride with the SAME trip count for all lanes of a wave (shuffles below)
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t t0 = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t iters = (total + stride - 1) / stride;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t t = t0 + it * stride;
        const bool live = t < total;
        const uint32_t tt = live ? t : total - 1;
        const uint32_t row = fd_div(tt, dwq);            // (b*C + c)*H + h
        const int xq = (int)(tt - row * wq);
        const uint32_t bc = fd_div(row, dH);
        const int h = (int)(row - bc * (uint32_t)H);
        const int b = (int)fd_div(bc, dC);
        const int c = (int)(bc - (uint32_t)b * (uint32_t)C);
        const float* P = params + (int64_t)b * PNNP_NPARAM;
        const float K = P[PNNP_P_K], ratio = P[PNNP_P_RATIO], wp = P[PNNP_P_WP], bl = P[PNNP_P_BL];
        const float span = wp - bl;
        Ctx ctx{k0, k1, crop_base + (uint32_t)b, off};

        uint32_t row_first = 0;
        // ---- row noise: ONE draw per (crop, channel, row).  The rows this block's 256 quads touch are drawn side by side by its first
        // threads -- thread j: row_first + j -- and handed round through LDS.  (Drawn by the first lane of each run of lanes sharing a row,
        // the Philox block and the Box-Muller ran once per WAVE with one lane alive: ~220 issue slots of the ~720 a dark quad costs.)
        float row_noise = 0.f;
        if (use_r && extras) {
            const uint32_t tb = blockIdx.x * blockDim.x + it * stride;
            row_first = fd_div(tb < total ? tb : total - 1, dwq);
            const uint32_t row_last = fd_div(tb + 255 < total ? tb + 255 : total - 1, dwq);
            if (it) __syncthreads();
            const uint32_t rj = row_first + threadIdx.x;
            if (rj <= row_last) {
                const uint32_t bcj = fd_div(rj, dH), bj = fd_div(bcj, dC);
                const uint32_t chj = rj - bj * (uint32_t)C * (uint32_t)H;            // c * H + h
                const uint4 r = philox4x32_10(chj, crop_base + bj, 0x40000000u, off, k0, k1);
                float n0, n1;
                box_muller2(r.x, r.y, n0, n1);
                row_n[threadIdx.x] = n0;
            }
            // (read back just before the stores, behind a barrier THERE: by then the drawing wave is long done and nobody waits for it)
        }
        // (lanes past the end stay in the loop -- the queues below are drained by ALL lanes of the wave -- on the last quad's inputs; they
        //  neither queue nor store)
        const int nx = min(4, W - 4 * xq);
        const int64_t base = (int64_t)row * W + 4 * xq;
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
        float o[4], yv[4], lam[4], rdn[4] = {0.f, 0.f, 0.f, 0.f}, shn[4] = {0.f, 0.f, 0.f, 0.f};
        // the quad's blocks (see the header): shot uniforms, read noise, first-round PTRS V (lazily), quantisation
        const uint4 b0 = philox4x32_10(e0, ctx.crop, 0u, off, k0, k1);
        const uint32_t U[4] = {b0.x, b0.y, b0.z, b0.w};
        bool big = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            yv[i] = fast_div(__fmul_rn(v[i], span), ratio);
            lam[i] = fast_div(__fmul_rn(mfm, yv[i]), K);
            big |= lam[i] >= 10.f;
        }
        uint32_t V[4] = {0u, 0u, 0u, 0u};
        if (use_p && big) {
            const uint4 b2 = philox4x32_10(e0, ctx.crop, 2u, off, k0, k1);
            V[0] = b2.x; V[1] = b2.y; V[2] = b2.z; V[3] = b2.w;
        }
        if (!use_p) { box_muller2(b0.x, b0.y, shn[0], shn[1]); box_muller2(b0.z, b0.w, shn[2], shn[3]); }
        // ---- Poisson counts of the quad.  Straight-line part: lam <= 0 -> 0; lam >= 10 -> PTRS round 0's quick acceptance.  Everything
        // data-dependent -- the inversion loop of lam < 10, PTRS's logarithm test and its retries -- is queued and run by the wave on
        // dense groups of 64 pixels: the 4 x 64 pixels a wave holds need 1 + 1 such passes on a mid-grey crop where running pixel i of
        // all lanes under its own branches needed 4 x (1 + 1 + retries), each with a seventh or a tenth of the lanes alive.
        float kp[4] = {0.f, 0.f, 0.f, 0.f};
        unsigned queued = 0;
        if (use_p) {
            int n_big = 0, n_small = 0;                                     // wave-uniform
            const bool wave_big = __builtin_amdgcn_ballot_w64(big) != 0;
            // inversion pixels: through the queue when that saves passes (<= 128 of the wave's 256: 1 - 2 dense passes), in place
            // otherwise (a dark crop: every pixel is one, the queue would only add its traffic to the same 4 passes)
            int tot_small = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) tot_small += __popcll(__builtin_amdgcn_ballot_w64(live && lam[i] > 0.f && lam[i] < 10.f));
            const bool queue_small = tot_small <= 128;
            if (!queue_small) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool sm = live && lam[i] > 0.f && lam[i] < 10.f;
                    const float ks = poisson_small(lam[i], u01(U[i]), sm);
                    kp[i] = sm ? ks : 0.f;
                }
            }
            if (wave_big || (queue_small && tot_small > 0)) {              // (a dark crop skips the queues altogether)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float l = lam[i];
                const bool pos = live && l > 0.f;                           // (NaN: count 0, as the oracle's !(lam > 0))
                const bool small0 = pos && l < 10.f, bigp = pos && !small0;
                const bool small = small0 && queue_small;
                bool q_big = false;
                if (wave_big) {
                    float kq;
                    const bool acc = ptrs_quick(l, U[i], V[i], kq);
                    kp[i] = (bigp && acc) ? kq : kp[i];
                    q_big = bigp && !acc;
                }
                const uint64_t mb = __builtin_amdgcn_ballot_w64(q_big), ms = __builtin_amdgcn_ballot_w64(small);
                const uint4 ent = make_uint4(__float_as_uint(l), U[i], V[i], e0 + i);
                const uint32_t tag = (uint32_t)(i * 64 + lane) | ((uint32_t)b << 8);
                if (q_big) {
                    const int p = n_big + __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
                    q_ent[p] = ent; q_tag[p] = tag;
                }
                if (small) {
                    const int p = 255 - (n_small + __builtin_amdgcn_mbcnt_hi((uint32_t)(ms >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ms, 0u)));
                    q_ent[p] = ent; q_tag[p] = tag;
                }
                n_big += __popcll(mb); n_small += __popcll(ms);
                queued |= (unsigned)(q_big || small) << i;
            }
            if (n_big | n_small) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                for (int j0 = 0; j0 < n_big; j0 += 64) {
                    const int j = j0 + lane;
                    if (j < n_big) {
                        const uint4 e = q_ent[j];
                        const uint32_t tg = q_tag[j];
                        const Ctx cj{k0, k1, crop_base + (tg >> 8), off};
                        q_res[tg & 255u] = poisson_ptrs(__uint_as_float(e.x), e.w, cj, e.y, e.z);
                    }
                }
                for (int j0 = 0; j0 < n_small; j0 += 64) {
                    const int j = j0 + lane;
                    const bool take = j < n_small;
                    const uint4 e = q_ent[255 - (take ? j : 0)];
                    const uint32_t tg = q_tag[255 - (take ? j : 0)];
                    const float k = poisson_small(__uint_as_float(e.x), u01(e.y), take);
                    if (take) q_res[tg & 255u] = k;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if ((queued >> i) & 1u) kp[i] = q_res[i * 64 + lane];
                __builtin_amdgcn_wave_barrier();                            // (the next iteration's pushes come after these reads)
            }
            }
        }
        if (!use_b) {
            const uint4 b1 = philox4x32_10(e0, ctx.crop, 1u, off, k0, k1);
            if (use_g) {
                const float lt = P[PNNP_P_LAM];
                rdn[0] = tukey_lambda(u01(b1.x), lt); rdn[1] = tukey_lambda(u01(b1.y), lt);
                rdn[2] = tukey_lambda(u01(b1.z), lt); rdn[3] = tukey_lambda(u01(b1.w), lt);
            } else {
                box_muller2(b1.x, b1.y, rdn[0], rdn[1]); box_muller2(b1.z, b1.w, rdn[2], rdn[3]);
            }
        }
        uint32_t Q[4] = {0u, 0u, 0u, 0u};
        if (extras && use_q) {
            const uint4 b3 = philox4x32_10(e0, ctx.crop, 3u, off, k0, k1);
            Q[0] = b3.x; Q[1] = b3.y; Q[2] = b3.z; Q[3] = b3.w;
        }
        if (use_r && extras) {
            __syncthreads();
            row_noise = __fdiv_rn(__fmul_rn(row_n[row - row_first], P[PNNP_P_SIGR]), mfm);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float yy = yv[i];
            float shot;
            if (use_p) {
                shot = fast_div(__fmul_rn(kp[i], K), mfm);
            } else {
                const float s = fast_sqrt(fmaxf(fast_div(yy, K), 1e-10f));
                shot = __fadd_rn(yy, fast_div(__fmul_rn(__fmul_rn(shn[i], s), K), mfm));
            }
            float acc = shot;
            if (!use_b) acc = __fadd_rn(acc, __fmul_rn(rdn[i], sig_read));
            if (extras) {
                if (use_r) acc = __fadd_rn(acc, row_noise);
                if (use_q) acc = __fadd_rn(acc, __fmul_rn(u01(Q[i]) - 0.5f, qscale));
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
        if (!live) continue;
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
    if ((int64_t)C * H * W >= (1ll << 32) || B >= (1 << 24)) return PNNP_E_INVALID;     // (element counter 32 bits; crop index 24 bits in the queue tag)
    const int64_t per_crop = (int64_t)C * H * ((W + 3) / 4);
    if ((int64_t)B * per_crop == 0) return PNNP_OK;
    if (per_crop >= (1ll << 31)) return PNNP_E_INVALID;
    auto fastdiv = [](uint32_t d) {
        uint32_t l = 0;
        while ((1ull << l) < d) ++l;
        const uint64_t m = (((unsigned __int128)1 << (32 + l)) + d - 1) / d - (1ull << 32);
        return FastDiv{(uint32_t)m, l, d};
    };
    const FastDiv dwq = fastdiv((uint32_t)((W + 3) / 4)), dH = fastdiv((uint32_t)H), dC = fastdiv((uint32_t)C);
#ifndef NS_BLOCKS_PER_CU
#define NS_BLOCKS_PER_CU 64      // short blocks, scheduled dynamically: a wave's time varies with its pixels' Poisson paths (measured 5 .. 64: 358 -> 196 us)
#endif
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
    const int Bmax = (int)(((1ll << 31) - 1) / per_crop);                          // crops per launch: the kernel's index is 31 bits
    for (int b0 = 0; b0 < B; b0 += Bmax) {
        const int Bl = B - b0 < Bmax ? B - b0 : Bmax;
        const int64_t total = (int64_t)Bl * per_crop;
        int64_t blocks = (total + 255) / 256;
        if (blocks > pnnp_device_cus() * NS_BLOCKS_PER_CU) blocks = pnnp_device_cus() * NS_BLOCKS_PER_CU;
        const int64_t eoff = (int64_t)b0 * C * H * W;
        hipLaunchKernelGGL(noise_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                           y + eoff, out + eoff, Bl, C, H, W, params + (int64_t)b0 * PNNP_NPARAM, flags, mfm, k0, k1, (uint32_t)offset,
                           crop_base + (uint32_t)b0, dwq, dH, dC);
    }
    return pnnp_launch_status();
}
