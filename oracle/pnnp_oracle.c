/*
 * pnnp_oracle.c -- ORACLE, TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C CPU restatement of
 *   (1) Bayer pack / unpack          utils/isp_ops.py:84-112  (bit-exact integer/byte work)
 *   (2) the physics noise sampler    data_process/process.py:591-673
 *       on the counter-based RNG the HIP kernel is specified to use.
 *
 * The reference draws from numpy / torch global RNG streams that cannot be reproduced
 * outside those libraries, so (2) pins the *distribution* to the reference statistically
 * (tests/golden/noise_stats.npz, captured from the reference itself) and serves as the
 * element-wise specification of the HIP sampler: same (seed, offset, crop, element) =>
 * same uniforms => same sample, up to libm rounding of log/exp/cos/lgamma.
 *
 * Algorithms restated from their publications:
 *   Philox4x32-10   Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11
 *   PTRS Poisson    W. Hoermann, "The transformed rejection method for generating Poisson
 *                   random variables", Insurance: Mathematics and Economics 12 (1993)
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* ------------------------------------------------------------------ (1) pack / unpack */
static const int PLANE_DY[4] = {0, 0, 1, 1};   /* R, G1, B, G2 */
static const int PLANE_DX[4] = {0, 1, 1, 0};

/* raw2bayer, utils/isp_ops.py:84-96.  is_f32 selects the input element type. */
void pnnp_oracle_pack(const void* src, int is_f32, int H, int W, float* dst, const double* black4,
                      double wp, int norm, int clip) {
    const int h = H / 2, w = W / 2;
    for (int c = 0; c < 4; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const long i = (long)(2 * y + PLANE_DY[c]) * W + 2 * x + PLANE_DX[c];
                const float v = is_f32 ? ((const float*)src)[i] : (float)((const uint16_t*)src)[i];
                float o;
                if (norm) {
                    double d = ((double)v - black4[c]) / (wp - black4[c]);
                    if (clip) d = d < 0 ? 0 : (d > 1 ? 1 : d);
                    o = (float)d;
                } else {
                    o = clip ? (v < 0 ? 0 : (v > 1 ? 1 : v)) : v;
                }
                dst[((long)c * h + y) * w + x] = o;
            }
}

/* bayer2raw, utils/isp_ops.py:98-112 */
void pnnp_oracle_unpack(const float* src, int h, int w, uint16_t* dst, int wp, int bl) {
    const float span = (float)(wp - bl), blf = (float)bl;
    for (int c = 0; c < 4; ++c)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                float v = src[((long)c * h + y) * w + x];
                v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
                volatile float m = v * span;        /* separate float32 roundings (no fma) */
                volatile float s = m + blf;
                dst[(long)(2 * y + PLANE_DY[c]) * (2 * w) + 2 * x + PLANE_DX[c]] = (uint16_t)(int)s;
            }
}

/* ------------------------------------------------------------------ (2) sampler */
enum { P_K = 0, P_SIGGS, P_SIGTL, P_LAM, P_SIGR, P_Q, P_RATIO, P_WP, P_BL, P_BIAS0, NPARAM = 16 };
enum { F_P = 1, F_G = 2, F_R = 4, F_Q = 8, F_D = 16, F_B = 32, F_ORI = 0x100, F_CLIP = 0x200, F_TORCH = 0x1000 };

typedef struct { uint32_t v[4]; } u4;

static u4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u4 o = {{c0, c1, c2, c3}};
    return o;
}

static float unit(uint32_t x) { return ((float)(x >> 9) + 0.5f) * 1.1920928955078125e-07f; }

static float gauss(uint32_t a, uint32_t b) {
    return sqrtf(-2.0f * logf(unit(a))) * cosf(6.28318530717958647692f * unit(b));
}
/* the second normal of the same Box-Muller pair (sin instead of cos): a pixel pair shares the two uniforms */
static float gauss2(uint32_t a, uint32_t b) {
    return sqrtf(-2.0f * logf(unit(a))) * sinf(6.28318530717958647692f * unit(b));
}

typedef struct { uint32_t k0, k1, crop, off; } ctx_t;

/* ln(k!) for integral k >= 0: table below 10, Stirling series above (truncation < 5e-9 at k = 10).  The HIP kernel evaluates the
   same expression (csrc/noise.hip log_factorial); the distribution is pinned against the reference's draws (tier B). */
static float log_factorial(float k) {
    static const float LF[10] = {0.f, 0.f, 0.69314718f, 1.79175947f, 3.17805383f, 4.78749174f, 6.57925121f, 8.52516136f, 10.60460290f, 12.80182748f};
    if (k < 10.f) return LF[(int)k];
    const float x = k + 1.f;
    const float r = 1.f / x;
    return (x - 0.5f) * logf(x) - x + 0.91893853f + 0.083333333f * r - 0.0027777778f * (r * r * r);
}

static float poisson(float lam, uint32_t elem, const ctx_t* c, uint32_t r0, uint32_t r1) {
    if (!(lam > 0.f)) return 0.f;
    if (lam < 10.f) {                       /* sequential inversion on one uniform */
        const float u = unit(r0);
        float p = expf(-lam), s = p, k = 0.f;
        /* the float32 CDF can saturate below the largest uniform (1 - 2^-24); without the s2 == s exit the loop would
           run on to a cap and return a far-out hot pixel about once per 1e7 dark pixels */
        while (u > s) { k += 1.f; p *= lam / k; const float s2 = s + p; if (s2 == s) break; s = s2; }
        return k;
    }
    /* Hoermann PTRS */
    const float slam = sqrtf(lam), loglam = logf(lam);
    const float b = 0.931f + 2.53f * slam;
    const float a = -0.059f + 0.02483f * b;
    const float inv_alpha = 1.1239f + 1.1328f / (b - 3.4f);
    const float vr = 0.9277f - 3.6224f / (b - 2.f);
    uint32_t x0 = r0, x1 = r1;
    u4 extra = {{0, 0, 0, 0}};
    for (int it = 0; it < 64; ++it) {
        if (it > 0) {
            if (it & 1) { extra = philox(elem, c->crop, 8u + ((it + 1) >> 1), c->off, c->k0, c->k1); x0 = extra.v[0]; x1 = extra.v[1]; }
            else { x0 = extra.v[2]; x1 = extra.v[3]; }
        }
        const float U = unit(x0) - 0.5f, V = unit(x1);
        const float us = 0.5f - fabsf(U);
        const float k = floorf((2.f * a / us + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return k;
        if (k < 0.f || (us < 0.013f && V > us)) continue;
        if (logf(V) + logf(inv_alpha) - logf(a / (us * us) + b) <= -lam + k * loglam - log_factorial(k)) return k;
    }
    return floorf(lam + 0.5f);
}

static float tukey(float u, float lam) {
    /* Tukey-lambda quantile (u^lam - (1-u)^lam)/lam (scipy.stats.tukeylambda ppf), written
       with expm1 so small |lam| does not cancel catastrophically in float32 */
    const float lu = logf(u), l1u = log1pf(-u);
    if (lam == 0.f) return lu - l1u;
    return (expm1f(lam * lu) - expm1f(lam * l1u)) / lam;
}

/* SNA_torch (data_process/process.py:562-588) with the HIP kernel's counter RNG; mirrors pnnp_sna_f32. */
void pnnp_oracle_sna(const float* gt, float* dn, float* dy, int C, int H, int W, const float* aug, float K, float wp, float bl,
                     float ratio, int black_lr, int ori, uint64_t seed, uint64_t offset, uint32_t crop) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32), off = (uint32_t)offset;
    const ctx_t cx = {k0, k1, crop, off};
    const float span = wp - bl;
    const long plane = (long)H * W, total = (long)C * plane;
    for (long t = 0; t < total; ++t) {
        const int c = (int)(t / plane) & 3;
        const uint32_t elem = (uint32_t)t;
        const u4 r = philox(elem, crop, 0x20000000u, off, k0, k1);
        volatile float g = gt[t] * span; g = g / ratio;
        volatile float y = g * aug[c];
        volatile float lam = y / K;
        volatile float n = poisson(lam, elem, &cx, r.v[0], r.v[1]) * K;
        if (black_lr) y = y - g;
        y = y * ratio; y = y / span;
        n = n / span;
        if (!ori) n = n * ratio;
        dn[t] = n; dy[t] = y;
    }
}

/* y, out: [B][C][H][W]; params: [B][NPARAM].  Mirrors pnnp_noise_sample_f32 (include/pnnp_hip.h). */
void pnnp_oracle_noise_sample(const float* y, float* out, int B, int C, int H, int W, const float* params,
                              unsigned flags, float mfm, uint64_t seed, uint64_t offset, uint32_t crop_base) {
    const int torch_mode = !!(flags & F_TORCH);
    const int up = !!(flags & F_P), ug = !!(flags & F_G), ur = !!(flags & F_R), uq = !!(flags & F_Q),
              ud = !!(flags & F_D), ub = !!(flags & F_B);
    const int extras = torch_mode || !ub;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32);
    const uint32_t off = (uint32_t)offset;
    for (int b = 0; b < B; ++b) {
        const float* P = params + (long)b * NPARAM;
        const float K = P[P_K], ratio = P[P_RATIO], wp = P[P_WP], bl = P[P_BL];
        const float span = wp - bl;
        const ctx_t cx = {k0, k1, crop_base + (uint32_t)b, off};
        const float sig_read = (ug ? P[P_SIGTL] : P[P_SIGGS]) / mfm;
        const float qscale = torch_mode ? P[P_Q] * span : 1.0f;
        const float lo = (flags & F_CLIP) ? 0.f : -(bl / wp);
        for (int c = 0; c < C; ++c)
            for (int h = 0; h < H; ++h) {
                float row = 0.f;
                if (ur && extras) {
                    const u4 r = philox((uint32_t)(c * H + h), cx.crop, 0x40000000u, off, k0, k1);
                    row = gauss(r.v[0], r.v[1]) * P[P_SIGR] / mfm;
                }
                /* Draws of a pixel QUAD (4 consecutive pixels of a row, the last quad of a row may be short) come from blocks keyed by
                   the quad's first element: slot 0 -> the shot uniforms U of pixels 0..3 (no 'p': two Box-Muller pairs, cos / sin),
                   slot 1 -> read noise (two Box-Muller pairs: (x,y) cos -> pixel 0, sin -> pixel 1; (z,w) -> pixels 2, 3; Tukey: one
                   uniform each), slot 2 -> the second uniform V of the first PTRS round, slot 3 -> quantisation.  Later PTRS rounds
                   of a pixel take blocks keyed by the pixel's own element, slots 9.. (poisson()). */
                u4 b0 = {{0, 0, 0, 0}}, b1 = {{0, 0, 0, 0}}, b2 = {{0, 0, 0, 0}}, b3 = {{0, 0, 0, 0}};
                for (int w = 0; w < W; ++w) {
                    const long i = (((long)b * C + c) * H + h) * W + w;
                    const uint32_t elem = (uint32_t)((c * H + h) * (long)W + w);
                    const int j = w & 3;
                    if (j == 0) {
                        b0 = philox(elem, cx.crop, 0u, off, k0, k1); b1 = philox(elem, cx.crop, 1u, off, k0, k1);
                        b2 = philox(elem, cx.crop, 2u, off, k0, k1); b3 = philox(elem, cx.crop, 3u, off, k0, k1);
                    }
                    float yy = y[i] * span;
                    yy = yy / ratio;
                    float shot;
                    if (up) {
                        const float lam = mfm * yy / K;
                        shot = poisson(lam, elem, &cx, b0.v[j], b2.v[j]) * K / mfm;
                    } else {
                        const float n = (j & 1) ? gauss2(b0.v[j & 2], b0.v[(j & 2) + 1]) : gauss(b0.v[j & 2], b0.v[(j & 2) + 1]);
                        float s = yy / K; s = sqrtf(s > 1e-10f ? s : 1e-10f);
                        shot = yy + n * s * K / mfm;
                    }
                    float acc = shot;
                    if (!ub) {
                        const float rd = ug ? tukey(unit(b1.v[j]), P[P_LAM])
                                            : ((j & 1) ? gauss2(b1.v[j & 2], b1.v[(j & 2) + 1]) : gauss(b1.v[j & 2], b1.v[(j & 2) + 1]));
                        acc = acc + rd * sig_read;
                    }
                    if (extras) {
                        if (ur) acc = acc + row;
                        if (uq) {
                            acc = acc + (unit(b3.v[j]) - 0.5f) * qscale;
                        }
                        if (ud) acc = acc + P[P_BIAS0 + (c & 3)];
                    }
                    float z = acc / span;
                    z = z < lo ? lo : (z > 1.f ? 1.f : z);
                    if (!(flags & F_ORI)) z = z * ratio;
                    if (flags & 0x4000u) z = z < 0.f ? 0.f : z;    /* trainer_SID.py:481-485 clamp */
                    if (flags & 0x2000u) z = z > 1.f ? 1.f : z;
                    out[i] = z;
                }
            }
    }
}

/* Raw Philox block, for the known-answer test against the published test vectors. */
void pnnp_oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    const u4 r = philox(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    memcpy(out, r.v, sizeof r.v);
}
