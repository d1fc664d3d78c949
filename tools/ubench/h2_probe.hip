// Stage A of the fp16x2 ("h2") family: what the gfx950 matrix core does with fp16 operands, before any kernel is
// built on it.  A float32 value a, scaled by a power of two s into fp16's range, splits into hi = f16(s a), lo = f16(s a - hi) (22-23
// significand bits); a product a b is then hi hi' + hi lo' + lo hi' -- three fp16 products instead of bf16x3's six.
//   (1) rate: v_mfma_f32_16x16x32_f16 against ..._bf16 on random operands, registers only, 2 waves per SIMD (clock / power limited);
//   (2) rounding of the accumulation (as tools/ubench/mfma_round.hip did for bf16);
//   (3) fp16 subnormal operands: kept or flushed?
//   (4) the split instructions: v_fma_mixlo_f16 / v_fma_mixhi_f16 (scale, round to nearest even, residual -- one instruction each) against the host;
//   (5) dot products of length K through the three schemes (fp32 MFMA 16x16x4, bf16x3, h2) against float64: relative L2 and signed mean.
//   hipcc --offload-arch=gfx950 -O3 -o h2_probe h2_probe.hip && ./h2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned hsh(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// ---------------------------------------------------------------- (1) rate
template <int F16>
__global__ void __launch_bounds__(512) rate_k(float* out, int iters, int nprod) {
    const unsigned h0 = hsh(threadIdx.x * 977u + blockIdx.x);
    u32x4 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            const unsigned h = hsh(h0 + i * 16 + e), g = hsh(h ^ 0x9e3779b9u);
            // two random values in [1, 2) with random sign per dword: fp16 0x3c00 | 10-bit mantissa, bf16 0x3f80 | 7-bit mantissa
            a[i][e] = F16 ? ((0x3c003c00u | (h & 0x03ff03ffu)) ^ (h & 0x80008000u)) : ((0x3f803f80u | (h & 0x007f007fu)) ^ (h & 0x80008000u));
            b[i][e] = F16 ? ((0x3c003c00u | (g & 0x03ff03ffu)) ^ (g & 0x80008000u)) : ((0x3f803f80u | (g & 0x007f007fu)) ^ (g & 0x80008000u));
        }
    f32x4 acc[16];
    for (int x = 0; x < 16; ++x) acc[x] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int p = 0; p < 6; ++p) {
                    if (p >= nprod) continue;
                    if (F16) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[(i + p) & 3]), __builtin_bit_cast(f16x8, b[(j + p) & 3]), acc[i * 4 + j], 0, 0, 0);
                    else acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[(i + p) & 3]), __builtin_bit_cast(bf16x8, b[(j + p) & 3]), acc[i * 4 + j], 0, 0, 0);
                }
    }
    float s = 0.f;
    for (int x = 0; x < 16; ++x) for (int e = 0; e < 4; ++e) s += acc[x][e];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int F16>
void rate(const char* name, int nprod) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 40000 * 3 / nprod;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate_k<F16>), dim3(256), dim3(512), 0, 0, out, iters, nprod);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double mf = 256.0 * 8 * iters * 16.0 * nprod;        // MFMAs
    printf("rate %-28s %8.3f ms  %7.1f TFLOP/s executed (%5.1f %% of 2516.6)  = %6.1f T float32-product-FLOP/s\n", name, ms, mf * 16384 / ms / 1e9,
           mf * 16384 / ms / 1e9 / 25.166, mf * 16384 / ms / 1e9 / nprod * (F16 ? 1.0 : 1.0));
    hipFree(out);
}

// ---------------------------------------------------------------- (2) rounding, (3) subnormals
__global__ void round_k(float* out, float c0, float v) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)1.0f; b[i] = (_Float16)(v / 32.0f); }
    f32x4 c4 = {c0, c0, c0, c0};
    f32x4 d4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
    f16x8 b5; for (int i = 0; i < 8; ++i) b5[i] = (_Float16)((i == 0 && threadIdx.x < 16) ? v - 31.0f * 0.0078125f : 0.0078125f);
    f32x4 d5 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b5, c4, 0, 0, 0);
    bf16x8 ab, bb; for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)1.0f; bb[i] = (__bf16)(v / 32.0f); }
    f32x4 d6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c4, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = d4[0]; out[1] = d5[0]; out[2] = d6[0]; }
}
__global__ void subn_k(float* out) {
    // A = one fp16 subnormal per lane's k = 0 (bits given), B = 2^10: the dot product is exact in float32 whatever the core does
    const unsigned short bits[4] = {0x0001, 0x0200, 0x03ff, 0x0400};      // 2^-24, 2^-15, largest subnormal, smallest normal
    for (int t = 0; t < 4; ++t) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
        if (threadIdx.x < 16) { a[0] = __builtin_bit_cast(_Float16, bits[t]); b[0] = (_Float16)1024.0f; }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
        // subnormal x subnormal-free B on the OTHER side too (B subnormal, A = 2^10)
        f32x4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c, 0, 0, 0);
        if (threadIdx.x == 0) { out[2 * t] = d[0]; out[2 * t + 1] = d2[0]; }
    }
}

// ---------------------------------------------------------------- (4) the split
__device__ __forceinline__ void split_h2(float a0, float a1, float s, unsigned& hi, unsigned& lo) {
    // hi = f16(a s) (RNE) for both values, packed; lo = f16(a s - hi): the fma is exact in float32 (Sterbenz), one rounding to fp16
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a0), "v"(s), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(a1), "v"(s), "v"(h));
    hi = h; lo = l;
}
__global__ void split_k(const float* x, float s, unsigned* hi, unsigned* lo, int n2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    unsigned h, l;
    split_h2(x[2 * i], x[2 * i + 1], s, h, l);
    hi[i] = h; lo[i] = l;
}
static unsigned short host_f16(float f) { _Float16 h = (_Float16)f; unsigned short u; __builtin_memcpy(&u, &h, 2); return u; }
static float host_f16f(unsigned short u) { _Float16 h; __builtin_memcpy(&h, &u, 2); return (float)h; }

// ---------------------------------------------------------------- (5) dot products
// One wave: D[16][16] = A[16][K] B[K][16].  Lane l feeds row / column l & 15, k-values 8 (l >> 4) .. + 7 of every 32-step.
template <int MODE>     // 0: fp32 MFMA 16x16x4, 1: bf16x3 (six products, smallest first), 2: h2 (three products)
__global__ void dot_k(const float* A, const float* B, float* D, int K, float sa, float sb) {
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * K + k + q], B[(k + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            float av[8], bv[8];
            for (int i = 0; i < 8; ++i) { av[i] = A[r * K + k + 8 * q + i]; bv[i] = B[(k + 8 * q + i) * 16 + r]; }
            if (MODE == 1) {
                bf16x8 ap[3], bp[3];
                for (int i = 0; i < 8; ++i) {
                    float t = av[i]; for (int p = 0; p < 3; ++p) { ap[p][i] = (__bf16)t; t -= (float)ap[p][i]; }
                    t = bv[i]; for (int p = 0; p < 3; ++p) { bp[p][i] = (__bf16)t; t -= (float)bp[p][i]; }
                }
#define M(PA, PB) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ap[PA], bp[PB], acc, 0, 0, 0)
                M(0, 2); M(2, 0); M(1, 1); M(0, 1); M(1, 0); M(0, 0);
#undef M
            } else {
                u32x4 ah, al, bh, bl;
                for (int i = 0; i < 4; ++i) { unsigned h_, l_; split_h2(av[2 * i], av[2 * i + 1], sa, h_, l_); ah[i] = h_; al[i] = l_; split_h2(bv[2 * i], bv[2 * i + 1], sb, h_, l_); bh[i] = h_; bl[i] = l_; }
#define M(X, Y) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, X), __builtin_bit_cast(f16x8, Y), acc, 0, 0, 0)
                M(ah, bl); M(al, bh); M(ah, bh);
#undef M
            }
        }
        if (MODE == 2) { const float inv = 1.0f / sa; const float inv2 = 1.0f / sb; for (int i = 0; i < 4; ++i) acc[i] = acc[i] * inv * inv2; }
    }
    for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + r] = acc[i];
}

static float pow2_scale(float amax) { int e; frexpf(amax, &e); return ldexpf(1.0f, 15 - e); }      // s amax in [2^14, 2^15)

int main() {
    float* out; hipMalloc(&out, 64);
    // (1)
    for (int rep = 0; rep < 2; ++rep) {
        rate<0>("bf16 16x16x32, 6 per pair", 6);
        rate<1>("f16  16x16x32, 6 per pair", 6);
        rate<1>("f16  16x16x32, 3 per pair", 3);
    }
    // (2)
    const float cases[6][2] = {{16777216.f, 3.f}, {-16777216.f, -3.f}, {16777216.f, 1.f}, {-16777216.f, -1.f}, {16777216.f, 5.f}, {-16777216.f, -5.f}};
    printf("%14s %4s | %14s %14s %14s | exact\n", "C", "v", "f16 equal", "f16 unequal", "bf16 equal");
    for (auto& cs : cases) {
        hipLaunchKernelGGL(round_k, dim3(1), dim3(64), 0, 0, out, cs[0], cs[1]);
        float h[3]; hipMemcpy(h, out, 12, hipMemcpyDeviceToHost);
        printf("%14.1f %4.0f | %14.1f %14.1f %14.1f | %.1f\n", cs[0], cs[1], h[0], h[1], h[2], (double)cs[0] + cs[1]);
    }
    // (3)
    {
        hipLaunchKernelGGL(subn_k, dim3(1), dim3(64), 0, 0, out);
        float h[8]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        const double want[4] = {ldexp(1.0, -24) * 1024, ldexp(1.0, -15) * 1024, (1023.0 / 1024) * ldexp(1.0, -14) * 1024, ldexp(1.0, -14) * 1024};
        for (int t = 0; t < 4; ++t) printf("subnormal operand %d: as A %.9g, as B %.9g, exact %.9g\n", t, h[2 * t], h[2 * t + 1], want[t]);
    }
    // (4)
    {
        const int n = 1 << 16;
        std::vector<float> x(n);
        for (int i = 0; i < n; ++i) {
            const unsigned h = hsh(i);
            const float m = 1.0f + (float)(hsh(h) & 0x7fffff) / 8388608.0f;
            x[i] = ldexpf((h & 1) ? -m : m, (int)((h >> 1) % 60) - 45);      // 2^-45 .. 2^15 before the scale
        }
        x[0] = 0.f; x[1] = -0.f; x[2] = 1e-41f; x[3] = 65504.f; x[4] = 32768.f - 1.f / 512; x[5] = 1.0f + 1.0f / 2048; x[6] = 1.0f + 3.0f / 2048;
        float* dx; unsigned *dh, *dl; hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2);
        hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
        const float s = 1.0f;
        hipLaunchKernelGGL(split_k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, s, dh, dl, n / 2);
        std::vector<unsigned short> hh(n), hl(n);
        hipMemcpy(hh.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hl.data(), dl, n * 2, hipMemcpyDeviceToHost);
        int bad_hi = 0, bad_lo = 0; double worst = 0;
        for (int i = 0; i < n; ++i) {
            const unsigned short eh = host_f16(x[i] * s);
            const unsigned short el = host_f16(x[i] * s - host_f16f(eh));
            if (eh != hh[i] && !(x[i] == 0.f)) { if (bad_hi < 5) printf("  hi mismatch x=%.9g got %04x want %04x\n", x[i], hh[i], eh); ++bad_hi; }
            if (el != hl[i] && (host_f16f(el) != 0.f || host_f16f(hl[i]) != 0.f)) { if (bad_lo < 5) printf("  lo mismatch x=%.9g got %04x want %04x\n", x[i], hl[i], el); ++bad_lo; }
            const double rec = (double)host_f16f(hh[i]) + (double)host_f16f(hl[i]);
            if (fabs(x[i]) >= ldexp(1.0, -3) && fabs(x[i]) < 65504) worst = fmax(worst, fabs(rec - (double)x[i] * s) / fabs((double)x[i] * s));
        }
        printf("split (v_fma_mixlo/hi_f16): %d hi and %d lo mismatches against the host's RNE conversion over %d values (2^-45 .. 2^15, zeros, float32 subnormal);\n"
               "  worst |hi + lo - x| / |x| for |x| >= 2^-3: %.3e (2^-22 = %.3e)\n", bad_hi, bad_lo, n, worst, ldexp(1.0, -22));
        printf("  x = 1e-41 (float32 subnormal): hi %04x lo %04x;  x = 65504: hi %04x lo %04x\n", hh[2], hl[2], hh[3], hl[3]);
    }
    // (5)
    {
        float *dA, *dB, *dD; const int KM = 36864;
        hipMalloc(&dA, 16 * KM * 4); hipMalloc(&dB, 16 * KM * 4); hipMalloc(&dD, 1024);
        printf("%-34s %7s | %-23s | %-23s | %-23s\n", "dot products D = A B vs float64", "K", "fp32 MFMA  L2 / mean", "bf16x3     L2 / mean", "h2         L2 / mean");
        for (int pattern = 0; pattern < 4; ++pattern)
            for (int K : {288, 1152, 4608, 36864}) {
                std::vector<float> A(16 * K), B(16 * K);
                double l2[3] = {0, 0, 0}, sm[3] = {0, 0, 0}, rn = 0, ra = 0;
                const int trials = 8;
                for (int tr = 0; tr < trials; ++tr) {
                    srand(1234 + tr * 77 + K + pattern * 13);
                    auto rnd = [&]() { return (float)((rand() & 0xffffff) + 0.5) / 16777216.0f; };
                    auto gauss = [&]() { return sqrtf(-2.0f * logf(rnd())) * cosf(6.2831853f * rnd()); };
                    for (int i = 0; i < 16 * K; ++i) {
                        const int k = i % K, kb = i / 16;
                        if (pattern == 0) { A[i] = gauss(); B[i] = gauss() * 0.05f; }                                          // random sign
                        else if (pattern == 1) { A[i] = rnd() + 0.5f; B[i] = (rnd() + 0.5f) * 0.05f; }                        // all positive
                        else if (pattern == 2) { A[i] = gauss() * powf(10.f, -4.f + 8.f * (float)((k * 7) % 64) / 63.f); B[i] = gauss() * 0.05f; }     // 8 decades along K
                        else { A[i] = gauss() * 1e-30f; B[i] = gauss() * 3e-4f; }                                               // far outside fp16's own range
                        (void)kb;
                    }
                    float amA = 0, amB = 0;
                    for (int i = 0; i < 16 * K; ++i) { amA = fmaxf(amA, fabsf(A[i])); amB = fmaxf(amB, fabsf(B[i])); }
                    hipMemcpy(dA, A.data(), 16 * K * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 16 * K * 4, hipMemcpyHostToDevice);
                    std::vector<double> ref(256, 0.0);
                    for (int r = 0; r < 16; ++r) for (int c = 0; c < 16; ++c) { double s = 0; for (int k = 0; k < K; ++k) s += (double)A[r * K + k] * (double)B[k * 16 + c]; ref[r * 16 + c] = s; }
                    for (int m = 0; m < 3; ++m) {
                        if (m == 0) hipLaunchKernelGGL(dot_k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1.f, 1.f);
                        if (m == 1) hipLaunchKernelGGL(dot_k<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1.f, 1.f);
                        if (m == 2) hipLaunchKernelGGL(dot_k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, pow2_scale(amA), pow2_scale(amB));
                        float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
                        for (int i = 0; i < 256; ++i) { const double d = D[i] - ref[i]; l2[m] += d * d; sm[m] += d; }
                    }
                    for (int i = 0; i < 256; ++i) { rn += ref[i] * ref[i]; ra += fabs(ref[i]); }
                }
                const char* pn[4] = {"random sign", "all positive", "8 decades along K", "A at 1e-30"};
                printf("%-34s %7d | %.2e / %+.2e | %.2e / %+.2e | %.2e / %+.2e\n", pn[pattern], K, sqrt(l2[0] / rn), sm[0] / ra, sqrt(l2[1] / rn), sm[1] / ra, sqrt(l2[2] / rn), sm[2] / ra);
            }
    }
    return 0;
}
