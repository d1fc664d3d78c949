// Can VALU work run UNDER the bf16 matrix pipe on gfx950?  One workgroup per CU; per SIMD: M waves that issue only v_mfma_f32_16x16x32_bf16
// (4 independent accumulator chains, operands in registers) and V waves that issue only the halo-staging arithmetic of csrc/conv_x3.hip
// (v_cvt_pk_bf16_f32 / shift / and / subtract chains), or waves that do BOTH interleaved (k VALU instructions behind every MFMA, fenced).
// Prints cycles per MFMA per SIMD for every mix: if the pipes are independent, "M + V waves" costs max(M alone, V alone), not the sum.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_coissue mfma_valu_coissue.hip && ./mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned cvt_pk(float a, float b) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// one "staging unit" = the split of one float pair into hi / mid / lo words: 11 VALU instructions
__device__ __forceinline__ void split_pair(float& a0, float& a1, unsigned& acc) {
    const unsigned h = cvt_pk(a0, a1);
    float r0 = a0 - __uint_as_float(h << 16), r1 = a1 - __uint_as_float(h & 0xffff0000u);
    const unsigned m = cvt_pk(r0, r1);
    float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
    const unsigned l = cvt_pk(s0, s1);
    acc ^= h ^ m ^ l; a0 += 1.0f; a1 += 0.5f;               // (+2: 13 VALU per unit, keeps the inputs changing)
}

// roles per wave (wave w sits on SIMD w & 3): NM MFMA-only waves per SIMD, NV VALU-only waves per SIMD, K VALU units interleaved in MFMA waves
template <int NM, int NV, int KPER8>            // KPER8: staging units per 8 MFMAs inside the MFMA waves (0 = none)
__global__ void __launch_bounds__((NM + NV) * 256) k(float* out, long long* t, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_m = wave < NM * 4;
    long long c0 = 0, c1 = 0;
    float res = 0.f;
    if (is_m) {
        f32x4 acc[4] = {};
        u32x4 a = {0x3f803f80u + lane, 0x3f813f80u, 0x3f823f80u, 0x3f833f80u}, b = {0x3f843f80u, 0x3f853f80u + lane, 0x3f863f80u, 0x3f873f80u};
        float p0 = 1.f + lane, p1 = 2.f + lane; unsigned x = 0;
        __syncthreads();
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                acc[g & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[g & 3], 0, 0, 0);
                if (KPER8 > 0 && (g * KPER8) / 8 != ((g + 1) * KPER8) / 8) split_pair(p0, p1, x);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        c1 = clock64();
        res = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + __uint_as_float(x & 0x3fffffff);
    } else {
        float p0 = 1.f + lane, p1 = 2.f + lane, q0 = 3.f, q1 = 4.f; unsigned x = 0;
        __syncthreads();
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { split_pair(p0, p1, x); split_pair(q0, q1, x); }
        }
        c1 = clock64();
        res = __uint_as_float(x & 0x3fffffff);
    }
    out[blockIdx.x * blockDim.x + tid] = res;
    if (lane == 0) t[blockIdx.x * 16 + wave] = c1 - c0;
}

template <int NM, int NV, int KPER8>
void run(const char* name, int iters) {
    float* out; long long* t;
    const int nth = (NM + NV) * 256;
    hipMalloc(&out, 256 * nth * 4); hipMalloc(&t, 256 * 16 * 8);
    hipMemset(t, 0, 256 * 16 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NM, NV, KPER8>), dim3(256), dim3(nth), 0, 0, out, t, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(256 * 16);
    hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < 256; ++b) { for (int w = 0; w < NM * 4; ++w) tm += h[b * 16 + w]; for (int w = NM * 4; w < (NM + NV) * 4; ++w) tv += h[b * 16 + w]; }
    if (NM) tm /= 256.0 * NM * 4; if (NV) tv /= 256.0 * NV * 4;
    const double mf = 8.0 * iters, vu = 8.0 * iters;
    printf("%-58s %7.3f ms |", name, ms);
    if (NM) printf(" M wave: %5.1f cycles per MFMA (%4.1f per SIMD)", tm / mf, tm / mf / NM);
    if (NV) printf(" | V wave: %5.1f cycles per 13-VALU unit", tv / vu);
    if (NM) printf(" | %6.0f TF", 256.0 * NM * 4 * mf * 16384 / ms / 1e9);
    printf("\n");
    hipFree(out); hipFree(t);
}

int main() {
    const int it = 20000;
    run<1, 0, 0>("1 MFMA wave / SIMD", it);
    run<2, 0, 0>("2 MFMA waves / SIMD", it);
    run<0, 1, 0>("1 VALU wave / SIMD", it);
    run<0, 2, 0>("2 VALU waves / SIMD", it);
    run<1, 1, 0>("1 MFMA wave + 1 VALU wave / SIMD", it);
    run<2, 1, 0>("2 MFMA waves + 1 VALU wave / SIMD", it);
    run<2, 2, 0>("2 MFMA waves + 2 VALU waves / SIMD", it);
    run<1, 0, 4>("1 wave / SIMD, 4 units per 8 MFMAs interleaved", it);
    run<1, 0, 8>("1 wave / SIMD, 8 units per 8 MFMAs interleaved", it);
    run<2, 0, 4>("2 waves / SIMD, 4 units per 8 MFMAs interleaved", it);
    run<2, 0, 8>("2 waves / SIMD, 8 units per 8 MFMAs interleaved", it);
    run<2, 0, 2>("2 waves / SIMD, 2 units per 8 MFMAs interleaved", it);
    return 0;
}
