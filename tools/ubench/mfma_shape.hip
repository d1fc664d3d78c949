// Micro-benchmark: the two bf16 MFMA shapes under the POWER limit -- v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 on
// random operands re-read from LDS (ds_read_b128), two waves per SIMD, the bf16x3 product pattern (6 MFMAs per operand pair), same
// FLOP, same LDS bytes and same accumulator registers per wave: a 64 x 64 output tile per wave.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip && ./mfma_shape [zero]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ unsigned rnd_bf16x2(unsigned i, int zero) {          // two bf16 in [1, 2) with random mantissa and sign
    if (zero) return 0u;
    const unsigned h = hash(i);
    return (0x3f803f80u | (h & 0x007f007fu)) ^ (h & 0x80008000u);
}

template <int SHAPE>   // 0: 32x32x16, 1: 16x16x32
__global__ void __launch_bounds__(512) k(float* out, int items, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* L = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8192; i += 512) L[i] = u32x4{rnd_bf16x2(4 * i, zero), rnd_bf16x2(4 * i + 1, zero), rnd_bf16x2(4 * i + 2, zero), rnd_bf16x2(4 * i + 3, zero)};
    __syncthreads();
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[4];
        for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
        for (int it = 0; it < items; ++it) {
            // two K = 16 steps = the K = 32 step of the other shape
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                u32x4 a[2][3], b[2][3];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        a[i][p] = L[((it * 2 + kk) & 7) * 768 + lane + 64 * (i * 3 + p)];
                        b[i][p] = L[((it * 2 + kk) & 7) * 768 + lane + 64 * (6 + i * 3 + p)];
                    }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#define M(PA, PB) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][PA]), __builtin_bit_cast(bf16x8, b[j][PB]), acc[i * 2 + j], 0, 0, 0)
                        M(0, 2); M(2, 0); M(1, 1); M(0, 1); M(1, 0); M(0, 0);
#undef M
                    }
            }
        }
        for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) s += acc[x][e];
    } else {
        f32x4 acc[16];
        for (int x = 0; x < 16; ++x) for (int e = 0; e < 4; ++e) acc[x][e] = 0.f;
        for (int it = 0; it < items; ++it) {
            u32x4 a[4][3], b[4][3];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = L[(it & 3) * 1536 + lane + 64 * (i * 3 + p)];
                    b[i][p] = L[(it & 3) * 1536 + lane + 64 * (12 + i * 3 + p)];
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#define M(PA, PB) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i][PA]), __builtin_bit_cast(bf16x8, b[j][PB]), acc[i * 4 + j], 0, 0, 0)
                    M(0, 2); M(2, 0); M(1, 1); M(0, 1); M(1, 0); M(0, 0);
#undef M
                }
        }
        for (int x = 0; x < 16; ++x) for (int e = 0; e < 4; ++e) s += acc[x][e];
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int SHAPE>
void run(const char* name, int items, int zero) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE>), dim3(256), dim3(512), 131072, 0, out, items, zero);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 256.0 * 8 * items * 96.0 * 16384;     // per wave and iteration: 96 x 16384 = 48 x 32768
    printf("%-22s %s operands  %8.3f ms  %7.1f TFLOP/s (bf16)  = %5.1f %% of 2516.6\n", name, zero ? "zero  " : "random", ms, flop / ms / 1e9, flop / ms / 1e9 / 25.166);
    hipFree(out);
}

int main(int argc, char** argv) {
    const int items = 20000;
    for (int zero = 0; zero < 2; ++zero) {
        run<0>("32x32x16 (48 / iter)", items, zero);
        run<1>("16x16x32 (96 / iter)", items, zero);
        run<0>("32x32x16 (48 / iter)", items, zero);
        run<1>("16x16x32 (96 / iter)", items, zero);
    }
    return 0;
}
