// Micro-benchmark: bf16 MFMA (32x32x16) issue with one / two waves per SIMD, with and without a workgroup barrier every
// 72 MFMAs, operands from registers or re-read from LDS (12 ds_read_b128 per 24 MFMAs, like csrc/conv_x3.hip).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_share mfma_share.hip && ./mfma_share
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// MODE bit0: barrier per item; bit1: operands from LDS; bit2: s_setprio 1 for waves 4-7
template <int NW, int MODE>
__global__ void __launch_bounds__(NW * 64) k(float* out, long long* t, int items) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* L = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += NW * 64) L[i] = u32x4{0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    __syncthreads();
    if ((MODE & 4) && wave >= 4) __builtin_amdgcn_s_setprio(1);
    f32x16 acc[4];
    for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
    u32x4 a[2][3], b[2][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) { a[i][p] = L[lane + 64 * (i * 3 + p)]; b[i][p] = L[lane + 64 * (6 + i * 3 + p)]; }
    long long busy = 0;
    const long long c0 = clock64();
    for (int it = 0; it < items; ++it) {
        if (MODE & 1) __syncthreads();
        const long long s0 = clock64();
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        a[i][p] = L[((it + tp) & 7) * 384 + lane + 64 * (i * 3 + p)];
                        b[i][p] = L[((it + tp) & 7) * 384 + lane + 64 * (6 + i * 3 + p) - 384 * 0];
                    }
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
        busy += clock64() - s0;
    }
    const long long c1 = clock64();
    float s = 0.f;
    for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) s += acc[x][e];
    out[blockIdx.x * NW * 64 + tid] = s;
    if (lane == 0) { t[(blockIdx.x * NW + wave) * 2] = c1 - c0; t[(blockIdx.x * NW + wave) * 2 + 1] = busy; }
}

template <int NW, int MODE>
void run(const char* name, int items) {
    float* out; long long* t;
    hipMalloc(&out, 256 * NW * 64 * 4); hipMalloc(&t, 256 * NW * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NW, MODE>), dim3(256), dim3(NW * 64), 65536, 0, out, t, items);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256 * NW * 2);
    hipMemcpy(h.data(), t, h.size() * 8, hipMemcpyDeviceToHost);
    const double mf = 72.0 * items;
    printf("%-44s %7.3f ms  %6.1f TF(bf16)  cycles/MFMA per SIMD: %5.1f   wave0 total %lld busy %lld | wave%d total %lld busy %lld\n", name, ms,
           256.0 * NW * mf * 32768 / ms / 1e9, (double)h[0] / (mf * (NW / 4)), h[0], h[1], NW - 1, h[(NW - 1) * 2], h[(NW - 1) * 2 + 1]);
    hipFree(out); hipFree(t);
}

int main() {
    const int items = 2000;
    run<4, 0>("4 waves, regs, no barrier", items);
    run<4, 1>("4 waves, regs, barrier/72", items);
    run<4, 2>("4 waves, LDS operands, no barrier", items);
    run<4, 3>("4 waves, LDS operands, barrier/72", items);
    run<8, 0>("8 waves, regs, no barrier", items);
    run<8, 1>("8 waves, regs, barrier/72", items);
    run<8, 2>("8 waves, LDS operands, no barrier", items);
    run<8, 3>("8 waves, LDS operands, barrier/72", items);
    run<8, 7>("8 waves, LDS, barrier, setprio(1) on waves 4-7", items);
    return 0;
}
