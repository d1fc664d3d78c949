// Micro-benchmark: does a wave's VALU / LDS work overlap its own in-flight MFMAs on gfx950, and what does the chip
// clock do under each mix?  One wave per SIMD (256 VGPR accumulators like the Winograd kernels), whole chip busy.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_overlap mfma_overlap.hip && ./mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NVALU, int NLDS, int INDEP>
__global__ void __launch_bounds__(256, 1) k(float* out, long long* t, int iters) {
    __shared__ float lds[16384];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += 256) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 acc[16];
    for (int x = 0; x < 16; ++x) for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
    float a = tid * 1e-3f, b = 1.0001f, v[8];
    for (int j = 0; j < 8; ++j) v[j] = tid + j;
    float l0 = 0.f;
    const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int xx = 0; xx < 16; ++xx) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int x = INDEP == 1 ? ((xx * 4 + rr) & 15) : (INDEP == 2 ? ((xx & ~1) | (rr & 1)) : xx), r = rr;      // 1: round-robin over 16 accumulators, 2: pairs a,b,a,b
                acc[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[x], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NVALU; ++j) v[j & 7] = fmaf(v[j & 7], 1.0001f, 0.5f);
#pragma unroll
                for (int j = 0; j < NLDS; ++j) l0 += lds[(tid * 4 + (it + x * 4 + r + j) * 64) & 16383];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = l0;
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int x = 0; x < 16; ++x) for (int e = 0; e < 16; ++e) s += acc[x][e];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NVALU, int NLDS, int INDEP>
void run(const char* name) {
    const int G = 256, iters = 2000;
    float* out; long long* t;
    hipMalloc(&out, G * 256 * 4); hipMalloc(&t, G * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NVALU, NLDS, INDEP>), dim3(G), dim3(256), 0, 0, out, t, 200);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NVALU, NLDS, INDEP>), dim3(G), dim3(256), 0, 0, out, t, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * G); hipMemcpy(h.data(), t, G * 16, hipMemcpyDeviceToHost);
    const double nm = 64.0 * iters;
    printf("%-28s %8.3f ms  %7.1f ns/MFMA  cyc-counter/MFMA %7.1f  wallclk/MFMA %7.2f  -> %5.1f TF executed\n", name, ms, ms * 1e6 / nm,
           h[0] / nm, h[1] / nm, G * 4 * nm * 4096.0 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(t);
}

int main() {
    run<0, 0, 0>("dep chain: mfma only");
    run<4, 0, 0>("dep chain: + 4 valu");
    run<8, 0, 0>("dep chain: + 8 valu");
    run<16, 0, 0>("dep chain: + 16 valu");
    run<0, 0, 1>("indep accs: mfma only");
    run<4, 0, 1>("indep accs: + 4 valu");
    run<8, 0, 1>("indep accs: + 8 valu");
    run<12, 0, 1>("indep accs: + 12 valu");
    run<16, 0, 1>("indep accs: + 16 valu");
    run<0, 0, 2>("pairs a,b,a,b: mfma only");
    run<4, 0, 2>("pairs a,b,a,b: + 4 valu");
    return 0;
}
