// Micro-benchmark 2: two waves per SIMD.  Waves 0-3 run only MFMAs, waves 4-7 only VALU (or LDS) work.
// If the SIMD overlaps them, each finishes in its own stand-alone time; if not, times add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: both roles, 1: MFMA waves only (others exit), 2: VALU waves only, 3: LDS waves instead of VALU, 4: LDS only
__global__ void __launch_bounds__(512) k(float* out, long long* t, int iters) {
    __shared__ float lds[16384];
    const int tid = threadIdx.x, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 512) lds[i] = i * 1e-6f;
    __syncthreads();
    float s = 0.f;
    const long long c0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (MODE == 0 || MODE == 1 || MODE == 3) {
            f32x16 acc[4];
            for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) acc[x][e] = 0.f;
            float a = tid * 1e-3f, b = 1.0001f;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r & 3], 0, 0, 0);
            for (int x = 0; x < 4; ++x) for (int e = 0; e < 16; ++e) s += acc[x][e];
        }
    } else {
        if (MODE == 0 || MODE == 2) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = tid + j;
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 16 * 16; ++r) v[r & 7] = fmaf(v[r & 7], 1.0001f, 0.5f);     // 16 VALU per partner MFMA
            for (int j = 0; j < 8; ++j) s += v[j];
        } else if (MODE == 3 || MODE == 4) {
            float4 l = make_float4(0, 0, 0, 0);
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int r = 0; r < 16 * 2; ++r) {                                                 // 2 ds_read_b128 per partner MFMA
                    const float4 q = *reinterpret_cast<const float4*>(&lds[((tid & 63) * 4 + (it + r) * 256) & 16383]);
                    l.x += q.x; l.y += q.y; l.z += q.z; l.w += q.w;
                }
            s += l.x + l.y + l.z + l.w;
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) t[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int MODE>
void run(const char* name) {
    const int G = 256, iters = 4000;
    float* out; long long* t;
    hipMalloc(&out, G * 512 * 4); hipMalloc(&t, G * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(G), dim3(512), 0, 0, out, t, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(G), dim3(512), 0, 0, out, t, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(G * 8); hipMemcpy(h.data(), t, G * 64, hipMemcpyDeviceToHost);
    printf("%-44s %8.3f ms | cycles per MFMA slot: mfma wave %7.1f   other wave %7.1f\n", name, ms, h[0] / (16.0 * iters), h[4] / (16.0 * iters));
    hipFree(out); hipFree(t);
}

int main() {
    run<1>("MFMA waves alone");
    run<2>("VALU waves alone (16 fma per slot)");
    run<0>("MFMA waves + VALU waves on the same SIMDs");
    run<4>("LDS waves alone (2 ds_read_b128 per slot)");
    run<3>("MFMA waves + LDS waves on the same SIMDs");
    return 0;
}
