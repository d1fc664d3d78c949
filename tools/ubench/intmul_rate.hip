// Micro-benchmark: issue rate of 32-bit integer multiplies on gfx950 (what a Philox4x32 round costs).
//   hipcc --offload-arch=gfx950 -O3 -o intmul_rate intmul_rate.hip && ./intmul_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, long long* t, int iters) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + 1u;
    unsigned long long w[4] = {a[0], a[1], a[2], a[3]};
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = a[i] * 0xD2511F53u + 1u;                       // v_mul_lo_u32 (+ add)
                else if (OP == 1) a[i] = __umulhi(a[i], 0xCD9E8D57u) + 1u;         // v_mul_hi_u32 (+ add)
                else if (OP == 2) a[i] = (a[i] ^ 0x9E3779B9u) + (a[i] >> 3);       // 3 simple VALU
                else if (OP == 3) { unsigned long long p = (unsigned long long)a[i] * 0xD2511F53u; a[i] = (unsigned)p ^ (unsigned)(p >> 32); }   // mad_u64_u32 / mul pair + xor
                else if (OP == 4) a[i] = __umul24(a[i], 0x511F53u) + 1u;           // v_mul_u32_u24
                else if (OP == 5) a[i] = __umulhi(a[i] << 8, 0x9E8D57u << 8) + 1u;
            }
        }
    }
    const long long c1 = clock64();
    unsigned s = 0;
    for (int i = 0; i < 8; ++i) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + (unsigned)w[0];
    if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;
}
template <int OP> void run(const char* name, int wgs_per_cu) {
    unsigned* out; long long* t; const int iters = 2000, blocks = 256 * wgs_per_cu;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&t, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, t, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, t, 8, hipMemcpyDeviceToHost);
    const double ops = 64.0 * iters;     // per wave
    printf("%-40s %d WG/CU: %7.3f ms   %.2f cycles per op-group per wave   (%.2f Tops/s chip-wide)\n", name, wgs_per_cu, ms, (double)c / ops,
           (double)blocks * 256 * ops / ms / 1e9);
}
int main() {
    for (int w : {1, 4}) {
        if (w == 1) { run<0>("v_mul_lo_u32 + add", 1); run<1>("v_mul_hi_u32 + add", 1); run<2>("xor + shift + add", 1); run<3>("64-bit product (hi ^ lo)", 1); run<4>("v_mul_u32_u24 + add", 1); }
        else { run<0>("v_mul_lo_u32 + add", 4); run<1>("v_mul_hi_u32 + add", 4); run<2>("xor + shift + add", 4); run<3>("64-bit product (hi ^ lo)", 4); run<4>("v_mul_u32_u24 + add", 4); }
    }
    return 0;
}
