// Micro-benchmark 3: can packed fp32 VALU FMAs with scalar (uniform) weights sustain near-peak rate?
// Each lane owns PX pixel pairs x 32 output channels (float2 accumulators = pixel pair); per reduction step k it reads one
// float2 per pair from LDS and does 32 v_pk_fma_f32 per pair with the weight w[k][n] broadcast to both halves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int PAIRS>
__global__ void __launch_bounds__(256) k(const float* __restrict__ w, float* out, int K, int iters) {
    __shared__ f2 xs[64 * 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 64 * 64; i += 256) xs[i] = f2{(float)i * 1e-4f, (float)i * 2e-4f};
    __syncthreads();
    f2 acc[PAIRS][32];
    for (int p = 0; p < PAIRS; ++p) for (int n = 0; n < 32; ++n) acc[p][n] = f2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        for (int kk = 0; kk < K; ++kk) {
            const float* wk = w + kk * 32;                       // uniform address -> scalar loads
            f2 x[PAIRS];
#pragma unroll
            for (int p = 0; p < PAIRS; ++p) x[p] = xs[((kk + it) * 64 + (tid & 63) + p * 7) & 4095];
#pragma unroll
            for (int n = 0; n < 32; ++n) {
                const float wv = wk[n];
#pragma unroll
                for (int p = 0; p < PAIRS; ++p) acc[p][n] = __builtin_elementwise_fma(x[p], f2{wv, wv}, acc[p][n]);
            }
        }
    }
    float s = 0.f;
    for (int p = 0; p < PAIRS; ++p) for (int n = 0; n < 32; ++n) s += acc[p][n].x + acc[p][n].y;
    out[blockIdx.x * 256 + tid] = s;
}

template <int PAIRS>
void run() {
    const int G = 256 * 4, K = 288, iters = 40;
    float *w, *out;
    hipMalloc(&w, K * 32 * 4); hipMemset(w, 0, K * 32 * 4); hipMalloc(&out, G * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAIRS>, dim3(G), dim3(256), 0, 0, w, out, K, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAIRS>, dim3(G), dim3(256), 0, 0, w, out, K, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)G * 256 * PAIRS * 2 * 32 * 2.0 * K * iters;
    printf("pairs/lane %d: %.3f ms  %.1f TFLOP/s\n", PAIRS, ms, flop / (ms * 1e-3) / 1e12);
}

int main() { run<1>(); run<2>(); run<3>(); return 0; }
