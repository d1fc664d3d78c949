// How fast does one CU STORE?  One 512-thread workgroup per CU (160 KB LDS asked for, as the persistent conv kernels do), every wave
// issues `n` back-to-back buffer stores of one kind, cycles by s_memtime around the issue loop + s_waitcnt vmcnt(0):
//   kind 0: buffer_store_dwordx4, lane l -> 16 contiguous bytes (1 KB per instruction, the conv epilogues' patch layout)
//   kind 1: buffer_store_dword,   lane l -> 4 contiguous bytes (256 B per instruction)
//   kind 2: buffer_store_dword in the 16x16 MFMA accumulator layout: lanes 0-15 = 64 contiguous bytes of one pixel, lane groups 4 pixels
//           apart ... i.e. 4 x 64-byte segments, pixel stride `cs` floats (what a store straight from the accumulators would do)
//   kind 3: buffer_store_dword in the 32x32 accumulator layout: lanes 0-31 = 128 contiguous bytes of one pixel, lanes 32-63 of the pixel 4 further
// usage: store_rate [workgroups = 256] [stores per wave = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(512, 1) store_kernel(float* dst, long long wg_stride, int n, int cs, long long* out) {
    extern __shared__ int lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* base = dst + (long long)blockIdx.x * wg_stride;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(wg_stride * 4), 0x00020000);
    unsigned vo; int step;
    if (KIND == 0) { vo = (wave * 64 + lane) * 16; step = 8 * 1024; }
    else if (KIND == 1) { vo = (wave * 64 + lane) * 4; step = 8 * 256; }
    else if (KIND == 2) { vo = ((wave * 64 + 4 * (lane >> 4)) * cs + (lane & 15)) * 4; step = 4; }      // r = 0..3 -> + cs floats; 16 columns on: + 64 B
    else if (KIND == 3) { vo = ((wave * 64 + 4 * (lane >> 5)) * cs + (lane & 31)) * 4; step = 4; }
    else if (KIND == 4) { vo = ((wave * 64 + (lane & 15)) * cs + (lane >> 4) * 4) * 4; step = 64; }       // b128: 16 pixels x 64 B per instruction (conv_x3s epilogue)
    else { vo = ((wave * 64 + (lane >> 3)) * cs + (lane & 7) * 4) * 4; step = 128; }                    // b128: 8 pixels x 128 B per instruction (round 3's LDS-patch epilogue)
    const u32x4 v = {(unsigned)tid, 1u, 2u, 3u};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    int so = 0;
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) { __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, so, 0); so += step; }
        else if (KIND == 1) { __builtin_amdgcn_raw_buffer_store_b32(v[0], rd, vo, so, 0); so += step; }
        else if (KIND == 2) {
            // (i & 3) = r (next pixel), (i >> 2) & 3 = 16-channel block (cs = 64: 4 blocks), then 16 pixels on, then the next 64-pixel group of the wave
            const int r = i & 3, j = (i >> 2) & 3, h = (i >> 4) & 3, rest = i >> 6;
            __builtin_amdgcn_raw_buffer_store_b32(v[0], rd, vo, ((r + 16 * h + 512 * rest) * cs + 16 * j) * 4, 0);
        } else if (KIND == 4) {
            const int j = i & 3, h = (i >> 2) & 3, rest = i >> 4;          // 4 x 16-channel blocks of a pixel, then 16 pixels on
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, ((16 * h + 512 * rest) * cs + 16 * j) * 4, 0);
        } else if (KIND == 5) {
            const int j = i & 1, h = (i >> 1) & 7, rest = i >> 4;          // 2 x 32-channel blocks, then 8 pixels on
            __builtin_amdgcn_raw_buffer_store_b128(v, rd, vo, ((8 * h + 512 * rest) * cs + 32 * j) * 4, 0);
        } else {
            const int r = i & 3, g = (i >> 2) & 3, j = (i >> 4) & 1, rest = i >> 5;       // rows r + 8 g (+ 4 by the lane half), 32-channel block j
            __builtin_amdgcn_raw_buffer_store_b32(v[0], rd, vo, ((r + 8 * g + 512 * rest) * cs + 32 * j) * 4, 0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0x0f70);
    const long long t2 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = t2 - t0; }
    if (lds[tid] == 0x7fffffff) out[0] = 0;
}
template <int KIND> void run(int wgs, int n, float* dst, long long stride, long long* out, const char* name, double bytes_per_store) {
    hipFuncSetAttribute((const void*)store_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(store_kernel<KIND>, dim3(wgs), dim3(512), 160 * 1024, 0, dst, stride, n, 64, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    long long* h = (long long*)malloc(wgs * 16 * 8); hipMemcpy(h, out, wgs * 16 * 8, hipMemcpyDeviceToHost);
    double issue = 0, done = 0; for (int i = 0; i < wgs * 8; ++i) { issue += h[2 * i]; done += h[2 * i + 1]; }
    issue /= wgs * 8; done /= wgs * 8;
    const double bytes = 8.0 * n * bytes_per_store;
    printf("%-34s wgs %3d: issue %8.0f ticks, drained %8.0f ticks (100 MHz) per wave for %d stores; kernel %.3f ms -> %.1f GB/s, %.2f B/ns per CU\n",
           name, wgs, issue, done, n, ms, bytes * wgs / ms / 1e6, bytes / (done * 10.0));
    free(h);
}
int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 256;
    const long long stride = 8ll * 1024 * 1024;        // floats per workgroup (32 MB)
    float* dst; hipMalloc(&dst, stride * 4 * wgs); long long* out; hipMalloc(&out, wgs * 16 * 8);
    run<0>(wgs, n, dst, stride, out, "dwordx4 (1 KB / instr)", 1024);
    run<1>(wgs, n, dst, stride, out, "dword contiguous (256 B / instr)", 256);
    run<2>(wgs, n, dst, stride, out, "dword 16x16 accumulator layout", 256);
    run<3>(wgs, n, dst, stride, out, "dword 32x32 accumulator layout", 256);
    run<0>(wgs, n / 4, dst, stride, out, "dwordx4, same bytes as the dwords", 1024);
    run<4>(wgs, n, dst, stride, out, "dwordx4, 16 px x 64 B (cs = 64)", 1024);
    run<5>(wgs, n, dst, stride, out, "dwordx4, 8 px x 128 B (cs = 64)", 1024);
    return 0;
}
