// A kernel that OCCUPIES k compute units for a given time and does nothing else (tools/squat_test.py, tests/test_gpu_overlap.py): one
// workgroup per CU (it asks for the whole LDS), spinning on the real-time counter.  Stands in for a collective kernel that is resident
// on some CUs while a persistent convolution grid is dispatched (the 1-GPU boxes cannot run RCCL with more than one rank).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libsquat.so squat.hip
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) squat_kernel(unsigned long long ticks, int* sink) {
    extern __shared__ int lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (sink && lds[threadIdx.x] == 0x7fffffff) sink[0] = 1;                 // (keeps the LDS allocation alive)
}
extern "C" int squat_launch(int cus, double microseconds, void* stream) {
    static bool once = false;
    const int lds = 160 * 1024;
    if (!once) { if (hipFuncSetAttribute((const void*)squat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1; once = true; }
    hipLaunchKernelGGL(squat_kernel, dim3(cus), dim3(256), lds, (hipStream_t)stream, (unsigned long long)(microseconds * 100.0), (int*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
