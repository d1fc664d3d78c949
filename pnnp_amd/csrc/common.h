// Shared helpers for libpnnp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pnnp_hip.h"

#define PNNP_WAVE 64

static inline int pnnp_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PNNP_OK : PNNP_E_LAUNCH;
}

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

template <typename T>
static inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// The library keeps no mutable state that changes results.  What it does cache are device facts (CU count, occupancy of a
// kernel) and the per-device "raise the dynamic-LDS limit of this kernel" call: idempotent, keyed by the CURRENT device
// (hipFuncSetAttribute and occupancy are per device), lock-free, safe to race (two threads may both do the one-time call).
#include <atomic>
constexpr int PNNP_MAX_DEVICES = 64;
static inline int pnnp_current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= PNNP_MAX_DEVICES) return -1;
    return dev;
}
struct PnnpPerDevice {                    // one int per device, 0 = not yet computed
    std::atomic<int> v[PNNP_MAX_DEVICES];
    template <class F>
    int get(F&& compute) {                // compute() returns the value (> 0) or <= 0 on failure (not cached)
        const int dev = pnnp_current_device();
        if (dev < 0) return compute();
        int x = v[dev].load(std::memory_order_relaxed);
        if (x > 0) return x;
        x = compute();
        if (x > 0) v[dev].store(x, std::memory_order_relaxed);
        return x;
    }
};
// raise a kernel's dynamic shared-memory limit once per device; returns PNNP_OK / PNNP_E_LAUNCH
template <class K>
static inline int pnnp_allow_lds(PnnpPerDevice& once, K kern, int bytes) {
    if (bytes <= 64 * 1024) return PNNP_OK;
    const int ok = once.get([&] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 0;
    });
    return ok > 0 ? PNNP_OK : PNNP_E_LAUNCH;
}

// grid of a persistent kernel with `tiles` work items: pnnp_get_persistent_split() workgroups per CU (csrc/capi.hip), at most one per tile
extern "C" int pnnp_get_persistent_split(void);
extern "C" int pnnp_device_cus(void);
static inline int pnnp_persistent_grid(int64_t tiles) {
    int cus = pnnp_device_cus();
    if (cus < 1) cus = 256;
    const int64_t wgs = (int64_t)cus * pnnp_get_persistent_split();
    return (int)(wgs < tiles ? wgs : tiles);
}

// amax slots of the fp16x2 family (csrc/h2.h): a wave's largest |stored value| -> atomicMax on the slot (non-negative floats order like their
// bit patterns).  Call once per wave at the end of a kernel, all 64 lanes active.
__device__ __forceinline__ void pnnp_amax_commit(float m, unsigned* slot) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
    // a wave whose maximum does not exceed what the slot already holds skips the atomic: a kernel with a large grid (maxpool_bwd: 130 000 waves)
    // would otherwise serialise that many atomics on one address (measured: 95 -> 164 us per launch); a stale read only costs an unnecessary atomic
    if ((threadIdx.x & 63) == 0 && m > 0.f && __float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(m));
}

// ... for streaming kernels whose waves all finish at about the same time (a one-crop layout pass: 8192 waves found the slot at 0 and serialised
// 8192 atomics on one address, 128 us for a 4 MB tensor): the block's waves reduce through LDS first and ONE thread issues the atomic.  Every thread
// of the block must call it (it contains a barrier); blocks of at most 1024 threads.
__device__ __forceinline__ void pnnp_amax_commit_block(float m, unsigned* slot) {
    __shared__ float pnnp_amax_red[16];
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) m = fmaxf(m, __shfl_xor(m, sft, 64));
    const int wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) pnnp_amax_red[wv] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) m = fmaxf(m, pnnp_amax_red[i]);
        if (m > 0.f && __float_as_uint(m) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(m));
    }
}

// Tensor.clamp / np.clip semantics: a NaN stays a NaN (fminf / fmaxf return the OTHER operand for a NaN, which would turn a diverged network's
// output into a finite loss or PSNR).  Comparisons with a NaN are false, so it falls through both selects.
__device__ __forceinline__ float pnnp_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
