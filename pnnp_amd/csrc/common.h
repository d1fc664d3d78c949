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
