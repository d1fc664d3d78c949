// Library-level entry points of libpnnp_hip.so.
#include "common.h"

extern "C" {

int pnnp_version(void) { return 100; }   // 0.1.0

const char* pnnp_error_string(int code) {
    switch (code) {
        case PNNP_OK: return "ok";
        case PNNP_E_INVALID: return "invalid argument";
        case PNNP_E_UNSUPPORTED: return "unsupported configuration";
        case PNNP_E_LAUNCH: return "kernel launch failed";
        case PNNP_E_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

int pnnp_device_cus(void) {            // compute units of the CURRENT device (cached per device); never below 1: grids and workspace
    static PnnpPerDevice cache;        // sizes are derived from it, and a failed query must not turn into a 0-block launch
    const int n = cache.get([] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        return n;
    });
    return n >= 1 ? n : 256;           // MI355X
}

}  // extern "C"
