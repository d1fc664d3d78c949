// Library-level entry points of libpnnp_hip.so.
#include "common.h"

extern "C" {

int pnnp_version(void) { return 100; }   // 0.1.0

int pnnp_abi_version(void) { return PNNP_ABI_VERSION; }
int pnnp_pack_job_bytes(void) { return (int)sizeof(PnnpPackJob); }

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

// How many workgroups the persistent forward / backward-data convolution kernels (csrc/conv_x3.hip, csrc/gemm_x3.hip) launch per CU.
// 1 (default): one workgroup per CU with an equal, static share of the tiles -- the fastest when the kernel has the chip to itself.
// n > 1: n x CUs workgroups of 1/n share each; a CU still holds one at a time (160 KB of LDS), the others wait in the dispatcher and
// go to whichever CU frees up first.  That is what makes running NEXT to another resident kernel safe (an RCCL collective on a side
// stream: trainer.BucketedAllReduce(overlap=True)): a workgroup that finds its CU occupied no longer carries a full share that it can
// only start when the others have finished (layer time x 2) -- the shares of the occupied CUs are picked up by the free ones, and the
// layer takes ~CUs / (CUs - k) as long with k CUs occupied.  Costs one pipeline fill per workgroup (a few microseconds each).
// Process-wide, takes effect at the next launch; the only setting the library keeps.
static std::atomic<int> g_persistent_split{1};
void pnnp_set_persistent_split(int n) { g_persistent_split.store(n < 1 ? 1 : (n > 16 ? 16 : n), std::memory_order_relaxed); }
int pnnp_get_persistent_split(void) { return g_persistent_split.load(std::memory_order_relaxed); }

}  // extern "C"
