// escat_static.hip -- the one entry point of libescat.so that allocates nothing (include/escat.h:
// escat_chain_csr_static_f32).  No counterpart in the reference; for callers that cannot call hipMalloc (HIP graph
// capture, tight serving loops, harnesses that forbid allocation).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "../../include/escat.h"
#include "escat_internal.h"

// ---- the one entry point that allocates nothing (include/escat.h): the multi-launch path's ping-pong vectors live in
// a file-scope device array instead of a caller workspace.  `scratch` is a host-side stand-in here, NOT a device
// address: its device address is fetched once per device with hipGetSymbolAddress and only that pointer is used.
static __device__ __align__(256) unsigned char scratch[ESCAT_STATIC_SCRATCH_BYTES];
static void *g_scratch_ptr[64];                     // per device; each GPU holds its own copy of the array
static std::mutex g_scratch_mu;

static int scratch_for_current_device(void **out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return escat_fail(ESCAT_ELAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= 64) return escat_fail(ESCAT_ERANGE, "device ordinal %d outside [0,64)", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) return escat_fail(ESCAT_ELAUNCH, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return ESCAT_OK;
}

extern "C" int escat_chain_csr_static_f32(const int32_t *rowptr, const int32_t *col, const float *val, int64_t n, int64_t nnz,
                                          const float *X, int C, int T, const escat_event *events, int n_events, int F, float *OUT,
                                          const int32_t *part_ptr, int n_parts, int max_part_rows, const escat_sell *sell,
                                          void *stream)
{
    if (n < 0 || C < 1 || C > ESCAT_MAX_CHANNELS) return escat_fail(ESCAT_EINVAL, "n=%lld, C=%d: bad size", (long long)n, C);
    void *ws = nullptr;
    int rc = scratch_for_current_device(&ws); if (rc) return rc;
    // Only the multi-launch path uses scratch (the register-resident and LDS-streaming paths keep their vectors in LDS):
    // the size is checked where the path is chosen, inside escat_chain_csr_f32, against the array's size handed over as
    // the workspace size -- a call that stays in LDS is never refused for its n.  Every element of the ping-pong vectors
    // is written by a step before the next one reads it: nothing to clear.
    rc = escat_chain_csr_f32(rowptr, col, val, n, nnz, X, C, T, events, n_events, F, OUT, part_ptr, n_parts, max_part_rows, sell,
                             ws, (size_t)ESCAT_STATIC_SCRATCH_BYTES, stream);
    if (rc == ESCAT_EWORKSPACE)
        return escat_fail(ESCAT_ERANGE, "escat_chain_csr_static_f32: n=%lld rows x C=%d channels need %zu B of scratch on the "
                                        "multi-launch path, the static buffer holds %u B (use escat_chain_csr_f32 with a workspace)",
                          (long long)n, C, escat_chain_workspace_bytes(n, C), (unsigned)ESCAT_STATIC_SCRATCH_BYTES);
    return rc;
}

