// abi.hip -- version + status strings of the C ABI (include/hpc_rll_hip.h).
#include <hip/hip_runtime.h>

#include "hpc_rll_hip.h"

extern "C" int hpc_rll_abi_version(void) { return 6; }

extern "C" const char* hpc_rll_status_string(int status) {
    switch (status) {
        case HPC_RLL_OK: return "ok";
        case HPC_RLL_EINVAL: return "hpc_rll: invalid argument (negative size or null pointer)";
        case HPC_RLL_EALIGN: return "hpc_rll: pointer is not 4-byte aligned";
        case HPC_RLL_EUNSUPPORTED: return "hpc_rll: shape/configuration not supported by the gfx950 kernels";
        case HPC_RLL_ETIMEOUT:
            return "hpc_rll: a persistent LSTM kernel gave up waiting for its co-resident workgroups (another process "
                   "held the GPU); the results of that call are invalid -- hpc_rll_clear_async_error() continues on the "
                   "step kernels";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "hpc_rll: unknown status";
}

extern "C" int hpc_rll_stream_is_capturing(void* stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const hipError_t e = hipStreamIsCapturing((hipStream_t)stream, &st);
    if (e != hipSuccess) return -(int)e - 1000;
    return st == hipStreamCaptureStatusActive ? 1 : 0;
}

extern "C" int hpc_rll_stream_synchronize(void* stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }
