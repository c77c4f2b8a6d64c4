// stream_write.hpp -- the store pattern that reaches MI355X's pure-write rate.
//
// Measured (tests/tools/micro/writebw2.hip, writebw3.hip; profiles/r04_writebw.txt): a 4 GiB output is written at
//   6.3-6.5 TB/s  by ONE workgroup of 256 threads per CU walking the output grid-stride with 16-byte stores (every sweep of
//                 the grid covers one contiguous MiB; hipMemsetD32Async: 6.6),
//   4.5-5.9 TB/s  by the same loop with 2 ... 256 workgroups per CU, with 2 / 6 / 8 / 16 waves per workgroup, or with a
//                 contiguous chunk per workgroup or per wave -- what every write-heavy kernel of this library did.
// What decides it is that the 256 x 16 bytes a workgroup stores together are ONE 4 KiB-ALIGNED block (2 / 6 KiB per workgroup,
// or 4 KiB starting 512 bytes off a boundary: 4.7-5.8) and that few waves per CU store at a time.
// Kernels whose output is mostly zeros (the one-hot gradients of the n-step TD losses: 1 / N of the elements) therefore
// write in two launches: this fill, then the few values.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wave.hpp"

namespace hpc_rll {

inline int device_cus() {   // CU count of the current device (cached per device; 0 on error)
    constexpr int kMax = 64;
    static int cus[kMax] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMax) return 0;
    if (cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        cus[dev] = v > 0 ? v : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

// `shift` = 16-byte elements between the 4 KiB boundary below y and y: the loop runs over indices from that boundary, so that
// every workgroup's 256 x 16 bytes are ONE 4 KiB-aligned block (the same loop on a base 512 bytes off such a boundary:
// 5.7 instead of 6.5 TB/s).
static __global__ __launch_bounds__(256) void stream_zero_kernel(vfloat4* __restrict__ y, size_t n4, unsigned shift,
                                                          float* __restrict__ tail, int ntail) {
    const vfloat4 z = {0.f, 0.f, 0.f, 0.f};
    const size_t nt = (size_t)gridDim.x * 256, end = n4 + shift;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < end; v += nt)
        if (v >= shift) y[v - shift] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}

// p[0, n) = 0.  p must be 16-byte aligned.
inline int launch_stream_zero(float* p, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    int cus = device_cus();
    if (cus <= 0) cus = 256;
    const size_t n4 = n / 4;
    const unsigned shift = (unsigned)((reinterpret_cast<uintptr_t>(p) & 4095) / 16);
    hipLaunchKernelGGL(stream_zero_kernel, dim3((unsigned)cus), dim3(256), 0, st, reinterpret_cast<vfloat4*>(p), n4, shift,
                       p + n4 * 4, (int)(n - n4 * 4));
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace hpc_rll
