// reduce.hip -- deterministic scalar reductions and gradient scaling (gfx950).
//
//   finalize_sums : per-workgroup partial sums [nacc][nblocks] -> nacc scalars (fp64 accumulation, fixed
//                   order; replaces the reference's cross-block float atomicAdd, e.g. td_lambda_kernel.h:38)
//   scale_rows    : out[i] = g[0] * in[i] for i < n_in, 0 for n_in <= i < n_out   (the "backward = upstream
//                   scalar x saved unit gradient" step of every scalar-loss op; the zero tail is the
//                   bootstrap row of grad_value, cf. td_lambda_kernel.h:46-50, vtrace_kernel.h:225-233)
#include <hip/hip_runtime.h>

#include "colscan.hpp"
#include "hpc_rll_hip.h"

namespace hpc_rll {
namespace {

struct Scales { float s[8]; };

__global__ __launch_bounds__(256) void finalize_sums_kernel(const float* __restrict__ partials, int nblocks,
                                                            int nacc, Scales scales, float* __restrict__ out) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < nacc; ++k) {
        // thread t adds partials t, t + 256, ... in that order; 16 of them are requested at a time (round 4: one memory round
        // trip per partial made this launch 10 us behind IQN's 8192 workgroups)
        double s = 0.0;
        constexpr int CH = 16;
        for (int i0 = threadIdx.x; i0 < nblocks; i0 += CH * 256) {
            float v[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) v[j] = i0 + j * 256 < nblocks ? partials[(size_t)k * nblocks + i0 + j * 256] : 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (i0 + j * 256 < nblocks) s += (double)v[j];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) red[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[k] = (float)((red[0] + red[1] + red[2] + red[3]) * (double)scales.s[k]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ g, const float* __restrict__ in,
                                                         float* __restrict__ out, long n_in, long n_out) {
    const float u = g[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (long)gridDim.x * 256)
        __builtin_nontemporal_store(i < n_in ? u * in[i] : 0.f, out + i);
}

__global__ __launch_bounds__(256) void scale_rows4_kernel(const float* __restrict__ g,
                                                          const vfloat4* __restrict__ in,
                                                          vfloat4* __restrict__ out, long n4_in, long n4_out) {
    const float u = g[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4_out; i += (long)gridDim.x * 256) {
        vfloat4 x = {0.f, 0.f, 0.f, 0.f};
        if (i < n4_in) x = in[i] * u;
        __builtin_nontemporal_store(x, out + i);
    }
}

}  // namespace

int finalize_sums(const float* partials, int nblocks, int nacc, const float* scales, float* out, hipStream_t st) {
    if (nacc < 1 || nacc > 8 || nblocks < 0 || !out) return HPC_RLL_EINVAL;
    Scales sc;
    for (int k = 0; k < 8; ++k) sc.s[k] = k < nacc ? scales[k] : 0.f;
    hipLaunchKernelGGL(finalize_sums_kernel, dim3(1), dim3(256), 0, st, partials, nblocks, nacc, sc, out);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

int scale_rows(const float* g, const float* in, float* out, long n_in, long n_out, hipStream_t st) {
    if (n_in < 0 || n_out < n_in) return HPC_RLL_EINVAL;
    if (n_out == 0) return HPC_RLL_OK;
    if (!g || !out || (n_in > 0 && !in)) return HPC_RLL_EINVAL;
    const bool v4 = (n_in % 4 == 0) && (n_out % 4 == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (v4) {
        long blocks = (n_out / 4 + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(scale_rows4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g,
                           reinterpret_cast<const vfloat4*>(in), reinterpret_cast<vfloat4*>(out), n_in / 4, n_out / 4);
    } else {
        long blocks = (n_out + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, in, out, n_in, n_out);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

}  // namespace hpc_rll

extern "C" int hpc_rll_scale_rows(const float* g, const float* in, float* out, int64_t n_in, int64_t n_out,
                                  void* stream) {
    return hpc_rll::scale_rows(g, in, out, (long)n_in, (long)n_out, (hipStream_t)stream);
}

// Floats of scratch a scalar-loss op over `n` columns / samples needs for its per-workgroup partial sums
// (never more than one workgroup per sample, never more than 8 sums per 64 samples).
extern "C" int64_t hpc_rll_partials_floats(int64_t n) { return n + 64; }
