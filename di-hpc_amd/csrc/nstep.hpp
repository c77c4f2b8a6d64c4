// nstep.hpp -- the n-step discounted reward sum shared by the per-sample TD kernels (dist_ops.hip, sample_ops.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace hpc_rll {
namespace {

// n-step discounted reward sum R = sum_t gamma^t r[t, b] for S samples at once, in the SAME order of operations as the
// plain loop `R = fmaf(f, r_t, R); f *= gamma` -- but the loads of four (one sample: eight) time steps x S samples are all issued before the
// first use.  In a loop with a run-time trip count the compiler waits for every load before the fma that consumes it: the
// plain form costs one dependent memory round trip PER STEP (nstep = 5: five round trips before the sample's rows can
// even be addressed), which is what these per-sample kernels were bound by at large batch.
template <int S>
__device__ __forceinline__ void nstep_returns(const float* __restrict__ reward, int B, int nstep, float gamma,
                                              const long (&bb)[S], float (&R)[S]) {
    constexpr int CH = S == 1 ? 8 : 4;          // time steps requested together (x S samples)
#pragma unroll
    for (int s = 0; s < S; ++s) R[s] = 0.f;
    float f = 1.f;
    for (int t0 = 0; t0 < nstep; t0 += CH) {
        float r[CH][S];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            const int t = t0 + k < nstep ? t0 + k : nstep - 1;     // clamped: the load is unconditional
#pragma unroll
            for (int s = 0; s < S; ++s) r[k][s] = reward[(size_t)t * B + bb[s]];
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) {
            if (t0 + k < nstep) {
#pragma unroll
                for (int s = 0; s < S; ++s) R[s] = fmaf(f, r[k][s], R[s]);
                f *= gamma;
            }
        }
    }
}
__device__ __forceinline__ float nstep_return1(const float* __restrict__ reward, int B, int nstep, float gamma, long b) {
    const long bb[1] = {b};
    float R[1];
    nstep_returns<1>(reward, B, nstep, gamma, bb, R);
    return R[0];
}

}  // namespace
}  // namespace hpc_rll
