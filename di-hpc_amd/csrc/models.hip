// models.hip -- the three AlphaStar actor-critic inference helpers of the reference's `hpc_models` extension
// (src/models/actor_critic.cu:8-83, include/hpc/rll/cuda/models/actor_critic_kernel.h:14-80), forward only.
// SURVEY.md 8f-4 ("next row"): tiny batch-parallel kernels; written wave64-first (one wave per (b, entity) dot
// product with a butterfly instead of a 32-thread block reduce; guarded, 64-bit indexing).
#include <hip/hip_runtime.h>

#include "hpc_rll_hip.h"
#include "wave.hpp"

namespace hpc_rll {
namespace {

inline int last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

// ae[b,:] += (sample_entity[b] == entity_num[b]) ? 0 : key_embeddings[b, sample_entity[b], :]
__global__ __launch_bounds__(256) void update_ae_kernel(const float* __restrict__ key, const int64_t* __restrict__ sample,
                                                        const int64_t* __restrict__ entity_num, float* __restrict__ ae,
                                                        long B, long E, long D) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < B * D; i += (long)gridDim.x * 256) {
        const long b = i / D, d = i - b * D;
        const long e = sample[b];
        if (e != entity_num[b] && e >= 0 && e < E) ae[i] += key[(b * E + e) * D + d];
    }
}

// gates = ih + hh + bias, order i,f,g,o (torch.nn.LSTM);  c = f*c + i*g ; h = o*tanh(c)
__global__ __launch_bounds__(256) void lstm_activation_kernel(const float* __restrict__ ih, const float* __restrict__ hh,
                                                              const float* __restrict__ bias, float* __restrict__ h,
                                                              float* __restrict__ c, long B, long H) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < B * H; i += (long)gridDim.x * 256) {
        const long b = i / H, j = i - b * H;
        float v[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = ih[(b * 4 + g) * H + j] + hh[(b * 4 + g) * H + j] + bias[g * H + j];
        const float ig = 1.f / (1.f + expf(-v[0])), fg = 1.f / (1.f + expf(-v[1]));
        const float gg = tanhf(v[2]), og = 1.f / (1.f + expf(-v[3]));
        const float nc = fg * c[i] + ig * gg;
        c[i] = nc;
        h[i] = og * tanhf(nc);
    }
}

// out[b,e] = mask[b,e] ? dot(mat[b,e,:], vec[b,:]) / div : mask_value / div ; one wave per (b,e)
__global__ __launch_bounds__(256) void pre_sample_kernel(const float* __restrict__ mat, const float* __restrict__ vec,
                                                         const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                         long B, long E, long H, float mask_value, float div) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < B * E; r += (long)gridDim.x * 4) {
        const long b = r / E;
        float s = 0.f;
        if (mask[r]) {
            for (long k = lane; k < H; k += 64) s = fmaf(mat[r * H + k], vec[b * H + k], s);
            s = wave_sum(s);
        } else {
            s = mask_value;
        }
        if (lane == 0) out[r] = s / div;
    }
}

inline unsigned blocks_for(long n, long per) {
    long g = (n + per - 1) / per;
    if (g > 4096) g = 4096;
    return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace
}  // namespace hpc_rll

using namespace hpc_rll;

extern "C" int hpc_rll_actor_critic_update_ae(const float* key_embeddings, const int64_t* sample_entity,
                                              const int64_t* entity_num, float* autoregressive_embedding, int64_t B,
                                              int64_t E, int64_t D, void* stream) {
    if (B < 0 || E < 0 || D < 0) return HPC_RLL_EINVAL;
    if (B * D == 0) return HPC_RLL_OK;
    if (!key_embeddings || !sample_entity || !entity_num || !autoregressive_embedding) return HPC_RLL_EINVAL;
    hipLaunchKernelGGL(update_ae_kernel, dim3(blocks_for(B * D, 256)), dim3(256), 0, (hipStream_t)stream,
                       key_embeddings, sample_entity, entity_num, autoregressive_embedding, (long)B, (long)E, (long)D);
    return last_error();
}

extern "C" int hpc_rll_actor_critic_lstm_activation(const float* ih, const float* hh, const float* bias, float* h,
                                                    float* c, int64_t B, int64_t H, void* stream) {
    if (B < 0 || H < 0) return HPC_RLL_EINVAL;
    if (B * H == 0) return HPC_RLL_OK;
    if (!ih || !hh || !bias || !h || !c) return HPC_RLL_EINVAL;
    hipLaunchKernelGGL(lstm_activation_kernel, dim3(blocks_for(B * H, 256)), dim3(256), 0, (hipStream_t)stream, ih, hh,
                       bias, h, c, (long)B, (long)H);
    return last_error();
}

extern "C" int hpc_rll_actor_critic_pre_sample(const float* mat, const float* vec, const uint8_t* mask, float* out,
                                               int64_t B, int64_t E, int64_t H, float mask_value, float div_factor,
                                               void* stream) {
    if (B < 0 || E < 0 || H < 0 || div_factor == 0.f) return HPC_RLL_EINVAL;
    if (B * E == 0) return HPC_RLL_OK;
    if (!mat || !vec || !mask || !out) return HPC_RLL_EINVAL;
    hipLaunchKernelGGL(pre_sample_kernel, dim3(blocks_for(B * E, 4)), dim3(256), 0, (hipStream_t)stream, mat, vec, mask,
                       out, (long)B, (long)E, (long)H, mask_value, div_factor);
    return last_error();
}
