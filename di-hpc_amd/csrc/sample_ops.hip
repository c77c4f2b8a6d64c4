// sample_ops.hip -- per-sample (batch-parallel) loss ops for gfx950: PPO and the q n-step TD errors.
//
// Replaces (under /root/reference):
//   PPOForward/Backward              src/rl_utils/ppo.cu:8-111, ppo_kernel.h:12-283
//   QNStepTdForward/Backward         src/rl_utils/q_nstep_td.cu, q_nstep_td_kernel.h:11-62
//   QNStepTdRescaleForward/Backward  src/rl_utils/q_nstep_td_rescale.cu, q_nstep_td_rescale_kernel.h:11-72
// Semantics: hpc_rll/origin/ppo.py:51-80, origin/td.py:9-22,280-291,326-354.
//
// One lane per sample, coalesced along B; scalar losses by deterministic two-stage reduction
// (wave butterfly -> LDS -> one partial per workgroup -> fixed-order fp64 finalize), no float atomics.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "colscan.hpp"
#include "hpc_rll_hip.h"
#include "nstep.hpp"
#include "ppo_op.hpp"
#include "stream_write.hpp"

namespace hpc_rll {

int categorical_forward(const float* logits, const int64_t* action, float* logp, float* ent, long rows, int N,
                        hipStream_t st);
int categorical_backward(const float* logits, const int64_t* action, const float* c1, const float* g1,
                         const float* c2, const float* g2, float* grad, long rows, int N, hipStream_t st);
struct PpoOp;
// categorical.hip: both policy heads and the per-sample loss in ONE launch; false = shape not covered (caller runs three)
bool ppo_forward_fused(const float* logits_new, const float* logits_old, const int64_t* action, const PpoOp& op, long rows,
                       int N, float* partials, const float* scales, float* out5, hipStream_t st, int* rc);
extern int g_ppo_fused;

namespace {

inline int last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

// Generic "one lane per sample + NACC sums" kernel: workgroups of NT threads walk the samples grid-stride.
// Round 5: the launch keeps its grid within the fold's 512 workgroups (colscan.hpp) -- batches above 131072 samples take
// 1024-thread workgroups, above 524288 the workgroups loop -- so the loss is finalised inside the launch at every batch size
// (q n-step TD at B = 262144 used to be 1024 workgroups + a finalize launch: two dependent launches for 17 us of work).
template <class Op, int NT>
__global__ __launch_bounds__(NT) void sample_kernel(const Op op, long n, float* __restrict__ partials,
                                                    const ScanFold fold) {
    constexpr int NACC = Op::NACC, NWV = NT / 64;
    __shared__ float red[NACC * NWV];
    float acc[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n; i += (long)gridDim.x * NT) op(i, acc);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[k * NWV + w] = s;
    }
    __syncthreads();
    float sum = 0.f;
    if (threadIdx.x < NACC) {
        if (NWV == 4) sum = (red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1]) + (red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3]);
        else
            for (int i = 0; i < NWV; ++i) sum += red[threadIdx.x * NWV + i];
    }
    publish_sums<NACC, NT>(sum, partials, fold);   // with a fold: the last workgroup also finalises the sums (colscan.hpp)
}
// launch + finalisation of the NACC sums into `out` (x scale[k])
template <class Op>
inline int launch_sample(const Op& op, long n, float* partials, int nacc, const float* scale, float* out, hipStream_t st) {
    const bool wide = (n + 255) / 256 > kFoldMaxGrid;
    const long nt = wide ? 1024 : 256;
    long blocks = (n + nt - 1) / nt;
    if (blocks > kFoldMaxGrid) blocks = kFoldMaxGrid;
    const ScanFold fold = make_fold(st, nacc, scale, out, blocks);
    if (wide) hipLaunchKernelGGL((sample_kernel<Op, 1024>), dim3((unsigned)blocks), dim3(1024), 0, st, op, n, partials, fold);
    else hipLaunchKernelGGL((sample_kernel<Op, 256>), dim3((unsigned)blocks), dim3(256), 0, st, op, n, partials, fold);
    const hipError_t e = hipGetLastError();
    const int rc = e == hipSuccess ? HPC_RLL_OK : (int)e;
    if (rc || fold.out) return rc;
    return finalize_sums(partials, (int)blocks, nacc, scale, out, st);
}

// ---------------------------------------------------------------------------------------------- q n-step TD
__device__ __forceinline__ float h_transform(float x, float eps) {
    const float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    return s * (sqrtf(fabsf(x) + 1.f) - 1.f) + eps * x;
}
__device__ __forceinline__ float h_inverse(float x, float eps) {
    const float s = (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f);
    const float t = (sqrtf(1.f + 4.f * eps * (fabsf(x) + 1.f + eps)) - 1.f) / (2.f * eps);
    return s * (t * t - 1.f);
}

struct QNStepOp {
    static constexpr int NACC = 1;
    const float *q, *next_q; const int64_t *action, *next_action; const float *reward, *done, *weight;
    float *td_err, *grad_buf;
    int nstep, B, N; float gamma, gamma_n, scale; int rescale;
    __device__ void operator()(long b, float (&acc)[NACC]) const {
        // Round 4: every scalar of the sample is requested before the first use, the rewards eight steps at a time (nstep.hpp).
        // The plain loop `R = fmaf(f, reward[t, b], R)` with its run-time trip count made the compiler wait for each reward before
        // the next was requested: nstep dependent memory round trips in front of a two-load kernel.  Same operations, same order.
        const long a = action[b], na = next_action[b];
        const float dn = done[b];
        const float w = weight ? weight[b] : 1.f;
        const float R = nstep_return1(reward, B, nstep, gamma, b);
        const float qsa = q[b * N + a];
        float tq = next_q[b * N + na];
        if (rescale) tq = h_inverse(tq, 1e-2f);
        float tgt = R + gamma_n * tq * (1.f - dn);
        if (rescale) tgt = h_transform(tgt, 1e-2f);
        const float d = qsa - tgt;
        td_err[b] = d * d;
        acc[0] = fmaf(d * d, w, acc[0]);
        grad_buf[b] = 2.f * d * w * scale;
    }
};

// grad[b, n] = (n == action[b]) ? g * buf[b] : 0     (rows of N; also used with an inner dimension K:
// grad[b, n, k] = (n == action[b]) ? g * buf[b*K + k] : 0)
__global__ __launch_bounds__(256) void onehot_scatter_kernel(const float* __restrict__ g, const float* __restrict__ buf,
                                                             const int64_t* __restrict__ action,
                                                             float* __restrict__ grad, long B, int N, int K) {
    const float u = g[0];
    const long total = B * N * K;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long b = i / ((long)N * K);
        const int rem = (int)(i - b * (long)N * K);
        const int n = rem / K, k = rem - n * K;
        __builtin_nontemporal_store(((long)n == action[b]) ? u * buf[b * K + k] : 0.f, grad + i);
    }
}

// The same gradient with 16-byte stores and no per-element division (the 4-byte kernel above spends a 64-bit divide
// and a dependent action load on every float it writes: 3.5 TB/s on a pure write stream).  The output is rows of
// L = N*K floats (one per sample; PLANES > 1: `planes` stacked (B, N) planes with K = 1, the IQN layout (tau, B, N),
// plane = blockIdx.y), all zero except the K floats at [a*K, a*K + K).  A workgroup takes `rb` consecutive samples =
// rb*L/4 quads; quad -> (sample, column) by a multiply-high with a host-made reciprocal (exact for the ranges used).
__global__ __launch_bounds__(256) void onehot_rows4_kernel(const float* __restrict__ g, const float* __restrict__ buf,
                                                           const int64_t* __restrict__ action, float* __restrict__ grad,
                                                           long B, int N, int K, int rb, unsigned l4, unsigned magic) {
    const float u = g[0];
    const long b0 = (long)blockIdx.x * rb;
    const int nb = (int)(B - b0 < rb ? B - b0 : rb);
    const unsigned quads = (unsigned)nb * l4;
    const int plane = blockIdx.y, planes = gridDim.y;
    const long L = (long)l4 * 4;
    float* __restrict__ out = grad + ((long)plane * B + b0) * L;
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (rb == 1 && planes == 1) {
        // Round 5, a workgroup per sample (rows of 4 KiB and more): the zeros of the whole row go out FIRST -- no load in front of
        // them -- and the K values are stored over them by the threads whose quads they fall in once the action has arrived (the
        // same thread to the same address: in program order).  QR-DQN backward at B = 262144: 0.373 -> 0.363 ms, same bits.  (Rows
    // shorter than a workgroup's block -- IQN: one value per 256-byte row -- lose with it, 0.118 -> 0.140 ms: every value quad is
    // a second, partial-line store.)
        const long a = action[b0];
        for (unsigned q = threadIdx.x; q < quads; q += 256) __builtin_nontemporal_store(zero4, reinterpret_cast<vfloat4*>(out + (long)q * 4));
        const int lo = (a >= 0 && a < (long)N) ? (int)a * K : -K - 4;
        for (unsigned q = threadIdx.x; q < quads; q += 256) {
            const int col = (int)q * 4;
            if (col + 4 > lo && col < lo + K) {
                vfloat4 v = zero4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = col + j - lo;
                    if (k >= 0 && k < K) v[j] = u * buf[b0 * K + k];
                }
                *reinterpret_cast<vfloat4*>(out + (long)q * 4) = v;
            }
        }
        return;
    }
    for (unsigned q = threadIdx.x; q < quads; q += 256) {
        const unsigned r = rb > 1 ? (l4 == 1 ? q : __umulhi(q, magic)) : 0u;
        const int col = (int)(q - r * l4) * 4;
        const long b = b0 + r;
        const long a = action[b];
        const int lo = (a >= 0 && a < (long)N) ? (int)a * K : -K - 4;   // an out-of-range action selects nothing
        vfloat4 v = zero4;
        if (col + 4 > lo && col < lo + K) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = col + j - lo;
                if (k >= 0 && k < K) v[j] = u * (planes > 1 ? buf[b * planes + plane] : buf[b * K + k]);
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(out + (long)q * 4));
    }
}

// Large outputs: the zeros come from the fill that reaches the part's write rate (stream_write.hpp), this kernel adds the
// K values per sample -- as WHOLE 128-byte lines: the K floats at [a*K, a*K + K) of the row are written together with the
// zeros around them up to the next line boundaries (inside the row), because a store that covers part of a line that is no
// longer in a cache is a read-modify-write at the memory (measured: 62 us for the 13 M values of C51 at B = 262144 written
// as they lie, 204 bytes per row).  32 lanes per sample, one 16-byte store each (K <= 64; more: the lanes loop).
__global__ __launch_bounds__(256) void onehot_values_kernel(const float* __restrict__ g, const float* __restrict__ buf,
                                                            const int64_t* __restrict__ action, float* __restrict__ grad,
                                                            long B, int N, int K) {
    const float u = g[0];
    const long b = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (b >= B) return;
    const int j = threadIdx.x & 31;
    const long a = action[b];
    if (a < 0 || a >= (long)N) return;
    const long L = (long)N * K;
    const uintptr_t row_s = reinterpret_cast<uintptr_t>(grad + b * L), row_e = row_s + (uintptr_t)L * 4;
    const uintptr_t lo = row_s + (uintptr_t)a * K * 4, hi = lo + (uintptr_t)K * 4;
    uintptr_t ss = lo & ~(uintptr_t)127, se = (hi + 127) & ~(uintptr_t)127;
    if (ss < row_s) ss = row_s;        // (rows are multiples of 16 bytes: L % 4 == 0 and a 16-byte aligned base)
    if (se > row_e) se = row_e;
    const float* __restrict__ bv = buf + b * K;
    for (uintptr_t p = ss + 16 * (uintptr_t)j; p < se; p += 512) {
        vfloat4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const long k = p + 4 * e >= lo ? (long)((p + 4 * e - lo) >> 2) : -1;
            v[e] = (k >= 0 && k < K) ? u * bv[k] : 0.f;
        }
        *reinterpret_cast<vfloat4*>(p) = v;
    }
}

}  // namespace

int g_onehot_fill_mb = 3072;  // hpc_rll_tune_set key 31: outputs of at least this many MiB are written as fill + values (0 = never)
int g_onehot_qpw = 0;         // hpc_rll_tune_set key 35: 16-byte quads per workgroup of the one-launch one-hot kernel (0 = by size)

// planes > 1: grad is (planes, B, N) and buf is (B, planes) (K must be 1); else grad is (B, N, K), buf (B, K).
int onehot_scatter(const float* g, const float* buf, const int64_t* action, float* grad, long B, int N, int K,
                   hipStream_t st, int planes) {
    const long total = B * N * K * (planes > 1 ? planes : 1);
    if (total == 0) return HPC_RLL_OK;
    const long L = (long)N * K;
    // (K >= 16: with one value per row of N floats -- q-TD, IQN -- the values pass touches every other line of the output again
    // and the one-launch kernel wins: 0.114 against 0.181 ms for IQN's 537 MB)
    if (g_onehot_fill_mb > 0 && total * 4 >= (long)g_onehot_fill_mb << 20 && (reinterpret_cast<uintptr_t>(grad) & 15) == 0 &&
        planes <= 1 && K >= 16 && L % 4 == 0) {
        const int rc = launch_stream_zero(grad, (size_t)total, st);
        if (rc) return rc;
        hipLaunchKernelGGL(onehot_values_kernel, dim3((unsigned)((B + 7) / 8)), dim3(256), 0, st, g, buf, action, grad, B, N, K);
        return last_error();
    }
    if (L % 4 == 0 && L / 4 < (1L << 20) && (reinterpret_cast<uintptr_t>(grad) & 15) == 0 && (planes <= 1 || K == 1) &&
        planes <= 65535) {
        const unsigned l4 = (unsigned)(L / 4);
        // Quads per workgroup (tune key 35; 0 = by size).  Round 4: 4096 (16 per thread, a dependent action / value load chain each)
        // -> 256 for outputs of 256 MiB and more: ONE quad per thread, the workgroup writes one 4 KiB block and retires (IQN
        // 0.114 -> 0.091 ms, QR-DQN 0.412 -> 0.358, C51 0.642 -> 0.623); 1024 below that (q-TD at 67 MB: 23.2 -> 16.9 us, 256: 20.0)
        const long qpw = g_onehot_qpw > 0 ? g_onehot_qpw : (total * 4 >= (256L << 20) ? 256 : 1024);
        const int rb = (long)l4 >= qpw ? 1 : (int)(qpw / l4);
        const unsigned magic = (unsigned)(((1ull << 32) + l4 - 1) / l4);   // q / l4 == umulhi(q, magic) for q < 2^32 / l4
        const long blocks = (B + rb - 1) / rb;
        hipLaunchKernelGGL(onehot_rows4_kernel, dim3((unsigned)blocks, planes > 1 ? planes : 1), dim3(256), 0, st, g, buf,
                           action, grad, B, N, K, rb, l4, l4 == 1 ? 0u : magic);
        return last_error();
    }
    if (planes > 1) return HPC_RLL_EUNSUPPORTED;   // (the IQN caller keeps its own 4-byte kernel for this case)
    long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(onehot_scatter_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, buf, action, grad, B, N, K);
    return last_error();
}

}  // namespace hpc_rll

using namespace hpc_rll;

// ws layout (floats): [coef_logp B | coef_ent B | gv_unit B | logp_new B | ent B | logp_old B | partials]
// (partials: 5 sums x the workgroups of the launch -- ceil(B / 256) for the three-launch forward, up to 4096 for the fused one)
extern "C" int64_t hpc_rll_ppo_workspace_floats(int B) {
    const int64_t blocks = ((int64_t)B + 255) / 256 + 1;
    return 6 * (int64_t)B + 8 * (blocks > 4097 ? blocks : 4097);
}

extern "C" int hpc_rll_ppo_forward(const float* logits_new, const float* logits_old, const int64_t* action,
                                   const float* value_new, const float* value_old, const float* adv,
                                   const float* ret, const float* weight, float* out5, float* ws, int B, int N,
                                   float clip_ratio, int use_value_clip, float dual_clip, float scale,
                                   void* stream) {
    if (B < 0 || N <= 0 || !out5) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(out5, 0, 5 * sizeof(float), st);
    if (!logits_new || !logits_old || !action || !value_new || !value_old || !adv || !ret || !ws)
        return HPC_RLL_EINVAL;
    float *coef_logp = ws, *coef_ent = ws + B, *gv_unit = ws + 2 * (size_t)B, *lpn = ws + 3 * (size_t)B,
          *ent = ws + 4 * (size_t)B, *lpo = ws + 5 * (size_t)B, *partials = ws + 6 * (size_t)B;
    PpoOp op{lpn, ent, lpo, value_new, value_old, adv, ret, weight, coef_logp, coef_ent, gv_unit,
             clip_ratio, dual_clip, scale, use_value_clip};
    // approx_kl and clipfrac are plain (unweighted) means over the LOCAL batch: scale by 1/B
    const float sc[5] = {scale, 0.5f * scale, scale, 1.f / (float)B, 1.f / (float)B};
    int rc = HPC_RLL_OK;
    if (g_ppo_fused && ppo_forward_fused(logits_new, logits_old, action, op, B, N, partials, sc, out5, st, &rc)) return rc;
    rc = categorical_forward(logits_new, action, lpn, ent, B, N, st);
    if (rc) return rc;
    rc = categorical_forward(logits_old, action, lpo, nullptr, B, N, st);
    if (rc) return rc;
    return launch_sample(op, (long)B, partials, 5, sc, out5, st);
}

extern "C" int hpc_rll_ppo_backward(const float* g_policy, const float* g_value, const float* g_ent,
                                    const float* logits_new, const int64_t* action, const float* ws,
                                    float* grad_logits_new, float* grad_value_new, int B, int N, void* stream) {
    if (B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!ws) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int rc = HPC_RLL_OK;
    if (grad_value_new) {
        if (!g_value) return HPC_RLL_EINVAL;
        rc = scale_rows(g_value, ws + 2 * (size_t)B, grad_value_new, B, B, st);
        if (rc) return rc;
    }
    if (grad_logits_new) {
        if (!logits_new || !action) return HPC_RLL_EINVAL;
        rc = categorical_backward(logits_new, action, ws, g_policy, ws + B, g_ent, grad_logits_new, B, N, st);
    }
    return rc;
}

// q n-step TD (rescale = 0) and with value rescaling (rescale = 1).
extern "C" int hpc_rll_q_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                                          const int64_t* next_n_action, const float* reward, const float* done,
                                          const float* weight, float* loss, float* td_err, float* grad_buf,
                                          float* partials, int nstep, int B, int N, float gamma, int rescale,
                                          float scale, void* stream) {
    if (nstep < 0 || B < 0 || N <= 0 || !loss) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!q || !next_n_q || !action || !next_n_action || (nstep && !reward) || !done || !td_err || !grad_buf ||
        !partials)
        return HPC_RLL_EINVAL;
    QNStepOp op{q, next_n_q, action, next_n_action, reward, done, weight, td_err, grad_buf,
                nstep, B, N, gamma, (float)pow((double)gamma, (double)nstep), scale, rescale};
    return launch_sample(op, (long)B, partials, 1, &scale, loss, st);
}

extern "C" int hpc_rll_q_nstep_td_backward(const float* grad_loss, const float* grad_buf, const int64_t* action,
                                           float* grad_q, int B, int N, void* stream) {
    if (B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!grad_loss || !grad_buf || !action || !grad_q) return HPC_RLL_EINVAL;
    return onehot_scatter(grad_loss, grad_buf, action, grad_q, B, N, 1, (hipStream_t)stream, 1);
}
