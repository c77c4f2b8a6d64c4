// dist_ops.hip -- distributional n-step TD errors for gfx950: C51 (dist), IQN and QR-DQN.
//
// Replaces (under /root/reference):
//   DistNStepTdForward/Backward        src/rl_utils/dist_nstep_td.cu:8-98,  dist_nstep_td_kernel.h:11-107
//   IQNNStepTDErrorForward/Backward    src/rl_utils/iqn_nstep_td_error.cu,  iqn_nstep_td_error_kernel.h:11-108
//   QRDQNNStepTDErrorForward/Backward  src/rl_utils/qrdqn_nstep_td_error.cu, qrdqn_nstep_td_error_kernel.h:11-106
// Semantics: hpc_rll/origin/td.py:56-143, 391-448, 480-517.
//
// One WAVE per sample b; the atoms / quantile pairs of that sample are spread over the 64 lanes, per-sample
// sums are wave butterflies, the batch loss goes through the usual deterministic two-stage reduction.
//   * C51: the reference scatters projected mass with float atomicAdd (dist_nstep_td_kernel.h:58-59, order
//     dependent).  Here each lane OWNS target atoms and gathers the (few) source atoms that land on them, in
//     source order: deterministic, no atomics, no scratch proj buffer round trip.
//   * IQN / QR-DQN: the reference materialises three (B,tau',tau) scratch tensors over three launches; here the
//     tau x tau' pair loop runs in registers, one launch, and only the (B,tau) unit gradient is written.
#include <hip/hip_runtime.h>

#include "colscan.hpp"
#include "hpc_rll_hip.h"
#include "nstep.hpp"

namespace hpc_rll {

int onehot_scatter(const float* g, const float* buf, const int64_t* action, float* grad, long B, int N, int K,
                   hipStream_t st, int planes = 1);

namespace {

inline int last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

// 4 waves (= 4 samples) per workgroup; workgroup partial = sum of its 4 per-sample weighted losses.
template <class F>
__device__ __forceinline__ void wave_per_sample(int B, float* __restrict__ partials, const ScanFold& fold, F&& body) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    float contrib = 0.f;
    if (b < B) contrib = body(b, lane);
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) s = (red[0] + red[1]) + (red[2] + red[3]);
    publish_sums<1, 256>(s, partials, fold);   // with a fold: the last workgroup also finalises the loss (colscan.hpp)
}

// ------------------------------------------------------------------------------------------------ C51
// buf[b, k] = -w_b * proj[b,k] / p[b,a,k] * scale    (unit gradient wrt dist[b, a_b, k])
__global__ __launch_bounds__(256) void dist_nstep_fwd_kernel(
    const float* __restrict__ dist, const float* __restrict__ next_dist, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, float* __restrict__ td_err, float* __restrict__ buf,
    float* __restrict__ partials, int nstep, int B, int N, int n_atom, float gamma, float gamma_n, float v_min,
    float v_max, float dz, float scale, const ScanFold fold) {
    wave_per_sample(B, partials, fold, [&](int b, int lane) -> float {
        const float R = nstep_return1(reward, B, nstep, gamma, b);
        const float nd_scale = (1.f - done[b]) * gamma_n;
        const float* __restrict__ p = dist + ((size_t)b * N + action[b]) * n_atom;
        const float* __restrict__ pn = next_dist + ((size_t)b * N + next_action[b]) * n_atom;
        const float w = weight ? weight[b] : 1.f;
        float ce = 0.f;  // sum_k proj_k * log p_k over this lane's atoms
        for (int k = lane; k < n_atom; k += 64) {
            float proj = 0.f;
            // gather: source atoms j whose floor / ceil position is k (source order => deterministic sum)
            for (int j = 0; j < n_atom; ++j) {
                // support_j: torch.linspace(v_min, v_max, n_atom) evaluates start + j*step below the midpoint
                // and end - (n-1-j)*step above it
                const float step = (v_max - v_min) / (float)(n_atom - 1);
                const float sup = (j < n_atom / 2) ? (v_min + step * (float)j) : (v_max - step * (float)(n_atom - 1 - j));
                // unfused mul/add/div, like the oracle's tensor ops (origin/td.py:95-98)
                float tz = __fadd_rn(R, __fmul_rn(nd_scale, sup));
                tz = fminf(fmaxf(tz, v_min), v_max);
                const float bp = __fdiv_rn(__fsub_rn(tz, v_min), dz);
                const float lo = floorf(bp), up = ceilf(bp);
                if ((int)lo == k) proj = fmaf(pn[j], up - bp, proj);
                if ((int)up == k) proj = fmaf(pn[j], bp - lo, proj);
            }
            const float pk = p[k];
            ce = fmaf(proj, logf(pk), ce);
            buf[(size_t)b * n_atom + k] = -w * proj / pk * scale;
        }
        ce = wave_sum(ce);
        if (lane == 0) td_err[b] = -ce;
        return -ce * w;
    });
}

// ------------------------------------------------------------------------------------------------ IQN
// q (tau,B,N), next_q (tau',B,N), replay_quantiles (tau,B).  buf[b,i] = unit gradient wrt q[i,b,a_b].
__global__ __launch_bounds__(256) void iqn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ rq, const float* __restrict__ weight, const float* __restrict__ value_gamma,
    float* __restrict__ td_err, float* __restrict__ buf, float* __restrict__ partials, int tau, int tau_p,
    int nstep, int B, int N, float gamma, float gamma_n, float kappa, float scale, const ScanFold fold, int bnt) {
    wave_per_sample(B, partials, fold, [&](int b, int lane) -> float {
        const float R = nstep_return1(reward, B, nstep, gamma, b);
        const float vg = (value_gamma ? value_gamma[b] : gamma_n) * (1.f - done[b]);
        const long a = action[b], na = next_action[b];
        const float w = weight ? weight[b] : 1.f;
        const float inv_tp = 1.f / (float)tau_p;
        float loss = 0.f;
        for (int i = lane; i < tau; i += 64) {
            const float qi = bnt ? q[((size_t)b * N + a) * tau + i] : q[((size_t)i * B + b) * N + a];
            const float rho = rq[(size_t)i * B + b];
            float li = 0.f, gi = 0.f;
            for (int j = 0; j < tau_p; ++j) {
                const float tgt = fmaf(vg, bnt ? next_q[((size_t)b * N + na) * tau_p + j] : next_q[((size_t)j * B + b) * N + na], R);
                const float e = tgt - qi;
                const float ae = fabsf(e);
                const float hub = (ae <= kappa) ? 0.5f * e * e : kappa * (ae - 0.5f * kappa);
                const float dh = (ae <= kappa) ? e : ((e > 0.f) ? kappa : -kappa);
                const float qw = fabsf(rho - ((e < 0.f) ? 1.f : 0.f)) / kappa;
                li = fmaf(qw, hub, li);
                gi = fmaf(qw, dh, gi);
            }
            loss += li;
            buf[(size_t)b * tau + i] = -gi * inv_tp * w * scale;   // de/dq = -1
        }
        loss = wave_sum(loss) * inv_tp;
        if (lane == 0) td_err[b] = loss;
        return loss * w;
    });
}

// grad_q[i,b,n] = (n == a_b) ? g * buf[b,i] : 0      for q laid out (tau,B,N)
__global__ __launch_bounds__(256) void iqn_bwd_kernel(const float* __restrict__ g, const float* __restrict__ buf,
                                                      const int64_t* __restrict__ action, float* __restrict__ grad,
                                                      int tau, int B, int N) {
    const float u = g[0];
    const long total = (long)tau * B * N;
    for (long x = (long)blockIdx.x * 256 + threadIdx.x; x < total; x += (long)gridDim.x * 256) {
        const int n = (int)(x % N);
        const long ib = x / N;
        const int b = (int)(ib % B);
        const int i = (int)(ib / B);
        __builtin_nontemporal_store(((long)n == action[b]) ? u * buf[(size_t)b * tau + i] : 0.f, grad + x);
    }
}

// ------------------------------------------------------------------------------------------------ QR-DQN
// q (B,N,tau), next_q (B,N,tau).  buf[b,i] = unit gradient wrt q[b,a_b,i].
__global__ __launch_bounds__(256) void qrdqn_fwd_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, const float* __restrict__ value_gamma, float* __restrict__ td_err,
    float* __restrict__ buf, float* __restrict__ partials, int tau, int nstep, int B, int N, float gamma,
    float gamma_n, float tau_value, float scale, const ScanFold fold) {
    wave_per_sample(B, partials, fold, [&](int b, int lane) -> float {
        const float R = nstep_return1(reward, B, nstep, gamma, b);
        const float vg = (value_gamma ? value_gamma[b] : gamma_n) * (1.f - done[b]);
        const float* __restrict__ qa = q + ((size_t)b * N + action[b]) * tau;
        const float* __restrict__ qn = next_q + ((size_t)b * N + next_action[b]) * tau;
        const float w = weight ? weight[b] : 1.f;
        const float inv_tau = 1.f / (float)tau;
        float loss = 0.f;
        for (int i = lane; i < tau; i += 64) {
            const float qi = qa[i];
            float li = 0.f, gi = 0.f;
            for (int j = 0; j < tau; ++j) {
                const float e = fmaf(vg, qn[j], R) - qi;
                const float ae = fabsf(e);
                const float u = (ae < 1.f) ? 0.5f * e * e : ae - 0.5f;       // smooth_l1, beta = 1
                const float du = (ae < 1.f) ? e : ((e > 0.f) ? 1.f : -1.f);
                const float qw = fabsf(tau_value - ((e <= 0.f) ? 1.f : 0.f));
                li = fmaf(qw, u, li);
                gi = fmaf(qw, du, gi);
            }
            loss += li;
            buf[(size_t)b * tau + i] = -gi * inv_tau * w * scale;
        }
        loss = wave_sum(loss) * inv_tau;
        if (lane == 0) td_err[b] = loss;
        return loss * w;
    });
}

// ------------------------------------------------------------------------------------------------ large batches
// The three wave-per-sample kernels above are written for the reference's test shapes (a handful of samples).  At
// B ~ 10^5 they are what sets the time, and they leave most of the machine idle:
//   * C51 recomputes the projection of ALL n_atom source atoms in every target lane (n_atom^2 / 64 projections per lane:
//     a division, floor, ceil each);
//   * IQN / QR-DQN use tau of the 64 lanes and re-load the tau' targets from memory in every lane's inner loop.
// Below: the same arithmetic in the same order per element (results identical up to the order of the final per-sample
// sum), with
//   * C51 (n_atom <= 64): lane j projects source atom j ONCE; a target lane then visits only the contiguous run of
//     sources that can reach it (found by binary search over the monotone floor positions), fetching them with
//     ds_bpermute -- ~5 sources instead of n_atom, each ~10 instructions instead of ~30;
//   * IQN / QR-DQN (tau, tau' <= 64): a sample is held by a GROUP of G = 2^k >= max(tau, tau') lanes, 64/G samples per
//     wave; lane j of the group builds target j once, the pair loop fetches it with one ds_bpermute.
template <int G> __device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int m = G / 2; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}
// 4 waves x 64/G samples per workgroup; workgroup partial = the per-sample contributions added in sample order.
template <int G, class F>
__device__ __forceinline__ void group_per_sample(int B, float* __restrict__ partials, const ScanFold& fold, F&& body) {
    constexpr int SPW = 64 / G;
    __shared__ float red[4 * SPW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gl = lane % G, gs = lane / G;
    const long b = ((long)blockIdx.x * 4 + w) * SPW + gs;
    const float contrib = body(b, b < (long)B, gl, gs * G);   // every lane runs the body (it shuffles): loads are guarded
    if (gl == 0) red[w * SPW + gs] = b < (long)B ? contrib : 0.f;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 4 * SPW; ++i) s += red[i];
    }
    publish_sums<1, 256>(s, partials, fold);
}

// One sample of the n_atom <= 64 projection: lane j holds source atom j's next-state mass `pn_j`, lane k accumulates the
// projected mass of target atom k.  Returns proj for this lane.
__device__ __forceinline__ float c51_project64(int lane, int n_atom, float R, float nd_scale, float pn_j, float v_min,
                                               float v_max, float dz) {
    // source atom j = lane (the expressions of dist_nstep_fwd_kernel, evaluated once per source)
    const int j = lane < n_atom ? lane : n_atom - 1;
    const float step = (v_max - v_min) / (float)(n_atom - 1);
    const float sup = (j < n_atom / 2) ? (v_min + step * (float)j) : (v_max - step * (float)(n_atom - 1 - j));
    float tz = __fadd_rn(R, __fmul_rn(nd_scale, sup));
    tz = fminf(fmaxf(tz, v_min), v_max);
    const float bp = __fdiv_rn(__fsub_rn(tz, v_min), dz);
    const float lo = floorf(bp), up = ceilf(bp);
    const int i_lo = (int)lo, i_up = (int)up;
    const int f_pn = __builtin_bit_cast(int, pn_j);
    const int f_al = __builtin_bit_cast(int, up - bp), f_au = __builtin_bit_cast(int, bp - lo);
    float proj = 0.f;   // target atom k = lane: sources in order, the floor hit before the ceil hit of each source
    if (nd_scale == 0.f) {
        // terminal sample (done = 1): tz = R for every source, so all of next_dist's mass lands on the SAME two atoms
        // and every lane holds the same i_lo / i_up / weights: one broadcast + two masked fmas per source
        const float al = up - bp, au = bp - lo;
        const bool hit_lo = lane == (int)lo, hit_up = lane == (int)up;
        for (int s = 0; s < n_atom; ++s) {
            const float s_pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(f_pn, s));
            if (hit_lo) proj = fmaf(s_pn, al, proj);
            if (hit_up) proj = fmaf(s_pn, au, proj);
        }
    } else if (nd_scale > 0.f) {
        // every step from j to floor(bp_j) is monotone (correctly rounded mul / add / div by positive numbers,
        // clamp, floor), so i_lo is non-decreasing over the source lanes and the sources that can touch target k
        // (floor = k, or floor = k-1 with ceil = k) are ONE contiguous run [first j: i_lo >= k-1, first j: i_lo > k):
        // two 7-step binary searches through ds_bpermute, then a loop as long as the longest run of the wave
        // (1/nd_scale + 2 sources).
        int b0 = 0, e0 = n_atom, b1 = 0, e1 = n_atom;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int m0 = (b0 + e0) >> 1, m1 = (b1 + e1) >> 1;
            const int v0 = __shfl(i_lo, m0 < n_atom ? m0 : n_atom - 1, 64);
            const int v1 = __shfl(i_lo, m1 < n_atom ? m1 : n_atom - 1, 64);
            if (b0 < e0) { if (v0 < lane - 1) b0 = m0 + 1; else e0 = m0; }
            if (b1 < e1) { if (v1 <= lane) b1 = m1 + 1; else e1 = m1; }
        }
        const int len = lane < n_atom ? b1 - b0 : 0;
        const int maxlen = __builtin_amdgcn_readfirstlane((int)wave_max((float)len));
        for (int c = 0; c < maxlen; ++c) {
            const int src = b0 + c < n_atom ? b0 + c : n_atom - 1;
            const int s_lo = __shfl(i_lo, src, 64), s_up = __shfl(i_up, src, 64);
            const float s_pn = __builtin_bit_cast(float, __shfl(f_pn, src, 64));
            const float s_al = __builtin_bit_cast(float, __shfl(f_al, src, 64));
            const float s_au = __builtin_bit_cast(float, __shfl(f_au, src, 64));
            if (c < len && s_lo == lane) proj = fmaf(s_pn, s_al, proj);
            if (c < len && s_up == lane) proj = fmaf(s_pn, s_au, proj);
        }
    } else {   // done > 1 (not a flag): no monotonicity to use, walk every source
        for (int s = 0; s < n_atom; ++s) {
            const int s_lo = __builtin_amdgcn_readlane(i_lo, s), s_up = __builtin_amdgcn_readlane(i_up, s);
            const float s_pn = __builtin_bit_cast(float, __builtin_amdgcn_readlane(f_pn, s));
            const float s_al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(f_al, s));
            const float s_au = __builtin_bit_cast(float, __builtin_amdgcn_readlane(f_au, s));
            if (s_lo == lane) proj = fmaf(s_pn, s_al, proj);
            if (s_up == lane) proj = fmaf(s_pn, s_au, proj);
        }
    }
    return proj;
}

// Round 3: S consecutive samples per wave (4 waves x S samples per workgroup).  A sample is a chain of dependent memory
// round trips -- action -> row address -> row -- and with one sample per wave the kernel was bound by that latency times
// the number of waves a SIMD can hold (0.26 ms for 248 + 59 MB at B = 262144): here the per-sample
// scalars of all S samples are fetched together, then the 2*S rows, then the S projections run back to back.  The
// arithmetic of a sample is unchanged (bit-identical td_err / buf); the workgroup partial adds 4*S contributions in order.
// (Dispatched with S = 1 since dist_nstep_fwd_batch_kernel took over the large batches; S = 4 measured 0.187 ms at
// B = 262144 against 0.069 ms for the batch kernel, tests/tools/r03_batch_probe.py.)
template <int S>
__global__ __launch_bounds__(256) void dist_nstep_fwd64_kernel(
    const float* __restrict__ dist, const float* __restrict__ next_dist, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, float* __restrict__ td_err, float* __restrict__ buf,
    float* __restrict__ partials, int nstep, int B, int N, int n_atom, float gamma, float gamma_n, float v_min,
    float v_max, float dz, float scale, const ScanFold fold) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long first = ((long)blockIdx.x * 4 + w) * S;
    long bb[S];
    long a[S], na[S];
    float dn[S], wt[S], R[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        bb[s] = first + s < (long)B ? first + s : (long)B - 1;       // clamped: every load below is unconditional
        a[s] = action[bb[s]];
        na[s] = next_action[bb[s]];
        dn[s] = done[bb[s]];
        wt[s] = weight ? weight[bb[s]] : 1.f;
    }
    nstep_returns<S>(reward, B, nstep, gamma, bb, R);
    const int jl = lane < n_atom ? lane : n_atom - 1;
    float pk[S], pnj[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        pk[s] = dist[((size_t)bb[s] * N + a[s]) * n_atom + jl];
        pnj[s] = next_dist[((size_t)bb[s] * N + na[s]) * n_atom + jl];
    }
    float contrib = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const float nd_scale = (1.f - dn[s]) * gamma_n;
        const float proj = c51_project64(lane, n_atom, R[s], nd_scale, pnj[s], v_min, v_max, dz);
        float ce = 0.f;
        const bool live = first + s < (long)B;
        if (lane < n_atom) {
            ce = proj * logf(pk[s]);
            if (live) buf[(size_t)bb[s] * n_atom + lane] = -wt[s] * proj / pk[s] * scale;
        }
        ce = wave_sum(ce);
        if (live) {
            if (lane == 0) td_err[bb[s]] = -ce;
            contrib += -ce * wt[s];
        }
    }
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x == 0) tot = (red[0] + red[1]) + (red[2] + red[3]);
    publish_sums<1, 256>(tot, partials, fold);
}

// The projection for large batches.  (A first version scattered with LDS float atomics, two ds_add_f32 per sample in the
// oracle's order: bit-identical to the sequential scatter, but ds_add_f32 runs at about one lane per two clocks on this
// part -- 110 of the kernel's 154 us at B = 262144, tests/tools/r03_batch_probe.py.)
// j -> floor(bp_j) is monotone for any sign of nd_scale (correctly rounded mul / add / div, clamp, floor), so the sources
// that share a floor atom are ONE contiguous run of lanes, and target atom k receives the floor weights of run(k) and the
// ceil weights of run(k-1) (a source whose position is integral has weight 0 both ways, like the reference's l == u case).
//   1. run boundaries: a lane whose neighbour (DPP wave shift) has another floor atom marks start / end of its run in a
//      64-entry LDS table indexed by atom (plain stores, no two lanes write the same word);
//   2. both weight streams are summed along the runs by a segmented doubling scan (ds_bpermute; offsets 8..32 only when
//      some run is that long: clamped returns make runs of 10+ sources common, terminal samples one run of all atoms);
//   3. target k fetches the totals at end(run(k)) and end(run(k-1)).
// Fixed cost whatever the run lengths (the search-and-gather form above walks the longest run of the wave with five
// ds_bpermute per step).  The sum of a run is formed in doubling order, not source order: deterministic, equal to the
// oracle's sequential scatter up to fp32 rounding of a few terms (tests/test_losses_gpu.py pins it at 4e-7 of the maximum).
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int x) { return __builtin_amdgcn_update_dpp(old, x, CTRL, 0xF, 0xF, false); }

__device__ __forceinline__ float c51_project_scan(int* __restrict__ se, int lane, int n_atom, float R, float nd_scale,
                                                  float pn_j, float v_min, float v_max, float dz, float inv_dz) {
    const bool src = lane < n_atom;
    const int j = src ? lane : n_atom - 1;
    const float step = (v_max - v_min) / (float)(n_atom - 1);
    const float sup = (j < n_atom / 2) ? (v_min + step * (float)j) : (v_max - step * (float)(n_atom - 1 - j));
    float tz = __fadd_rn(R, __fmul_rn(nd_scale, sup));
    tz = fminf(fmaxf(tz, v_min), v_max);
    // bp = (tz - v_min) / dz, correctly rounded, without the division sequence (two v_div_scale, v_rcp, five fmas, v_div_fmas,
    // v_div_fixup): dz is one number per launch, so its correctly rounded reciprocal comes from the host, and two residual
    // corrections of x * (1 / dz) give the rounded quotient (Markstein; the hardware sequence is the same two corrections on a
    // reciprocal it refines itself, plus scaling for exponents this quotient -- an atom position in [0, n_atom) -- cannot have).
    // Checked against exact rational arithmetic on 48000 positions incl. the near-integer ones.
    const float x = __fsub_rn(tz, v_min);
    const float q0 = __fmul_rn(x, inv_dz);
    const float q1 = fmaf(fmaf(-dz, q0, x), inv_dz, q0);
    const float bp = fmaf(fmaf(-dz, q1, x), inv_dz, q1);
    const float lo = floorf(bp), up = ceilf(bp);
    int i_lo = (int)lo;
    i_lo = i_lo < 0 ? 0 : i_lo > 63 ? 63 : i_lo;              // NaN inputs must not leave the table
    float yl = src ? __fmul_rn(pn_j, up - bp) : 0.f, yu = src ? __fmul_rn(pn_j, bp - lo) : 0.f;
    // Round 5: a source at an INTEGRAL position (up == lo: both weights are 0, the reference's l == u case) stays outside
    // the runs.  Those are the sources clamped to v_min / v_max -- a run of a dozen zeros at an end of the support in an
    // ordinary sample -- and they alone made "some run is longer than two" the common case: with them out, a sample whose
    // atoms land more than half an atom apart (gamma^n (1 - done) > 0.5) has runs of one or two sources and needs ONE shuffle
    // step instead of three.  By monotonicity an integral position is the first (or last) of its floor's sources, never
    // between two others, so the remaining sources of a floor are still contiguous; the sums are unchanged (zeros left out).
    const bool nz = src && up != lo;
    const int key = nz ? i_lo : -1;
    const int prev = dpp_mov<0x138>(-2, key), next = dpp_mov<0x130>(-2, key);     // wave_shr:1 / wave_shl:1
    const bool is_start = nz && (lane == 0 || prev != key), is_end = nz && (lane == n_atom - 1 || next != key);
    int2* __restrict__ se2 = reinterpret_cast<int2*>(se);
    se2[lane] = int2{0, -1};                                  // empty run
    __builtin_amdgcn_wave_barrier();
    if (is_start) se[2 * i_lo] = lane;
    if (is_end) se[2 * i_lo + 1] = lane;
    __builtin_amdgcn_wave_barrier();
    const int d = nz ? lane - se[2 * i_lo] : 0;               // distance to the start of my run
    const int2 mine = se2[lane];                              // run of target atom k = lane
    int2 below = se2[lane > 0 ? lane - 1 : 0];
    if (lane == 0) below = int2{0, -1};
    __builtin_amdgcn_wave_barrier();                          // the table is reused by the wave's next sample
    {
        const float tl = __shfl_up(yl, 1, 64), tu = __shfl_up(yu, 1, 64);
        if (1 <= d) { yl += tl; yu += tu; }
    }
    if (__builtin_amdgcn_ballot_w64(d >= 2)) {
#pragma unroll
        for (int off = 2; off <= 4; off <<= 1) {
            const float tl = __shfl_up(yl, off, 64), tu = __shfl_up(yu, off, 64);
            if (off <= d) { yl += tl; yu += tu; }
        }
        if (__builtin_amdgcn_ballot_w64(d >= 8)) {
#pragma unroll
            for (int off = 8; off <= 32; off <<= 1) {
                const float tl = __shfl_up(yl, off, 64), tu = __shfl_up(yu, off, 64);
                if (off <= d) { yl += tl; yu += tu; }
            }
        }
    }
    const float fsum = __shfl(yl, mine.y < 0 ? 0 : mine.y, 64), csum = __shfl(yu, below.y < 0 ? 0 : below.y, 64);
    return (mine.y >= mine.x ? fsum : 0.f) + (below.y >= below.x ? csum : 0.f);
}

// Large batches: SW consecutive samples per wave, per-sample scalars loaded coalesced by their owner lanes (as
// qrdqn_fwd_batch_kernel below), the rows of U samples requested together, projection by run sums (c51_project_scan).
template <int SW>
__global__ __launch_bounds__(256) void dist_nstep_fwd_batch_kernel(
    const float* __restrict__ dist, const float* __restrict__ next_dist, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, float* __restrict__ td_err, float* __restrict__ buf,
    float* __restrict__ partials, int nstep, int B, int N, int n_atom, float gamma, float gamma_n, float v_min,
    float v_max, float dz, float inv_dz, float scale, const ScanFold fold) {
    constexpr int U = 8;
    static_assert(SW % U == 0, "samples per wave");
    __shared__ float red[4];
    __shared__ __attribute__((aligned(8))) int runs[4][128];       // per wave: {start, end} lane of the run of every atom
    // the wave index as a scalar: sample numbers and the addresses built from them stay in the scalar unit
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long b0 = ((long)blockIdx.x * 4 + w) * SW;
    const long bown = b0 + lane % SW;
    const long bl = bown < (long)B ? bown : (long)B - 1;
    const int a_l = (int)action[bl], na_l = (int)next_action[bl];
    const float w_l = weight ? weight[bl] : 1.f;
    const float nd_l = (1.f - done[bl]) * gamma_n;
    const float R_l = nstep_return1(reward, B, nstep, gamma, bl);
    const int jl = lane < n_atom ? lane : n_atom - 1;
    auto bcast = [](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
    float mine = 0.f;
    for (int c = 0; c < SW; c += U) {
        float pk[U], pnj[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // sample c+u is owned by lane c+u; its index (clamped as in phase A: a sample past the end of the batch is sample
            // B - 1 again) and its row offsets are scalars: 64-bit element offsets, B * N * n_atom may pass 2^31
            const long bs = b0 + c + u < (long)B ? b0 + c + u : (long)B - 1;
            pk[u] = dist[(bs * N + __builtin_amdgcn_readlane(a_l, c + u)) * n_atom + jl];
            pnj[u] = next_dist[(bs * N + __builtin_amdgcn_readlane(na_l, c + u)) * n_atom + jl];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float proj = c51_project_scan(runs[w], lane, n_atom, bcast(R_l, c + u), bcast(nd_l, c + u), pnj[u],
                                                v_min, v_max, dz, inv_dz);
            // a sample past the end of the batch is sample B - 1 again (phase A clamps): stored a second time, without a branch
            // (round 5: 69 -> 63 us at B = 262144)
            const long bst = b0 + c + u < (long)B ? b0 + c + u : (long)B - 1;
            float ce = 0.f;
            if (lane < n_atom) {
                ce = proj * logf(pk[u]);
                // (-w proj) / p: v_rcp_f32 and one residual correction (within an ulp of the division sequence, a third of its
                // instructions)
                const float num = -bcast(w_l, c + u) * proj, rp = __builtin_amdgcn_rcpf(pk[u]);
                const float g0 = num * rp;
                float gq = fmaf(fmaf(-pk[u], g0, num), rp, g0);
                //  v_rcp_f32 flushes subnormal inputs: a probability below FLT_MIN with proj == 0 (common when
                // done = 1) would give 0 * inf = NaN where the division -- the small-batch kernels, the reference -- gives 0,
                // and an overflowing quotient turns into NaN in the correction step.  Those lanes take the division.
                if (pk[u] < 1.17549435e-38f || !(fabsf(g0) <= 3.402823466e+38f)) gq = num / pk[u];
                buf[(size_t)bst * n_atom + lane] = gq * scale;
            }
            const float tot = bcast(group_sum_last<64>(ce), 63);      // DPP sum: no LDS round trips
            if (lane % SW == c + u) mine = -tot;
        }
    }
    const bool own = lane < SW && bown < (long)B;
    if (own) td_err[bown] = mine;
    const float contrib = wave_sum(own ? mine * w_l : 0.f);
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x == 0) tot = (red[0] + red[1]) + (red[2] + red[3]);
    publish_sums<1, 256>(tot, partials, fold);
}

// BNT (round 6): q (B,N,tau), next_q (B,N,tau') -- the sample's quantiles are ONE contiguous row (128 bytes at tau = 32) instead of
// tau values a whole (B,N) plane apart: 2 lines per sample instead of 2 tau.  Same arithmetic, same buf (B,tau).
template <int G, bool BNT>
__global__ __launch_bounds__(256) void iqn_fwd_group_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ rq, const float* __restrict__ weight, const float* __restrict__ value_gamma,
    float* __restrict__ td_err, float* __restrict__ buf, float* __restrict__ partials, int tau, int tau_p,
    int nstep, int B, int N, float gamma, float gamma_n, float kappa, float scale, const ScanFold fold) {
    group_per_sample<G>(B, partials, fold, [&](long b, bool ok, int gl, int base) -> float {
        float R = 0.f, vg = 0.f, w = 0.f, qi = 0.f, rho = 0.f, tgt = 0.f;
        if (ok) {
            // every scalar of the sample is requested before the first use (the reward loop below waits for its loads)
            const long a = action[b], na = next_action[b];
            const float dn = done[b], vgm = value_gamma ? value_gamma[b] : gamma_n;
            w = weight ? weight[b] : 1.f;
            if (gl < tau) rho = rq[(size_t)gl * B + b];
            R = nstep_return1(reward, B, nstep, gamma, b);
            vg = vgm * (1.f - dn);
            if (gl < tau) qi = BNT ? q[((size_t)b * N + a) * tau + gl] : q[((size_t)gl * B + b) * N + a];
            if (gl < tau_p) tgt = fmaf(vg, BNT ? next_q[((size_t)b * N + na) * tau_p + gl] : next_q[((size_t)gl * B + b) * N + na], R);
        }
        const float inv_tp = 1.f / (float)tau_p;
        // The pair loop is VALU-bound (tau * tau' pairs per sample, a wave64 instruction costs 4 cycles): two targets per
        // iteration in packed fp32 (v_pk_add / v_pk_fma / v_pk_mul), and per pair only
        //   dh = med3(e, -kappa, kappa)            (the Huber derivative: e inside, +-kappa outside)
        //   hub = dh * (e - 0.5 * dh)              (= 0.5 e^2 inside, kappa (|e| - 0.5 kappa) outside: the same roundings
        //                                           as the two-branch form, element by element)
        //   qw = e < 0 ? |rho - 1| / kappa : |rho| / kappa      (both hoisted out of the loop)
        // -- 5.5 instructions per pair instead of ~13.  Even and odd targets accumulate separately (summation order).
        const float qneg = fabsf(rho - 1.f) / kappa, qpos = fabsf(rho) / kappa;
        const vfloat2 q2 = {qi, qi}, mh2 = {-0.5f, -0.5f};
        vfloat2 li2 = {0.f, 0.f}, gi2 = {0.f, 0.f};
        for (int j = 0; j < tau_p; j += 2) {
            const bool two = j + 1 < tau_p;                         // wave-uniform
            vfloat2 t2;
            t2.x = __shfl(tgt, base + j, 64);
            t2.y = __shfl(tgt, base + (two ? j + 1 : j), 64);
            const vfloat2 e = t2 - q2;
            vfloat2 dh, qw;
            dh.x = __builtin_amdgcn_fmed3f(e.x, -kappa, kappa);
            dh.y = __builtin_amdgcn_fmed3f(e.y, -kappa, kappa);
            const vfloat2 hub = dh * __builtin_elementwise_fma(mh2, dh, e);
            qw.x = e.x < 0.f ? qneg : qpos;
            qw.y = two ? (e.y < 0.f ? qneg : qpos) : 0.f;
            li2 = __builtin_elementwise_fma(qw, hub, li2);
            gi2 = __builtin_elementwise_fma(qw, dh, gi2);
        }
        const float li = li2.x + li2.y, gi = gi2.x + gi2.y;
        if (ok && gl < tau) buf[(size_t)b * tau + gl] = -gi * inv_tp * w * scale;   // de/dq = -1
        const float loss = group_sum<G>(gl < tau ? li : 0.f) * inv_tp;
        if (ok && gl == 0) td_err[b] = loss;
        return loss * w;
    });
}

// The QR-DQN pair arithmetic (round 5).  For e = target - q, the quantile Huber term of a pair is
//   qw * u,  u = du * (e - 0.5 du),  du = clamp(e, -1, 1),  qw = e <= 0 ? |tau - 1| : |tau|
// and its derivative qw * du.  Rounds 2-4 formed du with v_med3_f32 and qw with v_cmp + v_cndmask per element: 11 VALU
// instructions per PAIR of targets (5 packed + 6 scalar), and on gfx950 every v_cmp -> v_cndmask hand-over through VCC costs two
// wait states the compiler could fill only half of the time (221 s_nop in the unrolled loop of the batch kernel: 0.61 of the
// issue slots doing work).  Here the two signs are split by the CLAMP output modifier of the packed subtract:
//   dp = clamp01(t - q),  dn = clamp01(q - t)            (one of them is 0; du = dp - dn)
//   up = dp * (e - 0.5 dp),  un = dn * (-e - 0.5 dn)     (u = up + un, one of them is 0; same roundings as du * (e - 0.5 du))
// and the two quantile weights multiply the four SUMS once per sample instead of every term:
//   loss_i = |tau| * sum(up) + |tau - 1| * sum(un),   dloss_i = |tau| * sum(dp) - |tau - 1| * sum(dn)
// -- 9 packed instructions per pair, nothing through VCC.  Per-term values are those of the old form; the sums differ from
// it by where the weight is applied (fp32 rounding of a 32-term sum; tests pin the fp64 oracle at 2e-5).
struct QrSums { vfloat2 up, un, dp, dn; };
__device__ __forceinline__ vfloat2 pk_sub_clamp01(vfloat2 a, vfloat2 b) {
    vfloat2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void qr_pair(QrSums& s, vfloat2 t2, vfloat2 q2) {
    const vfloat2 mh2 = {-0.5f, -0.5f};
    const vfloat2 e = t2 - q2;
    const vfloat2 dp = pk_sub_clamp01(t2, q2), dn = pk_sub_clamp01(q2, t2);
    s.up = __builtin_elementwise_fma(dp, __builtin_elementwise_fma(mh2, dp, e), s.up);
    s.un = __builtin_elementwise_fma(dn, __builtin_elementwise_fma(mh2, dn, -e), s.un);
    s.dp += dp;
    s.dn += dn;
}
// (li, gi) of lane i from the four sums: even targets in .x, odd ones in .y
__device__ __forceinline__ void qr_finish(const QrSums& s, float qneg, float qpos, float& li, float& gi) {
    li = fmaf(qneg, s.un.x + s.un.y, qpos * (s.up.x + s.up.y));
    gi = fmaf(-qneg, s.dn.x + s.dn.y, qpos * (s.dp.x + s.dp.y));
}

template <int G>
__global__ __launch_bounds__(256) void qrdqn_fwd_group_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, const float* __restrict__ value_gamma, float* __restrict__ td_err,
    float* __restrict__ buf, float* __restrict__ partials, int tau, int nstep, int B, int N, float gamma,
    float gamma_n, float tau_value, float scale, const ScanFold fold) {
    group_per_sample<G>(B, partials, fold, [&](long b, bool ok, int gl, int base) -> float {
        float R = 0.f, w = 0.f, qi = 0.f, tgt = 0.f;
        if (ok) {
            const long a = action[b], na = next_action[b];
            const float dn = done[b], vgm = value_gamma ? value_gamma[b] : gamma_n;
            w = weight ? weight[b] : 1.f;
            R = nstep_return1(reward, B, nstep, gamma, b);
            const float vg = vgm * (1.f - dn);
            if (gl < tau) {
                qi = q[((size_t)b * N + a) * tau + gl];
                tgt = fmaf(vg, next_q[((size_t)b * N + na) * tau + gl], R);
            }
        }
        const float inv_tau = 1.f / (float)tau;
        // qr_pair above: two targets per iteration; an odd tail pairs the last target with e = 0 (contributes nothing)
        const float qneg = fabsf(tau_value - 1.f), qpos = fabsf(tau_value);
        const vfloat2 q2 = {qi, qi};
        QrSums sm = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
        for (int j = 0; j < tau; j += 2) {
            const bool two = j + 1 < tau;
            vfloat2 t2;
            t2.x = __shfl(tgt, base + j, 64);
            t2.y = __shfl(tgt, base + (two ? j + 1 : j), 64);
            if (!two) t2.y = qi;
            qr_pair(sm, t2, q2);
        }
        float li, gi;
        qr_finish(sm, qneg, qpos, li, gi);
        if (ok && gl < tau) buf[(size_t)b * tau + gl] = -gi * inv_tau * w * scale;
        const float loss = group_sum<G>(gl < tau ? li : 0.f) * inv_tau;
        if (ok && gl == 0) td_err[b] = loss;
        return loss * w;
    });
}

// Round 3, large batches: a wave takes SW CONSECUTIVE samples.  The group kernel above is a chain of dependent round trips
// per sample (action -> row address -> row) with 64/G samples in flight per wave: at B = 262144, tau = 32 it ran 0.128 ms
// for 16 rounds of resident waves -- ~8 us per round against ~1.7 us of pair-loop arithmetic.  Here
//   phase A: lane l holds the per-sample scalars of sample b0 + l (action, next action, done, weight, n-step return):
//            coalesced loads, ONE round trip for all SW samples;
//   phase B: the groups of G lanes walk the samples 64/G at a time; a sample's scalars come from its owner lane through
//            ds_bpermute, and the rows of U iterations are requested before the first pair loop runs.
// Per-sample arithmetic as in qrdqn_fwd_group_kernel (td_err / buf bit-identical); the workgroup partial adds its samples
// in butterfly order.
template <int G, int SW, bool FULL>
__global__ __launch_bounds__(256, 4) void qrdqn_fwd_batch_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, const float* __restrict__ value_gamma, float* __restrict__ td_err,
    float* __restrict__ buf, float* __restrict__ partials, int tau, int nstep, int B, int N, float gamma,
    float gamma_n, float tau_value, float scale, const ScanFold fold) {
    constexpr int SPW = 64 / G, NIT = SW / SPW, U = NIT < 4 ? NIT : 4;
    static_assert(SW % SPW == 0 && NIT % U == 0, "samples per wave");
    __shared__ float red[4];
    // per wave, two buffers of U iterations: the q row elements (lane = element) and the next-q row elements, which become the
    // targets in place
    __shared__ __attribute__((aligned(16))) float rq[4][2][U][64], tg[4][2][U][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gl = lane % G, gs = lane / G;
    const long b0 = ((long)blockIdx.x * 4 + w) * SW;
    // phase A (lanes >= SW repeat the first SW samples: harmless)
    const long bown = b0 + lane % SW;
    const long bl = bown < (long)B ? bown : (long)B - 1;
    const long row_l = ((long)bl * N + action[bl]) * tau, rown_l = ((long)bl * N + next_action[bl]) * tau;   // element offsets
    const float dn_l = done[bl], vgm_l = value_gamma ? value_gamma[bl] : gamma_n;
    const float w_l = weight ? weight[bl] : 1.f;
    const float R_l = nstep_return1(reward, B, nstep, gamma, bl);
    const float vg_l = vgm_l * (1.f - dn_l);
    const int glc = gl < tau ? gl : tau - 1;                  // clamped: the row loads are unconditional
    const float inv_tau = 1.f / (float)tau;
    const float qneg = fabsf(tau_value - 1.f), qpos = fabsf(tau_value);
    float mine = 0.f;                                         // td_err of the sample this lane owns
    // Round 5: the two rows of an iteration come by LDS-DMA (global_load_lds_dword: lane l's element lands in word l of the
    // wave's slot, no registers), and the rows of the NEXT U iterations are requested before the pair loops of the current
    // ones run.  Rounds 3-4 loaded them into registers right before use: a 2-3 us round trip exposed once per U iterations
    // with four waves per SIMD to cover it -- the pair loops themselves measured at the VALU issue rate (31 us of the kernel's
    // 57 at B = 262144, tests/tools/r05_qrdqn_ablate.py), the rest was this.  (Prefetching into registers instead: the
    // compiler spilled the prefetched values to scratch, one vmcnt(0) each.)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gl_ptr;
    auto rows = [&](int c, int pb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sl = (c + u) * SPW + gs;                // sample of this group, as a lane index of phase A
            __builtin_amdgcn_global_load_lds((gl_ptr)(q + __shfl(row_l, sl, 64) + glc), (lds_ptr)&rq[w][pb][u][0], 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gl_ptr)(next_q + __shfl(rown_l, sl, 64) + glc), (lds_ptr)&tg[w][pb][u][0], 4, 0, 0);
        }
    };
    rows(0, 0);
    for (int c = 0, pb = 0; c < NIT; c += U, pb ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the rows of these U iterations are in LDS
        // Targets stay in LDS: a group reads FOUR of them with one broadcast ds_read_b128 (a ds_bpermute per target plus its
        // lane arithmetic was half of the loop's instructions).  One wave writes and reads its own slots: LDS operations of a
        // wave execute in order, no barrier needed.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sl = (c + u) * SPW + gs;
            tg[w][pb][u][lane] = fmaf(__shfl(vg_l, sl, 64), tg[w][pb][u][lane], __shfl(R_l, sl, 64));
        }
        __builtin_amdgcn_wave_barrier();
        if (c + U < NIT) rows(c + U, pb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int sl = (c + u) * SPW + gs;
            const bool ok = b0 + sl < (long)B;
            const float* tgs = &tg[w][pb][u][gs * G];
            const float qvu = rq[w][pb][u][lane];
            const vfloat2 q2 = {qvu, qvu};
            QrSums sm = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // qr_pair: as qrdqn_fwd_group_kernel
            if constexpr (FULL) {                             // tau == G: no tail, fully unrolled
#pragma unroll
                for (int j = 0; j < G; j += 4) {
                    const vfloat4 t = *reinterpret_cast<const vfloat4*>(tgs + j);
                    qr_pair(sm, vfloat2{t.x, t.y}, q2);
                    qr_pair(sm, vfloat2{t.z, t.w}, q2);
                }
            } else {
                for (int j = 0; j < tau; j += 2) {
                    vfloat2 t = *reinterpret_cast<const vfloat2*>(tgs + j);
                    if (j + 1 >= tau) t.y = qvu;
                    qr_pair(sm, t, q2);
                }
            }
            float li, gi;
            qr_finish(sm, qneg, qpos, li, gi);
            const float wu = __shfl(w_l, sl, 64);             // outside the branch: a masked-off owner lane would read as 0
            // tau == G: stored WITHOUT a branch (the compiler sinks the dp / dn sums into a conditional store's block and keeps
            // all 32 of them alive until then: 64 registers, spills).  A group past the end of the batch holds sample B - 1
            // again (phase A clamps) and stores that sample's values a second time.
            const long bst = b0 + sl < (long)B ? b0 + sl : (long)B - 1;
            if (FULL || (ok && gl < tau)) buf[(size_t)bst * tau + gl] = -gi * inv_tau * wu * scale;
            const float loss = group_sum_last<G>(gl < tau ? li : 0.f);  // DPP: the group's LAST lane holds the sum
            const float theirs = __shfl(loss, (lane % SPW) * G + G - 1, 64);   // the group that holds my sample in this iteration
            if ((lane % SW) / SPW == c + u) mine = theirs * inv_tau;
            __builtin_amdgcn_sched_barrier(0);                // one sample's 32 target registers at a time (else: spills)
        }
    }
    const bool own = lane < SW && bown < (long)B;
    if (own) td_err[bown] = mine;
    const float contrib = wave_sum(own ? mine * w_l : 0.f);
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x == 0) tot = (red[0] + red[1]) + (red[2] + red[3]);
    publish_sums<1, 256>(tot, partials, fold);
}

// Round 5, large batches, tau a multiple of 4 (<= 64): FOUR quantiles per lane, a sample on G = 8 (tau <= 32) or 16 lanes,
// 64 / G samples per iteration of a wave.  qrdqn_fwd_batch_kernel above reads all tau targets into every one of a sample's
// tau lanes -- 4 KB of LDS reads per sample at tau = 32, and those reads, not the arithmetic, set its time: with a quarter of
// them the kernel ran 44 us instead of 58 at B = 262144 (tests/tools/r05_qrdqn_ablate.py; dropping the pair arithmetic
// from 11 to 9 instructions per pair changed nothing).  Here a target read serves four quantiles (1 KB per sample), the
// packed lanes are two QUANTILES against one broadcast target (op_sel picks the half of the register pair the target sits
// in), a quantile's terms add up in target order, rows move as 16 bytes per lane and the row / target / reduction overhead
// of an iteration is spread over four times the samples.  Per-term arithmetic: qr_pair.
template <bool HI>
__device__ __forceinline__ void qr_quad(QrSums& s, vfloat2 tp, vfloat2 q2) {
    const vfloat2 mh2 = {-0.5f, -0.5f};
    const float t = HI ? tp.y : tp.x;
    const vfloat2 e = vfloat2{t, t} - q2;
    vfloat2 dp, dn;
    if (HI) {
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(dp) : "v"(tp), "v"(q2));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(dn) : "v"(q2), "v"(tp));
    } else {
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(dp) : "v"(tp), "v"(q2));
        asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(dn) : "v"(q2), "v"(tp));
    }
    s.up = __builtin_elementwise_fma(dp, __builtin_elementwise_fma(mh2, dp, e), s.up);
    s.un = __builtin_elementwise_fma(dn, __builtin_elementwise_fma(mh2, dn, -e), s.un);
    s.dp += dp;
    s.dn += dn;
}

template <int G, int SW, bool FULL>
__global__ __launch_bounds__(256, 4) void qrdqn_fwd_quad_kernel(
    const float* __restrict__ q, const float* __restrict__ next_q, const int64_t* __restrict__ action,
    const int64_t* __restrict__ next_action, const float* __restrict__ reward, const float* __restrict__ done,
    const float* __restrict__ weight, const float* __restrict__ value_gamma, float* __restrict__ td_err,
    float* __restrict__ buf, float* __restrict__ partials, int tau, int nstep, int B, int N, float gamma,
    float gamma_n, float tau_value, float scale, const ScanFold fold) {
    constexpr int SPW = 64 / G, NIT = SW / SPW;
    static_assert(SW % SPW == 0 && SW <= 64 && G >= 8, "samples per wave");
    __shared__ float red[4];
    // per wave, two buffers: the q rows (lane l's four quantiles in words 4l ..) and the next-q rows, which become the targets
    // in place (a sample's targets contiguous: G * 4 words)
    __shared__ __attribute__((aligned(16))) float rq[4][2][256], tg[4][2][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, gl = lane % G, gs = lane / G;
    const long b0 = ((long)blockIdx.x * 4 + w) * SW;
    // phase A as qrdqn_fwd_batch_kernel: lane l holds the scalars of sample b0 + l (lanes >= SW repeat the first SW samples)
    const long bown = b0 + lane % SW;
    const long bl = bown < (long)B ? bown : (long)B - 1;
    const long row_l = ((long)bl * N + action[bl]) * tau, rown_l = ((long)bl * N + next_action[bl]) * tau;   // element offsets
    const float dn_l = done[bl], vgm_l = value_gamma ? value_gamma[bl] : gamma_n;
    const float w_l = weight ? weight[bl] : 1.f;
    const float R_l = nstep_return1(reward, B, nstep, gamma, bl);
    const float vg_l = vgm_l * (1.f - dn_l);
    const int nq4 = tau >> 2;                                 // lanes of a group that hold quantiles
    const int glc = gl < nq4 ? gl : nq4 - 1;                  // clamped: the row loads are unconditional
    const float inv_tau = 1.f / (float)tau;
    const float qneg = fabsf(tau_value - 1.f), qpos = fabsf(tau_value);
    float mine = 0.f;                                         // td_err of the sample this lane owns
    // The rows come by LDS-DMA (global_load_lds_dwordx4: lane l's 16 bytes land at word 4l of the wave's slot, no registers),
    // those of the NEXT iteration requested before the pair loops of the current one.  (The compiler puts s_waitcnt vmcnt(0) in
    // front of every LDS read that may alias a DMA destination, so those rows are in fact awaited before the pair loops start;
    // with the LDS reads written in assembly and the buf store deferred past the wait -- vmcnt counts stores on this part -- the
    // overlap is real and the kernel takes the same 51 us: five to six waves per SIMD hide the round trip either way.)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gl_ptr;
    auto rows = [&](int c, int pb) {
        const int sl = c * SPW + gs;                          // sample of this group, as a lane index of phase A
        __builtin_amdgcn_global_load_lds((gl_ptr)(q + __shfl(row_l, sl, 64) + 4 * glc), (lds_ptr)&rq[w][pb][0], 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gl_ptr)(next_q + __shfl(rown_l, sl, 64) + 4 * glc), (lds_ptr)&tg[w][pb][0], 16, 0, 0);
    };
    rows(0, 0);
#pragma unroll 1
    for (int c = 0, pb = 0; c < NIT; ++c, pb ^= 1) {
        const int sl = c * SPW + gs;
        const float vg = __shfl(vg_l, sl, 64), R = __shfl(R_l, sl, 64), wu = __shfl(w_l, sl, 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the rows of this iteration are in LDS
        vfloat4* const t4 = reinterpret_cast<vfloat4*>(&tg[w][pb][4 * lane]);
        vfloat4 nv = *t4;
        nv.x = fmaf(vg, nv.x, R);
        nv.y = fmaf(vg, nv.y, R);
        nv.z = fmaf(vg, nv.z, R);
        nv.w = fmaf(vg, nv.w, R);
        *t4 = nv;                                             // one wave writes and reads its own slots: in order, no barrier
        __builtin_amdgcn_wave_barrier();
        if (c + 1 < NIT) rows(c + 1, pb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        const vfloat4 q4 = *reinterpret_cast<const vfloat4*>(&rq[w][pb][4 * lane]);
        const vfloat2 q01 = {q4.x, q4.y}, q23 = {q4.z, q4.w};
        QrSums s01 = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, s23 = s01;
        const float* tgs = &tg[w][pb][gs * G * 4];
        auto four = [&](int j) {
            const vfloat4 t = *reinterpret_cast<const vfloat4*>(tgs + j);
            const vfloat2 ta = {t.x, t.y}, tb = {t.z, t.w};
            qr_quad<false>(s01, ta, q01);
            qr_quad<false>(s23, ta, q23);
            qr_quad<true>(s01, ta, q01);
            qr_quad<true>(s23, ta, q23);
            qr_quad<false>(s01, tb, q01);
            qr_quad<false>(s23, tb, q23);
            qr_quad<true>(s01, tb, q01);
            qr_quad<true>(s23, tb, q23);
        };
        if constexpr (FULL) {                                 // tau == 4 G: fully unrolled
#pragma unroll
            for (int j = 0; j < 4 * G; j += 4) four(j);
        } else {
            for (int j = 0; j < tau; j += 4) four(j);
        }
        // loss_i = |tau| sum(up) + |tau - 1| sum(un), dloss_i likewise (qr_finish, per quantile)
        vfloat4 li, gi;
        li.x = fmaf(qneg, s01.un.x, qpos * s01.up.x);
        li.y = fmaf(qneg, s01.un.y, qpos * s01.up.y);
        li.z = fmaf(qneg, s23.un.x, qpos * s23.up.x);
        li.w = fmaf(qneg, s23.un.y, qpos * s23.up.y);
        gi.x = fmaf(-qneg, s01.dn.x, qpos * s01.dp.x);
        gi.y = fmaf(-qneg, s01.dn.y, qpos * s01.dp.y);
        gi.z = fmaf(-qneg, s23.dn.x, qpos * s23.dp.x);
        gi.w = fmaf(-qneg, s23.dn.y, qpos * s23.dp.y);
        // tau == 4 G: stored without a branch (a conditional store's block attracts the sums and keeps their terms alive);
        // a group past the end of the batch holds sample B - 1 again (phase A clamps) and stores its values a second time
        const long bst = b0 + sl < (long)B ? b0 + sl : (long)B - 1;
        const float gsc = -inv_tau * wu * scale;
        if (FULL || gl < nq4)
            *reinterpret_cast<vfloat4*>(buf + (size_t)bst * tau + 4 * gl) = vfloat4{gi.x * gsc, gi.y * gsc, gi.z * gsc, gi.w * gsc};
        const float lsum = (li.x + li.y) + (li.z + li.w);
        const float loss = group_sum_last<G>(FULL || gl < nq4 ? lsum : 0.f);   // DPP: the group's LAST lane holds the sum
        const float theirs = __shfl(loss, (lane % SPW) * G + G - 1, 64);       // the group that holds my sample in this iteration
        if ((lane % SW) / SPW == c) mine = theirs * inv_tau;
    }
    const bool own = lane < SW && bown < (long)B;
    if (own) td_err[bown] = mine;
    const float contrib = wave_sum(own ? mine * w_l : 0.f);
    if (lane == 0) red[w] = contrib;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x == 0) tot = (red[0] + red[1]) + (red[2] + red[3]);
    publish_sums<1, 256>(tot, partials, fold);
}

inline int group_lanes(int n) { return n <= 8 ? 8 : n <= 16 ? 16 : n <= 32 ? 32 : 64; }

}  // namespace

int g_sample_batch = 0;   // hpc_rll_tune_set key 24: samples per wave of the large-batch QR-DQN forward (0 = by batch size, 1 = off)
}  // namespace hpc_rll

using namespace hpc_rll;

extern "C" int hpc_rll_dist_nstep_td_forward(const float* dist, const float* next_n_dist, const int64_t* action,
                                             const int64_t* next_n_action, const float* reward, const float* done,
                                             const float* weight, float* loss, float* td_err, float* buf,
                                             float* partials, int nstep, int B, int N, int n_atom, float gamma,
                                             float v_min, float v_max, float scale, void* stream) {
    if (nstep < 0 || B < 0 || N <= 0 || n_atom < 2 || !loss) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!dist || !next_n_dist || !action || !next_n_action || (nstep && !reward) || !done || !td_err || !buf ||
        !partials)
        return HPC_RLL_EINVAL;
    // large batches: SW samples per wave (dist_nstep_fwd_batch_kernel); widths from tests/tools/r03_batch_probe.py
    int sw = g_sample_batch;
    if (sw == 0) sw = B >= 262144 ? 16 : B >= 16384 ? 8 : 1;
    if (n_atom > 64) sw = 1;
    const int blocks = sw > 1 ? (int)(((long)B + 4 * sw - 1) / (4 * sw)) : (B + 3) / 4;
    // delta_z is a python double in the oracle, rounded to fp32 when it meets the fp32 tensor
    const float dz = (float)(((double)v_max - (double)v_min) / (double)(n_atom - 1));
    const float inv_dz = (float)(1.0 / (double)dz);            // the correctly rounded reciprocal (c51_project_scan)
    const ScanFold fold = make_fold(st, 1, &scale, loss, blocks);
    const float gamma_n = (float)pow((double)gamma, (double)nstep);
    if (sw > 1) {
#define HPC_RLL_C51_B(SW_)                                                                                            \
        if (sw == SW_)                                                                                                \
            hipLaunchKernelGGL(dist_nstep_fwd_batch_kernel<SW_>, dim3(blocks), dim3(256), 0, st, dist, next_n_dist,    \
                               action, next_n_action, reward, done, weight, td_err, buf, partials, nstep, B, N, n_atom, \
                               gamma, gamma_n, v_min, v_max, dz, inv_dz, scale, fold);
        HPC_RLL_C51_B(8) HPC_RLL_C51_B(16) HPC_RLL_C51_B(32) HPC_RLL_C51_B(64)
#undef HPC_RLL_C51_B
    } else if (n_atom <= 64)
        hipLaunchKernelGGL(dist_nstep_fwd64_kernel<1>, dim3(blocks), dim3(256), 0, st, dist, next_n_dist, action,
                           next_n_action, reward, done, weight, td_err, buf, partials, nstep, B, N, n_atom, gamma, gamma_n,
                           v_min, v_max, dz, scale, fold);
    else
        hipLaunchKernelGGL(dist_nstep_fwd_kernel, dim3(blocks), dim3(256), 0, st, dist, next_n_dist, action,
                           next_n_action, reward, done, weight, td_err, buf, partials, nstep, B, N, n_atom, gamma, gamma_n,
                           v_min, v_max, dz, scale, fold);
    const int rc = last_error();
    if (rc || fold.out) return rc;
    return finalize_sums(partials, blocks, 1, &scale, loss, st);
}

extern "C" int hpc_rll_dist_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action,
                                              float* grad_dist, int B, int N, int n_atom, void* stream) {
    if (B < 0 || N <= 0 || n_atom <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!grad_loss || !buf || !action || !grad_dist) return HPC_RLL_EINVAL;
    return onehot_scatter(grad_loss, buf, action, grad_dist, B, N, n_atom, (hipStream_t)stream);
}

namespace hpc_rll { namespace {
int iqn_forward_impl(const float* q, const float* next_n_q, const int64_t* action, const int64_t* next_n_action,
                     const float* reward, const float* done, const float* replay_quantiles, const float* weight,
                     const float* value_gamma, float* loss, float* td_err, float* buf, float* partials, int tau,
                     int tau_prime, int nstep, int B, int N, float gamma, float kappa, float scale, bool bnt, void* stream) {
    if (tau <= 0 || tau_prime <= 0 || nstep < 0 || B < 0 || N <= 0 || !loss || !(kappa > 0.f)) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!q || !next_n_q || !action || !next_n_action || (nstep && !reward) || !done || !replay_quantiles || !td_err ||
        !buf || !partials)
        return HPC_RLL_EINVAL;
    int blocks = (B + 3) / 4;
    const float gamma_n = (float)pow((double)gamma, (double)nstep);
    const int gmax = tau > tau_prime ? tau : tau_prime;
    if (gmax <= 64) blocks = (B + 4 * (64 / group_lanes(gmax)) - 1) / (4 * (64 / group_lanes(gmax)));
    const ScanFold fold = make_fold(st, 1, &scale, loss, blocks);
    if (gmax <= 64) {
        const int G = group_lanes(gmax);
#define HPC_RLL_IQN_G(G_)                                                                                               \
        if (G == G_) {                                                                                                  \
            if (bnt)                                                                                                    \
                hipLaunchKernelGGL((iqn_fwd_group_kernel<G_, true>), dim3(blocks), dim3(256), 0, st, q, next_n_q, action, \
                                   next_n_action, reward, done, replay_quantiles, weight, value_gamma, td_err, buf,     \
                                   partials, tau, tau_prime, nstep, B, N, gamma, gamma_n, kappa, scale, fold);          \
            else                                                                                                        \
                hipLaunchKernelGGL((iqn_fwd_group_kernel<G_, false>), dim3(blocks), dim3(256), 0, st, q, next_n_q, action, \
                                   next_n_action, reward, done, replay_quantiles, weight, value_gamma, td_err, buf,     \
                                   partials, tau, tau_prime, nstep, B, N, gamma, gamma_n, kappa, scale, fold);          \
        }
        HPC_RLL_IQN_G(8) HPC_RLL_IQN_G(16) HPC_RLL_IQN_G(32) HPC_RLL_IQN_G(64)
#undef HPC_RLL_IQN_G
    } else {
        hipLaunchKernelGGL(iqn_fwd_kernel, dim3(blocks), dim3(256), 0, st, q, next_n_q, action, next_n_action, reward,
                           done, replay_quantiles, weight, value_gamma, td_err, buf, partials, tau, tau_prime, nstep, B,
                           N, gamma, gamma_n, kappa, scale, fold, bnt ? 1 : 0);
    }
    const int rc = last_error();
    if (rc || fold.out) return rc;
    return finalize_sums(partials, blocks, 1, &scale, loss, st);
}
} }  // namespace hpc_rll::(anonymous)

extern "C" int hpc_rll_iqn_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                                            const int64_t* next_n_action, const float* reward, const float* done,
                                            const float* replay_quantiles, const float* weight,
                                            const float* value_gamma, float* loss, float* td_err, float* buf,
                                            float* partials, int tau, int tau_prime, int nstep, int B, int N,
                                            float gamma, float kappa, float scale, void* stream) {
    return iqn_forward_impl(q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma, loss,
                            td_err, buf, partials, tau, tau_prime, nstep, B, N, gamma, kappa, scale, false, stream);
}
// ABI 6: the same loss on q (B,N,tau), next_n_q (B,N,tau') -- the layout of the QR-DQN op; replay_quantiles stays (tau,B)
extern "C" int hpc_rll_iqn_nstep_td_forward_bnt(const float* q, const float* next_n_q, const int64_t* action,
                                                const int64_t* next_n_action, const float* reward, const float* done,
                                                const float* replay_quantiles, const float* weight,
                                                const float* value_gamma, float* loss, float* td_err, float* buf,
                                                float* partials, int tau, int tau_prime, int nstep, int B, int N,
                                                float gamma, float kappa, float scale, void* stream) {
    return iqn_forward_impl(q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma, loss,
                            td_err, buf, partials, tau, tau_prime, nstep, B, N, gamma, kappa, scale, true, stream);
}
// grad_q (B,N,tau): one-hot rows of tau values at [a_b * tau, a_b * tau + tau) -- the QR-DQN backward's shape
extern "C" int hpc_rll_iqn_nstep_td_backward_bnt(const float* grad_loss, const float* buf, const int64_t* action,
                                                 float* grad_q, int tau, int B, int N, void* stream) {
    if (tau <= 0 || B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!grad_loss || !buf || !action || !grad_q) return HPC_RLL_EINVAL;
    return onehot_scatter(grad_loss, buf, action, grad_q, B, N, tau, (hipStream_t)stream);
}

extern "C" int hpc_rll_iqn_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action,
                                             float* grad_q, int tau, int B, int N, void* stream) {
    if (tau <= 0 || B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!grad_loss || !buf || !action || !grad_q) return HPC_RLL_EINVAL;
    // (tau, B, N) = tau planes of one-hot rows: the 16-byte kernel of sample_ops.hip when the rows allow it
    const int rc = onehot_scatter(grad_loss, buf, action, grad_q, B, N, 1, (hipStream_t)stream, tau);
    if (rc != HPC_RLL_EUNSUPPORTED) return rc;
    long blocks = ((long)tau * B * N + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(iqn_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad_loss, buf,
                       action, grad_q, tau, B, N);
    return last_error();
}

extern "C" int hpc_rll_qrdqn_nstep_td_forward(const float* q, const float* next_n_q, const int64_t* action,
                                              const int64_t* next_n_action, const float* reward, const float* done,
                                              const float* weight, const float* value_gamma, float* loss,
                                              float* td_err, float* buf, float* partials, int tau, int nstep, int B,
                                              int N, float gamma, float tau_value, float scale, void* stream) {
    if (tau <= 0 || nstep < 0 || B < 0 || N <= 0 || !loss) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!q || !next_n_q || !action || !next_n_action || (nstep && !reward) || !done || !td_err || !buf || !partials)
        return HPC_RLL_EINVAL;
    int blocks = (B + 3) / 4;
    const float gamma_n = (float)pow((double)gamma, (double)nstep);
    if (tau <= 64) blocks = (B + 4 * (64 / group_lanes(tau)) - 1) / (4 * (64 / group_lanes(tau)));
    // large batches: SW consecutive samples per wave (qrdqn_fwd_batch_kernel); widths from tests/tools/r03_batch_probe.py
    int sw = g_sample_batch;
    if (sw == 0) sw = B >= 262144 ? 32 : B >= 32768 ? 8 : 1;
    if (tau > 64 || sw < 64 / group_lanes(tau)) sw = 1;
    if (sw > 1) blocks = (int)(((long)B + 4 * sw - 1) / (4 * sw));
    // tau a multiple of 4, rows 16-byte aligned: four quantiles per lane (qrdqn_fwd_quad_kernel)
    const bool quad = sw >= 8 && tau % 4 == 0 && tau >= 8 && ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(next_n_q) |
                                                                reinterpret_cast<uintptr_t>(buf)) & 15) == 0;
    if (quad) {
        const int G = tau <= 32 ? 8 : 16;
        if (sw < 64 / G) sw = 64 / G;
        blocks = (int)(((long)B + 4 * sw - 1) / (4 * sw));
    }
    const ScanFold fold = make_fold(st, 1, &scale, loss, blocks);
    if (quad) {
        const int G = tau <= 32 ? 8 : 16;
#define HPC_RLL_QR_Q(G_, SW_)                                                                                         \
        if (G == G_ && sw == SW_) {                                                                                   \
            if (tau == 4 * G_)                                                                                        \
                hipLaunchKernelGGL((qrdqn_fwd_quad_kernel<G_, SW_, true>), dim3(blocks), dim3(256), 0, st, q, next_n_q, \
                                   action, next_n_action, reward, done, weight, value_gamma, td_err, buf, partials, tau, \
                                   nstep, B, N, gamma, gamma_n, tau_value, scale, fold);                              \
            else                                                                                                      \
                hipLaunchKernelGGL((qrdqn_fwd_quad_kernel<G_, SW_, false>), dim3(blocks), dim3(256), 0, st, q, next_n_q, \
                                   action, next_n_action, reward, done, weight, value_gamma, td_err, buf, partials, tau, \
                                   nstep, B, N, gamma, gamma_n, tau_value, scale, fold);                              \
        }
        HPC_RLL_QR_Q(8, 8) HPC_RLL_QR_Q(8, 16) HPC_RLL_QR_Q(8, 32) HPC_RLL_QR_Q(8, 64)
        HPC_RLL_QR_Q(16, 8) HPC_RLL_QR_Q(16, 16) HPC_RLL_QR_Q(16, 32) HPC_RLL_QR_Q(16, 64)
#undef HPC_RLL_QR_Q
    } else if (sw > 1) {
        const int G = group_lanes(tau);
#define HPC_RLL_QR_B(G_, SW_)                                                                                         \
        if (G == G_ && sw == SW_) {                                                                                   \
            if (tau == G_)                                                                                            \
                hipLaunchKernelGGL((qrdqn_fwd_batch_kernel<G_, SW_, true>), dim3(blocks), dim3(256), 0, st, q, next_n_q, \
                                   action, next_n_action, reward, done, weight, value_gamma, td_err, buf, partials, tau, \
                                   nstep, B, N, gamma, gamma_n, tau_value, scale, fold);                              \
            else                                                                                                      \
                hipLaunchKernelGGL((qrdqn_fwd_batch_kernel<G_, SW_, false>), dim3(blocks), dim3(256), 0, st, q, next_n_q, \
                                   action, next_n_action, reward, done, weight, value_gamma, td_err, buf, partials, tau, \
                                   nstep, B, N, gamma, gamma_n, tau_value, scale, fold);                              \
        }
        HPC_RLL_QR_B(8, 8) HPC_RLL_QR_B(8, 16) HPC_RLL_QR_B(8, 32) HPC_RLL_QR_B(8, 64) HPC_RLL_QR_B(16, 8) HPC_RLL_QR_B(16, 16)
        HPC_RLL_QR_B(16, 32) HPC_RLL_QR_B(16, 64) HPC_RLL_QR_B(32, 8) HPC_RLL_QR_B(32, 16) HPC_RLL_QR_B(32, 32) HPC_RLL_QR_B(32, 64)
        HPC_RLL_QR_B(64, 8) HPC_RLL_QR_B(64, 16) HPC_RLL_QR_B(64, 32) HPC_RLL_QR_B(64, 64)
#undef HPC_RLL_QR_B
    } else if (tau <= 64) {
        const int G = group_lanes(tau);
#define HPC_RLL_QR_G(G_)                                                                                              \
        if (G == G_)                                                                                                  \
            hipLaunchKernelGGL(qrdqn_fwd_group_kernel<G_>, dim3(blocks), dim3(256), 0, st, q, next_n_q, action,        \
                               next_n_action, reward, done, weight, value_gamma, td_err, buf, partials, tau, nstep, B, N, \
                               gamma, gamma_n, tau_value, scale, fold);
        HPC_RLL_QR_G(8) HPC_RLL_QR_G(16) HPC_RLL_QR_G(32) HPC_RLL_QR_G(64)
#undef HPC_RLL_QR_G
    } else {
        hipLaunchKernelGGL(qrdqn_fwd_kernel, dim3(blocks), dim3(256), 0, st, q, next_n_q, action, next_n_action, reward,
                           done, weight, value_gamma, td_err, buf, partials, tau, nstep, B, N, gamma, gamma_n, tau_value,
                           scale, fold);
    }
    const int rc = last_error();
    if (rc || fold.out) return rc;
    return finalize_sums(partials, blocks, 1, &scale, loss, st);
}

extern "C" int hpc_rll_qrdqn_nstep_td_backward(const float* grad_loss, const float* buf, const int64_t* action,
                                               float* grad_q, int tau, int B, int N, void* stream) {
    if (tau <= 0 || B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (!grad_loss || !buf || !action || !grad_q) return HPC_RLL_EINVAL;
    return onehot_scatter(grad_loss, buf, action, grad_q, B, N, tau, (hipStream_t)stream);
}
