// gae.hip -- GAE forward + analytic backward for gfx950 (MI355X).
//
// Replaces the reference's GaeForward (src/rl_utils/gae.cu:8-28, gae_kernel.h:10-29: one thread
// per column walking T serially, 32-thread blocks, `value` fetched twice) and adds the backward the
// reference lacks (hpc_rll/rl_utils/gae.py:17-18 returns None).
//
// Maths.  With D_t = 1 + lambda*D_{t+1}, D_T = 0 and c_t = gamma*lambda*D_{t+1}/D_t:
//     forward : adv_t = delta_t + c_t * adv_{t+1},   delta_t = r_t + gamma*V_{t+1} - V_t
//     backward: d_t   = g_t + c_{t-1} * d_{t-1};  dL/dr_t = d_t;  dL/dV_t = -d_t + gamma*d_{t-1}
// Both are first-order affine recurrences with a LANE-UNIFORM coefficient, so a chunk [t0,t1) can be
// scanned locally from a zero carry and repaired afterwards:  adv_t = L_t + P_t * adv_{t1},
// P_t = prod_{s=t}^{t1-1} c_s.  That is what buys time-parallelism without a second HBM sweep.
//
// Mapping (HBM-bound, 12 B/sample each way; no data reuse -> no LDS staging of the payload):
//   * lane <-> V consecutive columns (V*4-byte coalesced loads along B, 64*V columns per wave row);
//   * wave <-> LC consecutive time steps, held entirely in VGPRs (all 2*LC+1 row loads of a chunk are
//     independent and issued back to back -> deep memory-level parallelism per wave);
//   * workgroup = NW waves covering NW*LC consecutive steps of one column tile; the NW chunk carries
//     (one L and one lane-uniform P per wave) are exchanged through a double-buffered LDS slot with
//     one barrier per NW*LC steps; the workgroup then walks the rest of T carrying V floats per lane.
//   * grid = ceil(B / (64*V)) workgroups, no inter-workgroup communication.
// HBM traffic = algorithmic + the one `value` row per chunk boundary that two waves both read
// (an L2 hit: same workgroup, same instant).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <atomic>
#include <initializer_list>
#include <new>
#include <type_traits>

#include "hpc_rll_hip.h"
#include "wave.hpp"

namespace hpc_rll {
namespace {

// ------------------------------------------------------------------------------------------------
// coefficient table: c_t = gamma*lambda*S(k-1)/S(k), k = T-t, S(k) = sum_{j<k} lambda^j (= D_t).
// Closed form in fp64 (expm1/log keep full relative precision as lambda -> 1).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double geom_sum(double lam, int k) {
    if (k <= 0) return 0.0;
    if (lam == 1.0) return (double)k;
    if (lam > 0.0) return -expm1((double)k * log(lam)) / (1.0 - lam);
    return (1.0 - pow(lam, (double)k)) / (1.0 - lam);
}

__global__ void gae_coef_kernel(float* __restrict__ coef, int T, float gamma, float lambda) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int k = T - t;
    const double lam = (double)lambda;
    coef[t] = (float)((double)gamma * lam * geom_sum(lam, k - 1) / geom_sum(lam, k));
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// HALF (V = 1 only): a wave covers 32 columns instead of 64 and its two 32-lane halves own two DIFFERENT chunks of
// the time axis ("virtual waves" 2w and 2w+1).  For narrow batches (B < 16384: fewer than 256 column tiles of 64) this
// doubles both the number of workgroups and the steps covered per barrier (NW*2*LC): the kernel is bound by the
// latency of its T / SPAN iterations there, not by bandwidth.
template <int V, int LC, int NW, bool NTL, bool NTS, bool HALF = false>
__global__ __launch_bounds__(NW * 64) void gae_fwd_kernel(const float* __restrict__ value,
                                                          const float* __restrict__ reward,
                                                          float* __restrict__ adv,
                                                          const float* __restrict__ coef, int T, int B,
                                                          float gamma) {
    static_assert(!HALF || V == 1, "half-wave tiles hold one column per lane");
    constexpr int NWV = HALF ? 2 * NW : NW;       // (virtual) waves per workgroup
    constexpr int TILE = HALF ? 32 : 64 * V;
    // one LDS object: [buf][wave][TILE] chunk-head values, then [buf][wave] chunk products
    __shared__ float lds[2 * NWV * TILE + 2 * NWV];
    float* const s_l0 = lds;
    float* const s_p0 = lds + 2 * NWV * TILE;

    const int lane = threadIdx.x & 63;
    const int wr = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cl = HALF ? (lane & 31) : lane;             // column lane inside the tile
    const int w = HALF ? 2 * wr + (lane >> 5) : wr;       // (virtual) wave: lane dependent when HALF
    const long col = (long)blockIdx.x * TILE + (long)cl * V;
    const bool col_ok = col < (long)B;  // dispatcher guarantees B % V == 0: packs never straddle B

    float carry[V];
#pragma unroll
    for (int k = 0; k < V; ++k) carry[k] = 0.f;

    constexpr int SPAN = NWV * LC;
    const int n_iter = (T + SPAN - 1) / SPAN;

    for (int it = 0; it < n_iter; ++it) {
        // chunks are aligned to the END of the trajectory; wave NWV-1 owns the latest chunk
        const int t1 = T - (it * NWV + (NWV - 1 - w)) * LC;  // exclusive end, <= T, may be <= 0
        const int t0 = t1 - LC;
        const int buf = it & 1;

        float L[LC][V];
        float P[LC];

        if (t0 >= 0) {
            // ---- fast path: the whole chunk is inside [0,T) ----
            Pack<V> vr[LC + 1], rr[LC];
            if (col_ok) {
                const float* vp = value + (size_t)t0 * B + col;
                const float* rp = reward + (size_t)t0 * B + col;
#pragma unroll
                for (int j = LC; j >= 0; --j) vr[j] = load_pack<V, NTL>(vp + (size_t)j * B);
#pragma unroll
                for (int j = LC - 1; j >= 0; --j) rr[j] = load_pack<V, NTL>(rp + (size_t)j * B);
            } else {
#pragma unroll
                for (int j = 0; j <= LC; ++j)
#pragma unroll
                    for (int k = 0; k < V; ++k) vr[j].v[k] = 0.f;
#pragma unroll
                for (int j = 0; j < LC; ++j)
#pragma unroll
                    for (int k = 0; k < V; ++k) rr[j].v[k] = 0.f;
            }
            float a[V];
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
            float p = 1.f;
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                const float c = coef[t0 + j];  // wave-uniform address -> scalar load
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float delta = fmaf(gamma, vr[j + 1].v[k], rr[j].v[k]) - vr[j].v[k];
                    a[k] = fmaf(c, a[k], delta);
                    L[j][k] = a[k];
                }
                p *= c;
                P[j] = p;
            }
        } else {
            // ---- ragged head of the trajectory (only in the last iteration) ----
            float a[V];
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
            float p = 1.f;
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                const int t = t0 + j;
                if (t >= 0) {
                    const float c = coef[t];
                    Pack<V> v0, v1, r;
                    if (col_ok) {
                        v0 = load_pack<V, NTL>(value + (size_t)t * B + col);
                        v1 = load_pack<V, NTL>(value + (size_t)(t + 1) * B + col);
                        r = load_pack<V, NTL>(reward + (size_t)t * B + col);
                    } else {
#pragma unroll
                        for (int k = 0; k < V; ++k) v0.v[k] = v1.v[k] = r.v[k] = 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        const float delta = fmaf(gamma, v1.v[k], r.v[k]) - v0.v[k];
                        a[k] = fmaf(c, a[k], delta);
                    }
                    p *= c;
                }
#pragma unroll
                for (int k = 0; k < V; ++k) L[j][k] = a[k];
                P[j] = p;
            }
        }

        // ---- publish this chunk's head (value at t0 with zero carry-in, and the product over the chunk)
#pragma unroll
        for (int k = 0; k < V; ++k) s_l0[(buf * NWV + w) * TILE + cl * V + k] = L[0][k];
        if (cl == 0) s_p0[buf * NWV + w] = P[0];
        __syncthreads();

        // ---- resolve carries: walk the NWV chunks from the latest to the earliest
        float A[V], Aw[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { A[k] = carry[k]; Aw[k] = 0.f; }
#pragma unroll
        for (int u = NWV - 1; u >= 0; --u) {
            if (u == w) {
#pragma unroll
                for (int k = 0; k < V; ++k) Aw[k] = A[k];
            }
            const float p0 = s_p0[buf * NWV + u];
#pragma unroll
            for (int k = 0; k < V; ++k) A[k] = fmaf(p0, A[k], s_l0[(buf * NWV + u) * TILE + cl * V + k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = A[k];

        // ---- repair and store
        if (col_ok) {
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                const int t = t0 + j;
                if (t >= 0) {
                    Pack<V> o;
#pragma unroll
                    for (int k = 0; k < V; ++k) o.v[k] = fmaf(P[j], Aw[k], L[j][k]);
                    store_pack<V, NTS>(adv + (size_t)t * B + col, o);
                }
            }
        }
        // no second barrier: the next iteration writes the other LDS buffer, and nobody can be two
        // iterations ahead because every wave must pass the next iteration's barrier first.
    }
}

// ------------------------------------------------------------------------------------------------
// forward, software-pipelined (round 3).  Same chunk / carry scheme and the SAME arithmetic order as gae_fwd_kernel
// (bit-identical results), but the 2*LC+1 row loads of iteration it+1 are issued BEFORE iteration it publishes its
// chunk head, waits at the barrier, resolves the carries and stores: a wave always has a full chunk of loads in
// flight, instead of alternating between a load phase and a store phase.  Costs a second live register set
// ((2*LC+1)*V loaded + (LC+1)*V computed), which the 128-thread workgroups of the streaming regime have to spare.
// Only full 64-column-per-V tiles (no half-wave variant: that regime is latency-, not bandwidth-bound).
// ------------------------------------------------------------------------------------------------
// FULLB: B is a multiple of the tile width, no lane is ever out of range -- no exec-masked region around the stores
// (which would make the compiler's vmcnt bookkeeping pessimistic again, see GUARD below).
template <int V, int LC, int NW, bool NTL, bool NTS, bool FULLB>
__global__ __launch_bounds__(NW * 64) void gae_fwd_pf_kernel(const float* __restrict__ value,
                                                             const float* __restrict__ reward,
                                                             float* __restrict__ adv,
                                                             const float* __restrict__ coef, int T, int B,
                                                             float gamma) {
    constexpr int TILE = 64 * V;
    __shared__ float lds[2 * NW * TILE + 2 * NW];
    float* const s_l0 = lds;
    float* const s_p0 = lds + 2 * NW * TILE;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (round 4 tried XCD-contiguous column tiles -- XCD x owning a contiguous eighth of the columns instead of every 8th tile:
    // no gain at C2 (tests/tools/r04_gae_xcd_probe.py); removed in round 5)
    const long col = (long)blockIdx.x * TILE + (long)lane * V;
    const bool col_ok = FULLB || col < (long)B;

    float carry[V];
#pragma unroll
    for (int k = 0; k < V; ++k) carry[k] = 0.f;

    constexpr int SPAN = NW * LC;
    const int n_full = T / SPAN;            // iterations in which every wave's chunk lies inside [0,T)
    const bool ragged = (T % SPAN) != 0;    // one more iteration with per-step guards

    Pack<V> vr[LC + 1], rr[LC];
#pragma unroll
    for (int j = 0; j <= LC; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) vr[j].v[k] = 0.f;
#pragma unroll
    for (int j = 0; j < LC; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) rr[j].v[k] = 0.f;

    auto chunk_t0 = [&](int it) { return T - (it * NW + (NW - 1 - w)) * LC - LC; };
    auto issue = [&](int it) {
        if (col_ok) {
            const int t0 = chunk_t0(it);
            const float* vp = value + (size_t)t0 * B + col;
            const float* rp = reward + (size_t)t0 * B + col;
#pragma unroll
            for (int j = LC; j >= 0; --j) vr[j] = load_pack<V, NTL>(vp + (size_t)j * B);
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) rr[j] = load_pack<V, NTL>(rp + (size_t)j * B);
        }
    };
    // publish the chunk head, resolve the carries through LDS, repair and store (identical to gae_fwd_kernel)
    // GUARD: per-step `t >= 0` tests (ragged iteration only).  In the full iterations the stores MUST be unconditional:
    // vmcnt counts loads and stores in issue order, and behind a conditional store the compiler has to assume it was
    // not issued -- its wait for the prefetched rows then also waits for the stores that WERE issued after them.
    auto finish = [&](auto guard_, int it, int t0, float (&L)[LC][V], float (&P)[LC]) {
        constexpr bool GUARD = decltype(guard_)::value != 0;
        const int buf = it & 1;
#pragma unroll
        for (int k = 0; k < V; ++k) s_l0[(buf * NW + w) * TILE + lane * V + k] = L[0][k];
        if (lane == 0) s_p0[buf * NW + w] = P[0];
        __syncthreads();
        float A[V], Aw[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { A[k] = carry[k]; Aw[k] = 0.f; }
#pragma unroll
        for (int u = NW - 1; u >= 0; --u) {
            if (u == w) {
#pragma unroll
                for (int k = 0; k < V; ++k) Aw[k] = A[k];
            }
            const float p0 = s_p0[buf * NW + u];
#pragma unroll
            for (int k = 0; k < V; ++k) A[k] = fmaf(p0, A[k], s_l0[(buf * NW + u) * TILE + lane * V + k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = A[k];
        if (col_ok) {
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                const int t = t0 + j;
                if (!GUARD || t >= 0) {
                    Pack<V> o;
#pragma unroll
                    for (int k = 0; k < V; ++k) o.v[k] = fmaf(P[j], Aw[k], L[j][k]);
                    store_pack<V, NTS>(adv + (size_t)t * B + col, o);
                }
            }
        }
    };

    if (n_full > 0) issue(0);
    for (int it = 0; it < n_full; ++it) {
        const int t0 = chunk_t0(it);
        float L[LC][V];
        float P[LC];
        {
            float a[V];
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
            float p = 1.f;
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                const float c = coef[t0 + j];
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float delta = fmaf(gamma, vr[j + 1].v[k], rr[j].v[k]) - vr[j].v[k];
                    a[k] = fmaf(c, a[k], delta);
                    L[j][k] = a[k];
                }
                p *= c;
                P[j] = p;
            }
        }
        if (it + 1 < n_full) issue(it + 1);      // next chunk's loads fly over the barrier, the resolve and the stores
        finish(std::integral_constant<int, 0>{}, it, t0, L, P);
    }
    if (ragged) {
        const int it = n_full;
        const int t0 = chunk_t0(it);
        float L[LC][V];
        float P[LC];
        float a[V];
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = 0.f;
        float p = 1.f;
#pragma unroll
        for (int j = LC - 1; j >= 0; --j) {
            const int t = t0 + j;
            if (t >= 0) {
                const float c = coef[t];
                Pack<V> v0, v1, r;
                if (col_ok) {
                    v0 = load_pack<V, NTL>(value + (size_t)t * B + col);
                    v1 = load_pack<V, NTL>(value + (size_t)(t + 1) * B + col);
                    r = load_pack<V, NTL>(reward + (size_t)t * B + col);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) v0.v[k] = v1.v[k] = r.v[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float delta = fmaf(gamma, v1.v[k], r.v[k]) - v0.v[k];
                    a[k] = fmaf(c, a[k], delta);
                }
                p *= c;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) L[j][k] = a[k];
            P[j] = p;
        }
        finish(std::integral_constant<int, 1>{}, it, t0, L, P);
    }
}

// ------------------------------------------------------------------------------------------------
// backward: d_t = g_t + c_{t-1} d_{t-1} (forward in time), chunks aligned to t = 0, wave 0 earliest.
// ------------------------------------------------------------------------------------------------
template <int V, int LC, int NW, bool NTL, bool NTS, bool HALF = false>
__global__ __launch_bounds__(NW * 64) void gae_bwd_kernel(const float* __restrict__ grad_adv,
                                                          float* __restrict__ grad_value,
                                                          float* __restrict__ grad_reward,
                                                          const float* __restrict__ coef, int T, int B,
                                                          float gamma) {
    static_assert(!HALF || V == 1, "half-wave tiles hold one column per lane");
    constexpr int NWV = HALF ? 2 * NW : NW;
    constexpr int TILE = HALF ? 32 : 64 * V;
    __shared__ float lds[2 * NWV * TILE + 2 * NWV];
    float* const s_l0 = lds;
    float* const s_p0 = lds + 2 * NWV * TILE;

    const int lane = threadIdx.x & 63;
    const int wr = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cl = HALF ? (lane & 31) : lane;
    const int w = HALF ? 2 * wr + (lane >> 5) : wr;
    const long col = (long)blockIdx.x * TILE + (long)cl * V;
    const bool col_ok = col < (long)B;

    float carry[V];
#pragma unroll
    for (int k = 0; k < V; ++k) carry[k] = 0.f;

    constexpr int SPAN = NWV * LC;
    const int n_iter = (T + SPAN - 1) / SPAN;

    for (int it = 0; it < n_iter; ++it) {
        const int t0 = (it * NWV + w) * LC;
        const int buf = it & 1;

        float L[LC][V];
        float Q[LC];
        {
            float a[V];
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
            float q = 1.f;
            if (t0 + LC <= T) {
                Pack<V> g[LC];
                if (col_ok) {
                    const float* gp = grad_adv + (size_t)t0 * B + col;
#pragma unroll
                    for (int j = 0; j < LC; ++j) g[j] = load_pack<V, NTL>(gp + (size_t)j * B);
                } else {
#pragma unroll
                    for (int j = 0; j < LC; ++j)
#pragma unroll
                        for (int k = 0; k < V; ++k) g[j].v[k] = 0.f;
                }
#pragma unroll
                for (int j = 0; j < LC; ++j) {
                    const int t = t0 + j;
                    const float c = (t > 0) ? coef[t - 1] : 0.f;
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        a[k] = fmaf(c, a[k], g[j].v[k]);
                        L[j][k] = a[k];
                    }
                    q *= c;
                    Q[j] = q;
                }
            } else {
#pragma unroll
                for (int j = 0; j < LC; ++j) {
                    const int t = t0 + j;
                    if (t < T) {
                        const float c = (t > 0) ? coef[t - 1] : 0.f;
                        Pack<V> g;
                        if (col_ok) {
                            g = load_pack<V, NTL>(grad_adv + (size_t)t * B + col);
                        } else {
#pragma unroll
                            for (int k = 0; k < V; ++k) g.v[k] = 0.f;
                        }
#pragma unroll
                        for (int k = 0; k < V; ++k) a[k] = fmaf(c, a[k], g.v[k]);
                        q *= c;
                    }
#pragma unroll
                    for (int k = 0; k < V; ++k) L[j][k] = a[k];
                    Q[j] = q;
                }
            }
        }

#pragma unroll
        for (int k = 0; k < V; ++k) s_l0[(buf * NWV + w) * TILE + cl * V + k] = L[LC - 1][k];
        if (cl == 0) s_p0[buf * NWV + w] = Q[LC - 1];
        __syncthreads();

        float A[V], Aw[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { A[k] = carry[k]; Aw[k] = 0.f; }
#pragma unroll
        for (int u = 0; u < NWV; ++u) {
            if (u == w) {
#pragma unroll
                for (int k = 0; k < V; ++k) Aw[k] = A[k];
            }
            const float q0 = s_p0[buf * NWV + u];
#pragma unroll
            for (int k = 0; k < V; ++k) A[k] = fmaf(q0, A[k], s_l0[(buf * NWV + u) * TILE + cl * V + k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = A[k];

        if (col_ok) {
            float prev[V];  // d_{t-1}
#pragma unroll
            for (int k = 0; k < V; ++k) prev[k] = Aw[k];
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                const int t = t0 + j;
                if (t < T) {
                    Pack<V> d, gv;
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        d.v[k] = fmaf(Q[j], Aw[k], L[j][k]);
                        gv.v[k] = fmaf(gamma, prev[k], -d.v[k]);
                        prev[k] = d.v[k];
                    }
                    if (grad_reward) store_pack<V, NTS>(grad_reward + (size_t)t * B + col, d);
                    if (grad_value) {
                        store_pack<V, NTS>(grad_value + (size_t)t * B + col, gv);
                        if (t == T - 1) {  // bootstrap row: dL/dV_T = gamma * d_{T-1}
                            Pack<V> last;
#pragma unroll
                            for (int k = 0; k < V; ++k) last.v[k] = gamma * d.v[k];
                            store_pack<V, NTS>(grad_value + (size_t)T * B + col, last);
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// (The mapping BASELINE.json's north_star names -- ONE TRAJECTORY PER WAVEFRONT, 64 lanes along time, a 6-step wavefront-shuffle
// scan per 64-step tile staged through LDS -- was built and measured in round 3: 2-8x slower than the lane-per-column kernels
// on every shape (they need ONE fma per element where the shuffle scan needs 6 shuffles + 6 fmas, no LDS round trip of the
// payload, and cover 256-512 steps per barrier): profiles/r03_gae_wpt_probe.txt.  The kernel lives on as a stand-alone
// measurement, tests/tools/micro/gae_wpt.hip; it is not part of the library any more.)
// ------------------------------------------------------------------------------------------------
// backward, software-pipelined (round 3): the mirror image of gae_fwd_pf_kernel.  Both gradients are written
// (grad_value and grad_reward non-null; the dispatcher falls back to gae_bwd_kernel otherwise).  Same arithmetic
// order as gae_bwd_kernel: bit-identical results.
// ------------------------------------------------------------------------------------------------
template <int V, int LC, int NW, bool NTL, bool NTS, bool FULLB>
__global__ __launch_bounds__(NW * 64) void gae_bwd_pf_kernel(const float* __restrict__ grad_adv,
                                                             float* __restrict__ grad_value,
                                                             float* __restrict__ grad_reward,
                                                             const float* __restrict__ coef, int T, int B,
                                                             float gamma) {
    constexpr int TILE = 64 * V;
    __shared__ float lds[2 * NW * TILE + 2 * NW];
    float* const s_l0 = lds;
    float* const s_p0 = lds + 2 * NW * TILE;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // (round 4 tried XCD-contiguous column tiles -- XCD x owning a contiguous eighth of the columns instead of every 8th tile:
    // no gain at C2 (tests/tools/r04_gae_xcd_probe.py); removed in round 5)
    const long col = (long)blockIdx.x * TILE + (long)lane * V;
    const bool col_ok = FULLB || col < (long)B;

    float carry[V];
#pragma unroll
    for (int k = 0; k < V; ++k) carry[k] = 0.f;

    constexpr int SPAN = NW * LC;
    const int n_full = T / SPAN;
    const bool ragged = (T % SPAN) != 0;

    Pack<V> g[LC];
#pragma unroll
    for (int j = 0; j < LC; ++j)
#pragma unroll
        for (int k = 0; k < V; ++k) g[j].v[k] = 0.f;

    auto chunk_t0 = [&](int it) { return (it * NW + w) * LC; };
    auto issue = [&](int it) {
        if (col_ok) {
            const float* gp = grad_adv + (size_t)chunk_t0(it) * B + col;
#pragma unroll
            for (int j = 0; j < LC; ++j) g[j] = load_pack<V, NTL>(gp + (size_t)j * B);
        }
    };
    auto finish = [&](auto guard_, int it, int t0, float (&L)[LC][V], float (&Q)[LC]) {
        constexpr bool GUARD = decltype(guard_)::value != 0;
        const int buf = it & 1;
#pragma unroll
        for (int k = 0; k < V; ++k) s_l0[(buf * NW + w) * TILE + lane * V + k] = L[LC - 1][k];
        if (lane == 0) s_p0[buf * NW + w] = Q[LC - 1];
        __syncthreads();
        float A[V], Aw[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { A[k] = carry[k]; Aw[k] = 0.f; }
#pragma unroll
        for (int u = 0; u < NW; ++u) {
            if (u == w) {
#pragma unroll
                for (int k = 0; k < V; ++k) Aw[k] = A[k];
            }
            const float q0 = s_p0[buf * NW + u];
#pragma unroll
            for (int k = 0; k < V; ++k) A[k] = fmaf(q0, A[k], s_l0[(buf * NW + u) * TILE + lane * V + k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = A[k];
        if (col_ok) {
            float prev[V];
#pragma unroll
            for (int k = 0; k < V; ++k) prev[k] = Aw[k];
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                const int t = t0 + j;
                if (!GUARD || t < T) {
                    Pack<V> d, gv;
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        d.v[k] = fmaf(Q[j], Aw[k], L[j][k]);
                        gv.v[k] = fmaf(gamma, prev[k], -d.v[k]);
                        prev[k] = d.v[k];
                    }
                    store_pack<V, NTS>(grad_reward + (size_t)t * B + col, d);
                    store_pack<V, NTS>(grad_value + (size_t)t * B + col, gv);
                    if ((GUARD || j == LC - 1) && t == T - 1) {  // bootstrap row: dL/dV_T = gamma * d_{T-1}
                        Pack<V> last;
#pragma unroll
                        for (int k = 0; k < V; ++k) last.v[k] = gamma * d.v[k];
                        store_pack<V, NTS>(grad_value + (size_t)T * B + col, last);
                    }
                }
            }
        }
    };

    if (n_full > 0) issue(0);
    for (int it = 0; it < n_full; ++it) {
        const int t0 = chunk_t0(it);
        float L[LC][V];
        float Q[LC];
        {
            float a[V];
#pragma unroll
            for (int k = 0; k < V; ++k) a[k] = 0.f;
            float q = 1.f;
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                const int t = t0 + j;
                const float c = (t > 0) ? coef[t - 1] : 0.f;
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    a[k] = fmaf(c, a[k], g[j].v[k]);
                    L[j][k] = a[k];
                }
                q *= c;
                Q[j] = q;
            }
        }
        if (it + 1 < n_full) issue(it + 1);
        finish(std::integral_constant<int, 0>{}, it, t0, L, Q);
    }
    if (ragged) {
        const int it = n_full;
        const int t0 = chunk_t0(it);
        float L[LC][V];
        float Q[LC];
        float a[V];
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = 0.f;
        float q = 1.f;
#pragma unroll
        for (int j = 0; j < LC; ++j) {
            const int t = t0 + j;
            if (t < T) {
                const float c = (t > 0) ? coef[t - 1] : 0.f;
                Pack<V> gg;
                if (col_ok) {
                    gg = load_pack<V, NTL>(grad_adv + (size_t)t * B + col);
                } else {
#pragma unroll
                    for (int k = 0; k < V; ++k) gg.v[k] = 0.f;
                }
#pragma unroll
                for (int k = 0; k < V; ++k) a[k] = fmaf(c, a[k], gg.v[k]);
                q *= c;
            }
#pragma unroll
            for (int k = 0; k < V; ++k) L[j][k] = a[k];
            Q[j] = q;
        }
        finish(std::integral_constant<int, 1>{}, it, t0, L, Q);
    }
}

// ------------------------------------------------------------------------------------------------
// host side: configuration choice + dispatch
// ------------------------------------------------------------------------------------------------
struct Cfg { int v, lc, nw, flags; bool half; bool pf; };

// Kernel timing (hpc_rll_ktime_begin / _end): while armed, every GAE launch goes through hipExtLaunchKernelGGL with a
// start/stop event pair that brackets the KERNEL ITSELF (the dispatch packet's begin / end timestamps -- what
// rocprofv3 --kernel-trace reports), not the gap to the neighbouring launches that a pair of hipEventRecord calls on
// the stream also counts.  A diagnostic for bench.py's live roofline figure; not thread-safe, not for capture.
struct KTime {
    bool on = false;
    int cap = 0, n = 0;
    hipEvent_t* ev = nullptr;
    int* kind = nullptr;
    std::atomic<int> last_cfg[2][6] = {};   // per direction: v, lc, nw, flags, half, pipelined (relaxed atomics: GAE launches
                                            // may come from several host threads; a diagnostic, no ordering implied)
};
KTime g_kt;

template <class K, class... A>
inline void launch(K kernel, dim3 grid, dim3 block, hipStream_t st, int kind, A... args) {
    if (g_kt.on && g_kt.n < g_kt.cap) {
        const int i = g_kt.n++;
        g_kt.kind[i] = kind;
        hipExtLaunchKernelGGL(kernel, grid, block, 0, st, g_kt.ev[2 * i], g_kt.ev[2 * i + 1], 0, args...);
    } else {
        hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
    }
}

inline bool aligned(const void* p, size_t a) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Widest pack the shape and pointers allow.
inline int max_vec(int B, std::initializer_list<const void*> ptrs) {
    for (int v : {4, 2}) {
        bool ok = (B % v) == 0;
        for (const void* p : ptrs) ok = ok && aligned(p, 4 * v);
        if (ok) return v;
    }
    return 1;
}

// Launch-configuration heuristic, tuned on MI355X with forward and backward launches ALTERNATING (the real
// access pattern; a kernel repeated back to back finds part of its input in the 256 MiB Infinity Cache and
// looks faster than it is).  Sweeps: profiles/r01/r01_gae_tuning_*.txt (every instantiation at T=1024,B=65536) and
// profiles/r01/r01_gae_heuristic_shapes.txt (candidate sets at 7 more shapes); summary in DESIGN.md.
//   * ~32 KiB of loads in flight per CU saturates HBM; beyond ~2048 waves more occupancy does not help.
//   * Stores are always NONTEMPORAL: a regular store leaves dirty lines in L2/MALL whose write-back lands in the
//     NEXT kernel (+35-45 us on the following launch).  For working sets beyond the Infinity Cache the forward also
//     loads nontemporally, which helps itself and the backward that follows it.
//   * streaming regime (one launch moves >= 300 MB): forward 2 columns/lane from B = 65536 up, 512 workgroups x
//     4 waves x 8 steps or >= 1024 workgroups x 8 waves x 16 steps; backward 2 columns/lane with 4 waves x 2 steps in the middle (4 columns/lane from B = 131072),
//     1 column/lane 8 x 8 for narrow batches, 2 columns/lane 16 x 16 for very wide ones.
//   * cache-resident regime: 1 column per lane and up to 16 waves x 16 steps so that ~2048 waves cover the chip even
//     when there are few column tiles (B=64 -> one workgroup walking T in 256-step strides).
inline Cfg choose_cfg(bool fwd, int T, int B, int vmax, int v, int lc, int nw, int flags) {
    auto wgs_for = [&](int vv) { return (B + 64 * vv - 1) / (64 * vv); };
    const bool all_auto = v == 0 && lc == 0 && nw == 0;
    const bool streaming = (12.0 * (double)T * (double)B) >= 300e6;
    int av, alc, anw, afl;
    bool apf = false;   // the software-pipelined kernels (round 3)
    // Round 3 sweeps (tests/tools/r03_gae_pf_sweep.py -> profiles/r03_gae_pf_sweep_*.txt; forward and backward
    // alternating, kernel begin/end timestamps, shipped round-2 configuration -> pipelined one in the same process):
    //   T=1024 B=32768  fwd (1,8,4) 65.7 -> (2,16,8)p 63.8      bwd (1,8,8) 59.5 -> (4,4,4)p 56.3
    //   T=1024 B=65536  fwd (2,8,2) 132.1 -> (2,4,2)p 128.8 / (4,4,8)p 128.7 (shipped: see below)   bwd (2,2,4) 116.1 -> (4,2,4)p 112.4
    //   T=1024 B=131072 fwd (2,16,8) 272.7, no pipelined gain   bwd (4,2,4) 305.1 -> (4,2,4)p 288.9
    //   T=256  B=262144 fwd (2,16,8) 131.6 -> (2,16,4)p 129.2   bwd (2,16,16) 135.4 -> (1,8,8)p 115.4
    // (plain loads in the forward are 0.3 us faster but cost the next backward 25 us: forward loads stay nontemporal)
    if (streaming && fwd) {
        av = (B >= 65536 && vmax >= 2) ? 2 : 1;
        const int wgs = wgs_for(av);
        if (wgs >= 1024) { alc = 16; anw = 8; }
        else if (wgs >= 512) { alc = 8; anw = av == 2 ? 2 : 4; }   // ~35 KiB of loads in flight per CU either way
        else { alc = 16; anw = 8; }
        afl = 3;
        if (vmax >= 2) {
            if (B < 65536) { if (B >= 32768) { av = 2; alc = 16; anw = 8; apf = true; } }   // narrower: keep the round-2 choice
            // B = 65536: (2,4,2)p is the fastest on a good day (124-127 us) but PLACEMENT-SENSITIVE: over ten independent
            // buffer sets in one process it reads 125 / 142 / 160 us (min / mean / max), the slow sets being slow for good
            // -- it is the physical placement of the OUTPUT buffer (swap `adv` for another allocation and the same inputs
            // run fast), and the r02 (2,8,2) configuration behaves the same (this is the "slow box" of round 2).  With 16-byte
            // stores and 8 waves per workgroup, (4,4,8)p reads 126 / 129 / 132 over the same sets
            // (tests/tools/r03_gae_placement_probe.py, profiles/r03_gae_placement_probe.txt).
            else if (B < 131072) {
                if (vmax >= 4) { av = 4; alc = 4; anw = 8; } else { av = 2; alc = 4; anw = 4; }
                apf = true;
            }
            else if (B >= 262144) { av = 2; alc = 16; anw = 4; apf = true; }
        }
    } else if (streaming) {
        if (B >= 262144) { av = 1; alc = 8; anw = 8; apf = true; }
        else if (B >= 65536 && vmax >= 4) { av = 4; alc = 2; anw = 4; apf = true; }
        else if (B >= 131072 && vmax >= 2) { av = 2; alc = 2; anw = 4; }
        // round 2 in-process A/B at B = 65536 (tests/tools/gae_bwd_ab.py): (2,2,4) 115.5 us, (4,2,4) 116.8, (4,4,4) 120.1
        else if (B >= 65536 && vmax >= 2) { av = 2; alc = 2; anw = 4; }
        else if (B >= 32768 && vmax >= 4) { av = 4; alc = 4; anw = 4; apf = true; }
        else { av = 1; alc = 8; anw = 8; }
        afl = 2;
    } else {
        av = 1; alc = 16; anw = 4;
        while (anw < 16 && wgs_for(1) * anw < 2048) anw <<= 1;
        afl = 2;
        // round 3 (profiles/r03_gae_pf_sweep_1024x16384.txt / _1024x8192.txt): with one workgroup per CU and several
        // iterations the pipelined kernels pay in the cache-resident regime too -- T=1024, B=16384: forward (1,16,8)
        // 32.9 -> 29.3 us, backward (1,16,8) 30.5 -> (1,8,8)p 28.7 us (plain loads: the data is in the Infinity Cache);
        // at B = 8192 (half-wave tiles, two iterations) there is nothing to pipeline: 17.8 / 16.7 vs 17.0 / 16.4 us
        if (T >= 512 && wgs_for(1) >= 256 && anw == 8) {
            apf = true;
            if (!fwd) alc = 8;
        }
    }
    if (v == 0) v = av;
    if (v > vmax) v = vmax;
    if (lc == 0) {
        lc = alc;
        if (lc == 16 && v == 4) lc = 8;  // (4,16) is not instantiated (VGPR budget)
    }
    if (nw == 0) {
        nw = anw;
        const int chunks = (T + lc - 1) / lc;
        while (nw > 1 && nw > chunks) nw >>= 1;
    }
    // narrow batches: half-wave tiles (32 columns, two time chunks per wave) when the 64-column tiling cannot give
    // every CU a workgroup and the trajectory has enough chunks for 32 virtual waves; explicit request: flags bit 2
    const int explicit_flags = flags;
    bool half = flags >= 0 && (flags & 4);
    if (flags < 0) {
        flags = afl;
        half = all_auto && !streaming && v == 1 && lc == 16 && nw == 16 && wgs_for(1) < 256 && T >= 512;
    }
    if (half && !(v == 1 && nw == 16 && (lc == 8 || lc == 16))) half = false;
    // software-pipelined forward (gae_fwd_pf_kernel): explicit request = flags bit 3; instantiated for
    // lc in {4,8,16} (not (4,16)), nw in {2,4,8}, nontemporal stores
    bool pf = !half && (explicit_flags >= 0 ? (explicit_flags & 8) != 0 : (all_auto && apf));
    if (pf && !(nw == 2 || nw == 4 || nw == 8)) pf = false;
    if (pf && fwd && !((lc == 4 || lc == 8 || lc == 16) && !(v == 4 && lc == 16) && !(v == 1 && lc == 4))) pf = false;
    if (pf && !fwd && !((lc == 2 || lc == 4 || lc == 8) && !(v == 1 && lc != 8))) pf = false;
    if (pf) flags |= 2;
    Cfg out{v, lc, nw, flags & 3, half, pf};
    return out;
}

template <int N> using I = std::integral_constant<int, N>;

// (V, LC, NW) triples that are instantiated (each in 4 nontemporal flavours).
template <class F>
inline void for_each_cfg(F&& f) {
#define HPC_RLL_NW_ROW(V, LC) \
    f(I<V>{}, I<LC>{}, I<1>{}); f(I<V>{}, I<LC>{}, I<2>{}); f(I<V>{}, I<LC>{}, I<4>{}); \
    f(I<V>{}, I<LC>{}, I<8>{}); f(I<V>{}, I<LC>{}, I<16>{});
    HPC_RLL_NW_ROW(1, 4) HPC_RLL_NW_ROW(1, 8) HPC_RLL_NW_ROW(1, 16)
    HPC_RLL_NW_ROW(2, 2) HPC_RLL_NW_ROW(2, 4) HPC_RLL_NW_ROW(2, 8) HPC_RLL_NW_ROW(2, 16)
    HPC_RLL_NW_ROW(4, 2) HPC_RLL_NW_ROW(4, 4) HPC_RLL_NW_ROW(4, 8)
#undef HPC_RLL_NW_ROW
}

#define HPC_RLL_GAE_DISPATCH(KERNEL, KIND, ...)                                                    \
    do {                                                                                           \
        bool hit = false;                                                                          \
        {                                                                                          \
            std::atomic<int>* lc_ = g_kt.last_cfg[KIND];                                           \
            lc_[0] = cfg.v; lc_[1] = cfg.lc; lc_[2] = cfg.nw; lc_[3] = cfg.flags; lc_[4] = cfg.half; lc_[5] = cfg.pf; \
        }                                                                                          \
        if (cfg.half) {                                                                            \
            const dim3 grid((unsigned)((B + 31) / 32)), block(1024);                               \
            const bool ntl = cfg.flags & 1;                                                        \
            if (cfg.lc == 16) {                                                                    \
                if (ntl) launch(KERNEL<1, 16, 16, true, true, true>, grid, block, st, KIND, __VA_ARGS__);   \
                else launch(KERNEL<1, 16, 16, false, true, true>, grid, block, st, KIND, __VA_ARGS__);      \
            } else {                                                                               \
                if (ntl) launch(KERNEL<1, 8, 16, true, true, true>, grid, block, st, KIND, __VA_ARGS__);    \
                else launch(KERNEL<1, 8, 16, false, true, true>, grid, block, st, KIND, __VA_ARGS__);       \
            }                                                                                      \
            hit = true;                                                                            \
        }                                                                                          \
        auto go = [&](auto V_, auto LC_, auto NW_) {                                               \
            constexpr int V = decltype(V_)::value, LC = decltype(LC_)::value,                      \
                          NW = decltype(NW_)::value;                                               \
            if (!hit && cfg.v == V && cfg.lc == LC && cfg.nw == NW) {                                \
                const dim3 grid((unsigned)((B + 64 * V - 1) / (64 * V))), block(NW * 64);          \
                switch (cfg.flags) {                                                               \
                    case 0: launch(KERNEL<V, LC, NW, false, false>, grid, block, st, KIND, __VA_ARGS__); break; \
                    case 1: launch(KERNEL<V, LC, NW, true, false>, grid, block, st, KIND, __VA_ARGS__); break;  \
                    case 2: launch(KERNEL<V, LC, NW, false, true>, grid, block, st, KIND, __VA_ARGS__); break;  \
                    default: launch(KERNEL<V, LC, NW, true, true>, grid, block, st, KIND, __VA_ARGS__); break;  \
                }                                                                                  \
                hit = true;                                                                        \
            }                                                                                      \
        };                                                                                         \
        for_each_cfg(go);                                                                          \
        if (!hit) return HPC_RLL_EUNSUPPORTED;                                                     \
    } while (0)

// the pipelined forward: (V, LC) x NW in {2,4,8}, nontemporal stores, both load flavours
template <class... A>
inline bool dispatch_fwd_pf(const Cfg& cfg, int B, hipStream_t st, A... args) {
    bool hit = false;
    auto go = [&](auto V_, auto LC_, auto NW_) {
        constexpr int V = decltype(V_)::value, LC = decltype(LC_)::value, NW = decltype(NW_)::value;
        if (!hit && cfg.v == V && cfg.lc == LC && cfg.nw == NW) {
            const dim3 grid((unsigned)((B + 64 * V - 1) / (64 * V))), block(NW * 64);
            const bool full = (B % (64 * V)) == 0;
            if (cfg.flags & 1) {
                if (full) launch(gae_fwd_pf_kernel<V, LC, NW, true, true, true>, grid, block, st, 0, args...);
                else launch(gae_fwd_pf_kernel<V, LC, NW, true, true, false>, grid, block, st, 0, args...);
            } else {
                if (full) launch(gae_fwd_pf_kernel<V, LC, NW, false, true, true>, grid, block, st, 0, args...);
                else launch(gae_fwd_pf_kernel<V, LC, NW, false, true, false>, grid, block, st, 0, args...);
            }
            hit = true;
        }
    };
#define HPC_RLL_PF_ROW(V, LC) go(I<V>{}, I<LC>{}, I<2>{}); go(I<V>{}, I<LC>{}, I<4>{}); go(I<V>{}, I<LC>{}, I<8>{});
    HPC_RLL_PF_ROW(1, 8) HPC_RLL_PF_ROW(1, 16) HPC_RLL_PF_ROW(2, 4) HPC_RLL_PF_ROW(2, 8) HPC_RLL_PF_ROW(2, 16)
    HPC_RLL_PF_ROW(4, 4) HPC_RLL_PF_ROW(4, 8)
#undef HPC_RLL_PF_ROW
    return hit;
}

template <class... A>
inline bool dispatch_bwd_pf(const Cfg& cfg, int B, hipStream_t st, A... args) {
    bool hit = false;
    auto go = [&](auto V_, auto LC_, auto NW_) {
        constexpr int V = decltype(V_)::value, LC = decltype(LC_)::value, NW = decltype(NW_)::value;
        if (!hit && cfg.v == V && cfg.lc == LC && cfg.nw == NW) {
            const dim3 grid((unsigned)((B + 64 * V - 1) / (64 * V))), block(NW * 64);
            const bool full = (B % (64 * V)) == 0;
            if (cfg.flags & 1) {
                if (full) launch(gae_bwd_pf_kernel<V, LC, NW, true, true, true>, grid, block, st, 1, args...);
                else launch(gae_bwd_pf_kernel<V, LC, NW, true, true, false>, grid, block, st, 1, args...);
            } else {
                if (full) launch(gae_bwd_pf_kernel<V, LC, NW, false, true, true>, grid, block, st, 1, args...);
                else launch(gae_bwd_pf_kernel<V, LC, NW, false, true, false>, grid, block, st, 1, args...);
            }
            hit = true;
        }
    };
#define HPC_RLL_PF_ROW(V, LC) go(I<V>{}, I<LC>{}, I<2>{}); go(I<V>{}, I<LC>{}, I<4>{}); go(I<V>{}, I<LC>{}, I<8>{});
    HPC_RLL_PF_ROW(1, 8) HPC_RLL_PF_ROW(2, 2) HPC_RLL_PF_ROW(2, 4) HPC_RLL_PF_ROW(2, 8)
    HPC_RLL_PF_ROW(4, 2) HPC_RLL_PF_ROW(4, 4) HPC_RLL_PF_ROW(4, 8)
#undef HPC_RLL_PF_ROW
    return hit;
}

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

}  // namespace
}  // namespace hpc_rll

using namespace hpc_rll;

extern "C" int hpc_rll_gae_coef(float* coef, int T, float gamma, float lambda, void* stream) {
    if (T < 0) return HPC_RLL_EINVAL;
    if (T == 0) return HPC_RLL_OK;
    if (!coef) return HPC_RLL_EINVAL;
    hipLaunchKernelGGL(gae_coef_kernel, dim3((T + 255) / 256), dim3(256), 0, (hipStream_t)stream, coef, T, gamma,
                       lambda);
    return check_launch();
}

extern "C" int hpc_rll_gae_forward_ex(const float* value, const float* reward, float* adv, const float* coef,
                                      int T, int B, float gamma, int vec, int lc, int nw, int flags, void* stream) {
    if (T < 0 || B < 0) return HPC_RLL_EINVAL;
    if (T == 0 || B == 0) return HPC_RLL_OK;
    if (!value || !reward || !adv || !coef) return HPC_RLL_EINVAL;
    if (!aligned(value, 4) || !aligned(reward, 4) || !aligned(adv, 4) || !aligned(coef, 4)) return HPC_RLL_EALIGN;
    if (flags >= 0 && (flags & ~15)) return HPC_RLL_EUNSUPPORTED;   // bits 4, 5: retired in round 5
    const Cfg cfg = choose_cfg(true, T, B, max_vec(B, {value, reward, adv}), vec, lc, nw, flags);
    hipStream_t st = (hipStream_t)stream;
    if (cfg.pf) {
        std::atomic<int>* lc_ = g_kt.last_cfg[0];
        lc_[0] = cfg.v; lc_[1] = cfg.lc; lc_[2] = cfg.nw; lc_[3] = cfg.flags; lc_[4] = 0; lc_[5] = 1;
        if (!dispatch_fwd_pf(cfg, B, st, value, reward, adv, coef, T, B, gamma)) return HPC_RLL_EUNSUPPORTED;
        return check_launch();
    }
    HPC_RLL_GAE_DISPATCH(gae_fwd_kernel, 0, value, reward, adv, coef, T, B, gamma);
    return check_launch();
}

extern "C" int hpc_rll_gae_forward(const float* value, const float* reward, float* adv, const float* coef, int T,
                                   int B, float gamma, void* stream) {
    return hpc_rll_gae_forward_ex(value, reward, adv, coef, T, B, gamma, 0, 0, 0, -1, stream);
}

extern "C" int hpc_rll_gae_backward_ex(const float* grad_adv, float* grad_value, float* grad_reward,
                                       const float* coef, int T, int B, float gamma, int vec, int lc, int nw,
                                       int flags, void* stream) {
    if (T < 0 || B < 0) return HPC_RLL_EINVAL;
    if (B == 0) return HPC_RLL_OK;
    if (T == 0) {  // grad_value has one row (the bootstrap value), which adv does not depend on
        if (grad_value) return (int)hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)B, (hipStream_t)stream);
        return HPC_RLL_OK;
    }
    if (!grad_adv || !coef) return HPC_RLL_EINVAL;
    if (!grad_value && !grad_reward) return HPC_RLL_OK;
    if (!aligned(grad_adv, 4) || !aligned(grad_value, 4) || !aligned(grad_reward, 4) || !aligned(coef, 4))
        return HPC_RLL_EALIGN;
    if (flags >= 0 && (flags & ~15)) return HPC_RLL_EUNSUPPORTED;   // bits 4, 5: retired in round 5
    const Cfg cfg = choose_cfg(false, T, B, max_vec(B, {grad_adv, grad_value, grad_reward}), vec, lc, nw, flags);
    hipStream_t st = (hipStream_t)stream;
    if (cfg.pf && grad_value && grad_reward) {
        std::atomic<int>* lc_ = g_kt.last_cfg[1];
        lc_[0] = cfg.v; lc_[1] = cfg.lc; lc_[2] = cfg.nw; lc_[3] = cfg.flags; lc_[4] = 0; lc_[5] = 1;
        if (!dispatch_bwd_pf(cfg, B, st, grad_adv, grad_value, grad_reward, coef, T, B, gamma)) return HPC_RLL_EUNSUPPORTED;
        return check_launch();
    }
    HPC_RLL_GAE_DISPATCH(gae_bwd_kernel, 1, grad_adv, grad_value, grad_reward, coef, T, B, gamma);
    return check_launch();
}

extern "C" int hpc_rll_gae_backward(const float* grad_adv, float* grad_value, float* grad_reward, const float* coef,
                                    int T, int B, float gamma, void* stream) {
    return hpc_rll_gae_backward_ex(grad_adv, grad_value, grad_reward, coef, T, B, gamma, 0, 0, 0, -1, stream);
}

// ---- diagnostics: per-launch kernel durations of the GAE kernels (see KTime above) ----------------------------------
extern "C" int hpc_rll_ktime_begin(int capacity) {
    if (capacity <= 0 || g_kt.on) return HPC_RLL_EINVAL;
    g_kt.ev = new (std::nothrow) hipEvent_t[2 * (size_t)capacity];
    g_kt.kind = new (std::nothrow) int[capacity];
    hipError_t e = (g_kt.ev && g_kt.kind) ? hipSuccess : hipErrorOutOfMemory;
    int made = 0;
    for (; e == hipSuccess && made < 2 * capacity; ++made) e = hipEventCreate(&g_kt.ev[made]);
    if (e != hipSuccess) {   // nothing half-initialised is left behind: the events made so far, both arrays
        for (int i = 0; i + 1 < made; ++i) (void)hipEventDestroy(g_kt.ev[i]);   // (the failed slot itself was never created)
        delete[] g_kt.ev;
        delete[] g_kt.kind;
        g_kt.ev = nullptr;
        g_kt.kind = nullptr;
        g_kt.cap = g_kt.n = 0;
        return (int)e;
    }
    g_kt.cap = capacity;
    g_kt.n = 0;
    g_kt.on = true;
    return HPC_RLL_OK;
}

// Waits for the recorded launches, writes their durations (milliseconds) and kinds (0 = forward, 1 = backward) in launch
// order, disarms.  Returns the number of launches recorded (<= max written) or a negative / HIP status.
extern "C" int hpc_rll_ktime_end(float* ms, int* kind, int max) {
    if (!g_kt.on) return HPC_RLL_EINVAL;
    g_kt.on = false;
    const int n = g_kt.n;
    int rc = n;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventSynchronize(g_kt.ev[2 * i + 1]);
        float t = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, g_kt.ev[2 * i], g_kt.ev[2 * i + 1]);
        if (e != hipSuccess) { rc = (int)e > 0 ? -1000 - (int)e : HPC_RLL_EINVAL; break; }
        if (i < max) {
            if (ms) ms[i] = t;
            if (kind) kind[i] = g_kt.kind[i];
        }
    }
    for (int i = 0; i < 2 * g_kt.cap; ++i) (void)hipEventDestroy(g_kt.ev[i]);
    delete[] g_kt.ev;
    delete[] g_kt.kind;
    g_kt.ev = nullptr;
    g_kt.kind = nullptr;
    g_kt.cap = g_kt.n = 0;
    return rc;
}

// The launch configuration the most recent GAE forward (dir = 0) / backward (dir = 1) call of this process used:
// out[6] = {columns per lane, steps per wave chunk, waves per workgroup, nontemporal flags, half-wave tiles, pipelined}.
extern "C" int hpc_rll_gae_last_config(int dir, int* out) {
    if ((dir != 0 && dir != 1) || !out) return HPC_RLL_EINVAL;
    for (int i = 0; i < 6; ++i) out[i] = g_kt.last_cfg[dir][i];
    return HPC_RLL_OK;
}
