// lstm_persist.hpp -- persistent small-batch path of the LayerNorm-LSTM (included by lstm.hip only).
//
// Regime: B <= 4 (the reference's own test shape is S=64, B=3, I=1792, H=384, L=3: tests/test_lstm.py:12-17).  There
// the per-step work is ~1 us of arithmetic but the two-launch step (skinny GEMM + cell kernel) costs ~18 us of kernel
// latency.  Here ONE kernel per layer walks all S steps:
//   * workgroup w owns JW hidden units -> the 4*JW gate columns of Wh, resident in LDS for the whole sequence
//     (forward: the columns; backward: the same units' ROWS of Wh for dh_prev = dHW @ Wh^T);
//   * the cell state c (backward: its adjoint dc, and dh) of the owned units lives in registers of the owning threads;
//   * per step two all-to-all exchanges between the workgroups go through memory as 64-bit {value, tag} words
//     (agent-scope relaxed atomics; the tag is the step number, so the data word itself is the ready flag and an
//     exchange costs one store + one load round trip -- ~0.5-0.65 us one way on MI355X, tests/tools/micro/pingpong.hip
//     -- not a counter barrier):
//       forward : (1) LayerNorm partials of h@Wh (sum, M2 -> combined with Chan's formula), (2) h_s;
//       backward: (1) the four LayerNorm-adjoint row sums, (2) dHW_s.
//     Every poll round issues all of a thread's loads back to back, lanes read consecutive words (8 lanes per 64-byte
//     sector), and buffers are double-buffered on the tag parity: a workgroup is at most one exchange ahead of the
//     slowest, which has then finished reading the buffer being overwritten.
//   * all workgroups must be co-resident.  That is REQUESTED, not assumed:
//       - at dispatch the runtime's own occupancy answer for the chosen instantiation
//         (hipOccupancyMaxActiveBlocksPerMultiprocessor x the CU count of the CURRENT device) must cover the grid,
//         otherwise the step kernels run;
//       - persistent launches of one process are chained per device with an event, so two of them (different streams)
//         never share the device half-resident;
//       - waits are bounded (kSpinLimit polls ~ seconds).  A wave that gives up sets a device abort word, which every
//         poll loop honours, writes a pinned HOST status word (system scope) and ENDS -- no trap, the HIP context
//         survives.  The next LSTM entry point reports HPC_RLL_ETIMEOUT (like an asynchronous HIP error: the results of
//         the call that timed out are invalid) until hpc_rll_clear_async_error(), which also switches the persistent
//         paths off for the rest of the process.  Only another PROCESS holding CUs for seconds can cause that.
// The saved-for-backward tensors (hw, gates, c, hseq, stats) are written exactly as the step-kernel path writes them,
// so forward/backward paths can be mixed.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>

#include "hpc_rll_hip.h"
#include "wave.hpp"

namespace hpc_rll {
int g_lstm_persist = 1;          // hpc_rll_tune_set key 3
constexpr int g_lstm_persist_max_b = 4;   // largest batch the persistent kernels take
constexpr int g_lstm_xchg_rep = 4;   // replicas of every exchange word (measured: 4 replicas -20 % per gather, 8 and more lose again)
namespace {

typedef unsigned long long u64;
constexpr long kSpinLimit = 1L << 22;
constexpr int kPersistMaxB = 4;

// ---- bounded waits without a trap -------------------------------------------------------------------------------
__device__ unsigned g_persist_abort = 0;                 // set by the first wave that gives up (sticky)
__device__ unsigned* g_persist_host_status = nullptr;    // pinned host word, see PersistRuntime
__device__ long g_persist_spin_limit = kSpinLimit;       // polls before giving up (test hook: hpc_rll_test_set_persist_spin_limit)

// one failed poll round: back off; every 1024 rounds look at the abort word / the limit
__device__ __forceinline__ void persist_poll_failed(long& spins) {
    if (__builtin_expect(((++spins) & 1023) == 0, 0)) {
        if (__hip_atomic_load(&g_persist_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
            __builtin_amdgcn_endpgm();
        if (spins >= g_persist_spin_limit) {
            __hip_atomic_store(&g_persist_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned* hs = g_persist_host_status;
            if (hs) __hip_atomic_store(hs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_endpgm();
        }
    }
    // the nap between polls: 64 cycles.  (512, ~0.2 us, shipped until round 4; 64 measured 1.5-5 % faster forward and 1-2.4 %
    // faster backward on three small-batch shapes, profiles/r04_persist_nap.txt)
    __builtin_amdgcn_s_sleep(1);
}

__device__ __forceinline__ void xchg_put(u64* p, float v, uint32_t tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every workgroup reads every exchange word: with 192 readers per line one poll round takes ~2.6 us although the data
// is already there (measured: 1.0 poll rounds per gather), against 0.5-0.65 us for a single reader.  Each word is
// therefore published to `nrep` = 4 replicas (a constant since round 5) and workgroup w reads replica w % nrep: nrep x more
// (fire-and-forget) stores, nrep x fewer readers per line.  Measured: 4 replicas -20 % per gather, 8 and more lose
// again to the extra stores; spreading the lines over memory channels instead made no difference.
constexpr int kMaxRep = 32;
__device__ __forceinline__ void xchg_put_all(u64* base, size_t rep_words, int nrep, int i, float v, uint32_t tag) {
    for (int r = 0; r < nrep; ++r) xchg_put(base + (size_t)r * rep_words + i, v, tag);
}


// wait until every word idx[i] with bit i of `valid` set carries `tag`.
// PAR = false: a load under `valid ? load : const` compiles to a branch per word with s_waitcnt vmcnt(0) inside, i.e.
// the NL words of a poll are fetched one round trip after the other.  For the few-word exchanges of row sums that
// trickle is what one wants (alternating-process A/B on one box: issuing them together made the step 15 % SLOWER --
// every waiting thread then hammers the memory path the producers' stores need).  PAR = true issues all NL loads
// unconditionally (an invalid slot reads word 0, result ignored) before the first check: one round trip per poll; used
// by the forward h gather (-2.5 % at B*H = 512, neutral at 4096); the backward dgate gather measured +6 % with it.
template <int NL, bool PAR = false>
__device__ __forceinline__ void xchg_get(const u64* base, const int (&idx)[NL], unsigned valid, uint32_t tag,
                                         float (&out)[NL], u64* poll_count = nullptr) {
    static_assert(NL <= 32, "validity mask is 32 bits");
    long spins = 0;
    while (true) {
        u64 w[NL];
        bool ok = true;
        if (PAR) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                w[i] = __hip_atomic_load(base + (((valid >> i) & 1u) ? idx[i] : 0), __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int i = 0; i < NL; ++i) ok = ok && (!((valid >> i) & 1u) || (uint32_t)(w[i] >> 32) == tag);
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                w[i] = ((valid >> i) & 1u)
                           ? __hip_atomic_load(base + idx[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                           : ((u64)tag << 32);
#pragma unroll
            for (int i = 0; i < NL; ++i) ok = ok && ((uint32_t)(w[i] >> 32) == tag);
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < NL; ++i) out[i] = __uint_as_float((uint32_t)w[i]);
            if (poll_count && threadIdx.x == 0 && blockIdx.x == 0) *poll_count += (u64)(spins + 1) * 100;
            return;
        }
        persist_poll_failed(spins);
    }
}

// all 256 threads: words [0, n) of `src` -> dst[0, n) (LDS), CH words per thread and poll round
template <int CH, bool PAR = false>
__device__ __forceinline__ void xchg_gather(const u64* src, int n, uint32_t tag, float* dst,
                                            u64* poll_count = nullptr) {
    for (int e0 = threadIdx.x; e0 < n; e0 += 256 * CH) {
        int idx[CH];
        unsigned valid = 0;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            idx[i] = e0 + 256 * i;
            if (idx[i] < n) valid |= 1u << i;
        }
        float v[CH];
        xchg_get<CH, PAR>(src, idx, valid, tag, v, poll_count);
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if ((valid >> i) & 1u) dst[idx[i]] = v[i];
    }
}

// DPP-only sums: dpp_add / group_sum_last live in wave.hpp
__device__ __forceinline__ float wave_sum_last(float x) { return group_sum_last<64>(x); }
// the same, broadcast to every lane of the group (one ds_bpermute)
template <int GL>
__device__ __forceinline__ float group_sum_all(float x) {
    x = group_sum_last<GL>(x);
    return GL > 16 ? __shfl(x, (int)(threadIdx.x & 63) | (GL - 1), 64) : x;
}

// x-branch LayerNorm statistics of every (s,b) row of xw (S*B, 4H) -> stats[row*4 + {0,1}] = mean, rstd
__global__ __launch_bounds__(256) void lstm_rowstats_kernel(const float* __restrict__ xw, int G,
                                                            float* __restrict__ stats) {
    __shared__ float red[16];
    const float* r = xw + (size_t)blockIdx.x * G;
    float s[1] = {0.f};
    for (int c = threadIdx.x; c < G; c += 256) s[0] += r[c];
    block_allsum<1>(s, red);
    const float mean = s[0] / (float)G;
    float v[1] = {0.f};
    for (int c = threadIdx.x; c < G; c += 256) v[0] += (r[c] - mean) * (r[c] - mean);
    block_allsum<1>(v, red);
    if (threadIdx.x == 0) {
        stats[(size_t)blockIdx.x * 4] = mean;
        stats[(size_t)blockIdx.x * 4 + 1] = rsqrtf(v[0] / (float)G + kLnEps);
    }
}

#define HPC_RLL_TICK(i)                                                        \
    if (a.prof && blockIdx.x == 0 && tid == 0) {                               \
        const u64 now_ = wall_clock64();                                       \
        a.prof[i] += now_ - tprev_;                                            \
        tprev_ = now_;                                                         \
    }

// The per-row sums exchange: workgroup w publishes word (b*NQ + k)*nwg + w; batch row b is reduced by its own group
// of GL = min(64, 256/NB) lanes, lane `part` taking workgroups part, part+GL, ...
template <int NB> struct RowGroup {
    static constexpr int GL = (256 / NB) < 64 ? (256 / NB) : 64;
    static constexpr int NI = 256 / GL;   // workgroups per lane (nwg <= 256)
};

struct PersistFwd {
    const float *xw, *wh, *bias, *gamma, *beta, *h0, *c0;
    float *hw, *gates, *c, *hseq, *stats;
    u64 *hx, *sx;
    int S, B, H, nwg, nrep /* replicas of every exchange word */;
    size_t hx_par, sx_par /* words between the two parity buffers */, hx_rep, sx_rep /* ... between replicas */;
    uint32_t tag_base;
    u64* prof;   // optional: 8 phase accumulators (100 MHz ticks) written by workgroup 0 (HPC_RLL_LSTM_PROFILE=1)
};

template <int NB, int JW>
__global__ __launch_bounds__(256) void lstm_persist_fwd_kernel(PersistFwd a) {
    extern __shared__ float smem[];
    constexpr int CW = 4 * JW;
    constexpr int GL = RowGroup<NB>::GL, NI = RowGroup<NB>::NI;
    const int H = a.H, G = 4 * H, B = a.B, nwg = a.nwg;
    float* Wl = smem;              // [CW][H]   column c = gate*JW + jj of this workgroup's units
    float* hs = Wl + CW * H;       // [NB][H]   h_{s-1}
    float* pre = hs + NB * H;      // [NB][CW]  this workgroup's slice of h @ Wh
    float* lnst = pre + NB * CW;   // [NB][2]   mean, rstd of the h-branch rows
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const int j0 = blockIdx.x * JW;
    const size_t myrep = (size_t)(blockIdx.x % a.nrep);
    const int nvalid = (H - j0) < JW ? (H - j0) : JW;

    for (int e = tid; e < CW * H; e += 256) {
        const int jj = e % JW, gg = (e / JW) & 3, k = e / CW;
        Wl[(gg * JW + jj) * H + k] = (jj < nvalid) ? a.wh[(size_t)k * G + gg * H + j0 + jj] : 0.f;
    }
    for (int e = tid; e < NB * H; e += 256) hs[e] = (e < B * H) ? a.h0[e] : 0.f;

    // cell-phase ownership: thread (cb, cjj) keeps c of unit cj for batch row cb in a register
    const int cb = tid / JW, cjj = tid % JW, cj = j0 + cjj;
    const bool cell = tid < NB * JW && cb < B && cjj < nvalid;
    float gx[4], gh[4], bsum[4], creg = 0.f;
    if (cell) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int col = gg * H + cj;
            gx[gg] = a.gamma[col];
            gh[gg] = a.gamma[G + col];
            bsum[gg] = (a.beta[col] + a.beta[G + col]) + a.bias[col];
        }
        creg = a.c0[(size_t)cb * H + cj];
    }
    const float inv_g = 1.f / (float)G;
    // row-group role of this thread in the LayerNorm reduction
    const int rb = tid / GL, rpart = tid % GL;
    const bool rgrp = rb < B;
    __syncthreads();

    // vmcnt retires in order, so a poll's wait also waits for every plain load/store the thread issued before it.
    // Plain traffic is therefore issued right AFTER a poll has completed (it then overlaps the next compute phase):
    // the saved-for-backward stores of step s-1 and the x-branch loads of step s go after the h gather of step s.
    float sv_g[4] = {0.f, 0.f, 0.f, 0.f}, sv_p[4] = {0.f, 0.f, 0.f, 0.f}, sv_c = 0.f, sv_h = 0.f;   // step s-1
    float sv_mean = 0.f, sv_rstd = 0.f;
    auto flush_saved = [&](int sp) {
        if (cell) {
            const size_t row = (size_t)sp * B + cb;
            float* gr = a.gates + row * G;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                gr[gg * H + cj] = sv_g[gg];
                a.hw[row * G + gg * H + cj] = sv_p[gg];
            }
            a.c[row * H + cj] = sv_c;
            a.hseq[row * H + cj] = sv_h;
        }
        if (blockIdx.x == 0 && rgrp && rpart == GL - 1) {
            a.stats[((size_t)sp * B + rb) * 4 + 2] = sv_mean;
            a.stats[((size_t)sp * B + rb) * 4 + 3] = sv_rstd;
        }
    };
    u64 tprev_ = a.prof ? wall_clock64() : 0;
    for (int s = 0; s < a.S; ++s) {
        const uint32_t tag = a.tag_base + (uint32_t)s + 1u;
        const int par = (int)(tag & 1u);
        if (s > 0) {   // h_{s-1}: published by its owners with tag-1
            xchg_gather<8, true>(a.hx + (size_t)((tag - 1u) & 1u) * a.hx_par + myrep * a.hx_rep, B * H, tag - 1u, hs,
                           a.prof ? a.prof + 6 : nullptr);
            __syncthreads();
            flush_saved(s - 1);
        }
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, mx = 0.f, rx = 0.f;
        if (cell) {
            const float* xr = a.xw + ((size_t)s * B + cb) * G;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) xv[gg] = xr[gg * H + cj];
            const float* st = a.stats + ((size_t)s * B + cb) * 4;
            mx = st[0];
            rx = st[1];
        }
        HPC_RLL_TICK(0)
        // ---- this workgroup's slice of h @ Wh: wave g <-> gate g, lanes split k
        float acc[NB][JW];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) acc[b][jj] = 0.f;
#pragma unroll 4
        for (int k = lane; k < H; k += 64) {
            float wv[JW];
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) wv[jj] = Wl[(g * JW + jj) * H + k];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float hv = hs[b * H + k];
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) acc[b][jj] = fmaf(hv, wv[jj], acc[b][jj]);
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) {
                const float t = wave_sum_last(acc[b][jj]);
                if (lane == 63) pre[b * CW + g * JW + jj] = t;
            }
        __syncthreads();
        HPC_RLL_TICK(1)
        // ---- LayerNorm partials of the owned columns -> exchange
        if (tid < B) {
            float pv[CW], s1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                pv[c] = ((c % JW) < nvalid) ? pre[tid * CW + c] : 0.f;
                s1 += pv[c];
            }
            const float m = s1 / (4.f * (float)nvalid);
#pragma unroll
            for (int c = 0; c < CW; ++c) m2 += ((c % JW) < nvalid) ? (pv[c] - m) * (pv[c] - m) : 0.f;
            u64* dst = a.sx + (size_t)par * a.sx_par;
            xchg_put_all(dst, a.sx_rep, a.nrep, (tid * 2) * nwg + blockIdx.x, s1, tag);
            xchg_put_all(dst, a.sx_rep, a.nrep, (tid * 2 + 1) * nwg + blockIdx.x, m2, tag);
        }
        // ---- row b is combined by its lane group: Chan's formula over the workgroups' (sum, M2) pairs
        {
            int idx[2 * NI];
            unsigned valid = 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int w = rpart + GL * i;
                idx[i] = (rb * 2) * nwg + w;
                idx[NI + i] = (rb * 2 + 1) * nwg + w;
                if (rgrp && w < nwg) valid |= (1u << i) | (1u << (NI + i));
            }
            float v[2 * NI];
            xchg_get<2 * NI>(a.sx + (size_t)par * a.sx_par + myrep * a.sx_rep, idx, valid, tag, v,
                             a.prof ? a.prof + 7 : nullptr);
            HPC_RLL_TICK(2)
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) s1 += ((valid >> i) & 1u) ? v[i] : 0.f;
            const float mean = group_sum_all<GL>(s1) * inv_g;
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int w = rpart + GL * i;
                const int units = (H - w * JW) < JW ? (H - w * JW) : JW;
                const float n = 4.f * (float)units;
                const float d = v[i] / n - mean;
                t += ((valid >> i) & 1u) ? v[NI + i] + n * d * d : 0.f;
            }
            t = group_sum_last<GL>(t);
            if (rgrp && rpart == GL - 1) {
                const float rstd = rsqrtf(t * inv_g + kLnEps);
                lnst[rb * 2] = mean;
                lnst[rb * 2 + 1] = rstd;
                sv_mean = mean;
                sv_rstd = rstd;
            }
        }
        __syncthreads();
        HPC_RLL_TICK(3)
        // ---- cell
        if (cell) {
            const float mh = lnst[cb * 2], rh = lnst[cb * 2 + 1];
            float av[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float p = pre[cb * CW + gg * JW + cjj];
                av[gg] = ((xv[gg] - mx) * rx * gx[gg] + (p - mh) * rh * gh[gg]) + bsum[gg];
                sv_p[gg] = p;
            }
            const float ig = 1.f / (1.f + expf(-av[0]));
            const float fg = 1.f / (1.f + expf(-av[1]));
            const float og = 1.f / (1.f + expf(-av[2]));
            const float ug = tanhf(av[3]);
            creg = fg * creg + ig * ug;
            const float h = og * tanhf(creg);
            if (s + 1 < a.S) xchg_put_all(a.hx + (size_t)par * a.hx_par, a.hx_rep, a.nrep, cb * H + cj, h, tag);
            sv_g[0] = ig; sv_g[1] = fg; sv_g[2] = og; sv_g[3] = ug;
            sv_c = creg;
            sv_h = h;
        }
        HPC_RLL_TICK(4)
    }
    if (a.S > 0) flush_saved(a.S - 1);
}

struct PersistBwd {
    const float *d_out /* (S,B,H) or null */, *dhn, *dcn /* (B,H) or null */;
    const float *gates, *c, *c0, *xw, *hw, *stats, *gamma, *wh;
    float *dgate, *dxw, *dhw, *dh0, *dc0;
    u64 *big, *sums;
    int S, B, H, nwg, nrep;
    size_t big_par, sums_par, big_rep, sums_rep;
    uint32_t tag_base;
    u64* prof;
};

// Backward of one layer, all S steps.  Thread (cb, cjj) of workgroup w owns unit cj = w*JW + cjj of batch row cb: dh
// and dc of that unit stay in its registers from step to step.
template <int NB, int JW>
__global__ __launch_bounds__(256) void lstm_persist_bwd_kernel(PersistBwd a) {
    extern __shared__ float smem[];
    constexpr int GL = RowGroup<NB>::GL, NI = RowGroup<NB>::NI;
    const int H = a.H, G = 4 * H, B = a.B, nwg = a.nwg;
    float* Wt = smem;                   // [JW][G]   rows of Wh of the owned units
    float* dl = Wt + JW * G;            // [NB][G]   dHW_s, all columns
    float* rp = dl + NB * G;            // [NB][JW][4] per-thread LayerNorm-adjoint partials
    float* rtot = rp + NB * JW * 4;     // [NB][4]   the four row sums / G
    float* gp = rtot + NB * 4;          // [4][NB*JW] per-wave partials of dh_prev
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.x * JW;
    const size_t myrep = (size_t)(blockIdx.x % a.nrep);
    const int nvalid = (H - j0) < JW ? (H - j0) : JW;
    for (int e = tid; e < JW * G; e += 256) {
        const int jj = e / G;
        Wt[e] = (jj < nvalid) ? a.wh[(size_t)(j0 + jj) * G + (e - jj * G)] : 0.f;
    }
    for (int e = tid; e < NB * G; e += 256) dl[e] = 0.f;
    const int cb = tid / JW, cjj = tid % JW, cj = j0 + cjj;
    const bool cell = tid < NB * JW && cb < B && cjj < nvalid;
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f};
    float dh_carry = 0.f, dc_carry = 0.f;
    if (cell) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            gx[gg] = a.gamma[gg * H + cj];
            gh[gg] = a.gamma[G + gg * H + cj];
        }
        if (a.dhn) dh_carry = a.dhn[(size_t)cb * H + cj];
        if (a.dcn) dc_carry = a.dcn[(size_t)cb * H + cj];
    }
    const float inv_g = 1.f / (float)G;
    const int rb = tid / GL, rpart = tid % GL;
    const bool rgrp = rb < B;
    // Saved tensors of one step.  vmcnt retires in order, so a poll's wait also waits for every plain load/store
    // issued before it: the loads for step s-1 and the dHW/dXW/dgate stores of step s are issued right after the
    // second poll of step s has completed and overlap the dh_prev product.
    float sg[4], sc_new = 0.f, sc_prev = 0.f, sx[4], sh[4], sst[4], sdo = 0.f;
    float o_dhw[4] = {0.f, 0.f, 0.f, 0.f}, o_dxw[4] = {0.f, 0.f, 0.f, 0.f}, o_da[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int gg = 0; gg < 4; ++gg) sg[gg] = sx[gg] = sh[gg] = sst[gg] = 0.f;
    auto load_saved = [&](int s) {
        if (!cell) return;
        const size_t row = (size_t)s * B + cb;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            sg[gg] = a.gates[row * G + gg * H + cj];
            sx[gg] = a.xw[row * G + gg * H + cj];
            sh[gg] = a.hw[row * G + gg * H + cj];
            sst[gg] = a.stats[row * 4 + gg];
        }
        sc_new = a.c[row * H + cj];
        sc_prev = s == 0 ? a.c0[(size_t)cb * H + cj] : a.c[(row - B) * H + cj];
        sdo = a.d_out ? a.d_out[row * H + cj] : 0.f;
    };
    if (a.S > 0) load_saved(a.S - 1);
    __syncthreads();

    u64 tprev_ = a.prof ? wall_clock64() : 0;
    for (int s = a.S - 1; s >= 0; --s) {
        const uint32_t tag = a.tag_base + (uint32_t)(a.S - 1 - s) + 1u;
        const int par = (int)(tag & 1u);
        // ---- gate adjoints of the owned units and their LayerNorm-adjoint partial sums
        float da[4] = {0.f, 0.f, 0.f, 0.f}, xh[4], hh[4];
        if (cell) {
            const float ig = sg[0], fg = sg[1], og = sg[2], ug = sg[3];
            const float dh = sdo + dh_carry;
            const float tc = tanhf(sc_new);
            const float dc = dc_carry + dh * og * (1.f - tc * tc);
            da[0] = dc * ug * ig * (1.f - ig);
            da[1] = dc * sc_prev * fg * (1.f - fg);
            da[2] = dh * tc * og * (1.f - og);
            da[3] = dc * ig * (1.f - ug * ug);
            dc_carry = dc * fg;
            float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                xh[gg] = (sx[gg] - sst[0]) * sst[1];
                hh[gg] = (sh[gg] - sst[2]) * sst[3];
                const float dyx = da[gg] * gx[gg], dyh = da[gg] * gh[gg];
                r[0] += dyx; r[1] += dyx * xh[gg];
                r[2] += dyh; r[3] += dyh * hh[gg];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rp[(cb * JW + cjj) * 4 + q] = r[q];
        }
        __syncthreads();
        if (tid < 4 * B) {   // thread (b, q): sum over the owned units, publish
            const int b = tid >> 2, q = tid & 3;
            float t = 0.f;
            for (int jj = 0; jj < nvalid; ++jj) t += rp[(b * JW + jj) * 4 + q];
            xchg_put_all(a.sums + (size_t)par * a.sums_par, a.sums_rep, a.nrep, (b * 4 + q) * nwg + blockIdx.x, t, tag);
        }
        HPC_RLL_TICK(0)
        {   // row b is summed by its lane group
            int idx[4 * NI];
            unsigned valid = 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int w = rpart + GL * i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    idx[q * NI + i] = (rb * 4 + q) * nwg + w;
                    if (rgrp && w < nwg) valid |= 1u << (q * NI + i);
                }
            }
            float v[4 * NI];
            xchg_get<4 * NI>(a.sums + (size_t)par * a.sums_par + myrep * a.sums_rep, idx, valid, tag, v,
                             a.prof ? a.prof + 6 : nullptr);
            HPC_RLL_TICK(1)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) t += ((valid >> (q * NI + i)) & 1u) ? v[q * NI + i] : 0.f;
                t = group_sum_last<GL>(t);
                if (rgrp && rpart == GL - 1) rtot[rb * 4 + q] = t * inv_g;
            }
        }
        __syncthreads();
        HPC_RLL_TICK(2)
        // ---- dXW, dHW of the owned columns; publish dHW
        if (cell) {
            const float r0 = rtot[cb * 4], r1 = rtot[cb * 4 + 1], r2 = rtot[cb * 4 + 2], r3 = rtot[cb * 4 + 3];
            u64* dst = a.big + (size_t)par * a.big_par;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int col = gg * H + cj;
                const float dyx = da[gg] * gx[gg], dyh = da[gg] * gh[gg];
                o_dhw[gg] = sst[3] * (dyh - r2 - hh[gg] * r3);
                o_dxw[gg] = sst[1] * (dyx - r0 - xh[gg] * r1);
                o_da[gg] = da[gg];
                xchg_put_all(dst, a.big_rep, a.nrep, cb * G + col, o_dhw[gg], tag);
            }
        }
        HPC_RLL_TICK(3)
        // ---- all of dHW_s -> LDS
        xchg_gather<24>(a.big + (size_t)par * a.big_par + myrep * a.big_rep, B * G, tag, dl,
                        a.prof ? a.prof + 7 : nullptr);
        __syncthreads();
        if (cell) {
            const size_t row = (size_t)s * B + cb;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int col = gg * H + cj;
                a.dhw[row * G + col] = o_dhw[gg];
                a.dxw[row * G + col] = o_dxw[gg];
                a.dgate[row * G + col] = o_da[gg];
            }
        }
        if (s > 0) load_saved(s - 1);
        HPC_RLL_TICK(4)
        // ---- dh_prev of the owned units = dHW_s @ Wh^T (rows of Wh in LDS): all 256 threads split the 4H columns
        {
            float acc[NB][JW];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) acc[b][jj] = 0.f;
#pragma unroll 2
            for (int c = tid; c < G; c += 256) {
                float wv_[JW];
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) wv_[jj] = Wt[jj * G + c];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float dv = dl[b * G + c];
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) acc[b][jj] = fmaf(dv, wv_[jj], acc[b][jj]);
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    const float t = wave_sum_last(acc[b][jj]);
                    if (lane == 63) gp[wv * NB * JW + b * JW + jj] = t;
                }
        }
        __syncthreads();
        if (cell) {
            const int o = cb * JW + cjj;
            dh_carry = (gp[o] + gp[NB * JW + o]) + (gp[2 * NB * JW + o] + gp[3 * NB * JW + o]);
        }
        HPC_RLL_TICK(5)
    }
    if (cell) {
        a.dh0[(size_t)cb * H + cj] = dh_carry;
        a.dc0[(size_t)cb * H + cj] = dc_carry;
    }
}

// ---- host side of the residency protocol ----------------------------------------------------------------------
constexpr int kMaxDevices = 64;
struct PersistRuntime {
    std::mutex mu;
    unsigned* host_status = nullptr;      // hipHostMalloc'ed (mapped, portable): written by a wave that gave up
    bool dev_ready[kMaxDevices] = {};     // g_persist_host_status set on that device
    int cus[kMaxDevices] = {};
    hipEvent_t chain[kMaxDevices] = {};   // last persistent launch on the device
    bool chain_armed[kMaxDevices] = {};
    std::map<std::tuple<int, const void*, size_t>, int> occupancy;   // (device, kernel, lds) -> blocks per CU
    bool disabled = false;                // after a timeout was reported and cleared
};
inline PersistRuntime& persist_rt() {
    static PersistRuntime* r = new PersistRuntime();
    return *r;
}
inline int persist_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return -1;
    return dev;
}
// CU count of the CURRENT device (cached per device)
inline int persist_cu_count() {
    const int dev = persist_device();
    if (dev < 0) return 0;
    PersistRuntime& r = persist_rt();
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.cus[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 0;
        r.cus[dev] = v > 0 ? v : -1;
    }
    return r.cus[dev] > 0 ? r.cus[dev] : 0;
}
// sticky asynchronous status: nonzero once any persistent kernel of this process gave up waiting
inline int persist_async_status() {
    PersistRuntime& r = persist_rt();
    return (r.host_status && *(volatile unsigned*)r.host_status) ? HPC_RLL_ETIMEOUT : HPC_RLL_OK;
}
// May a persistent kernel be launched on `st` now?  Sets up the pinned status word / the device symbol on first use
// (a synchronous copy: not possible while `st` is being captured into a graph -> the step kernels run that time).
inline bool persist_runtime_ready(hipStream_t st) {
    PersistRuntime& r = persist_rt();
    if (r.disabled || persist_async_status()) return false;
    const int dev = persist_device();
    if (dev < 0) return false;
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.dev_ready[dev]) return true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) return false;
    if (cs != hipStreamCaptureStatusNone) return false;   // (the one-time set-up below is a synchronous copy)
    if (!r.host_status) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return false; }
        r.host_status = (unsigned*)p;
        *r.host_status = 0u;
    }
    void* dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, r.host_status, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_persist_host_status), &dptr, sizeof(dptr)) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (hipEventCreateWithFlags(&r.chain[dev], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
    r.dev_ready[dev] = true;
    return true;
}
// Does the runtime itself say that `blocks` workgroups of kernel `k` (256 threads, `lds` bytes) fit the device at once?
template <class K> inline bool persist_resident_t(K k, int threads, int blocks, size_t lds);
template <class K> inline bool persist_resident(K k, int blocks, size_t lds) { return persist_resident_t(k, 256, blocks, lds); }
template <class K> inline bool persist_resident_t(K k, int threads, int blocks, size_t lds) {
    const int dev = persist_device();
    const int cus = persist_cu_count();
    if (dev < 0 || cus <= 0) return false;
    PersistRuntime& r = persist_rt();
    const auto key = std::make_tuple(dev, (const void*)k, lds);
    {
        std::lock_guard<std::mutex> lk(r.mu);
        auto it = r.occupancy.find(key);
        if (it != r.occupancy.end()) return (long)it->second * cus >= blocks;
    }
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, lds) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    std::lock_guard<std::mutex> lk(r.mu);
    r.occupancy[key] = per_cu;
    return (long)per_cu * cus >= blocks;
}
// Persistent launches of this process run one after the other on a device, whatever their streams.
inline void persist_chain_before(hipStream_t st) {
    const int dev = persist_device();
    PersistRuntime& r = persist_rt();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (dev < 0 || hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.chain_armed[dev]) (void)hipStreamWaitEvent(st, r.chain[dev], 0);
}
inline void persist_chain_after(hipStream_t st) {
    const int dev = persist_device();
    PersistRuntime& r = persist_rt();
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (dev < 0 || hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.dev_ready[dev] && hipEventRecord(r.chain[dev], st) == hipSuccess) r.chain_armed[dev] = true;
}
template <class K, class A> inline int persist_launch(K k, dim3 grid, size_t lds, const A& a, hipStream_t st) {
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    persist_chain_before(st);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    persist_chain_after(st);
    return 0;
}

// HPC_RLL_LSTM_PROFILE=1: per-phase time of workgroup 0, printed after every layer (synchronises; debugging only)
static u64* g_prof_buf = nullptr;
inline u64* persist_prof_peek() { return g_prof_buf; }
inline u64* persist_prof() {
    u64*& buf = g_prof_buf;
    static int on = -1;
    if (on < 0) {
        const char* e = getenv("HPC_RLL_LSTM_PROFILE");
        on = (e && e[0] == '1') ? 1 : 0;
        if (on && hipMalloc((void**)&buf, 8 * sizeof(u64)) != hipSuccess) on = 0;
    }
    if (on) (void)hipMemset(buf, 0, 8 * sizeof(u64));
    return on ? buf : nullptr;
}
inline void persist_prof_report(const char* what, int layer, int S, hipStream_t st) {
    u64* buf = persist_prof_peek();
    if (!buf) return;
    u64 h[8];
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[lstm persist %s] layer %d, us/step by phase:", what, layer);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %.2f", (double)h[i] / 100.0 / (S > 0 ? S : 1));
    fprintf(stderr, "\n");
}

struct PersistCfg { int nb, jw, nwg; size_t lds; };

// is the persistent path applicable?  (forward and backward use the same LDS budget: 4*JW*H weights + a row buffer)
inline bool persist_cfg(int B, int H, int row_floats /* per batch row staged in LDS */, PersistCfg* out) {
    // measured (tests/tools/lstm_small_probe.py, L=1, S=64): B <= 4 wins at every H (H=512: 0.68/1.02 vs 0.88/1.16 ms,
    // H=1024: 0.75/1.24 vs 1.24/1.92); at B = 8 the exchanged volume (B*H and B*4H words per workgroup and step) makes it
    // a tie or a loss against the split-K step kernels (H=256: 0.97/1.23 vs 0.74/0.89)
    if (!g_lstm_persist || B < 1 || B > g_lstm_persist_max_b || H < 1 || H > 1024) return false;
    PersistCfg c;
    c.nb = B <= 1 ? 1 : B <= 2 ? 2 : 4;
    c.jw = H <= 256 ? 1 : H <= 512 ? 2 : 4;
    c.nwg = (H + c.jw - 1) / c.jw;
    c.lds = ((size_t)4 * c.jw * H + (size_t)c.nb * row_floats + (size_t)c.nb * 8 * c.jw + 64 * c.nb + 64) * sizeof(float);
    if (c.lds > 144 * 1024 || c.nwg > 256) return false;   // residency itself: persist_fwd_ok / persist_bwd_ok
    *out = c;
    return true;
}

template <int NB, int JW>
inline int launch_persist_fwd_t(const PersistCfg& c, const PersistFwd& a, hipStream_t st) {
    return persist_launch(lstm_persist_fwd_kernel<NB, JW>, dim3(c.nwg), c.lds, a, st);
}
template <int NB, int JW>
inline int resident_persist_fwd_t(const PersistCfg& c, const PersistFwd&, hipStream_t) {
    return persist_resident(lstm_persist_fwd_kernel<NB, JW>, c.nwg, c.lds) ? 1 : 0;
}

#define HPC_RLL_PERSIST_DISPATCH(FN, c, a, st)                                         \
    do {                                                                               \
        if (c.jw == 1) {                                                               \
            if (c.nb == 1) return FN<1, 1>(c, a, st);                                  \
            if (c.nb == 2) return FN<2, 1>(c, a, st);                                  \
            return FN<4, 1>(c, a, st);                                                 \
        } else if (c.jw == 2) {                                                        \
            if (c.nb == 1) return FN<1, 2>(c, a, st);                                  \
            if (c.nb == 2) return FN<2, 2>(c, a, st);                                  \
            return FN<4, 2>(c, a, st);                                                 \
        } else {                                                                       \
            if (c.nb == 1) return FN<1, 4>(c, a, st);                                  \
            if (c.nb == 2) return FN<2, 4>(c, a, st);                                  \
            return FN<4, 4>(c, a, st);                                                 \
        }                                                                              \
    } while (0)

inline int launch_persist_fwd(const PersistCfg& c, const PersistFwd& a, hipStream_t st) {
    HPC_RLL_PERSIST_DISPATCH(launch_persist_fwd_t, c, a, st);
}

template <int NB, int JW>
inline int launch_persist_bwd_t(const PersistCfg& c, const PersistBwd& a, hipStream_t st) {
    return persist_launch(lstm_persist_bwd_kernel<NB, JW>, dim3(c.nwg), c.lds, a, st);
}
inline int launch_persist_bwd(const PersistCfg& c, const PersistBwd& a, hipStream_t st) {
    HPC_RLL_PERSIST_DISPATCH(launch_persist_bwd_t, c, a, st);
}
template <int NB, int JW>
inline int resident_persist_bwd_t(const PersistCfg& c, const PersistBwd&, hipStream_t) {
    return persist_resident(lstm_persist_bwd_kernel<NB, JW>, c.nwg, c.lds) ? 1 : 0;
}
// The persistent kernels may run for this shape NOW: runtime set up, no pending async failure, and the runtime's own
// occupancy figure covers the grid.
inline int persist_fwd_resident_i(const PersistCfg& c, hipStream_t st) {
    const PersistFwd a{};
    HPC_RLL_PERSIST_DISPATCH(resident_persist_fwd_t, c, a, st);
}
inline int persist_bwd_resident_i(const PersistCfg& c, hipStream_t st) {
    const PersistBwd a{};
    HPC_RLL_PERSIST_DISPATCH(resident_persist_bwd_t, c, a, st);
}
inline bool persist_fwd_ok(const PersistCfg& c, hipStream_t st) {
    return persist_runtime_ready(st) && persist_fwd_resident_i(c, st) == 1;
}
inline bool persist_bwd_ok(const PersistCfg& c, hipStream_t st) {
    return persist_runtime_ready(st) && persist_bwd_resident_i(c, st) == 1;
}

// Exchange buffers in the workspace: a "big" region (forward: h, B*H words; backward: dHW, B*4H words) and a "sums"
// region (<= 256 workgroups x NB x 4 words), each kMaxRep replicas x two parities.
struct XchgLayout { size_t big_rep, sums_rep, big_par, sums_par, total_words; };
inline XchgLayout xchg_layout(int B, int H) {
    XchgLayout x{0, 0, 0, 0, 0};
    if (B < 1 || B > kPersistMaxB || H < 1 || H > 1024) return x;
    const int nb = B <= 1 ? 1 : B <= 2 ? 2 : 4;
    x.big_rep = ((size_t)nb * 4 * H + 15) / 16 * 16;
    x.sums_rep = (size_t)256 * nb * 4;
    x.big_par = kMaxRep * x.big_rep;
    x.sums_par = kMaxRep * x.sums_rep;
    x.total_words = 2 * (x.big_par + x.sums_par);
    return x;
}

}  // namespace
}  // namespace hpc_rll
