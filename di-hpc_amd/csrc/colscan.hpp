// colscan.hpp -- generic reverse-time column scan for (T,B)-layout return ops on gfx950.
//
// All trajectory "return" recurrences on the hot path (TD-lambda, V-trace, UPGO; GAE has its own
// specialised kernels in gae.hip) are first-order affine maps walked backwards in time,
//       s_t = b_t + a_t * s_{t+1},          t = T-1 .. 0,   s_T = init(column)
// with per-element (a_t, b_t) computed from the row's inputs.  Affine maps compose associatively, so a
// chunk [t0,t1) can be scanned from a zero carry (L_t, and the running product P_t of the a's) and
// repaired once the true s_{t1} is known:  s_t = L_t + P_t * s_{t1}.
//
// Mapping (same skeleton as gae.hip): lane <-> V consecutive columns (coalesced along B), wave <-> LC
// consecutive steps held in VGPRs, workgroup = NW waves covering NW*LC steps; chunk heads (L,P per lane) go
// through a double-buffered LDS slot, one barrier per NW*LC steps.  Each workgroup also reduces NACC scalar
// sums (loss terms) deterministically: lane sums -> wave butterfly -> LDS -> one partial per workgroup;
// the last workgroup to finish adds the partials in a fixed order (fp64).  No float atomics anywhere
// (the reference uses cross-block atomicAdd: td_lambda_kernel.h:38, vtrace_kernel.h:215-222).
//
// The reference walks each column with one thread (td_lambda_kernel.h:17-32, vtrace_kernel.h:161-180,
// upgo_kernel.h:17-36) -- no time parallelism at all.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

#include "wave.hpp"

namespace hpc_rll {

// Round 2: (a) rows t and t+1 of a wave's chunk share value[t+1] (UPGO also reward[t+1], value[t+2]): `load` is told
// whether row t+1 is held by the same wave and `link` copies the shared fields from it -- LC+1 value rows per chunk
// instead of 2*LC; (b) the fixed-order fp64 sum of the per-workgroup partials runs in the LAST workgroup to finish
// (ticket = a {tag, count} word in the call's own scratch, CAS-counted, self-resetting so that a captured graph can be
// replayed) instead of a second launch -- one launch per scan.
//
// An Op provides:
//   static constexpr int NACC;                         number of scalar sums
//   template<int V> struct Row;                        per-row register payload
//   template<int V> void init(long col, bool ok, float (&carry)[V]) const;          s_T
//   template<int V> void load(Row<V>&, int t, long col, bool ok, bool next_in_regs) const;   issue the row's loads
//   template<int V> void link(Row<V>& row, const Row<V>& next_row) const;           fields shared with row t+1
//   template<int V> void coeffs(const Row<V>&, int t, float (&a)[V], float (&b)[V]) const;
//   template<int V> void finish(const Row<V>&, int t, long col, bool ok, const float (&s)[V],
//                               const float (&s_next)[V], float (&acc)[NACC]) const;   outputs + sums
struct ScanFin {          // fold the final sum into the scan launch: out[k] = scale[k] * sum of the workgroup partials
    float* out;           // nullptr: leave the partials to the caller
    float scale[4];
    uint32_t tag;         // unique per launch, never 0
};

template <class Op, int V, int LC, int NW>
__global__ __launch_bounds__(NW * 64) void colscan_rev_kernel(const Op op, int T, int B,
                                                              float* __restrict__ partials, const ScanFin fin) {
    constexpr int TILE = 64 * V;
    constexpr int NACC = Op::NACC;
    __shared__ float lds[4 * NW * TILE + NW * (NACC > 0 ? NACC : 1)];
    float* const s_l0 = lds;                      // [buf][wave][TILE]
    float* const s_p0 = lds + 2 * NW * TILE;      // [buf][wave][TILE]
    float* const s_red = lds + 4 * NW * TILE;     // [NACC][NW]

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long col = (long)blockIdx.x * TILE + (long)lane * V;
    const bool ok = col < (long)B;

    float carry[V];
    op.template init<V>(col, ok, carry);
    float acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int k = 0; k < (NACC > 0 ? NACC : 1); ++k) acc[k] = 0.f;

    constexpr int SPAN = NW * LC;
    const int n_iter = (T + SPAN - 1) / SPAN;
    for (int it = 0; it < n_iter; ++it) {
        const int t1 = T - (it * NW + (NW - 1 - w)) * LC;
        const int t0 = t1 - LC;
        const int buf = it & 1;

        typename Op::template Row<V> rows[LC];
        float L[LC][V], P[LC][V];
#pragma unroll
        for (int j = LC - 1; j >= 0; --j)
            if (t0 + j >= 0) op.template load<V>(rows[j], t0 + j, col, ok, j < LC - 1);
#pragma unroll
        for (int j = LC - 2; j >= 0; --j)
            if (t0 + j >= 0) op.template link<V>(rows[j], rows[j + 1]);
        {
            float a[V], p[V];
#pragma unroll
            for (int k = 0; k < V; ++k) { a[k] = 0.f; p[k] = 1.f; }
#pragma unroll
            for (int j = LC - 1; j >= 0; --j) {
                if (t0 + j >= 0) {
                    float ca[V], cb[V];
                    op.template coeffs<V>(rows[j], t0 + j, ca, cb);
#pragma unroll
                    for (int k = 0; k < V; ++k) {
                        a[k] = fmaf(ca[k], a[k], cb[k]);
                        p[k] *= ca[k];
                    }
                }
#pragma unroll
                for (int k = 0; k < V; ++k) { L[j][k] = a[k]; P[j][k] = p[k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s_l0[(buf * NW + w) * TILE + lane * V + k] = L[0][k];
            s_p0[(buf * NW + w) * TILE + lane * V + k] = P[0][k];
        }
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
#pragma unroll
            for (int k = 0; k < V; ++k)
                A[k] = fmaf(s_p0[(buf * NW + u) * TILE + lane * V + k], A[k],
                            s_l0[(buf * NW + u) * TILE + lane * V + k]);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = A[k];

        float s_next[V];
#pragma unroll
        for (int k = 0; k < V; ++k) s_next[k] = Aw[k];
#pragma unroll
        for (int j = LC - 1; j >= 0; --j) {
            if (t0 + j >= 0) {
                float s[V];
#pragma unroll
                for (int k = 0; k < V; ++k) s[k] = fmaf(P[j][k], Aw[k], L[j][k]);
                op.template finish<V>(rows[j], t0 + j, col, ok, s, s_next, acc);
#pragma unroll
                for (int k = 0; k < V; ++k) s_next[k] = s[k];
            }
        }
    }

    if (NACC > 0) {
        // deterministic workgroup reduction of the NACC running sums
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            const float sum = wave_sum(acc[k]);
            if (lane == 0) s_red[k * NW + w] = sum;
        }
        __syncthreads();
        if (threadIdx.x < NACC) {
            float sum = 0.f;
            for (int i = 0; i < NW; ++i) sum += s_red[threadIdx.x * NW + i];
            partials[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = sum;
        }
        if (fin.out) {
            typedef unsigned long long u64;
            __shared__ int s_last;
            __shared__ double s_sum[NW];
            u64* const ticket = reinterpret_cast<u64*>(
                (reinterpret_cast<uintptr_t>(partials + (size_t)NACC * gridDim.x) + 7) & ~(uintptr_t)7);
            __threadfence();   // release this workgroup's partials
            __syncthreads();
            if (threadIdx.x == 0) {
                u64 old = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned count;
                while (true) {   // the word is garbage until the first arrival stamps it with this launch's tag
                    count = ((uint32_t)(old >> 32) == fin.tag) ? (uint32_t)old + 1u : 1u;
                    const u64 want = ((u64)fin.tag << 32) | count;
                    if (__hip_atomic_compare_exchange_strong(ticket, &old, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT))
                        break;
                }
                s_last = count == gridDim.x;
            }
            __syncthreads();
            if (s_last) {   // every partial has been released: add them in a fixed order, fp64
                __threadfence();
                for (int k = 0; k < NACC; ++k) {
                    double sum = 0.0;
                    for (unsigned i = threadIdx.x; i < gridDim.x; i += NW * 64)
                        sum += (double)__hip_atomic_load(partials + (size_t)k * gridDim.x + i, __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m, 64);
                    if (lane == 0) s_sum[w] = sum;
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        double tot = 0.0;
                        for (int i = 0; i < NW; ++i) tot += s_sum[i];
                        fin.out[k] = (float)(tot * (double)fin.scale[k]);
                    }
                    __syncthreads();
                }
                if (threadIdx.x == 0)   // tag 0 is never issued: a replay of a captured launch starts from a clean word
                    __hip_atomic_store(ticket, (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

struct ScanCfg { int v, lc, nw; };

// V=1 or 2 (the fat rows of V-trace/UPGO do not fit V=4), LC = 8, NW up to 16 for small B.
inline ScanCfg scan_cfg(int T, int B, bool can_v2) {
    ScanCfg c;
    c.v = (can_v2 && (B + 127) / 128 >= 512) ? 2 : 1;
    c.lc = 8;
    const int wgs = (B + 64 * c.v - 1) / (64 * c.v);
    const int chunks = (T + c.lc - 1) / c.lc;
    c.nw = 4;
    while (c.nw < 16 && wgs * c.nw < 2048) c.nw <<= 1;
    while (c.nw > 1 && c.nw > chunks) c.nw >>= 1;
    return c;
}

// launch tag for the folded final sum: unique per launch in this process, never 0
inline uint32_t scan_next_tag() {
    static std::atomic<uint32_t> t{1};
    uint32_t v = t.fetch_add(1);
    if (v == 0) v = t.fetch_add(1);
    return v;
}

// `out` (device, NACC floats) != nullptr: the last workgroup also writes out[k] = scale[k] * sum of partials
template <class Op, bool ALLOW_V2 = true>
inline void launch_colscan(const Op& op, const ScanCfg& c, int T, int B, float* partials, hipStream_t st,
                           float* out = nullptr, const float* scale = nullptr) {
    const unsigned grid = (unsigned)((B + 64 * c.v - 1) / (64 * c.v));
    ScanFin fin{out, {0.f, 0.f, 0.f, 0.f}, 0u};
    if (out) {
        for (int k = 0; k < Op::NACC && k < 4; ++k) fin.scale[k] = scale[k];
        fin.tag = scan_next_tag();
    }
#define HPC_RLL_SCAN_CASE(V_, NW_)                                                                          \
    if (c.v == V_ && c.nw == NW_) {                                                                         \
        hipLaunchKernelGGL((colscan_rev_kernel<Op, V_, 8, NW_>), dim3(grid), dim3(NW_ * 64), 0, st, op, T, B, \
                           partials, fin);                                                                  \
        return;                                                                                             \
    }
    HPC_RLL_SCAN_CASE(1, 1) HPC_RLL_SCAN_CASE(1, 2) HPC_RLL_SCAN_CASE(1, 4) HPC_RLL_SCAN_CASE(1, 8)
    HPC_RLL_SCAN_CASE(1, 16)
    if constexpr (ALLOW_V2) {
        HPC_RLL_SCAN_CASE(2, 1) HPC_RLL_SCAN_CASE(2, 2) HPC_RLL_SCAN_CASE(2, 4) HPC_RLL_SCAN_CASE(2, 8)
        HPC_RLL_SCAN_CASE(2, 16)
    }
#undef HPC_RLL_SCAN_CASE
}

inline int scan_num_blocks(int T, int B, bool can_v2) {
    const ScanCfg c = scan_cfg(T, B, can_v2);
    return (B + 64 * c.v - 1) / (64 * c.v);
}

// reduce.hip
int finalize_sums(const float* partials, int nblocks, int nacc, const float* scales /*host, nacc*/, float* out,
                  hipStream_t st);
int scale_rows(const float* g, const float* in, float* out, long n_in, long n_out, hipStream_t st);

}  // namespace hpc_rll
