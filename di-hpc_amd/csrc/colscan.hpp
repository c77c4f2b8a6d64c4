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
// a second tiny kernel (reduce.hip) adds the partials in a fixed order.  No float atomics anywhere
// (the reference uses cross-block atomicAdd: td_lambda_kernel.h:38, vtrace_kernel.h:215-222).
//
// The reference walks each column with one thread (td_lambda_kernel.h:17-32, vtrace_kernel.h:161-180,
// upgo_kernel.h:17-36) -- no time parallelism at all.
#pragma once
#include <hip/hip_runtime.h>

#include "wave.hpp"

namespace hpc_rll {

// Round 2: rows t and t+1 of a wave's chunk share value[t+1] (UPGO also reward[t+1], value[t+2]): `load` is told
// whether row t+1 is held by the same wave and `link` copies the shared fields from it -- LC+1 value rows per chunk
// instead of 2*LC.
// The fixed-order sum of the workgroup partials is folded into the LAST workgroup to finish (ScanFold): one launch per
// forward instead of scan + finalize (a dependent kernel boundary costs ~1.7 us here, MI355X_MICROARCH.md, the
// finalize kernel itself ~3).  First attempt, NOT kept: arrival ticket behind `__threadfence()` -- on this multi-XCD
// part an agent-scope RELEASE writes the XCD's whole L2 back, and the scan has just left its output dirty there:
// TD-lambda forward at T=256,B=16384 went from 17.9 us to 319 us, V-trace 0.75 -> 1.01 ms.  What runs now needs no
// release: the partial is written with a relaxed AGENT-scope atomic store (write-through to the memory side), the
// thread waits for that one store (`s_waitcnt vmcnt(0)`), the workgroup barriers, and only then takes the ticket with a
// relaxed agent-scope atomic; the last workgroup reads the partials with agent-scope atomic loads (which bypass its
// XCD's L2).  Nothing else of the workgroup's output has to be visible to another workgroup.
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
// SUB (V = 1 only): sub-wave tiles for narrow batches -- a wave's 64 lanes are SUB groups of 64/SUB columns and the
// groups own SUB DIFFERENT chunks of the time axis ("virtual waves", as gae.hip's half-wave tiles): SUB x more
// workgroups and SUB x more steps per barrier.  At the reference's TD-lambda test shape (T=1024, B=64) the 64-column
// tiling is ONE workgroup walking 8 barriers.
// Where the last workgroup leaves the NACC sums (x scale[k]); out == nullptr: partials only (the caller finalises).
// `ticket` is zero before the launch and is left at zero by it.
struct ScanFold { float* out; unsigned* ticket; float scale[8]; };

// scan_ops.hip: the fold of a launch on `st` (its ticket, see there), or a partials-only fold when none is available
// `grid`: workgroups of the launch.  Above kFoldMaxGrid the fold is NOT used (partials only, the caller runs the separate
// finalize launch): every workgroup takes the ticket with an agent-scope atomic on ONE address, and those serialise at
// ~12 ns each on this part -- 32768 workgroups (QR-DQN at B = 262144) spent 0.41 ms of a 0.04 ms kernel there
// (tests/tools/r03_td_ab.py).  (Round 4 also tried a two-level counter tree for the large grids: neutral against the finalize
// launch, profiles/r04_fold_tree.txt; removed in round 5.)
constexpr long kFoldMaxGrid = 512;
ScanFold make_fold(hipStream_t st, int nacc, const float* scale, float* out, long grid = 0);

// Workgroup epilogue shared by every kernel that ends in NACC loss sums: thread k < NACC holds the workgroup's sum k in
// `sum`.  Stores partials[k][blockIdx.x]; with a fold the last workgroup of the grid to arrive adds all partials in a
// fixed order (fp64) and writes out[k] = total * scale[k].  Called by ALL NT threads (it barriers).
template <int NACC, int NT>
__device__ __forceinline__ void publish_sums(float sum, float* __restrict__ partials, const ScanFold& fold) {
    if (threadIdx.x < NACC) {
        float* slot = partials + (size_t)threadIdx.x * gridDim.x + blockIdx.x;
        if (fold.out) {
            __hip_atomic_store(slot, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this store has reached the memory side
        } else {
            *slot = sum;
        }
    }
    if (!fold.out) return;
    constexpr int NWAVES = NT / 64;
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, wr = threadIdx.x >> 6;
    __syncthreads();   // every partial of this workgroup is out
    if (threadIdx.x == 0) {
        s_last = __hip_atomic_fetch_add(fold.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;   // uniform: the last workgroup to arrive adds all partials
    // Round 4: ALL NACC * ceil(grid / NT) partials of a thread are requested before the first use, the NACC sums share one
    // barrier -- the first version walked the sums one after the other (a memory round trip and two barriers each: ~7 us of
    // serial tail for PPO's five sums, more than its kernels' work at B = 65536).  Same additions in the same order.
    constexpr int MAXI = (int)((kFoldMaxGrid + NT - 1) / NT);
    float v[NACC][MAXI];
#pragma unroll
    for (int k = 0; k < NACC; ++k)
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const unsigned i = threadIdx.x + j * NT;
            v[k][j] = i < gridDim.x ? __hip_atomic_load(partials + (size_t)k * gridDim.x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                    : 0.f;
        }
    __shared__ double s_fin2[NACC][NWAVES];
#pragma unroll
    for (int k = 0; k < NACC; ++k) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < MAXI; ++j)
            if (threadIdx.x + j * NT < gridDim.x) s += (double)v[k][j];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (lane == 0) s_fin2[k][wr] = s;
    }
    __syncthreads();
    if (threadIdx.x < NACC) {
        double tot = 0.0;
        for (int i = 0; i < NWAVES; ++i) tot += s_fin2[threadIdx.x][i];
        fold.out[threadIdx.x] = (float)(tot * (double)fold.scale[threadIdx.x]);
    }
    if (threadIdx.x == 0) __hip_atomic_store(fold.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class Op, int V, int LC, int NW, int SUB = 1>
__global__ __launch_bounds__(NW * 64) void colscan_rev_kernel(const Op op, int T, int B,
                                                              float* __restrict__ partials, const ScanFold fold) {
    static_assert(SUB == 1 || V == 1, "sub-wave tiles hold one column per lane");
    constexpr int NWV = NW * SUB;                 // virtual waves per workgroup
    constexpr int TILE = 64 * V / SUB;
    constexpr int NACC = Op::NACC;
    __shared__ float lds[4 * NWV * TILE + NW * (NACC > 0 ? NACC : 1)];
    float* const s_l0 = lds;                      // [buf][virtual wave][TILE]
    float* const s_p0 = lds + 2 * NWV * TILE;     // [buf][virtual wave][TILE]
    float* const s_red = lds + 4 * NWV * TILE;    // [NACC][NW]

    const int lane = threadIdx.x & 63;
    const int wr = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int cl = SUB == 1 ? lane : (lane & (TILE - 1));          // column lane inside the tile
    const int w = SUB == 1 ? wr : SUB * wr + lane / TILE;         // (virtual) wave: lane dependent when SUB > 1
    const long col = (long)blockIdx.x * TILE + (long)cl * V;
    const bool ok = col < (long)B;

    float carry[V];
    op.template init<V>(col, ok, carry);
    float acc[NACC > 0 ? NACC : 1];
#pragma unroll
    for (int k = 0; k < (NACC > 0 ? NACC : 1); ++k) acc[k] = 0.f;

    constexpr int SPAN = NWV * LC;
    const int n_iter = (T + SPAN - 1) / SPAN;
    for (int it = 0; it < n_iter; ++it) {
        const int t1 = T - (it * NWV + (NWV - 1 - w)) * LC;
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
            s_l0[(buf * NWV + w) * TILE + cl * V + k] = L[0][k];
            s_p0[(buf * NWV + w) * TILE + cl * V + k] = P[0][k];
        }
        __syncthreads();

        float A[V], Aw[V];
#pragma unroll
        for (int k = 0; k < V; ++k) { A[k] = carry[k]; Aw[k] = 0.f; }
#pragma unroll
        for (int u = NWV - 1; u >= 0; --u) {
            if (u == w) {
#pragma unroll
                for (int k = 0; k < V; ++k) Aw[k] = A[k];
            }
#pragma unroll
            for (int k = 0; k < V; ++k)
                A[k] = fmaf(s_p0[(buf * NWV + u) * TILE + cl * V + k], A[k],
                            s_l0[(buf * NWV + u) * TILE + cl * V + k]);
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
            if (lane == 0) s_red[k * NW + wr] = sum;
        }
        __syncthreads();
        float sum = 0.f;
        if (threadIdx.x < NACC)
            for (int i = 0; i < NW; ++i) sum += s_red[threadIdx.x * NW + i];
        publish_sums<NACC, NW * 64>(sum, partials, fold);
    }
}

struct ScanCfg { int v, lc, nw, sub; };

// V=1 or 2 (the fat rows of V-trace/UPGO do not fit V=4), LC = 8, NW up to 16 for small B.
constexpr int g_scan_wave_target = 4096;   // waves a scan launch aims for (fills NW up to 16); in-process sweep (r02_scan_sweep.py): TD-lambda C3 16.9 -> 16.0 us vs 2048
// lc16 (round 5, TD-lambda only: its row payload is four values): SIXTEEN steps per wave where the 8-step form would walk a
// one-workgroup-per-CU grid through two or more dependent iterations (load round trip -> chunk heads through LDS -> barrier ->
// stores, each ~3 us): T = 256 at B = 16384 becomes ONE iteration of 16 waves x 16 steps.
inline ScanCfg scan_cfg(int T, int B, bool can_v2, bool lc16 = false) {
    ScanCfg c;
    c.v = (can_v2 && (B + 127) / 128 >= 512) ? 2 : 1;
    c.lc = 8;
    const int wgs = (B + 64 * c.v - 1) / (64 * c.v);
    const int chunks = (T + c.lc - 1) / c.lc;
    c.nw = 4;
    while (c.nw < 16 && wgs * c.nw < g_scan_wave_target) c.nw <<= 1;
    while (c.nw > 1 && c.nw > chunks) c.nw >>= 1;
    // narrow batches: sub-wave tiles (2, 4 or 8 groups per wave) while the grid is below one workgroup per CU and barriers
    // remain to be saved
    c.sub = 1;
    if (c.v == 1 && c.nw == 16)
        while (c.sub < 8 && (long)wgs * c.sub < 256 && chunks >= 2 * c.nw * c.sub) c.sub <<= 1;
    if (lc16 && c.v == 1 && c.sub == 1 && c.nw == 16 && wgs <= 512 && T > c.nw * c.lc) c.lc = 16;
    return c;
}
inline unsigned scan_grid(const ScanCfg& c, int B) {
    const int tile = 64 * c.v / c.sub;
    return (unsigned)((B + tile - 1) / tile);
}

template <class Op, bool ALLOW_V2 = true, bool ALLOW_LC16 = false>
inline void launch_colscan(const Op& op, const ScanCfg& c, int T, int B, float* partials, hipStream_t st,
                           const ScanFold& fold = ScanFold{nullptr, nullptr, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}) {
    const unsigned grid = scan_grid(c, B);
    if constexpr (ALLOW_LC16) {
        if (c.lc == 16 && c.v == 1 && c.nw == 16 && c.sub == 1) {
            hipLaunchKernelGGL((colscan_rev_kernel<Op, 1, 16, 16>), dim3(grid), dim3(1024), 0, st, op, T, B, partials, fold);
            return;
        }
    }
    if (c.sub == 2 && c.v == 1 && c.nw == 16) {
        hipLaunchKernelGGL((colscan_rev_kernel<Op, 1, 8, 16, 2>), dim3(grid), dim3(1024), 0, st, op, T, B, partials, fold);
        return;
    }
    if (c.sub == 4 && c.v == 1 && c.nw == 16) {
        hipLaunchKernelGGL((colscan_rev_kernel<Op, 1, 8, 16, 4>), dim3(grid), dim3(1024), 0, st, op, T, B, partials, fold);
        return;
    }
    if (c.sub == 8 && c.v == 1 && c.nw == 16) {   // 8-column tiles: 128 virtual waves = 1024 steps per barrier
        hipLaunchKernelGGL((colscan_rev_kernel<Op, 1, 8, 16, 8>), dim3(grid), dim3(1024), 0, st, op, T, B, partials, fold);
        return;
    }
#define HPC_RLL_SCAN_CASE(V_, NW_)                                                                          \
    if (c.v == V_ && c.nw == NW_) {                                                                         \
        hipLaunchKernelGGL((colscan_rev_kernel<Op, V_, 8, NW_>), dim3(grid), dim3(NW_ * 64), 0, st, op, T, B, \
                           partials, fold);                                                                 \
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

inline int scan_num_blocks(int T, int B, bool can_v2) { return (int)scan_grid(scan_cfg(T, B, can_v2), B); }

// reduce.hip
int finalize_sums(const float* partials, int nblocks, int nacc, const float* scales /*host, nacc*/, float* out,
                  hipStream_t st);
int scale_rows(const float* g, const float* in, float* out, long n_in, long n_out, hipStream_t st);

}  // namespace hpc_rll
