// pad_group.hip -- grouped padding of PACKED ragged rows entirely on the device.
//
// Reference behaviour being reproduced: hpc_rll/rl_utils/padding.py:20-45 with group > 1 -- sort the inputs by element
// count, split the sorted list into `group` buckets (oracle_split_group: hpc_rll/origin/padding.py:11-50,
// src/rl_utils/padding.cu:44-108; sample_split_group: padding.cu:8-43), pad every bucket to its own width.  There the
// sort is a python sorted() over tensor objects and the split an O(group * n^2) host DP with stack VLAs: unusable at the
// ~1M rows of BASELINE.json configs[4].  Here, for flat values + a device vector of lengths (Padding1DPacked):
//   1. histogram of the lengths (bins 0..max_len);
//   2. ONE workgroup turns the histogram into runs of equal lengths and runs the split policy on the runs
//      (pad_group.hpp: same cuts as the element-level DP, including its tie rule) -> the plan: number of groups, the cut
//      positions in sorted order, every group's width and its offset in the concatenated output;
//   3. stable LSD radix sort (8-bit digits, 1 pass for max_len < 256, 2 below 65536) of (length, row index) -> `order`,
//      python's sorted() order: ascending length, original order among equal lengths; deterministic (no atomics decide
//      an output position);
//   4. one launch writes all groups: one wave per sorted row, rows fetched through `order`, outputs back to back.
// Integer / byte work, HBM-bound, bit exact.  The only host synchronisation is the one the return convention forces:
// the caller must know the bucket shapes to allocate them (3 * group + 4 integers come back).
#include <hip/hip_runtime.h>

#include "hpc_rll_hip.h"
#include "pad_group.hpp"

namespace hpc_rll {
namespace {

constexpr int kMaxBins = 16385;        // lengths 0..16384: the run DP is O(group * D^2) in one workgroup
constexpr int kSortChunk = 1024;       // rows per workgroup of the radix passes: 4 slots x 256 threads

inline int last_err() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

// ------------------------------------------------------------------------------------------------ 1. histogram
__global__ __launch_bounds__(256) void len_hist_kernel(const int64_t* __restrict__ lengths, long n, int max_len,
                                                       int32_t* __restrict__ hist, int32_t* __restrict__ keys,
                                                       int64_t* __restrict__ plan) {
    extern __shared__ int32_t s_hist[];      // max_len + 1 bins when they fit (<= 8192), else global atomics
    const int bins = max_len + 1;
    const bool in_lds = bins <= 8192;
    if (in_lds) {
        for (int b = threadIdx.x; b < bins; b += 256) s_hist[b] = 0;
        __syncthreads();
    }
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int64_t l = lengths[i];
        if (l < 0 || l > max_len) { bad = true; l = l < 0 ? 0 : max_len; }
        keys[i] = (int32_t)l;
        if (in_lds) atomicAdd(&s_hist[(int)l], 1);     // integer counts: order independent
        else atomicAdd(&hist[(int)l], 1);
    }
    if (bad) plan[1] = 1;                              // status word: a length outside [0, max_len] (benign race: all write 1)
    if (in_lds) {
        __syncthreads();
        for (int b = threadIdx.x; b < bins; b += 256) {
            const int32_t c = s_hist[b];
            if (c) atomicAdd(&hist[b], c);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 2. the plan
// plan (int64): [0] ng, [1] status, [2 .. 2+G] cut positions (sorted order), [3+G .. 3+2G) group widths,
// [3+2G .. 4+3G) offsets of the groups in the concatenated output (G+1 entries), G = `group`.
// ws (int64): val[bins] | E[bins+1] | f[2][bins+1] | arg (int32) [(G+1) * (bins+1)]
__global__ __launch_bounds__(1024) void split_plan_kernel(const int32_t* __restrict__ hist, int max_len, long n, int group,
                                                          int mode, uint64_t seed, int64_t* __restrict__ ws,
                                                          int64_t* __restrict__ plan) {
    const int bins = max_len + 1;
    int64_t* val = ws;
    int64_t* E = val + bins;
    int64_t* f0 = E + bins + 1;
    int64_t* f1 = f0 + bins + 1;
    int32_t* arg = reinterpret_cast<int32_t*>(f1 + bins + 1);
    __shared__ int s_cnt[1024];
    __shared__ int64_t s_sum[1024];
    __shared__ int s_D;
    const int t = threadIdx.x;
    // ---- runs: compact the non-empty bins (each thread owns a contiguous slice of bins)
    const int per = (bins + 1023) / 1024;
    const int b0 = t * per, b1 = min(bins, b0 + per);
    int c = 0;
    int64_t sm = 0;
    for (int b = b0; b < b1; ++b) { const int h = hist[b]; c += h != 0; sm += h; }
    s_cnt[t] = c;
    s_sum[t] = sm;
    __syncthreads();
    if (t == 0) {                      // 1024 serial adds: microseconds, once per call
        int a = 0;
        int64_t s = 0;
        for (int i = 0; i < 1024; ++i) {
            const int ci = s_cnt[i];
            const int64_t si = s_sum[i];
            s_cnt[i] = a;
            s_sum[i] = s;
            a += ci;
            s += si;
        }
        s_D = a;
        E[0] = 0;
    }
    __syncthreads();
    {
        int r = s_cnt[t];
        int64_t e = s_sum[t];
        for (int b = b0; b < b1; ++b) {
            const int h = hist[b];
            if (h) { e += h; val[r] = b; E[r + 1] = e; ++r; }
        }
    }
    __syncthreads();
    const int D = s_D;
    const int G = group;
    const int M = (long)group < n ? group : (int)n;
    int64_t* pos = plan + 2;
    int64_t* gmax = plan + 3 + G;
    int64_t* goff = plan + 3 + 2 * G;
    if (mode == 1) {
        // ---- sample policy (padding.cu:8-43): group-1 random cuts in [1, n-2], consecutive duplicates re-drawn,
        // sorted; a cut whose bucket has the same width as the previous bucket is skipped exactly as the reference skips it
        // (its rows join the next accepted bucket; after the last cut, the previous one); a repeated cut position is
        // dropped (the reference would emit an empty bucket of width -1).  Same rule and same splitmix64 stream as the host's
        // hpc_rll_sample_split_group: identical boundaries for identical (lengths, group, seed).
        if (t == 0) {
            int64_t cut[64];
            int nc = 0;
            if (n >= 3) {
                int64_t last = -1;
                for (int i = 0; i < G - 1 && nc < 63; ++i) {
                    int64_t now = last;
                    if (n == 3) now = 1;
                    else while (now == last) {
                        uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
                        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                        z ^= z >> 31;
                        now = (int64_t)(z % (uint64_t)(n - 2)) + 1;
                    }
                    cut[nc++] = now;
                    last = now;
                }
                for (int i = 1; i < nc; ++i) {       // insertion sort, <= 63 entries
                    const int64_t x = cut[i];
                    int j = i - 1;
                    while (j >= 0 && cut[j] > x) { cut[j + 1] = cut[j]; --j; }
                    cut[j + 1] = x;
                }
            }
            cut[nc++] = n - 1;
            int ng = 0;
            int64_t last_idx = -1;
            int r = 0;
            for (int i = 0; i < nc; ++i) {
                const int64_t idx = cut[i];
                if (idx <= last_idx) continue;
                while (E[r + 1] <= idx) ++r;                        // run holding sorted element idx = the bucket's maximum
                const int64_t w = val[r];
                if (ng > 0 && gmax[ng - 1] == w) continue;   // the reference's skip: last_idx stays (padding.cu:34-35)
                gmax[ng] = w;
                pos[ng] = last_idx + 1;
                ++ng;
                last_idx = idx;
            }
            pos[ng] = n;
            plan[0] = ng;
        }
    } else if (D < M) {
        if (t == 0) {
            split_runs_few(val, E, D, (int64_t)n, M, pos, gmax);
            plan[0] = M;
        }
    } else {
        // ---- DP over the run boundaries, one layer per group, states spread over the workgroup
        for (int r = t; r <= D; r += 1024) f0[r] = r == 0 ? 0 : kSplitInf;
        __syncthreads();
        int64_t* fp = f0;
        int64_t* fc = f1;
        for (int j = 1; j <= M; ++j) {
            int32_t* aj = arg + (size_t)j * (bins + 1);
            for (int r = t; r <= D; r += 1024) {
                int64_t best = kSplitInf;
                int32_t a = 0;
                if (r >= j) split_runs_state(val, E, fp, r, j - 1, &best, &a);
                fc[r] = best;
                aj[r] = a;
            }
            __syncthreads();
            int64_t* tmp = fp; fp = fc; fc = tmp;
        }
        if (t == 0) {
            int r = D;
            pos[M] = n;
            for (int j = M; j >= 1; --j) {
                gmax[j - 1] = val[r - 1];
                r = arg[(size_t)j * (bins + 1) + r];
                pos[j - 1] = E[r];
            }
            plan[0] = M;
        }
    }
    __syncthreads();
    if (t == 0) {
        const int ng = (int)plan[0];
        int64_t off = 0;
        for (int g = 0; g < ng; ++g) { goff[g] = off; off += (pos[g + 1] - pos[g]) * gmax[g]; }
        goff[ng] = off;
        for (int g = ng; g < G; ++g) { pos[g + 1] = n; gmax[g] = 0; goff[g + 1] = off; }
    }
}

// ------------------------------------------------------------------------------------------------ 3. stable radix sort
__global__ __launch_bounds__(256) void radix_hist_kernel(const int32_t* __restrict__ keys, long n, int shift,
                                                         int32_t* __restrict__ table, int nchunks) {
    __shared__ int32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * kSortChunk;
    for (int s = 0; s < 4; ++s) {
        const long i = base + s * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(keys[i] >> shift) & 255], 1);
    }
    __syncthreads();
    table[(size_t)threadIdx.x * nchunks + blockIdx.x] = s_h[threadIdx.x];   // digit-major: the scan order of a stable sort
}

// exclusive scan of m int32 entries in place, one workgroup (the tile sums of the three-phase scan below; small m)
__global__ __launch_bounds__(1024) void scan_i32_kernel(int32_t* __restrict__ a, long m) {
    __shared__ int64_t s_part[1024];
    const int t = threadIdx.x;
    const long per = (m + 1023) / 1024;
    const long lo = t * per, hi = lo + per < m ? lo + per : m;
    int64_t s = 0;
    for (long i = lo; i < hi; ++i) s += a[i];
    s_part[t] = s;
    __syncthreads();
    if (t < 64) {          // one wave scans the 1024 partials: 16 per lane
        int64_t loc[16];
        int64_t ls = 0;
        for (int k = 0; k < 16; ++k) { loc[k] = ls; ls += s_part[t * 16 + k]; }
        int64_t incl = ls;
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t y = __shfl_up(incl, d, 64);
            if (t >= d) incl += y;
        }
        const int64_t off = incl - ls;
        for (int k = 0; k < 16; ++k) s_part[t * 16 + k] = off + loc[k];
    }
    __syncthreads();
    int64_t run = s_part[t];
    for (long i = lo; i < hi; ++i) { const int32_t v = a[i]; a[i] = (int32_t)run; run += v; }
}

// Round 5: the digit table of a radix pass (256 x n / 1024 counters: 1 MB at 2^20 rows) used to be scanned by the ONE workgroup
// above, every thread walking 256 consecutive counters (a wave touches 64 lines per step): 0.42 ms of the 1.07 ms the grouped
// pad of 2^20 rows took (rocprofv3, tests/tools/r05_group_pad_profile.sh).  Now three phases over tiles of kScanTile counters:
// tile sums (one workgroup per tile, 16 consecutive counters per thread: whole lines per wave), the one-workgroup scan of the
// tile sums, tile rescan with its offset.  Integer arithmetic: the same result, whatever the order.
constexpr int kScanTile = 4096;
__device__ __forceinline__ int32_t tile_thread_load(const int32_t* __restrict__ a, long m, long i0, int32_t (&v)[16]) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    int32_t s = 0;
    if (i0 + 16 <= m) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const vint4 x = reinterpret_cast<const vint4*>(a + i0)[q];
            v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = i0 + k < m ? a[i0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    return s;
}
__global__ __launch_bounds__(256) void scan_tile_sums_kernel(const int32_t* __restrict__ a, long m, int32_t* __restrict__ sums) {
    __shared__ int32_t s_w[4];
    int32_t v[16];
    int32_t s = tile_thread_load(a, m, (long)blockIdx.x * kScanTile + 16L * threadIdx.x, v);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(256) void scan_tile_apply_kernel(int32_t* __restrict__ a, long m, const int32_t* __restrict__ sums) {
    __shared__ int32_t s_w[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long i0 = (long)blockIdx.x * kScanTile + 16L * threadIdx.x;
    int32_t v[16];
    const int32_t mine = tile_thread_load(a, m, i0, v);
    int32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t y = __shfl_up(incl, d, 64);
        if (lane >= d) incl += y;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    int32_t run = sums[blockIdx.x] + incl - mine;
    for (int u = 0; u < w; ++u) run += s_w[u];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (i0 + k < m) a[i0 + k] = run;
        run += v[k];
    }
}
// the launches (tile_sums: ceil(m / kScanTile) int32 of scratch)
inline void scan_i32(int32_t* a, long m, int32_t* tile_sums, hipStream_t st) {
    const long nt = (m + kScanTile - 1) / kScanTile;
    if (nt <= 1) {
        hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, a, m);
        return;
    }
    hipLaunchKernelGGL(scan_tile_sums_kernel, dim3((unsigned)nt), dim3(256), 0, st, (const int32_t*)a, m, tile_sums);
    hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, tile_sums, nt);
    hipLaunchKernelGGL(scan_tile_apply_kernel, dim3((unsigned)nt), dim3(256), 0, st, a, m, (const int32_t*)tile_sums);
}

// scatter of one pass.  Row e of a chunk sits in slot s = e / 256 (thread t = e % 256): slots are processed in order,
// waves in order inside a slot, lanes in order inside a wave -- the rank of a row among the rows of its chunk with the
// same digit is exact, so the pass is STABLE.
__global__ __launch_bounds__(256) void radix_scatter_kernel(const int32_t* __restrict__ keys_in,
                                                            const int64_t* __restrict__ idx_in,
                                                            int32_t* __restrict__ keys_out, int64_t* __restrict__ idx_out,
                                                            long n, int shift, const int32_t* __restrict__ table,
                                                            int nchunks) {
    __shared__ int32_t s_base[256];
    __shared__ int32_t s_w[4][256];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    s_base[t] = table[(size_t)t * nchunks + blockIdx.x];
    const long base = (long)blockIdx.x * kSortChunk;
    for (int s = 0; s < 4; ++s) {
        s_w[0][t] = 0; s_w[1][t] = 0; s_w[2][t] = 0; s_w[3][t] = 0;
        __syncthreads();
        const long i = base + s * 256 + t;
        const bool ok = i < n;
        const int32_t key = ok ? keys_in[i] : 0;
        const int d = (key >> shift) & 255;
        unsigned long long mask = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bb = __ballot((d >> b) & 1);
            mask &= ((d >> b) & 1) ? bb : ~bb;
        }
        const int rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (ok && rank == 0) s_w[w][d] = __popcll(mask);
        __syncthreads();
        if (ok) {
            int off = s_base[d] + rank;
            for (int u = 0; u < w; ++u) off += s_w[u][d];
            keys_out[off] = key;
            idx_out[off] = idx_in ? idx_in[i] : (int64_t)i;
        }
        __syncthreads();
        s_base[t] += s_w[0][t] + s_w[1][t] + s_w[2][t] + s_w[3][t];
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ 4. all groups, one launch
// One wave per sorted row p (rows dealt round robin over the waves: neighbouring waves write neighbouring rows at the same
// time): the row's group, source row and length are wave-uniform, the lanes copy / fill the row's `width` columns; consecutive
// rows of a group are consecutive in the output.  table: rows {pointer, 1, 1, length} of hpc_rll_packed_table.
// Round 5: the metadata of the wave's NEXT row (order[p'] -> table row: two dependent round trips) is requested before the
// current row is copied, so a row costs one exposed round trip (its data) instead of three.  (Also tried: 64 consecutive rows
// per wave with the metadata fetched per lane and four rows copied at a time -- 0.52 against 0.40 ms: the round-robin order is
// what keeps the partial lines at the row boundaries of neighbouring rows together in time.)
template <int G_MAX>
__global__ __launch_bounds__(256) void pad_groups_kernel(const int64_t* __restrict__ table, const int64_t* __restrict__ order,
                                                         const int64_t* __restrict__ plan, int G, float* __restrict__ out,
                                                         int32_t* __restrict__ mask, float value) {
    __shared__ int64_t s_pos[G_MAX + 1], s_off[G_MAX + 1];
    __shared__ int32_t s_w[G_MAX];
    const int ng = (int)plan[0];
    if ((int)threadIdx.x <= ng) {
        s_pos[threadIdx.x] = plan[2 + threadIdx.x];
        s_off[threadIdx.x] = plan[3 + 2 * G + threadIdx.x];
        if ((int)threadIdx.x < ng) s_w[threadIdx.x] = (int32_t)plan[3 + G + threadIdx.x];
    }
    __syncthreads();
    const int64_t n = s_pos[ng];
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int g = 0;
    int64_t p = wave;
    if (p >= n) return;
    const int64_t* tr = table + order[p] * 4;
    const float* sp = reinterpret_cast<const float*>(tr[0]);
    int len = (int)tr[3];
    while (true) {
        const int64_t pn = p + nwaves;
        const float* spn = nullptr;
        int lenn = 0;
        if (pn < n) {                                          // the next row's metadata: in flight during this row's copy
            const int64_t* trn = table + order[pn] * 4;
            spn = reinterpret_cast<const float*>(trn[0]);
            lenn = (int)trn[3];
        }
        while (g + 1 < ng && p >= s_pos[g + 1]) ++g;       // rows only move forward: amortised O(1)
        const int wdt = s_w[g];
        const int64_t base = s_off[g] + (p - s_pos[g]) * (int64_t)wdt;
        for (int c = lane; c < wdt; c += 64) {
            const bool in = c < len;
            const float x = in ? sp[c] : value;
            __builtin_nontemporal_store(x, out + base + c);
            __builtin_nontemporal_store(in ? 1 : 0, mask + base + c);
        }
        if (pn >= n) break;
        p = pn;
        sp = spn;
        len = lenn;
    }
}

}  // namespace
}  // namespace hpc_rll

using namespace hpc_rll;

extern "C" int64_t hpc_rll_pad1d_group_workspace_int64(int64_t n, int max_len, int group) {
    if (n < 0 || max_len < 0 || max_len >= kMaxBins || group < 1) return HPC_RLL_EINVAL;
    const int64_t bins = (int64_t)max_len + 1;
    const int64_t nchunks = (n + kSortChunk - 1) / kSortChunk;
    // hist (int32) | run / DP workspace | radix table (int32) | keys a,b (int32) | idx tmp (int64)
    const int64_t hist = (bins + 1) / 2;
    const int64_t dp = bins + (bins + 1) + 2 * (bins + 1) + ((int64_t)(group + 1) * (bins + 1) + 1) / 2;
    const int64_t radix = (256 * (nchunks > 0 ? nchunks : 1) + 1) / 2;
    const int64_t keys = 2 * ((n + 1) / 2);
    const int64_t tiles = ((256 * (nchunks > 0 ? nchunks : 1) + kScanTile - 1) / kScanTile + 1) / 2;   // tile sums of the digit-table scan
    return hist + dp + radix + keys + n + tiles + 8;
}

extern "C" int hpc_rll_pad1d_group_plan(const int64_t* lengths, int64_t n, int max_len, int group, int mode, uint64_t seed,
                                        int64_t* ws, int64_t* plan, int64_t* order, void* stream) {
    if (n < 0 || max_len < 0 || group < 1 || (mode != 0 && mode != 1)) return HPC_RLL_EINVAL;
    if (max_len >= kMaxBins || group > 63 || n > 0x7fffffffL) return HPC_RLL_EUNSUPPORTED;
    if (!ws || !plan || (n > 0 && (!lengths || !order))) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int64_t bins = (int64_t)max_len + 1;
    const int nchunks = (int)((n + kSortChunk - 1) / kSortChunk);
    int32_t* hist = reinterpret_cast<int32_t*>(ws);
    int64_t* dp = ws + (bins + 1) / 2;
    const int64_t dp_words = bins + (bins + 1) + 2 * (bins + 1) + ((int64_t)(group + 1) * (bins + 1) + 1) / 2;
    //  the digit table is read with 16-byte loads (tile_thread_load): it starts on an EVEN int64 word of the
    // workspace (one pad word where hist + dp is odd, inside the 8 spare words hpc_rll_pad1d_group_workspace_int64 adds)
    int32_t* table = reinterpret_cast<int32_t*>(dp + dp_words + ((((bins + 1) / 2) + dp_words) & 1));
    int32_t* keys_a = table + 2 * ((256 * (int64_t)(nchunks > 0 ? nchunks : 1) + 1) / 2);
    int32_t* keys_b = keys_a + 2 * ((n + 1) / 2);
    int64_t* idx_tmp = reinterpret_cast<int64_t*>(keys_b + 2 * ((n + 1) / 2));
    int32_t* tile_sums = reinterpret_cast<int32_t*>(idx_tmp + n);
    hipError_t e = hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)bins, st);
    if (e == hipSuccess) e = hipMemsetAsync(plan, 0, sizeof(int64_t) * (size_t)(3 * group + 4), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return HPC_RLL_OK;
    int blocks = (int)((n + 2047) / 2048);
    if (blocks > 1024) blocks = 1024;
    const size_t lds = bins <= 8192 ? sizeof(int32_t) * (size_t)bins : 0;
    hipLaunchKernelGGL(len_hist_kernel, dim3(blocks), dim3(256), lds, st, lengths, (long)n, max_len, hist, keys_a, plan);
    hipLaunchKernelGGL(split_plan_kernel, dim3(1), dim3(1024), 0, st, hist, max_len, (long)n, group, mode, seed, dp, plan);
    // stable LSD radix sort of (length, row): passes of 8 bits
    const int passes = max_len < 256 ? 1 : 2;
    const int32_t* kin = keys_a;
    int32_t* kout = keys_b;
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        const bool last = p == passes - 1;
        hipLaunchKernelGGL(radix_hist_kernel, dim3(nchunks), dim3(256), 0, st, kin, (long)n, shift, table, nchunks);
        scan_i32(table, (long)256 * nchunks, tile_sums, st);
        hipLaunchKernelGGL(radix_scatter_kernel, dim3(nchunks), dim3(256), 0, st, kin, p == 0 ? (const int64_t*)nullptr : idx_tmp,
                           kout, last ? order : idx_tmp, (long)n, shift, table, nchunks);
        const int32_t* tmp = kin;
        kin = kout;
        kout = const_cast<int32_t*>(tmp);
    }
    return last_err();
}

extern "C" int hpc_rll_pad1d_group_forward(const int64_t* table, const int64_t* order, const int64_t* plan, int group,
                                           float* out, int32_t* mask, int64_t total_out, int value, void* stream) {
    if (group < 1 || total_out < 0) return HPC_RLL_EINVAL;
    if (group > 63) return HPC_RLL_EUNSUPPORTED;
    if (total_out == 0) return HPC_RLL_OK;
    if (!table || !order || !plan || !out || !mask) return HPC_RLL_EINVAL;
    const int64_t blocks = 256 * 16;        // 16 workgroups of 4 waves per CU, rows dealt round robin
    hipLaunchKernelGGL(pad_groups_kernel<64>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, order, plan,
                       group, out, mask, (float)value);
    return last_err();
}
