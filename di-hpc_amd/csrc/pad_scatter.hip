// pad_scatter.hip -- ragged Pad/Unpad (1-3D) and ScatterConnection for gfx950: integer/index work, bit exact.
//
// Replaces (under /root/reference):
//   Pad{1,2,3}DForward, GroupPad{1,2,3}DForward, Unpad{1,2,3}DForward, sample/oracle_split_group
//       src/rl_utils/padding.cu:8-582, padding_kernel.h:92-247
//   ScatterConnectionForward/Backward
//       src/torch_utils/network/scatter_connection.cu:8-73, scatter_connection_kernel.h:15-106
// Semantics: hpc_rll/origin/padding.py:53-173 and origin/scatter_connection.py:49-65.
//
// Pad/Unpad.  Reference: one block per tensor, per-call cudaMalloc + synchronous cudaMemcpy of pointer tables +
// cudaFree (padding.cu:118-138).  Here: ONE launch over the dense output index space for any rank 1..3 (a 1-D
// tensor is the shape (1,1,len) so the contiguous axis is innermost), the caller hands a device table
// {pointer|offset, d0, d1, d2} per tensor, nothing is allocated or synchronised.  Unpad writes ONE flat buffer
// (the python layer returns views of it) and finds the owning tensor of each flat element by binary search.
//
// ScatterConnection.  Reference: memset of the whole output, then one thread per (b,m,n) storing 4 bytes with
// stride H*W between consecutive n (uncoalesced); `cover` is a last-writer-wins race, `add` uses float atomics.
// Here the scatter is turned into a GATHER driven by a per-cell owner list:
//   1. index kernel (per b, in LDS): head[cell] = smallest m at that cell, next[m] = next larger m at the same
//      cell, last[cell] = largest m.  Integer only, deterministic.
//   2. output kernel: lanes <-> consecutive cells (the contiguous axis of (B,N,H,W)), 4 channels per step read as
//      one float4 of x[b,m,n:n+4]; `cover` takes x[b,last[cell]] (the CPU oracle's sequential "largest m wins"),
//      `add` walks the chain in ascending m (the oracle's sequential summation order -> bit exact), empty cells
//      get 0.  Every output element is written exactly once with a nontemporal store: no memset pass, no atomics.
//   3. backward: grad_x[b,m,n] = grad_out[b,n,cell_m].  Planes of grad_out are staged through LDS with
//      coalesced loads (a 4-byte gather from HBM would fetch the whole plane anyway, sector by sector).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hpc_rll_hip.h"
#include "wave.hpp"
#include "pad_group.hpp"

namespace hpc_rll {
namespace {

inline int last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

// ------------------------------------------------------------------------------------------------ pad
// table[i] = {src pointer, d0, d1, d2}; output (n, m0, m1, m2); one thread per output element.
__global__ __launch_bounds__(256) void pad_kernel(const int64_t* __restrict__ table, float* __restrict__ new_x,
                                                  int32_t* __restrict__ mask, long n, unsigned m0, unsigned m1,
                                                  unsigned m2, float fill, int ifill) {
    const unsigned inner = m0 * m1 * m2;
    const long total = n * (long)inner;
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        const long i = o / inner;
        unsigned rem = (unsigned)(o - i * inner);
        const unsigned c = rem % m2; rem /= m2;
        const unsigned b = rem % m1;
        const unsigned a = rem / m1;
        const int64_t* __restrict__ e = table + i * 4;
        const unsigned d0 = (unsigned)e[1], d1 = (unsigned)e[2], d2 = (unsigned)e[3];
        const bool in = a < d0 && b < d1 && c < d2;
        float v = fill;
        if (in) v = reinterpret_cast<const float*>(e[0])[((size_t)a * d1 + b) * d2 + c];
        __builtin_nontemporal_store(v, new_x + o);
        __builtin_nontemporal_store(in ? 1 : ifill, mask + o);
    }
}

// Same mapping, 4 consecutive output elements per thread: one index decomposition + carries instead of four, and
// 16-byte nontemporal stores (the output is 3x the input bytes: the kernel is store bound).  total % 4 == 0 required.
__global__ __launch_bounds__(256) void pad4_kernel(const int64_t* __restrict__ table, float* __restrict__ new_x,
                                                   int32_t* __restrict__ mask, long n, unsigned m0, unsigned m1,
                                                   unsigned m2, float fill, int ifill) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    const unsigned inner = m0 * m1 * m2;
    const long total4 = n * (long)inner / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
        const long o = q * 4;
        long i = o / inner;
        unsigned rem = (unsigned)(o - i * inner);
        unsigned c = rem % m2; rem /= m2;
        unsigned b = rem % m1;
        unsigned a = rem / m1;
        const int64_t* __restrict__ e = table + i * 4;
        const float* src = reinterpret_cast<const float*>(e[0]);
        unsigned d0 = (unsigned)e[1], d1 = (unsigned)e[2], d2 = (unsigned)e[3];
        vfloat4 v;
        vint4 mk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = a < d0 && b < d1 && c < d2;
            v[k] = in ? src[((size_t)a * d1 + b) * d2 + c] : fill;
            mk[k] = in ? 1 : ifill;
            if (++c == m2) {
                c = 0;
                if (++b == m1) {
                    b = 0;
                    if (++a == m0) {   // next tensor
                        a = 0;
                        ++i;
                        if (i < n) {
                            e = table + i * 4;
                            src = reinterpret_cast<const float*>(e[0]);
                            d0 = (unsigned)e[1]; d1 = (unsigned)e[2]; d2 = (unsigned)e[3];
                        }
                    }
                }
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(new_x + o));
        __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(mask + o));
    }
}

// 1-D rows (m0 = m1 = 1: Padding1D and the packed API, BASELINE.json configs[4]) without the generic kernel's index
// arithmetic: pad4_kernel spends a 64-bit division and four 32-bit divisions / remainders per 16 bytes it writes, and
// at n = 2^20 rows (1.07 GB of output, beyond the Infinity Cache) that, not HBM, set its time (0.39 ms = 3.6 TB/s,
// profiles/r02_suite_c5_kernel_stats.csv).  Here: one thread per output quad, row = floor(o / L) from one fp64
// multiply by 1/L with an exact fix-up, then four columns with a carry into the next row.
__device__ __forceinline__ unsigned div_by(unsigned long o, unsigned L, double inv) {
    unsigned i = (unsigned)((double)o * inv);
    const unsigned long p = (unsigned long)i * L;
    if (p > o) --i;
    else if (p + L <= o) ++i;
    return i;
}

__global__ __launch_bounds__(256) void pad1d4_kernel(const int64_t* __restrict__ table, float* __restrict__ new_x,
                                                     int32_t* __restrict__ mask, long n, unsigned L, double inv,
                                                     float fill, int ifill) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    const long total4 = n * (long)L / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
        const unsigned long o = (unsigned long)q * 4;
        unsigned i = div_by(o, L, inv);
        unsigned c = (unsigned)(o - (unsigned long)i * L);
        const int64_t* __restrict__ e = table + (size_t)i * 4;
        const float* src = reinterpret_cast<const float*>(e[0]);
        unsigned len = (unsigned)e[3];
        vfloat4 v;
        vint4 mk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = c < len;
            v[k] = in ? src[c] : fill;
            mk[k] = in ? 1 : ifill;
            if (++c == L) {   // next row
                c = 0;
                ++i;
                if ((long)i < n) {
                    e = table + (size_t)i * 4;
                    src = reinterpret_cast<const float*>(e[0]);
                    len = (unsigned)e[3];
                }
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(new_x + o));
        __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(mask + o));
    }
}
// (four independent quads per thread and iteration, to overlap the table -> pointer -> data round trips, measured SLOWER:
// 396 vs 351 us at n = 2^20 -- the kernel is bound by its 4-byte source reads, not by their latency; see below.)

// PACKED rows (one flat buffer, row i at elements [off_i, off_i + len_i), off = exclusive scan of the lengths): a
// workgroup owns RB consecutive rows, whose source range is ONE contiguous span of `flat` -- staged into LDS with aligned
// 16-byte loads -- and writes its RB * L output floats (and mask ints) as 16-byte quads reading LDS.  The per-thread
// kernels above fetch the source with four 4-byte loads per quad, each touching the same cache lines again: the
// ablation (tests/tools/micro/padbw.hip, n = 2^20, L = 127) shows 328 us for the full kernel, 170 us without the source
// reads (= the rate of two plain fills of the outputs), and 234-247 us with the LDS staging.
// table rows {address of the row, 1, 1, length} as built by hpc_rll_packed_table(base = flat, stride = 4).
__global__ __launch_bounds__(256) void pad1d_packed_kernel(const float* __restrict__ flat, const int64_t* __restrict__ table,
                                                           float* __restrict__ new_x, int32_t* __restrict__ mask, long n,
                                                           unsigned L, double inv, int RB, float fill, int ifill) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    extern __shared__ float tile[];                   // RB * L + 8 floats, then RB + 1 int64 offsets (8-byte aligned)
    long* s_off = reinterpret_cast<long*>(tile + (((size_t)RB * L + 8 + 1) & ~(size_t)1));
    const uintptr_t base = reinterpret_cast<uintptr_t>(flat);
    for (long r0 = (long)blockIdx.x * RB; r0 < n; r0 += (long)gridDim.x * RB) {
        const int nr = (int)(n - r0 < RB ? n - r0 : RB);
        __syncthreads();                              // the previous round's readers of tile / s_off are done
        for (int r = threadIdx.x; r < nr; r += 256) {
            const int64_t* e = table + (size_t)(r0 + r) * 4;
            s_off[r] = (long)(((uintptr_t)e[0] - base) >> 2);
            if (r == nr - 1) s_off[nr] = s_off[r] + e[3];
        }
        __syncthreads();
        const long lo = s_off[0], hi = s_off[nr];
        // 16-byte chunks aligned by ADDRESS.  The first / last chunk may reach up to 12 bytes outside [lo, hi): an aligned
        // 16-byte chunk never crosses a page, and it contains a valid element, so the read cannot fault; the extra lanes
        // are never used.
        const long lo4 = lo - (long)(((base >> 2) + (unsigned long)lo) & 3UL);
        // a length beyond max_len (the caller's precondition, unchecked) must not overrun the tile: such a workgroup
        // reads its rows straight from memory
        // (the staging loop writes WHOLE 16-byte chunks, so the bound is on the span rounded up to a chunk: when RB * L is
        // no multiple of 4 the last chunk would otherwise reach up to 3 floats past the tile, into s_off -- ADVICE r03)
        const bool fits = ((hi - lo4 + 3) & ~3L) <= (((long)RB * L + 8) & ~3L) && hi >= lo;
        if (fits)
            for (long p = lo4 + (long)threadIdx.x * 4; p < hi; p += 1024)
                *reinterpret_cast<vfloat4*>(tile + (p - lo4)) = *reinterpret_cast<const vfloat4*>(flat + p);
        __syncthreads();
        const unsigned long obase = (unsigned long)r0 * L, oend = obase + (unsigned long)nr * L;
        // Every element finds its row with its own multiply (div_by) and reads the two row offsets and its value from
        // LDS: four independent chains per thread, no divergent row-crossing branch.
        for (unsigned long o = (obase & ~3UL) + (unsigned long)threadIdx.x * 4; o < oend; o += 1024) {
            vfloat4 v;
            vint4 mk;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long oo = o + k;
                bool in = false;
                float val = fill;
                if (oo >= obase && oo < oend) {
                    const unsigned rel = (unsigned)(oo - obase);
                    const unsigned rr = div_by(rel, L, inv);
                    const unsigned c = rel - rr * L;
                    const long so = s_off[rr];
                    in = c < (unsigned)(s_off[rr + 1] - so);
                    if (in) val = fits ? tile[so - lo4 + c] : flat[so + c];
                }
                v[k] = val;
                mk[k] = in ? 1 : ifill;
            }
            if (o >= obase && o + 4 <= oend) {
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(new_x + o));
                __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(mask + o));
            } else {                                   // a quad shared with the neighbouring workgroup's rows
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (o + k >= obase && o + k < oend) { new_x[o + k] = v[k]; mask[o + k] = mk[k]; }
            }
        }
    }
}

// Round 4: WAVE-synchronous tiles in OUTPUT space (tests/tools/micro/padbw.hip: 226 us = 6.2 TB/s at n = 2^20, L = 127, where
// the workgroup kernel above takes 276-293 and two plain fills of the outputs 159).  A wave owns 1024 consecutive elements
// (256 aligned quads) of the flat (n * L) output stream, whatever rows they fall in.  Because the rows are packed, the source
// of those elements is ONE contiguous span of at most 1024 floats: staged into the wave's own LDS slice with aligned
// 16-byte loads -- no workgroup barrier, no ragged quads at tile ends (only the very last quad of the tensor can be
// partial), one float multiply instead of a division per quad and two row lookups per quad instead of two per element.
// Row of a quad: t = c_lo + 4 q < 1024 + L, rr = (int)((t + 0.5) * (1 / L)) in fp32 -- exact: (t + 0.5) / L is at least
// 0.5 / L away from an integer and the rounding error is below 4 (1024 + L) / L * 2^-24 < 0.5 / L for L <= 16384.
// 32 <= L <= 16384 (at most 34 rows per tile); same results as pad1d_packed_kernel bit for bit.
__global__ __launch_bounds__(256) void pad1d_packed_wave_kernel(const float* __restrict__ flat, const int64_t* __restrict__ table,
                                                                float* __restrict__ new_x, int32_t* __restrict__ mask, long n,
                                                                unsigned L, float inv_l, float fill, int ifill) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile_all[4][1032];
    __shared__ int s_rel_all[4][40], s_len_all[4][40];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* const tile = tile_all[wv];
    int* const s_rel = s_rel_all[wv];
    int* const s_len = s_len_all[wv];
    const uintptr_t base = reinterpret_cast<uintptr_t>(flat);
    const unsigned long total = (unsigned long)n * L;
    const long ntiles = (long)((total + 1023) / 1024);
    for (long w = (long)blockIdx.x * 4 + wv; w < ntiles; w += (long)gridDim.x * 4) {
        const unsigned long o0 = (unsigned long)w * 1024, o1 = o0 + 1024 < total ? o0 + 1024 : total;
        const long r_lo = (long)(o0 / L), r_hi = (long)((o1 - 1) / L);
        const unsigned c_lo = (unsigned)(o0 - (unsigned long)r_lo * L), c_hi = (unsigned)(o1 - 1 - (unsigned long)r_hi * L);
        const int nr = (int)(r_hi - r_lo + 1);
        long my_off = 0;
        int my_len = 0;
        if (lane < nr) {
            const int64_t* e = table + (size_t)(r_lo + lane) * 4;
            my_off = (long)(((uintptr_t)e[0] - base) >> 2);
            const int64_t l = e[3];
            my_len = (int)(l < (int64_t)L ? l : (int64_t)L);   // a length beyond max_len (unchecked precondition) is truncated
        }
        const long off_lo = __shfl(my_off, 0, 64), off_hi = __shfl(my_off, nr - 1, 64);
        const int len_lo = __shfl(my_len, 0, 64), len_hi = __shfl(my_len, nr - 1, 64);
        const long span_lo = off_lo + ((int)c_lo < len_lo ? (int)c_lo : len_lo);
        const long span_hi = off_hi + ((int)c_hi + 1 < len_hi ? (int)c_hi + 1 : len_hi);
        // 16-byte chunks aligned by ADDRESS: the first / last chunk may reach up to 12 bytes outside [span_lo, span_hi) but
        // never across a page, and each contains a valid element: the read cannot fault, the extra lanes are never used
        const long lo4 = span_lo - (long)(((base >> 2) + (unsigned long)span_lo) & 3UL);
        // rows truncated to L keep the span <= 1024 floats; a table whose offsets are not the packed ones (hand-made) could
        // exceed the slice: such a tile reads its elements straight from memory
        const bool fits = span_hi - lo4 <= 1028 && span_hi >= span_lo;
        if (lane <= nr) {
            s_rel[lane] = lane < nr ? (int)(my_off - lo4) : 0;
            s_len[lane] = lane < nr ? my_len : 0;
        }
        if (fits) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const long p = lo4 + 4 * (long)(lane + 64 * j);
                if (p < span_hi) *reinterpret_cast<vfloat4*>(tile + 4 * (lane + 64 * j)) = *reinterpret_cast<const vfloat4*>(flat + p);
            }
        }
        __builtin_amdgcn_wave_barrier();   // (same-wave LDS operations execute in order: the slice is visible below)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + 64 * j;
            const unsigned long o = o0 + 4 * (unsigned long)q;
            if (o >= o1) break;
            const unsigned t = c_lo + 4 * (unsigned)q;
            const unsigned rr = (unsigned)(((float)t + 0.5f) * inv_l);
            const unsigned c = t - rr * L;
            const int rel0 = s_rel[rr], len0 = s_len[rr], rel1 = s_rel[rr + 1], len1 = s_len[rr + 1];
            vfloat4 v;
            vint4 mk;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned cc = c + k;
                const bool nxt = cc >= L;
                if (nxt) cc -= L;
                const int ln = nxt ? len1 : len0, rl = nxt ? rel1 : rel0;
                const bool in = (int)cc < ln;
                float val = fill;
                if (in) val = fits ? tile[rl + (int)cc] : flat[lo4 + rl + (int)cc];
                v[k] = val;
                mk[k] = in ? 1 : ifill;
            }
            if (o + 4 <= o1) {
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(new_x + o));
                __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(mask + o));
            } else {   // the last quad of the tensor when n * L is no multiple of 4
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (o + k < o1) { new_x[o + k] = v[k]; mask[o + k] = mk[k]; }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the slice is rewritten by the next tile
    }
}

// Inverse: the workgroup's RB padded rows are read as 16-byte quads, their valid prefixes collected in LDS at their
// packed positions, and the contiguous span of `flat` is written with aligned 16-byte stores (element-wise at the two
// ragged ends, which belong to the neighbouring workgroups).  table rows {flat offset (elements), 1, 1, length}.
__global__ __launch_bounds__(256) void unpad1d_packed_kernel(const float* __restrict__ padded, const int64_t* __restrict__ table,
                                                             float* __restrict__ flat, long n, long total, unsigned L, double inv,
                                                             int RB) {
    extern __shared__ float tile[];
    long* s_off = reinterpret_cast<long*>(tile + (((size_t)RB * L + 8 + 1) & ~(size_t)1));
    const uintptr_t base = reinterpret_cast<uintptr_t>(flat);
    for (long r0 = (long)blockIdx.x * RB; r0 < n; r0 += (long)gridDim.x * RB) {
        const int nr = (int)(n - r0 < RB ? n - r0 : RB);
        __syncthreads();
        for (int r = threadIdx.x; r < nr; r += 256) {
            const int64_t* e = table + (size_t)(r0 + r) * 4;
            s_off[r] = e[0];
            if (r == nr - 1) s_off[nr] = e[0] + (e[3] < (int64_t)L ? e[3] : (int64_t)L);
        }
        __syncthreads();
        const long lo = s_off[0];
        const long hi = s_off[nr] < total ? s_off[nr] : total;
        const long lo4 = lo - (long)(((base >> 2) + (unsigned long)lo) & 3UL);
        const bool fits = ((s_off[nr] - lo4 + 3) & ~3L) <= (((long)RB * L + 8) & ~3L) && s_off[nr] >= lo;   // see pad1d_packed_kernel
        const unsigned long obase = (unsigned long)r0 * L, oend = obase + (unsigned long)nr * L;
        for (unsigned long o = (obase & ~3UL) + (unsigned long)threadIdx.x * 4; o < oend; o += 1024) {
            const unsigned long first = o < obase ? obase : o;
            unsigned rr = div_by(first - obase, L, inv);
            unsigned c = (unsigned)(first - obase - (unsigned long)rr * L);
            long so = s_off[rr] - lo4;
            unsigned len = (unsigned)(s_off[rr + 1] - s_off[rr]);
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (o >= obase && o + 4 <= oend) {
                if (c < len || c + 4 > L) {            // (a quad entirely inside one row's padding is not fetched)
                    const vfloat4 t = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(padded + o));
                    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (o + k >= obase && o + k < oend) v[k] = padded[o + k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long oo = o + k;
                if (oo >= obase && oo < oend) {
                    if (c < len) {
                        if (fits) tile[so + c] = v[k];
                        else if (so + lo4 + c < total) flat[so + lo4 + c] = v[k];
                    }
                    if (++c == L) {
                        c = 0;
                        ++rr;
                        if ((int)rr < nr) { so = s_off[rr] - lo4; len = (unsigned)(s_off[rr + 1] - s_off[rr]); }
                    }
                }
            }
        }
        __syncthreads();
        for (long p = lo4 + (long)threadIdx.x * 4; fits && p < hi; p += 1024) {
            const float* srcp = tile + (p - lo4);
            if (p >= lo && p + 4 <= hi) {
                vfloat4 t;
                t.x = srcp[0]; t.y = srcp[1]; t.z = srcp[2]; t.w = srcp[3];
                __builtin_nontemporal_store(t, reinterpret_cast<vfloat4*>(flat + p));
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (p + k >= lo && p + k < hi) flat[p + k] = srcp[k];
            }
        }
    }
}

// inverse for 1-D rows: one thread per quad of the PADDED tensor (16-byte loads, no binary search per element);
// table[i] = {flat offset of row i (elements), 1, 1, length}
__global__ __launch_bounds__(256) void unpad1d4_kernel(const float* __restrict__ padded, const int64_t* __restrict__ table,
                                                       float* __restrict__ flat, long n, long total, unsigned L, double inv) {
    const long total4 = n * (long)L / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
        const unsigned long o = (unsigned long)q * 4;
        unsigned i = div_by(o, L, inv);
        unsigned c = (unsigned)(o - (unsigned long)i * L);
        const int64_t* __restrict__ e = table + (size_t)i * 4;
        long off = e[0];
        unsigned len = (unsigned)e[3];
        if (c >= len && c + 4 <= L) continue;          // a quad that lies entirely in one row's padding
        const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(padded + o));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c < len && off + c < total) flat[off + c] = v[k];
            if (++c == L) {
                c = 0;
                ++i;
                if ((long)i < n) {
                    e = table + (size_t)i * 4;
                    off = e[0];
                    len = (unsigned)e[3];
                } else {
                    len = 0;
                }
            }
        }
    }
}

// table[i] = {flat offset of tensor i (elements), d0, d1, d2}; one thread per element of the flat output.
__global__ __launch_bounds__(256) void unpad_kernel(const float* __restrict__ padded,
                                                    const int64_t* __restrict__ table, float* __restrict__ flat,
                                                    long n, long total, unsigned m0, unsigned m1, unsigned m2) {
    for (long o = (long)blockIdx.x * 256 + threadIdx.x; o < total; o += (long)gridDim.x * 256) {
        long lo = 0, hi = n - 1;  // last i with offset_i <= o (empty tensors share an offset: skip them)
        while (lo < hi) {
            const long mid = (lo + hi + 1) >> 1;
            if (table[mid * 4] <= o) lo = mid; else hi = mid - 1;
        }
        const int64_t* __restrict__ e = table + lo * 4;
        const unsigned d0 = (unsigned)e[1], d1 = (unsigned)e[2], d2 = (unsigned)e[3];
        unsigned rem = (unsigned)(o - e[0]);
        // a caller-supplied `total` beyond the sum of the shapes, or a shape larger than the padded tensor (the packed
        // API takes both from the caller unchecked): nothing to read for this element -- never index outside `padded`
        if ((unsigned long)rem >= (unsigned long)d0 * d1 * d2) continue;
        const unsigned c = rem % d2; rem /= d2;
        const unsigned b = rem % d1;
        const unsigned a = rem / d1;
        if (a >= m0 || b >= m1 || c >= m2) continue;
        flat[o] = padded[(((size_t)lo * m0 + a) * m1 + b) * m2 + c];
    }
}

// ------------------------------------------------------------------------------------------------ packed table
// lengths (n,) int64 -> table rows {base + stride * exclusive_sum(lengths)[i], 1, 1, lengths[i]}.  Three tiny launches:
// per-chunk sums (kScanChunk elements per workgroup), one workgroup scanning the chunk sums, per-chunk scan + write.
constexpr int kScanChunk = 2048;   // 256 threads x 8 elements

__device__ __forceinline__ int64_t wave_incl_scan_i64(int64_t x, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    return x;
}

// inclusive scan over the 256 threads of a workgroup; returns the thread's inclusive prefix, *total = workgroup sum
__device__ __forceinline__ int64_t block_incl_scan_i64(int64_t x, int64_t* lds /* 4 */, int64_t* total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t incl = wave_incl_scan_i64(x, lane);
    if (lane == 63) lds[w] = incl;
    __syncthreads();
    int64_t off = 0;
    for (int i = 0; i < w; ++i) off += lds[i];
    *total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return incl + off;
}

__global__ __launch_bounds__(256) void packed_chunk_sums_kernel(const int64_t* __restrict__ lengths, long n,
                                                                int64_t* __restrict__ sums) {
    __shared__ int64_t lds[4];
    const long base = (long)blockIdx.x * kScanChunk;
    int64_t s = 0;
    for (int k = 0; k < 8; ++k) {
        const long i = base + k * 256 + threadIdx.x;
        if (i < n) s += lengths[i];
    }
    int64_t total;
    block_incl_scan_i64(s, lds, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// exclusive scan of the chunk sums in place (one workgroup, carried over rounds of 256)
__global__ __launch_bounds__(256) void packed_scan_sums_kernel(int64_t* __restrict__ sums, long nchunks) {
    __shared__ int64_t lds[4];
    int64_t carry = 0;
    for (long r = 0; r < nchunks; r += 256) {
        const long i = r + threadIdx.x;
        const int64_t v = i < nchunks ? sums[i] : 0;
        int64_t total;
        const int64_t incl = block_incl_scan_i64(v, lds, &total);
        if (i < nchunks) sums[i] = carry + incl - v;
        carry += total;
    }
}

// RAW_SUMS: `sums` holds the chunk totals as packed_chunk_sums_kernel left them (no scan launch in between): the workgroup
// adds up the chunks before its own (<= 1024 of them; one dependent launch and its gap less, ~8 us of a 280 us pad call)
template <bool RAW_SUMS>
__global__ __launch_bounds__(256) void packed_table_kernel(const int64_t* __restrict__ lengths, long n,
                                                           const int64_t* __restrict__ sums, int64_t base,
                                                           int64_t stride, int64_t* __restrict__ table) {
    __shared__ int64_t lds[4];
    int64_t before = 0;
    if (RAW_SUMS) {
        int64_t v = 0;
        for (long j = threadIdx.x; j < (long)blockIdx.x; j += 256) v += sums[j];
        (void)block_incl_scan_i64(v, lds, &before);
    } else {
        before = sums[blockIdx.x];
    }
    // thread t owns 8 CONSECUTIVE elements so that its serial prefix is cheap
    const long first = (long)blockIdx.x * kScanChunk + (long)threadIdx.x * 8;
    int64_t len[8];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        len[k] = first + k < n ? lengths[first + k] : 0;
        s += len[k];
    }
    int64_t total;
    int64_t off = before + block_incl_scan_i64(s, lds, &total) - s;
    // Round 4: the rows go out through LDS so that a wave's stores are 2 KiB contiguous (lane = row, two 16-byte stores each).  The
    // first version stored a thread's eight rows itself -- 32 eight-byte stores at a 256-byte lane stride: 19 us for 32 MB at
    // n = 2^20, a tenth of the packed Pad1D call.  Index i + i / 8: the thread stride becomes 72 bytes (no 16-way bank conflict).
    typedef long vlong2 __attribute__((ext_vector_type(2)));
    __shared__ int64_t s_off[kScanChunk + kScanChunk / 8], s_len[kScanChunk + kScanChunk / 8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = threadIdx.x * 8 + k;
        s_off[i + threadIdx.x] = base + stride * off;
        s_len[i + threadIdx.x] = len[k];
        off += len[k];
    }
    __syncthreads();
    const long row0 = (long)blockIdx.x * kScanChunk;
    const bool al16 = (reinterpret_cast<uintptr_t>(table) & 15) == 0;
#pragma unroll
    for (int r = 0; r < kScanChunk / 256; ++r) {
        const int i = r * 256 + threadIdx.x;
        if (row0 + i < n) {
            int64_t* const row = table + (row0 + i) * 4;
            const vlong2 a = {(long)s_off[i + (i >> 3)], 1L}, b = {1L, (long)s_len[i + (i >> 3)]};
            if (al16) {
                reinterpret_cast<vlong2*>(row)[0] = a;
                reinterpret_cast<vlong2*>(row)[1] = b;
            } else {   // a table at 8 mod 16
                row[0] = a.x; row[1] = 1; row[2] = 1; row[3] = b.y;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ scatter
// idx layout per b (int32): [head HW | last HW | next M]
// One workgroup per b walks the entities in chunks of 256 in ascending m.  Entity m finds its predecessor at the same
// cell among the earlier lanes of its chunk (LDS scan) or, failing that, in last[cell] as left by the earlier chunks
// (this workgroup is the only writer of its b); it links next[prev] = m, or becomes head[cell].  last[cell] = max m
// (integer atomicMax: deterministic).  O(M * 256) instead of the first version's O(M^2) pair scan, 1 KB of LDS
// whatever M is (M ~ 1e4 entities per map).
__global__ __launch_bounds__(256) void scatter_index_kernel(const int64_t* __restrict__ location,
                                                            int32_t* __restrict__ idx, int M, int H, int W, int parts) {
    // parts (round 4): bit 0 = head + next are wanted (add), bit 1 = last is wanted (cover, or add with more than one chunk, which
    // links its chains across chunks through last).  The tables that are not wanted are neither initialised nor written: add at
    // M <= 256 leaves out half of the block's bytes and every atomic.
    typedef int vint4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) int32_t s_cell[256];
    const int b = blockIdx.x;
    const int HW = H * W;
    const bool want_chain = parts & 1, want_last = parts & 2;
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    int32_t* head = idx + (size_t)b * (2 * HW + M);
    int32_t* last = head + HW;
    int32_t* next = last + HW;
    // the tables of this b start at -1: 16-byte stores when every block is 16-byte aligned
    const int words = 2 * HW + M;
    const bool al = (words & 3) == 0 && (HW & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0;
    auto fill = [&](int32_t* p, int n) {
        if (al && (n & 3) == 0) {
            const vint4 m1 = {-1, -1, -1, -1};
            for (int i = threadIdx.x; i < n / 4; i += 256) reinterpret_cast<vint4*>(p)[i] = m1;
        } else {
            for (int i = threadIdx.x; i < n; i += 256) p[i] = -1;
        }
    };
    if (want_chain && want_last) fill(head, words);
    else if (want_chain) { fill(head, HW); fill(next, M); }
    else fill(last, HW);
    __syncthreads();
    for (int m0 = 0; m0 < M; m0 += 256) {
        const int m = m0 + threadIdx.x;
        int32_t c = -1;
        if (m < M) {
            const long y = loc[2 * m], x = loc[2 * m + 1];
            // out-of-range locations are dropped (the reference would write out of bounds)
            if (y >= 0 && y < H && x >= 0 && x < W) c = (int32_t)(y * W + x);
        }
        if (!want_chain) {                                   // cover: the largest m of a cell, nothing else
            if (c >= 0) atomicMax(&last[c], m);
            continue;
        }
        s_cell[threadIdx.x] = c;
        const int32_t before = (c >= 0 && want_last) ? last[c] : -1;   // largest m of the EARLIER chunks at this cell
        __syncthreads();
        // the latest EARLIER lane of the chunk at the same cell.  Round 4: every lane of a wave reads the same four cells per step
        // (a broadcast read) and keeps the last match below its own index -- no data-dependent branch.  The first version walked
        // backwards from its own lane with a break: up to 255 DEPENDENT LDS round trips per thread, ~10 us per workgroup and most
        // of this kernel's 94 us at B = 4096, M = 256.
        int pk = -1;
        const int kend = (((int)threadIdx.x >> 6) + 1) * 16;     // wave-uniform: quads up to the end of this wave's lanes
        for (int k4 = 0; k4 < kend; ++k4) {
            const vint4 v = reinterpret_cast<const vint4*>(s_cell)[k4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (v[j] == c && 4 * k4 + j < (int)threadIdx.x) pk = 4 * k4 + j;
        }
        if (c >= 0) {
            int32_t prev = pk >= 0 ? m0 + pk : -1;
            if (prev < 0) prev = before;
            if (prev >= 0) next[prev] = m;   // exactly one writer per slot
            else head[c] = m;                // exactly one writer per cell
            if (want_last) atomicMax(&last[c], m);
        }
        __syncthreads();   // the next chunk reads last[] (global memory written by this workgroup: barrier + L1 write-through)
    }
}

// grid: (cell blocks, n groups, B); lanes <-> consecutive cells; NPT channels per thread step of 4.
template <bool ADD>
__global__ __launch_bounds__(256) void scatter_out_kernel(const float* __restrict__ x,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ out, int M, int N, int HW,
                                                          int n_per_block, bool vec) {
    const int b = blockIdx.z;
    const int cell = blockIdx.x * 256 + threadIdx.x;
    if (cell >= HW) return;
    const int32_t* __restrict__ head = idx + (size_t)b * (2 * HW + M);
    const int32_t* __restrict__ last = head + HW;
    const int32_t* __restrict__ next = last + HW;
    const float* __restrict__ xb = x + (size_t)b * M * N;
    float* __restrict__ ob = out + (size_t)b * N * HW + cell;
    const int n0 = blockIdx.y * n_per_block;
    const int n1 = min(N, n0 + n_per_block);
    const int32_t first = ADD ? head[cell] : last[cell];
    for (int n = n0; n < n1; n += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int32_t m = first; m >= 0; m = ADD ? next[m] : -1) {
            const float* __restrict__ src = xb + (size_t)m * N + n;
            if (vec) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (n + k < n1) acc[k] += src[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n + k < n1) __builtin_nontemporal_store(acc[k], ob + (size_t)(n + k) * HW);
    }
}

// 4 consecutive cells x 4 channels per thread step: four float4 gathers, a 4x4 register transpose, four float4
// nontemporal stores (1 KiB per wave-store instead of 256 B).  Needs HW % 4 == 0, N % 4 == 0, 16-byte aligned x/out.
template <bool ADD>
__global__ __launch_bounds__(1024) void scatter_out4_kernel(const float* __restrict__ x,
                                                           const int32_t* __restrict__ idx,
                                                           float* __restrict__ out, int M, int N, int HW,
                                                           int n_per_block) {
    const int b = blockIdx.z;
    const int cell = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // blockDim.x = 256 or 1024 (16 KiB runs per plane)
    const int32_t* __restrict__ head = idx + (size_t)b * (2 * HW + M);
    const int32_t* __restrict__ last = head + HW;
    const int32_t* __restrict__ next = last + HW;
    // (round 4 also built the tables inside this kernel, as scatter_out_lds_kernel does: at C5 `add` 859 -> 875 us -- this kernel
    // has no staging phase the build's round trip and barriers could hide behind; removed in round 5)
    if (cell >= HW) return;
    int32_t first[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) first[c] = ADD ? head[cell + c] : last[cell + c];
    const float* __restrict__ xb = x + (size_t)b * M * N;
    float* __restrict__ ob = out + (size_t)b * N * HW + cell;
    const int n0 = blockIdx.y * n_per_block;
    const int n1 = min(N, n0 + n_per_block);
    auto nxt = [&](int32_t m) -> int32_t { return next[m]; };
    // 16 channels per iteration: all 16 gathers of an iteration are independent and issued together (a dependent
    // gather costs ~2 us under streaming load; serialising them made the first version latency bound)
    constexpr int U = 4;
    for (int n = n0; n < n1; n += 4 * U) {
        vfloat4 acc[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                vfloat4 a = {0.f, 0.f, 0.f, 0.f};
                if (first[c] >= 0 && n + 4 * u < n1)
                    a = *reinterpret_cast<const vfloat4*>(xb + (size_t)first[c] * N + n + 4 * u);
                acc[u][c] = a;
            }
        if (ADD) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                for (int32_t m = first[c] >= 0 ? nxt(first[c]) : -1; m >= 0; m = nxt(m))
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (n + 4 * u < n1) acc[u][c] += *reinterpret_cast<const vfloat4*>(xb + (size_t)m * N + n + 4 * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (n + 4 * u < n1) {
                    vfloat4 o = {acc[u][0][q], acc[u][1][q], acc[u][2][q], acc[u][3][q]};
                    __builtin_nontemporal_store(o, reinterpret_cast<vfloat4*>(ob + (size_t)(n + 4 * u + q) * HW));
                }
            }
    }
}

// LDS-staged output kernel (round 2).  Workgroup = (b, group of NPB channels): x[b, :, n0:n0+NPB] (M rows) and the cell
// owner table are staged in LDS once, then the workgroup's output -- NPB consecutive planes of out[b], ONE contiguous
// span of NPB*HW floats -- is cut into 16 contiguous pieces, one per wave, written front to back with 16-byte
// nontemporal stores: every wave streams tens of KiB of consecutive addresses (tests/tools/micro/writebw.hip: 5.5 TB/s
// with 64 KiB contiguous per wave against 5.1 with 1 KiB pieces strided by the plane size, which is what
// scatter_out4_kernel issues).  Per float4 of output: one 16-byte LDS read of the four owners, four 4-byte LDS gathers
// from the padded x tile (row stride NPB+1: conflict-free for distinct m), `add` walks the chain in ascending m.
// Needs HW % 4 == 0, 16-byte aligned x/out, HW ints + M*(NPB+1) floats of LDS.
// BUILD (round 4, tune key 37): the owner table and the chain links are built HERE, in LDS, from `location` -- no index
// launch, no index in memory.  cover: the largest m at a cell = an LDS atomic max per entity; add: the head = an LDS atomic
// min (empty = -1 = the largest unsigned), next[m] = the smallest later entity at the same cell, found with broadcast reads of
// the cell list (M <= 1024; 1024 / M threads share an entity's walk over the M / 4 quads).  The same table scatter_index_kernel leaves in memory, so the
// same output bits.  At C5 the index launch took 64 of the forward's 880 us; a workgroup spends ~1 us here.
constexpr int kScatterCollRows = 32;   // extra x-tile rows for the cells with several entities (add, in-kernel tables)
template <bool ADD, bool BUILD>
__global__ __launch_bounds__(1024) void scatter_out_lds_kernel(const float* __restrict__ x,
                                                              const int32_t* __restrict__ idx,
                                                              float* __restrict__ out, int M, int N, int HW, int npb,
                                                              const int64_t* __restrict__ location, int W, int pf_wgs) {
    typedef int vint4 __attribute__((ext_vector_type(4)));
    extern __shared__ float s_dyn[];
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * npb;
    const int nn = min(npb, N - n0);                       // channels of this workgroup
    const int ld = npb + 1;
    float* xs = s_dyn;                                     // [M][ld]
    const int tile_rows = M + (ADD && BUILD ? kScatterCollRows : 0);
    int32_t* s_first = reinterpret_cast<int32_t*>(s_dyn + (((size_t)tile_rows * ld + 3) & ~(size_t)3));   // [HW] head (add) / last (cover), 16-byte aligned
    int32_t* s_next = s_first + HW;                        // [M]   (add only)
    const float* __restrict__ xb = x + (size_t)b * M * N + n0;
    // the x tile (float4 along the channels when aligned)
    auto stage_x = [&]() {
        if ((nn & 3) == 0 && (N & 3) == 0 && (n0 & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
            const int q = nn >> 2;                             // float4 per row
            for (int e = threadIdx.x; e < M * q; e += 1024) {
                const int m = e / q, j = e - m * q;
                const vfloat4 t = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(xb + (size_t)m * N) + j);
                float* d = xs + m * ld + 4 * j;
                d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
            }
        } else {
            for (int e = threadIdx.x; e < M * nn; e += 1024) {
                const int m = e / nn, j = e - m * nn;
                xs[m * ld + j] = xb[(size_t)m * N + j];
            }
        }
    };
    // add with in-kernel tables (round 5): a cell that holds TWO OR MORE entities -- 0.2 % of the cells at 256 entities on a 64 x 64
    // map -- gets a row of its own in the x tile, the chain's sum per channel (ascending m, as everywhere), and points at it: the
    // stream loop then treats `add` like `cover`, ONE branch-free gather per cell.  Up to kScatterCollRows such cells per batch
    // element; a batch element with more keeps the chain walk in the loop.
    int32_t* s_coll = s_next + 2 * ((M + 3) & ~3);         // [kScatterCollRows] cells, [kScatterCollRows] chain heads, then the counter
    int32_t* s_ncoll = s_coll + 2 * kScatterCollRows;
    if (ADD && BUILD && threadIdx.x == 0) *s_ncoll = 0;
    if (BUILD) {
        int32_t* s_cell = s_next + ((M + 3) & ~3);         // [M rounded to 4]   (add only)
        const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
        const int H = HW / W;
        const int m4 = (M + 3) & ~3;
        // the first entity of every thread is requested before the x tile: one memory round trip for both
        long y0 = -1, x0 = -1;
        if ((int)threadIdx.x < M) { y0 = loc[2 * threadIdx.x]; x0 = loc[2 * threadIdx.x + 1]; }
        const vint4 m1 = {-1, -1, -1, -1};
        for (int c4 = threadIdx.x; c4 < (HW >> 2); c4 += 1024) reinterpret_cast<vint4*>(s_first)[c4] = m1;
        if (ADD)
            for (int m = threadIdx.x; m < M; m += 1024) s_next[m] = -1;
        stage_x();
        __syncthreads();
        for (int m = threadIdx.x; m < m4; m += 1024) {
            int32_t c = -2;                                // the padding of the cell list matches nothing
            if (m < M) {
                const long y = m == (int)threadIdx.x ? y0 : loc[2 * m], xx = m == (int)threadIdx.x ? x0 : loc[2 * m + 1];
                c = (y >= 0 && y < H && xx >= 0 && xx < W) ? (int32_t)(y * W + xx) : -1;   // out of range: dropped, as in scatter_index_kernel
                if (c >= 0) {
                    if (ADD) atomicMin(reinterpret_cast<unsigned*>(s_first) + c, (unsigned)m);
                    else atomicMax(s_first + c, m);
                }
            }
            if (ADD) s_cell[m] = c;
        }
        if (ADD) {
            __syncthreads();
            // M <= 1024: the 1024 threads are P = 1024 / mr groups of mr (M rounded to whole waves) -- thread (p, m) walks the p-th
            // part of the cell list for entity m and files the smallest later entity it finds with an LDS atomic min (-1 = none
            // = the largest unsigned)
            const int mr = (M + 63) & ~63, P = 1024 / mr;
            const int p = threadIdx.x / mr, m = threadIdx.x - p * mr;
            if (p < P) {
                const int quads = m4 >> 2, q0 = (int)((long)quads * p / P), q1 = (int)((long)quads * (p + 1) / P);
                const int32_t c = m < M ? s_cell[m] : -1;
                int nk = -1;
                for (int q = q1 - 1; q >= q0; --q) {       // descending: the last hit kept is the smallest later entity
                    const vint4 v = reinterpret_cast<const vint4*>(s_cell)[q];
#pragma unroll
                    for (int j = 3; j >= 0; --j)
                        if (v[j] == c && 4 * q + j > m) nk = 4 * q + j;
                }
                if (c >= 0 && nk >= 0) atomicMin(reinterpret_cast<unsigned*>(s_next) + m, (unsigned)nk);
            }
        }
    } else {
        const int32_t* __restrict__ head = idx + (size_t)b * (2 * HW + M);
        const int32_t* __restrict__ first_g = ADD ? head : head + HW;
        const int32_t* __restrict__ next_g = head + 2 * HW;
        // ---- stage: owner table, chain links
        for (int c = threadIdx.x; c < HW; c += 1024) s_first[c] = first_g[c];
        if (ADD)
            for (int m = threadIdx.x; m < M; m += 1024) s_next[m] = next_g[m];
        stage_x();
    }
    __syncthreads();
    bool walk = ADD;                                        // the stream loop walks chains (uniform)
    if (ADD && BUILD) {
        const int32_t* s_cell = s_next + ((M + 3) & ~3);
        for (int m = threadIdx.x; m < M; m += 1024) {
            const int32_t c = s_cell[m];
            if (c >= 0 && s_first[c] == m && s_next[m] >= 0) {   // m heads a chain of two or more
                const int i = atomicAdd(s_ncoll, 1);
                if (i < kScatterCollRows) { s_coll[i] = c; s_coll[kScatterCollRows + i] = m; }
            }
        }
        __syncthreads();
        const int nc = *s_ncoll;
        walk = nc > kScatterCollRows;
        if (!walk) {
            for (int e = threadIdx.x; e < nc * nn; e += 1024) {
                const int ci = e / nn, nch = e - ci * nn;
                int32_t m = s_coll[kScatterCollRows + ci];
                float a = xs[m * ld + nch];
                for (m = s_next[m]; m >= 0; m = s_next[m]) a += xs[m * ld + nch];
                xs[(M + ci) * ld + nch] = a;
                if (nch == 0) s_first[s_coll[ci]] = M + ci;
            }
        }
        __syncthreads();
    }
    // Round 5: what THIS workgroup stages came slowly -- its loads queue behind the other workgroups' stores (the memory system
    // is store-saturated: ~12 us until the tile is there, of a ~30 us lifetime) -- so before it starts streaming, a workgroup
    // touches the lines that workgroup L + pf_wgs will stage (one dword per 128-byte line of its x tile rows and of its
    // locations).  pf_wgs is a multiple of 8, so that workgroup runs on the SAME XCD (dispatch is round-robin over the eight)
    // and finds them in this L2; the launcher picks half the resident set, i.e. the workgroup that starts about when these
    // loads have returned.  At C5: 0.778 -> 0.724 ms (profiles/r05_scatter_pf.txt; the distance swept 64 ... 512 workgroups).
    float pfv = 0.f;
    if (pf_wgs > 0) {
        const unsigned L2 = blockIdx.y * gridDim.x + blockIdx.x + (unsigned)pf_wgs;
        const unsigned b2 = L2 / gridDim.x, bx2 = L2 - b2 * gridDim.x;
        if (b2 < gridDim.y) {
            const int n02 = (int)bx2 * npb, nl = (min(npb, N - n02) + 31) >> 5;      // 128-byte lines per tile row
            const float* xp = x + (size_t)b2 * M * N + n02;
            for (int e = threadIdx.x; e < M * nl; e += 1024) {
                const int m = e / nl, j = e - m * nl;
                pfv += xp[(size_t)m * N + 32 * j];
            }
            if (BUILD) {
                const float* lp = reinterpret_cast<const float*>(location + (size_t)b2 * M * 2);
                for (int e = 1023 - (int)threadIdx.x; e * 32 < M * 4; e += 1024) pfv += lp[e * 32];
            }
        }
    }
    // ---- stream the span: wave w writes float4 units [w*per, (w+1)*per) of the nn*HW/4 units of this workgroup
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;
    const long units = (long)nn * hw4;
    const long per = (units + 15) / 16;
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    // Round 4: the loop was INSTRUCTION-bound, not write-bound (~40 vector instructions per 16-byte store, among them a
    // 64-bit division by the plane size: 1.07 G stores x 40 / 64 lanes x 4 cycles / 1024 SIMDs = 1.1 ms at C5, the time it
    // took).  Now the lane's (channel, cell) position advances incrementally (one division before the loop), and a quad of
    // cells without an owner -- 78 % of them at 256 entities on a 64 x 64 map -- is one LDS read, three ANDs and the store.
    // (Also tried in round 4, not kept: ONE 4 KiB output block per short-lived 256-thread workgroup with no staging at all --
    // owner quad from L2, owned cells gathered from x 4 bytes at a time -- the shape that fixed the one-hot gradients: 1.47 ms
    // cover / 2.37 ms add against 0.90 / 0.89 here; the staged x tile and owner table are what make this kernel.)
    // Round 5 (tests/tools/micro/scatter_sweep.hip, profiles/r05_scatter_sweep.txt).  The store pattern is NOT the bound: the same
    // launch shape with nothing but these stores runs at 6.4 TB/s (0.67 ms at C5), with the x tile reads added 6.0-6.4.  In-kernel
    // clocks: a wave needed 0.43 us per 1 KiB piece (two DEPENDENT LDS round trips -- owner quad, then the gathers -- and ~40
    // instructions, eight waves per SIMD) against 0.13 us for a store-only wave, and the kernel's time followed the lifetime of
    // its workgroups (two 1024-thread slots per CU: ~14 us until the staged tile is there, ~14 us of issue, ~14 us until the last
    // store is acknowledged, per 512 KB), not the memory system's rate.  So the loop is UNROLLED BY FOUR: the owner quads of four
    // consecutive 1 KiB pieces are read together, `cover` gathers branch-free (an unowned cell reads row 0 and drops it: all
    // gathers of the four pieces are in flight at once), then four stores: issue 14 -> 6 us per wave, 0.83 -> 0.785 ms at C5.
    // (Also measured there, not adopted: persistent workgroups with double-buffered staging -- __syncthreads drains the stores --
    // and with a loader wave + bare s_barrier: 0.767 ms at best, for a second kernel structure; 512-thread workgroups; 16-byte
    // tile rows; throttled store issue; one 4 KiB / 16 KiB / 64 KiB block per workgroup in sweep order: all slower.)
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    long u = u0 + lane;
    int n = (int)(u / hw4);
    int c4 = (int)(u - (long)n * hw4);                       // float4 index inside the plane
    constexpr int UNR = 4;
    if ((hw4 % (64 * UNR)) == 0 && (per % (64 * UNR)) == 0) {   // a wave's four pieces never straddle a plane or its share
        for (; u < u1; u += 64 * UNR) {
            int4 f[UNR];
#pragma unroll
            for (int k = 0; k < UNR; ++k) f[k] = *reinterpret_cast<const int4*>(s_first + 4 * (c4 + 64 * k));
            vfloat4 o[UNR];
            int any = 0;
#pragma unroll
            for (int k = 0; k < UNR; ++k) { o[k] = zero4; any |= ~(f[k].x & f[k].y & f[k].z & f[k].w); }
            if (any < 0) {                                    // some cell of the lane's quads has an owner (owners are >= 0, empty is -1)
                const float* xn = xs + n;
#pragma unroll
                for (int k = 0; k < UNR; ++k) {
                    const int32_t fi[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (ADD && walk) {
                            float a = 0.f;
                            if (fi[c] >= 0) {
                                a = xn[fi[c] * ld];
                                for (int32_t m = s_next[fi[c]]; m >= 0; m = s_next[m]) a += xn[m * ld];
                            }
                            o[k][c] = a;
                        } else {                              // cover: the owner; add: the entity, or the row that holds the cell's sum
                            const float a = xn[max(fi[c], 0) * ld];
                            o[k][c] = fi[c] >= 0 ? a : 0.f;
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < UNR; ++k) __builtin_nontemporal_store(o[k], ob + u + 64 * k);
            c4 += 64 * UNR;
            if (c4 >= hw4) { c4 -= hw4; ++n; }
        }
        asm volatile("" :: "v"(pfv));
        return;
    }
    for (; u < u1; u += 64) {
        const int4 f = *reinterpret_cast<const int4*>(s_first + 4 * c4);
        vfloat4 o = zero4;
        if ((f.x & f.y & f.z & f.w) >= 0) {                   // some cell of the quad has an owner
            const int32_t fi[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
                if (fi[c] >= 0) {
                    a = xs[fi[c] * ld + n];
                    if (ADD && walk)
                        for (int32_t m = s_next[fi[c]]; m >= 0; m = s_next[m]) a += xs[m * ld + n];
                }
                o[c] = a;
            }
        }
        __builtin_nontemporal_store(o, ob + u);
        c4 += 64;
        while (c4 >= hw4) { c4 -= hw4; ++n; }
    }
    asm volatile("" :: "v"(pfv));
}

// (Round 4 also tried WAVE tiles -- no LDS, no workgroup barrier, a wave owning 1024 cells x 16 channels with the owners in
// registers, the structure that took the packed pad kernel from 5.0 to 6.2 TB/s: slower here on every shape; removed in round 5.)

// backward: workgroup = (b, group of NG channels); stage NG planes of grad_out in LDS, gather per entity.
// The planes are one contiguous span of grad_out: staged with nontemporal float4 loads, 1024 threads and up to
// 128 KB per workgroup so that every thread has several 16-byte loads in flight (a pure read streams at 7 TB/s on this
// chip; the first version used 4-byte loads from 256 threads and reached 4.4).
__global__ __launch_bounds__(1024) void scatter_bwd_lds_kernel(const float* __restrict__ grad_out,
                                                               const int64_t* __restrict__ location,
                                                               float* __restrict__ grad_x, int M, int N, int H,
                                                               int W, int NG, int xcd_order) {
    extern __shared__ __attribute__((aligned(16))) float s_plane[];  // NG * HW
    // xcd_order (round 4, tune key 38): the workgroups of ONE batch element write 16-byte pieces of the same 128-byte lines of
    // grad_x (an entity's row is N floats; a workgroup owns NG of them).  Dispatched in launch order they land on eight different
    // XCDs -- eight L2s each holding a partial line, each written back masked.  Workgroup i runs on XCD i % 8: with
    // L = (i % 8) * (total / 8) + i / 8 as the logical index, the channel groups of a batch element take consecutive slots of one
    // XCD and their pieces meet in ONE L2 before the line leaves it.
    unsigned bi = blockIdx.y, ci = blockIdx.x;
    if (xcd_order == 2) {
        // round 6: XCD x takes the batch elements b = 8 q + x -- the eight XCDs read EIGHT NEIGHBOURING batch elements at any time
        // (one moving window of grad_out) instead of eight streams total/8 apart, whose relative placement in the memory
        // channels made the kernel 12 % slower on some allocations of grad_out (profiles/r06_scatter_bwd_probe.txt)
        const unsigned nx = gridDim.x, id = blockIdx.x + nx * blockIdx.y, slot = id >> 3;
        bi = (slot / nx) * 8 + (id & 7u);
        ci = slot % nx;
    } else if (xcd_order) {
        const unsigned nx = gridDim.x, total = nx * gridDim.y, id = blockIdx.x + nx * blockIdx.y;
        const unsigned L = (id & 7u) * (total >> 3) + (id >> 3);
        bi = L / nx;
        ci = L - bi * nx;
    }
    const int b = (int)bi;
    const int n0 = (int)ci * NG;
    const int ng = min(NG, N - n0);
    const int HW = H * W;
    const float* __restrict__ g = grad_out + ((size_t)b * N + n0) * HW;
    const int total = ng * HW;
    // Round 5: the location of the thread's first (entity, channel) item is requested BEFORE the planes -- read after the
    // barrier, as it was, every workgroup spent a memory round trip between its staging and its stores (C5 0.8155 -> 0.803 ms, 32 x 32
    // maps 0.2125 -> 0.203, same box, profiles/r05_scatter_pf.txt).  Held RAW: a cell computed here would make the planes wait for it.
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    constexpr int KP = 1;                                  // (four items: registers past 64, one workgroup per CU)
    long yp[KP], xp[KP];                                   // raw: nothing waits for them before the planes are requested
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int i = threadIdx.x + 1024 * j;
        yp[j] = xp[j] = -1;
        if (i < M * ng) {
            const int m = i / ng;
            yp[j] = loc[2 * m];
            xp[j] = loc[2 * m + 1];
        }
    }
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        for (int i = threadIdx.x * 4; i + 3 < total; i += 4096)
            *reinterpret_cast<vfloat4*>(s_plane + i) = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(g + i));
        for (int i = (total & ~3) + threadIdx.x; i < total; i += 1024) s_plane[i] = g[i];
    } else {
        for (int i = threadIdx.x; i < total; i += 1024) s_plane[i] = g[i];
    }
    __syncthreads();
    float* __restrict__ gx = grad_x + (size_t)b * M * N + n0;
    // thread <-> (entity m, channel k) with k fastest so that each entity's ng outputs are contiguous
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int i = threadIdx.x + 1024 * j;
        if (i < M * ng) {
            const int m = i / ng, k = i - m * ng;
            const bool ok = yp[j] >= 0 && yp[j] < H && xp[j] >= 0 && xp[j] < W;
            gx[(size_t)m * N + k] = ok ? s_plane[k * HW + (int)(yp[j] * W + xp[j])] : 0.f;
        }
    }
    for (int i = threadIdx.x + 1024 * KP; i < M * ng; i += 1024) {
        const int m = i / ng, k = i - m * ng;
        const long y = loc[2 * m], xx = loc[2 * m + 1];
        const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
        gx[(size_t)m * N + k] = ok ? s_plane[k * HW + (int)(y * W + xx)] : 0.f;
    }
}

// (Round 4 also tried this backward as a PERSISTENT software-pipelined kernel -- eight loader waves keeping four planes in
// flight by LDS-DMA, four gathering waves, the gathered block written once: 1.25 ms against 0.865 at C5; removed in round 5.)

// fallback for planes too large for LDS: direct gather
__global__ __launch_bounds__(256) void scatter_bwd_direct_kernel(const float* __restrict__ grad_out,
                                                                 const int64_t* __restrict__ location,
                                                                 float* __restrict__ grad_x, long total, int M,
                                                                 int N, int H, int W) {
    const long HW = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int n = (int)(i % N);
        const long bm = i / N;
        const long b = bm / M;
        const long y = location[2 * bm], xx = location[2 * bm + 1];
        const bool ok = y >= 0 && y < H && xx >= 0 && xx < W;
        grad_x[i] = ok ? grad_out[(b * N + n) * HW + y * W + xx] : 0.f;
    }
}

}  // namespace
}  // namespace hpc_rll

namespace hpc_rll { int g_pad_wave = 1; }   // hpc_rll_tune_set key 28: packed Pad1D on wave tiles in output space (0 = the round-3 workgroup kernel)
namespace hpc_rll {
// path switches that tests flip (hpc_rll_tune_set, tune.hip)
int g_scatter_lds_fwd = 1;   // key 17: 1 = LDS-staged streaming forward kernel where it applies, 0 = cells-per-thread kernel everywhere
int g_scatter_npb = 0;       // key 18: channels per workgroup of the LDS-staged kernel (0 = by LDS budget)
int g_scatter_build = 1;     // key 37: 1 = owner table / chain links built inside the forward kernel, 0 = index launch
int g_scatter_bwd_xcd = 1;   // key 38: 1 = XCD-major workgroup order of the backward where its pieces are below a sector pair, 0 = launch order
constexpr int g_scatter_threads = 1024;     // threads per workgroup of the cells-per-thread kernel on maps of >= 4096 cells
int g_scatter_bwd_tile = 0;                 // hpc_rll_tune_set key 40: scatter backward by spatial tiles: 0 = by rule, 1 = never, 2 = wherever it applies
constexpr int g_scatter_bwd_lds_kb = 64;    // planes staged per backward workgroup (profiles/r04_scatter_bwd_lds.txt)
}
using namespace hpc_rll;

extern "C" int hpc_rll_pad_forward(const int64_t* table, float* new_x, int32_t* mask, int64_t n, int m0, int m1,
                                   int m2, int value, void* stream) {
    if (n < 0 || m0 < 0 || m1 < 0 || m2 < 0) return HPC_RLL_EINVAL;
    const long inner = (long)m0 * m1 * m2;
    if (inner >= (1L << 31)) return HPC_RLL_EUNSUPPORTED;
    const long total = n * inner;
    if (total == 0) return HPC_RLL_OK;
    if (!table || !new_x || !mask) return HPC_RLL_EINVAL;
    const bool v4 = (total % 4) == 0 && (reinterpret_cast<uintptr_t>(new_x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(mask) & 15) == 0;
    long blocks = ((v4 ? total / 4 : total) + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    if (v4 && m0 == 1 && m1 == 1 && total < (1L << 32))
        hipLaunchKernelGGL(pad1d4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, new_x, mask,
                           (long)n, (unsigned)m2, 1.0 / (double)m2, (float)value, value);
    else if (v4)
        hipLaunchKernelGGL(pad4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, new_x, mask,
                           (long)n, (unsigned)m0, (unsigned)m1, (unsigned)m2, (float)value, value);
    else
        hipLaunchKernelGGL(pad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, table, new_x, mask,
                           (long)n, (unsigned)m0, (unsigned)m1, (unsigned)m2, (float)value, value);
    return last_error();
}

extern "C" int hpc_rll_unpad_forward(const float* padded, const int64_t* table, float* flat, int64_t n,
                                     int64_t total, int m0, int m1, int m2, void* stream) {
    if (n < 0 || total < 0 || m0 < 0 || m1 < 0 || m2 < 0) return HPC_RLL_EINVAL;
    if (total == 0 || n == 0) return HPC_RLL_OK;
    if (!padded || !table || !flat) return HPC_RLL_EINVAL;
    const long padded_total = (long)n * m0 * m1 * m2;
    if (m0 == 1 && m1 == 1 && m2 > 0 && (padded_total % 4) == 0 && padded_total < (1L << 32) &&
        (reinterpret_cast<uintptr_t>(padded) & 15) == 0) {
        long b4 = (padded_total / 4 + 255) / 256;
        if (b4 > 256L * 16) b4 = 256L * 16;
        hipLaunchKernelGGL(unpad1d4_kernel, dim3((unsigned)b4), dim3(256), 0, (hipStream_t)stream, padded, table, flat,
                           (long)n, (long)total, (unsigned)m2, 1.0 / (double)m2);
        return last_error();
    }
    long blocks = (total + 255) / 256;
    if (blocks > 256L * 16) blocks = 256L * 16;
    hipLaunchKernelGGL(unpad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, padded, table, flat,
                       (long)n, (long)total, (unsigned)m0, (unsigned)m1, (unsigned)m2);
    return last_error();
}

// rows per workgroup of the LDS-staged packed kernels: 16 (tests/tools/micro/padbw.hip at n = 2^20, L = 127: 16 rows and
// one workgroup per 16 rows 270 us, 64 rows 303 us, the per-thread kernel 328 us), fewer when 60 KB of LDS hold fewer
static inline int packed_rows_per_wg(int L) {
    const long cap = (60L * 1024 / 4 - 160) / (L > 0 ? L : 1);
    return (int)(cap > 16 ? 16 : cap);
}

extern "C" int hpc_rll_pad1d_packed_forward(const float* flat, const int64_t* table, float* new_x, int32_t* mask, int64_t n,
                                            int max_len, int value, void* stream) {
    if (n < 0 || max_len < 0) return HPC_RLL_EINVAL;
    if (n == 0 || max_len == 0) return HPC_RLL_OK;
    if (!flat || !table || !new_x || !mask) return HPC_RLL_EINVAL;
    const int RB = packed_rows_per_wg(max_len);
    const bool ok = RB >= 1 && (reinterpret_cast<uintptr_t>(new_x) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(flat) & 3) == 0;
    if (!ok) return hpc_rll_pad_forward(table, new_x, mask, n, 1, 1, max_len, value, stream);
    if (max_len >= 32 && max_len <= 16384 && g_pad_wave) {   // wave tiles in output space (round 4)
        const long ntiles = (long)(((unsigned long)n * (unsigned long)max_len + 1023) / 1024);
        long wb = (ntiles + 3) / 4;
        // (fewer, persistent workgroups -- the shape that makes a pure fill fast, profiles/r04_writebw.txt -- are SLOWER here:
        // 1 / 2 / 4 / 8 workgroups per CU 612 / 466 / 324 / 315 us against 270 for the API call: a tile is a chain of table
        // lookups, source loads, LDS and stores that needs many tiles in flight per CU)
        if (wb > (1L << 20)) wb = 1L << 20;   // (every wave ONE tile: the 8192-workgroup cap of the first version, four tiles per wave, was 2.5 % slower)
        hipLaunchKernelGGL(pad1d_packed_wave_kernel, dim3((unsigned)wb), dim3(256), 0, (hipStream_t)stream, flat, table, new_x, mask,
                           (long)n, (unsigned)max_len, 1.0f / (float)max_len, (float)value, value);
        return last_error();
    }
    long blocks = (n + RB - 1) / RB;
    if (blocks > (1L << 20)) blocks = 1L << 20;
    const size_t lds = ((((size_t)RB * max_len + 8 + 1) & ~(size_t)1)) * 4 + (size_t)(RB + 1) * 8;
    hipLaunchKernelGGL(pad1d_packed_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, flat, table, new_x, mask,
                       (long)n, (unsigned)max_len, 1.0 / (double)max_len, RB, (float)value, value);
    return last_error();
}

extern "C" int hpc_rll_unpad1d_packed_forward(const float* padded, const int64_t* table, float* flat, int64_t n, int64_t total,
                                              int max_len, void* stream) {
    if (n < 0 || total < 0 || max_len < 0) return HPC_RLL_EINVAL;
    if (n == 0 || total == 0 || max_len == 0) return HPC_RLL_OK;
    if (!padded || !table || !flat) return HPC_RLL_EINVAL;
    const int RB = packed_rows_per_wg(max_len);
    const bool ok = RB >= 1 && (reinterpret_cast<uintptr_t>(padded) & 15) == 0 && (reinterpret_cast<uintptr_t>(flat) & 3) == 0;
    if (!ok) return hpc_rll_unpad_forward(padded, table, flat, n, total, 1, 1, max_len, stream);
    long blocks = (n + RB - 1) / RB;
    if (blocks > (1L << 20)) blocks = 1L << 20;
    const size_t lds = ((((size_t)RB * max_len + 8 + 1) & ~(size_t)1)) * 4 + (size_t)(RB + 1) * 8;
    hipLaunchKernelGGL(unpad1d_packed_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, padded, table, flat,
                       (long)n, (long)total, (unsigned)max_len, 1.0 / (double)max_len, RB);
    return last_error();
}

extern "C" int64_t hpc_rll_packed_table_scratch_int64(int64_t n) {
    return n <= 0 ? 1 : (n + kScanChunk - 1) / kScanChunk;
}

extern "C" int hpc_rll_packed_table(const int64_t* lengths, int64_t n, int64_t base, int64_t stride, int64_t* table,
                                    int64_t* scratch, void* stream) {
    if (n < 0) return HPC_RLL_EINVAL;
    if (n == 0) return HPC_RLL_OK;
    if (!lengths || !table || !scratch) return HPC_RLL_EINVAL;
    const long nchunks = (long)((n + kScanChunk - 1) / kScanChunk);
    if (nchunks > 0x7fffffffL) return HPC_RLL_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(packed_chunk_sums_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, lengths, (long)n, scratch);
    if (nchunks <= 1024) {
        hipLaunchKernelGGL(packed_table_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, st, lengths, (long)n, scratch, base,
                           stride, table);
        return last_error();
    }
    hipLaunchKernelGGL(packed_scan_sums_kernel, dim3(1), dim3(256), 0, st, scratch, nchunks);
    hipLaunchKernelGGL(packed_table_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, st, lengths, (long)n, scratch, base,
                       stride, table);
    return last_error();
}

// ---- group splitting policies (host logic; reference: padding.cu:8-108).  sizes: n x dim, row-major, the list is
// already sorted by numel.  Outputs: group_shapes (<= group rows of dim ints), positions (<= group+1 ints).
// Returns the number of groups (>= 1) or a negative error.
namespace {
// Lists sorted by element count: cost(k,i) = key[i-1] * (i-k) with key nondecreasing satisfies the quadrangle
// inequality, so the smallest optimal split point is monotone in i and each DP layer is a divide-and-conquer in
// O(n log n) instead of O(n^2) (the reference's O(group * n^2) DP with stack VLAs breaks long before
// n ~ 1e6).  Ties resolve to the smallest split point, exactly like the quadratic DP.
//   rank 1: key = the length                 -> the exact padded-element count (padding.cu:44-108)
//   rank 2/3 (long lists only): key = numel  -> the cost model of hpc_rll/origin/padding.py:16-17 (arr[end] * count);
//     the per-dimension maxima of padding.cu are not monotone in a numel-sorted list, their exact DP stays quadratic
//     and is kept for n <= 512.  The group SHAPES are always the true per-dimension maxima.
// E: element count of the first k items (E[k] = k for the element-level DP, the run boundaries for the run-level one)
void dc_layer(const int64_t* key, const int64_t* E, const int64_t* prev, int64_t* cur, int32_t* arg, int lo, int hi, int klo,
              int khi, int64_t INF) {
    if (lo > hi) return;
    const int mid = (lo + hi) >> 1;
    int64_t best = INF;
    int32_t bk = klo;
    const int kend = khi < mid - 1 ? khi : mid - 1;
    for (int k = klo; k <= kend; ++k) {
        if (prev[k] >= INF) continue;
        const int64_t c = prev[k] + key[mid - 1] * (E[mid] - E[k]);
        if (c < best) { best = c; bk = k; }
    }
    cur[mid] = best;
    arg[mid] = bk;
    dc_layer(key, E, prev, cur, arg, lo, mid - 1, klo, best >= INF ? khi : bk, INF);
    dc_layer(key, E, prev, cur, arg, mid + 1, hi, best >= INF ? klo : bk, khi, INF);
}

// monotone DP over m items (elements, or runs of equal keys) into M groups; ps[0..M] = item boundaries
void dc_split(const int64_t* key, const int64_t* E, int m, int M, std::vector<int32_t>& ps) {
    const int64_t INF = kSplitInf;
    std::vector<int32_t> pos((size_t)(m + 1) * (M + 1), 0);
    std::vector<int64_t> prev(m + 1, INF), cur(m + 1, INF);
    std::vector<int32_t> arg(m + 1, 0);
    prev[0] = 0;
    for (int j = 1; j <= M; ++j) {
        std::fill(cur.begin(), cur.end(), INF);
        dc_layer(key, E, prev.data(), cur.data(), arg.data(), 1, m, 0, m - 1, INF);
        for (int i = 1; i <= m; ++i) pos[(size_t)i * (M + 1) + j] = arg[i];
        prev.swap(cur);
    }
    ps.assign(M + 1, 0);
    int lp = m;
    ps[M] = m;
    for (int j = M; j >= 1; --j) { lp = pos[(size_t)lp * (M + 1) + j]; ps[j - 1] = lp; }
}
}  // namespace

namespace hpc_rll { int g_split_algo = 0; }   // hpc_rll_tune_set key 22: 0 = runs of equal keys (default), 1 = the
                                              // round-2 paths (quadratic element DP / divide-and-conquer over elements)

extern "C" int hpc_rll_oracle_split_group(const int32_t* sizes, int n, int dim, int group, int32_t* group_shapes,
                                          int32_t* positions) {
    if (!sizes || n <= 0 || dim <= 0 || dim > 3 || group <= 0 || !group_shapes || !positions) return HPC_RLL_EINVAL;
    const int M = group < n ? group : n;  // more groups than tensors is meaningless (the reference would walk off)
    const int64_t INF = kSplitInf;
    // lists sorted by element count (the documented precondition; hpc_rll/rl_utils/padding.py sorts): DP on the key
    std::vector<int64_t> key(n);
    bool sorted = true;
    for (int i = 0; i < n; ++i) {
        int64_t e = 1;
        for (int d = 0; d < dim; ++d) e *= sizes[(size_t)i * dim + d];
        key[i] = e;
        if (i && key[i] < key[i - 1]) { sorted = false; break; }
    }
    // rank 1: key = the length, the DP cost is the exact padded-element count of padding.cu:44-108 for any n.
    // rank 2/3: key = numel is the cost model of hpc_rll/origin/padding.py:16-17 (arr[end] * count); padding.cu's
    // per-dimension maxima are not monotone in a numel-sorted list, their exact DP stays quadratic and is kept for
    // n <= 512.  The group SHAPES are always the true per-dimension maxima.
    const bool by_key = sorted && (dim == 1 || n > 512);
    if (!by_key && n > 20000) return HPC_RLL_EUNSUPPORTED;   // the quadratic DP would take hours: sort the list first
    std::vector<int32_t> ps;
    if (by_key && g_split_algo == 0) {
        // runs of equal keys (pad_group.hpp): O(n) + O(M D log D) for D distinct keys
        std::vector<int64_t> val, E(1, 0);
        for (int i = 0; i < n; ++i) {
            if (val.empty() || key[i] != val.back()) { val.push_back(key[i]); E.push_back(E.back()); }
            ++E.back();
        }
        const int D = (int)val.size();
        std::vector<int64_t> pe(M + 1, 0);
        if (D < M) {
            std::vector<int32_t> gm(M);
            split_runs_few(val.data(), E.data(), D, (int64_t)n, M, pe.data(), gm.data());
        } else {
            std::vector<int32_t> pr;
            dc_split(val.data(), E.data(), D, M, pr);
            for (int g = 0; g <= M; ++g) pe[g] = E[pr[g]];
        }
        ps.assign(pe.begin(), pe.end());
    } else if (by_key && (dim > 1 || n > 512)) {
        std::vector<int64_t> E(n + 1);
        for (int i = 0; i <= n; ++i) E[i] = i;
        dc_split(key.data(), E.data(), n, M, ps);
    } else {
        std::vector<int32_t> pos((size_t)(n + 1) * (M + 1), 0);
        auto P = [&](int i, int j) -> int32_t& { return pos[(size_t)i * (M + 1) + j]; };
        std::vector<int64_t> cost((size_t)(n + 1) * (M + 1), INF);
        auto C = [&](int i, int j) -> int64_t& { return cost[(size_t)i * (M + 1) + j]; };
        C(0, 0) = 0;
        std::vector<int64_t> elems(n);  // elems[k] = prod_d max_{k<=t<=i-1} sizes[t][d]
        for (int i = 1; i <= n; ++i) {
            int32_t mx[3] = {0, 0, 0};
            for (int k = i - 1; k >= 0; --k) {
                int64_t e = 1;
                for (int d = 0; d < dim; ++d) {
                    mx[d] = std::max(mx[d], sizes[(size_t)k * dim + d]);
                    e *= mx[d];
                }
                elems[k] = e;
            }
            for (int j = 1; j <= M; ++j) {
                int64_t best = INF;
                int32_t arg = 0;
                for (int k = 0; k < i; ++k) {
                    if (C(k, j - 1) >= INF) continue;
                    const int64_t c = C(k, j - 1) + elems[k] * (i - k);
                    if (c < best) { best = c; arg = k; }  // strict: the smallest k wins ties, like the reference
                }
                C(i, j) = best;
                P(i, j) = arg;
            }
        }
        int lp = n, lc = M;
        ps.push_back(n);
        while (lp > 0) { lp = P(lp, lc); --lc; ps.push_back(lp); }
        std::reverse(ps.begin(), ps.end());
    }
    const int ng = (int)ps.size() - 1;
    for (int g = 0; g < ng; ++g) {
        for (int d = 0; d < dim; ++d) {
            int32_t m = 0;
            for (int t = ps[g]; t < ps[g + 1]; ++t) m = std::max(m, sizes[(size_t)t * dim + d]);
            group_shapes[g * dim + d] = m;
        }
    }
    for (int g = 0; g <= ng; ++g) positions[g] = ps[g];
    return ng;
}

extern "C" int hpc_rll_sample_split_group(const int32_t* sizes, int n, int dim, int group, uint64_t seed,
                                          int32_t* group_shapes, int32_t* positions) {
    if (!sizes || n <= 0 || dim <= 0 || dim > 3 || group <= 0 || !group_shapes || !positions) return HPC_RLL_EINVAL;
    auto next_rand = [&seed]() {  // splitmix64
        uint64_t z = (seed += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    std::vector<int> cut;
    if (n >= 3) {  // the reference draws from [1, n-2] and divides by zero for n = 2 (padding.cu:17)
        int last = -1;
        for (int i = 0; i < group - 1; ++i) {
            int now = last;
            if (n == 3) now = 1;
            else while (now == last) now = (int)(next_rand() % (uint64_t)(n - 2)) + 1;
            cut.push_back(now);
            last = now;
        }
        std::sort(cut.begin(), cut.end());
    }
    cut.push_back(n - 1);
    int ng = 0, last_idx = -1;
    for (int idx : cut) {
        if (idx <= last_idx) continue;
        int32_t shape[3] = {-1, -1, -1};
        for (int t = last_idx + 1; t <= idx; ++t)
            for (int d = 0; d < dim; ++d) shape[d] = std::max(shape[d], sizes[(size_t)t * dim + d]);
        // same padded shape as the previous group: the reference skips the cut WITHOUT moving last_idx (padding.cu:34-35),
        // so these rows join the NEXT accepted group -- or, after the final cut, the previous one (its end is N).  Mirrored
        // exactly.  What is not reproduced: a repeated cut position (idx <= last_idx) makes the reference emit an
        // EMPTY group of shape -1 (its loop over last_idx+1 .. idx is empty), which GroupPad then cannot allocate; dropped.
        if (ng > 0 && std::memcmp(shape, group_shapes + (ng - 1) * dim, sizeof(int32_t) * dim) == 0) continue;
        for (int d = 0; d < dim; ++d) group_shapes[ng * dim + d] = shape[d];
        positions[ng] = last_idx + 1;
        ++ng;
        last_idx = idx;
    }
    positions[ng] = n;
    return ng;
}

// ---- ScatterConnection
extern "C" int64_t hpc_rll_scatter_workspace_ints(int B, int M, int H, int W) {
    return (int64_t)B * (2 * (int64_t)H * W + M);
}

extern "C" int hpc_rll_scatter_connection_forward(const float* x, const int64_t* location, float* out,
                                                  int32_t* ws, int B, int M, int N, int H, int W, int add,
                                                  void* stream) {
    if (B < 0 || M < 0 || N < 0 || H < 0 || W < 0) return HPC_RLL_EINVAL;
    const long HW = (long)H * W;
    if ((size_t)B * N * HW == 0) return HPC_RLL_OK;
    if (!out || !ws || (M > 0 && (!x || !location))) return HPC_RLL_EINVAL;
    if (HW >= (1L << 31) || B > 65535) return HPC_RLL_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int rc = HPC_RLL_OK;
    bool indexed = false;
    auto build_index = [&]() {   // the paths that read the index from memory
        if (indexed) return;
        const int parts = add ? (M > 256 ? 3 : 1) : 2;
        hipLaunchKernelGGL(scatter_index_kernel, dim3(B), dim3(256), 0, st, location, ws, M, H, W, parts);
        rc = last_error();
        indexed = true;
    };
    const bool v4 = (HW % 4) == 0 && (N % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    // LDS-staged streaming kernel: the owner table (HW ints) and an M x NPB tile of x must fit in LDS with room for
    // three workgroups per CU (so that one workgroup's staging overlaps the others' stores).  In-process A/B
    // (tests/tools/r02_scatter_probe.py, profiles/r02_scatter_probe.json): configs[4] cover 0.953 -> 0.919 ms (4.81 ->
    // 4.99 TB/s), reference test shape (16x16 maps) cover 51 -> 46 us, add 68 -> 53 us; `add` on large maps is a tie at
    // 64 channels per workgroup (0.893 ms, 5.13 TB/s) and a loss at 32, so it keeps the cells-per-thread kernel there.
    // in-kernel index build (round 4, key 37): cover for every M (an LDS atomic per entity); add where the LDS kernel is taken
    // anyway (maps up to 8 KB) and the quadratic chain search is small against the workgroup's output
    // (add: measured -13 % at 32 x 32 maps with 128 entities, +15 % at the reference's 16 x 16 test shape with 256, where four
    // workgroups per batch element each repeat a build that outweighs their 64 KB of output)
    const bool build = g_scatter_build && W > 0 && (!add || (M <= 256 && HW >= 1024));
    // (`add` on large maps, rounds 4-5 before the prefetch: the LDS kernel + build at 32 / 64 channels per workgroup lost to the
    // cells-per-thread kernel behind the index launch, 0.837 / 0.89 against 0.828 ms -- DESIGN.md 4.5)
    // (round 5, with the staging prefetch: at C5 the LDS kernel + build at 32 channels per workgroup runs `add` in 0.811 ms against
    // 0.823 for the cells-per-thread kernel behind its index launch -- and has no second launch: taken wherever the tables are built
    // in the kernel)
    const bool lds_pays = !add || HW * 4 <= 8 * 1024 || g_scatter_npb != 0 || build;
    if (g_scatter_lds_fwd && lds_pays && (HW % 4) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && M > 0 && B <= 65535 &&
        (size_t)HW * 4 <= 32 * 1024) {
        const size_t fixed = (size_t)HW * 4 + (add ? (size_t)M * 4 : 0) +
                             (add && build ? (size_t)((M + 3) & ~3) * 4 + 16 + ((size_t)2 * kScatterCollRows + 2) * 4 : 0);   // + cell list, collision list
        const size_t rows = (size_t)M + (add && build ? kScatterCollRows : 0);   // x tile rows (+ the rows of the cells with several entities)
        const bool big = g_scatter_npb != 0;
        const size_t cap = (size_t)(big ? 100 : add && build ? 60 : 52) * 1024;   // two workgroups per CU
        int npb = 0;
        static const int kNpb[5] = {64, 32, 16, 8, 4};
        for (int i = 0; i < 5 && !npb; ++i) {
            const int c = g_scatter_npb ? g_scatter_npb : kNpb[i];
            if (c <= 64 && c >= 1 && rows * (c + 1) * 4 + 16 + fixed <= cap) npb = c;
            if (g_scatter_npb) break;
        }
        if (npb > N) npb = (N + 3) / 4 * 4;
        if (npb >= 1 && rows * (npb + 1) * 4 + 16 + fixed <= cap) {
            const size_t lds = ((rows * (npb + 1) + 3) & ~(size_t)3) * 4 + fixed;
            const void* k = add ? (build ? (const void*)scatter_out_lds_kernel<true, true> : (const void*)scatter_out_lds_kernel<true, false>)
                                : (build ? (const void*)scatter_out_lds_kernel<false, true> : (const void*)scatter_out_lds_kernel<false, false>);
            if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return last_error();
            if (!build) {
                build_index();
                if (rc) return rc;
            }
            const dim3 grid((N + npb - 1) / npb, B);
            // staging prefetch distance (see the kernel): half of the workgroups the chip holds -- two 1024-thread workgroups per
            // CU, fewer when LDS limits -- rounded to a multiple of 8
            static const int pf_env = getenv("HPC_RLL_SCATTER_PF") ? atoi(getenv("HPC_RLL_SCATTER_PF")) : -1;
            const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
            // CU count of the device the launch goes to (was the literal 256), read once per device
            static int cu_cache[16] = {0};
            int devid = 0;
            (void)hipGetDevice(&devid);
            int cus = devid >= 0 && devid < 16 ? cu_cache[devid] : 0;
            if (cus <= 0) {
                if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, devid) != hipSuccess || cus <= 0) cus = 256;
                if (devid >= 0 && devid < 16) cu_cache[devid] = cus;
            }
            const int pf_wgs = pf_env >= 0 ? pf_env : (cus * per_cu / 2) & ~7;
            if (add && build) hipLaunchKernelGGL((scatter_out_lds_kernel<true, true>), grid, dim3(1024), lds, st, x, ws, out, M, N, (int)HW, npb, location, W, pf_wgs);
            else if (add) hipLaunchKernelGGL((scatter_out_lds_kernel<true, false>), grid, dim3(1024), lds, st, x, ws, out, M, N, (int)HW, npb, location, W, pf_wgs);
            else if (build) hipLaunchKernelGGL((scatter_out_lds_kernel<false, true>), grid, dim3(1024), lds, st, x, ws, out, M, N, (int)HW, npb, location, W, pf_wgs);
            else hipLaunchKernelGGL((scatter_out_lds_kernel<false, false>), grid, dim3(1024), lds, st, x, ws, out, M, N, (int)HW, npb, location, W, pf_wgs);
            return last_error();
        }
    }
    const int tpb = (v4 && HW >= 4096) ? g_scatter_threads : 256;   // threads per block of the 4-wide kernel
    const int cell_blocks = (int)((HW + (v4 ? 4 * tpb - 1 : 255)) / (v4 ? 4 * tpb : 256));
    // enough workgroups to cover the chip: split the channel axis when B * cell_blocks is small
    int n_per_block = N;
    while (n_per_block > 4 && (long)B * cell_blocks * ((N + n_per_block - 1) / n_per_block) < 2048) n_per_block = (n_per_block / 2 + 3) / 4 * 4;
    const dim3 grid(cell_blocks, (N + n_per_block - 1) / n_per_block, B);
    if (v4) {
        build_index();
        if (rc) return rc;
        if (add) hipLaunchKernelGGL(scatter_out4_kernel<true>, grid, dim3(tpb), 0, st, x, ws, out, M, N, (int)HW, n_per_block);
        else hipLaunchKernelGGL(scatter_out4_kernel<false>, grid, dim3(tpb), 0, st, x, ws, out, M, N, (int)HW, n_per_block);
        return last_error();
    }
    build_index();
    if (rc) return rc;
    const bool vec = (N % 4) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (add) hipLaunchKernelGGL(scatter_out_kernel<true>, grid, dim3(256), 0, st, x, ws, out, M, N, (int)HW, n_per_block, vec);
    else hipLaunchKernelGGL(scatter_out_kernel<false>, grid, dim3(256), 0, st, x, ws, out, M, N, (int)HW, n_per_block, vec);
    return last_error();
}

// Round 6: backward by SPATIAL tiles.  Workgroup = (batch element, TR map rows): it stages rows y0 .. y0+TR-1 of ALL N planes (N
// contiguous pieces of TR*W floats, 1 KB each at C5) and owns every entity whose location falls into those rows -- it writes
// their whole N-float rows of grad_x (256 contiguous bytes at C5: full lines, one writer per line) instead of a 16-byte
// piece of every entity's row from each of N/NG workgroups that have to meet in one XCD's L2.  Every workgroup scans the
// batch element's M locations (4 KB, from L2) to find its entities; the rows of out-of-range entities (zeros) are dealt over the tiles.
// LDS: plane n at s_tile + n*cells, cell c at position c ^ (4*(n & swz)) -- a gather of one cell over the planes (stride
// cells: one bank) spreads over 8 banks, the float4 staging stays aligned.  The gather result is the plane kernel's, bit for bit.
__global__ __launch_bounds__(1024) void scatter_bwd_tile_kernel(const float* __restrict__ grad_out,
                                                                const int64_t* __restrict__ location,
                                                                float* __restrict__ grad_x, int M, int N, int H, int W,
                                                                int TR, int swz, int xcd_order) {
    extern __shared__ __attribute__((aligned(16))) float s_tile[];   // [N][cells], then the entity list (int32 x 2 per entry) and its counter
    unsigned bi = blockIdx.y, ti = blockIdx.x;
    if (xcd_order == 2) {   // the tiles of a batch element on one XCD, neighbouring batch elements on the eight XCDs
        const unsigned nx = gridDim.x, id = blockIdx.x + nx * blockIdx.y, slot = id >> 3;
        bi = (slot / nx) * 8 + (id & 7u);
        ti = slot % nx;
    } else if (xcd_order) {   // the tiles of a batch element on one XCD (they read neighbouring 1 KB pieces of the same planes)
        const unsigned nx = gridDim.x, total = nx * gridDim.y, id = blockIdx.x + nx * blockIdx.y;
        const unsigned L = (id & 7u) * (total >> 3) + (id >> 3);
        bi = L / nx;
        ti = L - bi * nx;
    }
    const int b = (int)bi, y0 = (int)ti * TR;
    const int rows = min(TR, H - y0), cells = TR * W, live = rows * W;
    const int HW = H * W;
    int* const s_cnt = reinterpret_cast<int*>(s_tile + (size_t)N * cells);
    int* const s_list = s_cnt + 4;                       // [M][2]: entity, cell inside the tile (-1: write zeros)
    if (threadIdx.x == 0) *s_cnt = 0;
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    // locations first (raw), planes next: nothing waits before every request is out
    long yy[4], xx[4];
    const int nl = (M + 1023) / 1024;                    // (M <= 4096 on this path)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = threadIdx.x + 1024 * j;
        yy[j] = xx[j] = 0;
        if (j < nl && m < M) { yy[j] = loc[2 * m]; xx[j] = loc[2 * m + 1]; }
    }
    const float* __restrict__ g = grad_out + (size_t)b * N * HW + (size_t)y0 * W;
    const int q4 = live >> 2;                            // float4 per plane piece (W % 4 == 0 on this path)
    const int total4 = N * q4;
    for (int i = threadIdx.x; i < total4; i += 1024) {
        const int n = i / q4, c = (i - n * q4) * 4;
        const vfloat4 v = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(g + (size_t)n * HW + c));
        *reinterpret_cast<vfloat4*>(s_tile + n * cells + (c ^ (4 * (n & swz)))) = v;
    }
    __syncthreads();   // (the counter's zero is visible)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = threadIdx.x + 1024 * j;
        if (j < nl && m < M) {
            const bool ok = yy[j] >= 0 && yy[j] < H && xx[j] >= 0 && xx[j] < W;
            const bool mine = ok ? (yy[j] >= y0 && yy[j] < y0 + rows) : (unsigned)m % gridDim.x == ti;   // zero rows: dealt over the tiles
            if (mine) {
                const int slot = atomicAdd(s_cnt, 1);
                s_list[2 * slot] = m;
                s_list[2 * slot + 1] = ok ? (int)((yy[j] - y0) * W + xx[j]) : -1;
            }
        }
    }
    __syncthreads();
    const int cnt = *s_cnt;
    float* __restrict__ gx = grad_x + (size_t)b * M * N;
    for (int i = threadIdx.x; i < cnt * N; i += 1024) {
        const int e = i / N, k = i - e * N;
        const int m = s_list[2 * e], c = s_list[2 * e + 1];
        gx[(size_t)m * N + k] = c >= 0 ? s_tile[k * cells + (c ^ (4 * (k & swz)))] : 0.f;
    }
}

extern "C" int hpc_rll_scatter_connection_backward(const float* grad_out, const int64_t* location, float* grad_x,
                                                   int B, int M, int N, int H, int W, void* stream) {
    if (B < 0 || M < 0 || N < 0 || H < 0 || W < 0) return HPC_RLL_EINVAL;
    if ((size_t)B * M * N == 0) return HPC_RLL_OK;
    if (!location || !grad_x) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    // an empty map: every location is out of range, every gradient row is zero (grad_out has no elements and may be null)
    if (HW == 0) return (int)hipMemsetAsync(grad_x, 0, sizeof(float) * (size_t)B * M * N, st);
    if (!grad_out) return HPC_RLL_EINVAL;
    const long plane_bytes = HW * 4;
    // Which kernel (round 6, profiles/r06_scatter_bwd_probe.txt).  The plane kernel writes NG*4-byte pieces of every entity's row
    // from N/NG workgroups; where those pieces are 16 bytes or less (maps of 4096 cells and more) AND the entity rows are a
    // sizeable part of the traffic (M*16 >= H*W: C5 exactly), the spatial-tile kernel -- full rows from one writer -- is faster
    // (M = 1024, N = 128: 1086 -> 946 us; C5: 741 -> 724 us mean of 50 buffer sets on five boxes) and half as sensitive to where
    // grad_out and grad_x lie relative to each other (C5: +5 % instead of +12 % on an unlucky pair).  Few entities per map
    // (B = 8192, M = 64: 656 vs 710 us) and small maps (32 x 32: 249 vs 288) keep the plane kernel.  Key 40 forces either.
    const long pieces = std::min<long>(N, std::max<long>(1, (long)g_scatter_bwd_lds_kb * 1024 / std::max<long>(plane_bytes, 1))) * 4;
    const bool tile_rule = g_scatter_bwd_tile == 2 || (g_scatter_bwd_tile == 0 && pieces <= 16 && (long)M * 16 >= HW);
    if (tile_rule && (W % 4) == 0 && M <= 4096 && B <= 65535 && (reinterpret_cast<uintptr_t>(grad_out) & 15) == 0 &&
        (long)N * W * 4 <= 64 * 1024) {
        int TR = (int)(64L * 1024 / ((long)N * W * 4));
        if (TR > H) TR = H;
        int p2 = 1;
        while (p2 * 2 <= TR) p2 *= 2;                    // a power of two (the swizzle XORs inside a plane piece)
        TR = p2;
        const int cells = TR * W;
        int swz = 0;
        if ((cells & (cells - 1)) == 0 && cells >= 8) { swz = cells / 4 - 1; if (swz > 63) swz = 63; }
        const size_t lds = (size_t)N * cells * 4 + 16 + (size_t)M * 8;
        // plane pieces of a tile: 1 KB and more where entity rows are a sixteenth of the planes (C5), 512 bytes only where they are a
        // quarter (N = 128, M = 1024: 1070 -> 945 us; N = 128, M = 256 keeps the plane kernel: 714 vs 747)
        const bool pays = ((long)cells * 4 >= 1024 && (long)M * 16 >= HW) || ((long)cells * 4 >= 512 && (long)M * 4 >= HW);
        if (pays || g_scatter_bwd_tile == 2) {
            const dim3 grid((H + TR - 1) / TR, B);
            if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)scatter_bwd_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)lds) != hipSuccess)
                return last_error();
            hipLaunchKernelGGL(scatter_bwd_tile_kernel, grid, dim3(1024), lds, st, grad_out, location, grad_x, M, N, H, W, TR, swz, 0);
            return last_error();
        }
    }
    if (plane_bytes <= 128 * 1024 && B <= 65535) {
        int NG = (int)std::min<long>(N, std::max<long>(1, (long)g_scatter_bwd_lds_kb * 1024 / plane_bytes));
        if (NG > 32) NG = 32;
        const dim3 grid((N + NG - 1) / NG, B);
        const size_t lds = (size_t)NG * plane_bytes;
        if (lds > 64 * 1024 &&
            hipFuncSetAttribute((const void*)scatter_bwd_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
                hipSuccess)
            return last_error();
        // measured (profiles/r04_scatter_bwd_xcd.txt): 16-byte pieces (64 x 64 maps, NG = 4) -8 % at C5 and -31 % at M = 1024, N = 128;
        // 64-byte pieces (32 x 32 maps) +6 %, whole lines (16 x 16) +4 % -- only pieces below a 64-byte sector pair profit
        const int xcd_order = (g_scatter_bwd_xcd == 2 || (g_scatter_bwd_xcd == 1 && NG * 4 <= 32)) &&
                              ((long)grid.x * grid.y) % 8 == 0 && grid.x > 1;
        // (round 6) with B a multiple of 8 the XCDs take NEIGHBOURING batch elements (order 2: -1 % on every shape measured)
        const int xo2 = (xcd_order && B % 8 == 0) ? 2 : xcd_order;
        hipLaunchKernelGGL(scatter_bwd_lds_kernel, grid, dim3(1024), lds, st, grad_out, location, grad_x, M, N, H, W, NG, xo2);
    } else {
        const long total = (long)B * M * N;
        long blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(scatter_bwd_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, st, grad_out, location,
                           grad_x, total, M, N, H, W);
    }
    return last_error();
}
