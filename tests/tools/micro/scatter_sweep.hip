// Round 5: what bounds the ScatterConnection forward at C5 (B = 4096, M = 256, N = 64, 64 x 64 maps:
// 4.29 GB of output written once)?  The shipped kernel (scatter_out_lds_kernel, csrc/pad_scatter.hip) stands at 5.35 TB/s
// against write streams measured at 6.3-6.6 TB/s (profiles/r04_writebw.txt).  This binary times, on the SAME output buffer:
//   S0  the shipped kernel's EXACT store pattern and launch shape with nothing else (no staging, no LDS reads, no gathers):
//       grid (N / 32, B) x 1024 threads, 50 KB of dynamic LDS (two workgroups per CU), every wave streams its own contiguous
//       32 KB with 16-byte nontemporal stores;
//   S1  the best pure-write loop known (256 workgroups x 256 threads, grid-stride, 4 KiB per workgroup and sweep);
//   S2  G workgroups x 256 threads, one 16 KiB PLANE per workgroup and iteration (four 4 KiB store steps), static stride G;
//   S3  the same with PL = 4 planes (64 KiB contiguous) per iteration;
//   C   `cover` prototypes on the S2 / S3 patterns: the owner table of batch element b is built in LDS per iteration (one LDS
//       atomic max per entity, double-buffered table, one barrier), owned cells gather x[b, m, n] straight from memory, the
//       next iteration's locations are requested before the barrier.  Verified against a host loop (last m wins).
// Prints GB/s of OUTPUT bytes (4.29 GB) per kernel, best and median of 7.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef int vint4 __attribute__((ext_vector_type(4)));
__device__ int g_throttle = 0;   // > 0: at most that many vector-memory operations outstanding per wave in the stream loops (1, 2, 4, 8, 16)
__device__ long long* g_prof = nullptr;   // [4096 sampled workgroups][3]: entry, first store, last store (100 MHz wall clock)
#define PROF(i) do { if (g_prof && threadIdx.x == 0 && blockIdx.x == 0) g_prof[(size_t)blockIdx.y * 3 + (i)] = wall_clock64(); } while (0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- S0: the shipped pattern, stores only
__global__ __launch_bounds__(1024) void s0_kernel(float* __restrict__ out, int N, int HW, int npb) {
    extern __shared__ float s_dyn[];
    const int b = blockIdx.y, n0 = blockIdx.x * npb;
    const int nn = min(npb, N - n0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;
    const long units = (long)nn * hw4, per = (units + 15) / 16;
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    if (threadIdx.x == 5000) s_dyn[0] = 0.f;   // (keeps the dynamic LDS allocation alive)
    const vfloat4 v = {1.f, 2.f, 3.f, 4.f};
    for (long u = u0 + lane; u < u1; u += 64) __builtin_nontemporal_store(v, ob + u);
}

// ---- S1: grid-stride 16-byte stores
__global__ __launch_bounds__(256) void s1_kernel(float* __restrict__ out, long n4) {
    const vfloat4 v = {1.f, 2.f, 3.f, 4.f};
    vfloat4* o = reinterpret_cast<vfloat4*>(out);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, o + i);
}

// ---- S2 / S3: PL planes of HW floats per workgroup and iteration
template <int PL, bool NT>
__global__ __launch_bounds__(256) void s23_kernel(float* __restrict__ out, int HW, long nblk) {
    const vfloat4 v = {1.f, 2.f, 3.f, 4.f};
    const int steps = PL * (HW >> 10);
    for (long k = blockIdx.x; k < nblk; k += gridDim.x) {
        vfloat4* o = reinterpret_cast<vfloat4*>(out + (size_t)k * PL * HW) + threadIdx.x;
        for (int j = 0; j < steps; ++j) {
            if (NT) __builtin_nontemporal_store(v, o + 256 * j);
            else o[256 * j] = v;
        }
    }
}

// ---- C: cover on the sweep patterns.  HW % 1024 == 0, M <= 256, N % PL == 0.
template <int PL>
__global__ __launch_bounds__(256) void cover_sweep_kernel(const float* __restrict__ x, const int64_t* __restrict__ loc,
                                                          float* __restrict__ out, int M, int N, int HW, int W, long nblk) {
    extern __shared__ int tab[];   // [2][HW]
    const int H = HW / W, ng = N / PL, tid = threadIdx.x;
    for (int i = tid; i < 2 * HW; i += 256) tab[i] = -1;
    auto cell_of = [&](long k) -> int {
        if (k >= nblk || tid >= M) return -1;
        const int b = (int)(k / ng);
        const long y = loc[((size_t)b * M + tid) * 2], xx = loc[((size_t)b * M + tid) * 2 + 1];
        return (y >= 0 && y < H && xx >= 0 && xx < W) ? (int)(y * W + xx) : -1;
    };
    long k = blockIdx.x;
    int cell = cell_of(k);
    int buf = 0;
    __syncthreads();
    for (; k < nblk; k += gridDim.x) {
        int* const t = tab + buf * HW;
        if (cell >= 0) atomicMax(t + cell, tid);
        const int cell_next = cell_of(k + gridDim.x);      // in flight over the barrier and the stores
        __syncthreads();
        const int b = (int)(k / ng), n0 = (int)(k % ng) * PL;
        const float* __restrict__ xb = x + (size_t)b * M * N + n0;
        vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
        const int hw4 = HW >> 2;
        for (int j = 0; j < (HW >> 10); ++j) {
            const int c4 = tid + 256 * j;
            const vint4 f = reinterpret_cast<const vint4*>(t)[c4];
            reinterpret_cast<vint4*>(t)[c4] = vint4{-1, -1, -1, -1};   // ready for the build two iterations from now
            vfloat4 o[PL];
#pragma unroll
            for (int p = 0; p < PL; ++p) o[p] = vfloat4{0.f, 0.f, 0.f, 0.f};
            if ((f.x & f.y & f.z & f.w) >= 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (f[c] >= 0) {
                        if (PL == 4) {
                            const vfloat4 g = *reinterpret_cast<const vfloat4*>(xb + (size_t)f[c] * N);
#pragma unroll
                            for (int p = 0; p < PL; ++p) o[p][c] = g[p];
                        } else {
#pragma unroll
                            for (int p = 0; p < PL; ++p) o[p][c] = xb[(size_t)f[c] * N + p];
                        }
                    }
            }
#pragma unroll
            for (int p = 0; p < PL; ++p) __builtin_nontemporal_store(o[p], ob + (size_t)p * hw4 + c4);
        }
        cell = cell_next;
        buf ^= 1;
    }
}


// ---- L: the shipped `cover` kernel (scatter_out_lds_kernel<false, true>, csrc/pad_scatter.hip) with ablations:
// ABL 0 = as shipped; 1 = the streaming loop stores zeros without touching LDS (staging + table build still run);
// 2 = no x tile staging; 3 = branch-free gathers (four LDS reads per quad, always); 4 = no staging and no build at all
// (empty table: every quad takes the one-read fast path); 5 = as shipped but the x tile is read with PLAIN loads; 6 = the x
// tile is loaded but only one word per thread goes to LDS; 7 = the x tile is read as one contiguous 32 KB (wrong data);
// 8 = x staged, no table build (empty table).
template <int ABL>
__global__ __launch_bounds__(1024) void lds_cover_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW,
                                                         int npb, const int64_t* __restrict__ location, int W) {
    extern __shared__ float s_dyn[];
    PROF(0);
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * npb;
    const int nn = min(npb, N - n0);
    const int ld = npb + 1;
    float* xs = s_dyn;
    int32_t* s_first = reinterpret_cast<int32_t*>(s_dyn + (((size_t)M * ld + 3) & ~(size_t)3));
    const float* __restrict__ xb = x + (size_t)b * M * N + n0;
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    const int H = HW / W;
    long y0 = -1, x0 = -1;
    if (ABL != 4 && (int)threadIdx.x < M) { y0 = loc[2 * threadIdx.x]; x0 = loc[2 * threadIdx.x + 1]; }
    const vint4 m1 = {-1, -1, -1, -1};
    for (int c4 = threadIdx.x; c4 < (HW >> 2); c4 += 1024) reinterpret_cast<vint4*>(s_first)[c4] = m1;
    if (ABL != 2 && ABL != 4) {
        const int q = nn >> 2;
        for (int e = threadIdx.x; e < M * q; e += 1024) {
            const int m = e / q, j = e - m * q;
            const float* src = ABL == 7 ? x + ((size_t)b * (N / npb) + blockIdx.x) * M * npb + 4 * e : xb + (size_t)m * N + 4 * j;
            const vfloat4 t = ABL == 5 ? *reinterpret_cast<const vfloat4*>(src) : __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(src));
            float* d = xs + m * ld + 4 * j;
            if (ABL == 6) { xs[threadIdx.x] = t.x + t.y + t.z + t.w; }
            else { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
        }
    }
    __syncthreads();
    if (ABL != 4 && ABL != 8) {
        const int m = threadIdx.x;
        if (m < M) {
            const int c = (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) ? (int)(y0 * W + x0) : -1;
            if (c >= 0) atomicMax(s_first + c, m);
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;
    const long units = (long)nn * hw4;
    const long per = (units + 15) / 16;
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    long u = u0 + lane;
    int n = (int)(u / hw4);
    int c4 = (int)(u - (long)n * hw4);
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int thr = g_throttle;
    PROF(1);
    for (; u < u1; u += 64) {
        vfloat4 o = zero4;
        if (ABL != 1) {
            const int4 f = *reinterpret_cast<const int4*>(s_first + 4 * c4);
            if (ABL == 3) {
                const int32_t fi[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = xs[max(fi[c], 0) * ld + n];
                    o[c] = fi[c] >= 0 ? a : 0.f;
                }
            } else if ((f.x & f.y & f.z & f.w) >= 0) {
                const int32_t fi[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f;
                    if (fi[c] >= 0) a = xs[fi[c] * ld + n];
                    o[c] = a;
                }
            }
        }
        __builtin_nontemporal_store(o, ob + u);
        if (thr == 1) __builtin_amdgcn_s_waitcnt(0x0F71);
        else if (thr == 2) __builtin_amdgcn_s_waitcnt(0x0F72);
        else if (thr == 4) __builtin_amdgcn_s_waitcnt(0x0F74);
        else if (thr == 8) __builtin_amdgcn_s_waitcnt(0x0F78);
        else if (thr == 16) __builtin_amdgcn_s_waitcnt(0x4F70);
        c4 += 64;
        while (c4 >= hw4) { c4 -= hw4; ++n; }
    }
    PROF(2);
}

// ---- L6 / L7: the shipped structure with NT threads per workgroup (512: three workgroups per CU by LDS) and an x tile with
// 16-byte aligned rows (stride npb + 4: ONE ds_write_b128 per staged float4 instead of four ds_write_b32), plain loads.
template <int NT, bool VEC>
__global__ __launch_bounds__(NT) void lds_cover_nt_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW,
                                                          int npb, const int64_t* __restrict__ location, int W) {
    extern __shared__ float s_dyn[];
    constexpr int NWV = NT / 64;
    const int b = blockIdx.y, n0 = blockIdx.x * npb, nn = min(npb, N - n0);
    const int ld = VEC ? npb + 4 : npb + 1;
    float* xs = s_dyn;
    int32_t* s_first = reinterpret_cast<int32_t*>(s_dyn + (((size_t)M * ld + 3) & ~(size_t)3));
    const float* __restrict__ xb = x + (size_t)b * M * N + n0;
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    const int H = HW / W;
    long y0 = -1, x0 = -1;
    if ((int)threadIdx.x < M) { y0 = loc[2 * threadIdx.x]; x0 = loc[2 * threadIdx.x + 1]; }
    const vint4 m1 = {-1, -1, -1, -1};
    for (int c4 = threadIdx.x; c4 < (HW >> 2); c4 += NT) reinterpret_cast<vint4*>(s_first)[c4] = m1;
    {
        const int q = nn >> 2;
        for (int e = threadIdx.x; e < M * q; e += NT) {
            const int m = e / q, j = e - m * q;
            const vfloat4 t = *(reinterpret_cast<const vfloat4*>(xb + (size_t)m * N) + j);
            float* d = xs + m * ld + 4 * j;
            if (VEC) *reinterpret_cast<vfloat4*>(d) = t;
            else { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
        }
    }
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += NT) {
        const long y = m == (int)threadIdx.x ? y0 : loc[2 * m], xx = m == (int)threadIdx.x ? x0 : loc[2 * m + 1];
        const int c = (y >= 0 && y < H && xx >= 0 && xx < W) ? (int)(y * W + xx) : -1;
        if (c >= 0) atomicMax(s_first + c, m);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;
    const long units = (long)nn * hw4, per = (units + NWV - 1) / NWV;
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    long u = u0 + lane;
    int n = (int)(u / hw4);
    int c4 = (int)(u - (long)n * hw4);
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (; u < u1; u += 64) {
        vfloat4 o = zero4;
        const int4 f = *reinterpret_cast<const int4*>(s_first + 4 * c4);
        if ((f.x & f.y & f.z & f.w) >= 0) {
            const int32_t fi[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
                if (fi[c] >= 0) a = xs[fi[c] * ld + n];
                o[c] = a;
            }
        }
        __builtin_nontemporal_store(o, ob + u);
        c4 += 64;
        while (c4 >= hw4) { c4 -= hw4; ++n; }
    }
}

// ---- L8: PERSISTENT workgroups, double-buffered staging: a workgroup walks items (b, channel group) it = blockIdx, + gridDim,
// ...; the next item's locations and x tile are requested BEFORE the current item is streamed and land in registers meanwhile;
// its owner table is cleared before and built after the stream.  The staging latency (what L2 above removes: 0.81 -> 0.70 ms)
// is paid once per workgroup instead of once per 512 KB of output.  M <= NT, x tile of at most XR float4 per thread.
template <int NT, int XR>
__global__ __launch_bounds__(NT) void lds_cover_persist_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW,
                                                               int npb, const int64_t* __restrict__ location, int W, int nitems) {
    extern __shared__ float s_dyn[];
    constexpr int NWV = NT / 64;
    const int ld = npb + 4, ngr = N / npb, H = HW / W, q = npb >> 2;
    const size_t xs_floats = ((size_t)M * ld + 3) & ~(size_t)3;
    float* const xs0 = s_dyn;
    float* const xs1 = s_dyn + xs_floats;
    int32_t* const tab0 = reinterpret_cast<int32_t*>(s_dyn + 2 * xs_floats);
    int32_t* const tab1 = tab0 + HW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const vint4 m1 = {-1, -1, -1, -1};
    vfloat4 xr[XR];
    int cell = -1;
    auto request = [&](int it) {     // global loads of item `it` into registers
        const int b = it / ngr, n0 = (it % ngr) * npb;
        if (tid < M) {
            const long y = location[((size_t)b * M + tid) * 2], xx = location[((size_t)b * M + tid) * 2 + 1];
            cell = (y >= 0 && y < H && xx >= 0 && xx < W) ? (int)(y * W + xx) : -1;
        } else cell = -1;
        const float* __restrict__ xb = x + (size_t)b * M * N + n0;
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + r * NT;
            if (e < M * q) { const int m = e / q, j = e - m * q; xr[r] = *(reinterpret_cast<const vfloat4*>(xb + (size_t)m * N) + j); }
        }
    };
    auto commit = [&](float* xs, int32_t* tab) {   // registers -> LDS tile, owner table (two barriers)
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + r * NT;
            if (e < M * q) { const int m = e / q, j = e - m * q; *reinterpret_cast<vfloat4*>(xs + m * ld + 4 * j) = xr[r]; }
        }
        __syncthreads();           // every thread's table clear (before its stream) and tile writes are done
        if (cell >= 0) atomicMax(tab + cell, tid);
        __syncthreads();
    };
    int it = blockIdx.x;
    if (it >= nitems) return;
    for (int c4 = tid; c4 < (HW >> 2); c4 += NT) reinterpret_cast<vint4*>(tab0)[c4] = m1;
    request(it);
    commit(xs0, tab0);
    int cur = 0;
    const int hw4 = HW >> 2;
    const long units = (long)npb * hw4, per = (units + NWV - 1) / NWV;
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (; it < nitems; it += gridDim.x) {
        float* const xs = cur ? xs1 : xs0;
        int32_t* const tab = cur ? tab1 : tab0;
        float* const xsn = cur ? xs0 : xs1;
        int32_t* const tabn = cur ? tab0 : tab1;
        const bool more = it + (int)gridDim.x < nitems;
        if (more) {
            request(it + gridDim.x);
            for (int c4 = tid; c4 < hw4; c4 += NT) reinterpret_cast<vint4*>(tabn)[c4] = m1;
        }
        const int b = it / ngr, n0 = (it % ngr) * npb;
        vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
        const long u0 = wave * per, u1 = min(units, u0 + per);
        long u = u0 + lane;
        int n = (int)(u / hw4);
        int c4 = (int)(u - (long)n * hw4);
        for (; u < u1; u += 64) {
            vfloat4 o = zero4;
            const int4 f = *reinterpret_cast<const int4*>(tab + 4 * c4);
            if ((f.x & f.y & f.z & f.w) >= 0) {
                const int32_t fi[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f;
                    if (fi[c] >= 0) a = xs[fi[c] * ld + n];
                    o[c] = a;
                }
            }
            __builtin_nontemporal_store(o, ob + u);
            c4 += 64;
            while (c4 >= hw4) { c4 -= hw4; ++n; }
        }
        if (more) commit(xsn, tabn);
        cur ^= 1;
    }
}

// ---- S0r: S0 (stores only, the shipped launch shape) + every workgroup READS its x tile (MODE 0: the shipped pattern, 256 row
// halves of 128 B at a 256 B stride; MODE 1: one contiguous 32 KB) at its start and does nothing with it: what does mixing
// 6 % of reads into the write stream cost by itself?  MODE 2: the reads come from a 4 MB (L2 / Infinity Cache resident) buffer.
template <int MODE>
__global__ __launch_bounds__(1024) void s0r_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW, int npb) {
    extern __shared__ float s_dyn[];
    PROF(0);
    const int b = blockIdx.y, n0 = blockIdx.x * npb;
    const int nn = min(npb, N - n0);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;
    const long units = (long)nn * hw4, per = (units + 15) / 16;
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    vfloat4 acc = {0.f, 0.f, 0.f, 0.f};
    const int q = nn >> 2;
    for (int e = threadIdx.x; e < M * q; e += 1024) {
        const int m = e / q, j = e - m * q;
        const float* p = MODE == 0 ? x + (size_t)b * M * N + n0 + (size_t)m * N + 4 * j
                       : MODE == 1 ? x + ((size_t)b * (N / npb) + blockIdx.x) * M * npb + 4 * e
                                   : x + ((size_t)((b * (N / npb) + blockIdx.x) & 127) * M * npb) + 4 * e;
        acc += *reinterpret_cast<const vfloat4*>(p);
    }
    s_dyn[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    __syncthreads();
    const float t = s_dyn[(threadIdx.x + 1) & 1023];
    vfloat4 v = {1.f, 2.f, 3.f, 4.f};
    if (t == 12345.678f) v.x = t;   // (never: keeps the loads)
    PROF(1);
    for (long u = u0 + lane; u < u1; u += 64) __builtin_nontemporal_store(v, ob + u);
    PROF(2);
}

// ---- L9: persistent workgroups with a LOADER WAVE.  What the in-kernel clocks above show: a load issued on a chip whose memory
// system is saturated with stores returns after 10-14 us (whatever the issuing wave itself has in flight), a workgroup that
// starts with dependent loads spends a third of its life waiting, and so does any wave that executes s_waitcnt vmcnt(0) (which
// __syncthreads implies) with stores in flight.  Here the last wave of a workgroup never stores: it fetches the NEXT item's
// locations and x tile, fills the other LDS buffers and builds the owner table while the other waves stream the current item;
// the waves meet once per item at a bare s_barrier (LDS counters only).  The streaming waves never wait for memory.
template <int NT, int UNR = 1>
__global__ __launch_bounds__(NT) void lds_cover_spec_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW,
                                                            int npb, const int64_t* __restrict__ location, int W, int nitems) {
    extern __shared__ float s_dyn[];
    constexpr int NWS = NT / 64 - 1;     // streaming waves
    const int ld = npb, ngr = N / npb, H = HW / W, q = npb >> 2, hw4 = HW >> 2;   // (M <= 256: four entities per loader lane)
    const size_t xs_floats = ((size_t)M * ld + 3) & ~(size_t)3;
    float* const xs0 = s_dyn;
    float* const xs1 = s_dyn + xs_floats;
    int32_t* const tab0 = reinterpret_cast<int32_t*>(s_dyn + 2 * xs_floats);
    int32_t* const tab1 = tab0 + HW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const bool loader = wave == NWS;
    auto load_item = [&](int it, float* xs, int32_t* tab) {   // the loader wave: everything item `it` needs, into LDS
        const int b = it / ngr, n0 = (it % ngr) * npb;
        const vint4 m1 = {-1, -1, -1, -1};
        // the x tile by LDS-DMA: rows of npb floats, UNPADDED (a DMA piece lands as 64 lanes x 16 contiguous bytes); every piece is
        // requested before anything is waited for -- one memory round trip (10-14 us here) per item, no registers
        const float* __restrict__ xb = x + (size_t)b * M * N + n0;
        typedef __attribute__((address_space(3))) void* lds_ptr;
        typedef const __attribute__((address_space(1))) void* gl_ptr;
        const int total = M * q;                        // 16-byte pieces of the tile
        for (int e0 = 0; e0 < total; e0 += 64) {
            const int e = e0 + lane, m = e / q, j = e - m * q;
            if (e < total) __builtin_amdgcn_global_load_lds((gl_ptr)(xb + (size_t)m * N + 4 * j), (lds_ptr)(xs + e0 * 4), 16, 0, 0);
        }
        long ly[4], lx[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lane + 64 * r;
            ly[r] = lx[r] = -1;
            if (m < M) { ly[r] = location[((size_t)b * M + m) * 2]; lx[r] = location[((size_t)b * M + m) * 2 + 1]; }
        }
        for (int c4 = lane; c4 < hw4; c4 += 64) reinterpret_cast<vint4*>(tab)[c4] = m1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = lane + 64 * r;
            if (ly[r] >= 0 && ly[r] < H && lx[r] >= 0 && lx[r] < W) atomicMax(tab + (int)(ly[r] * W + lx[r]), m);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA pieces have landed
    };
    int it = blockIdx.x;
    if (it >= nitems) return;
    if (loader) load_item(it, xs0, tab0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int cur = 0;
    const int dbg = g_throttle;   // (debug: 16 = the loader loads nothing after the first item, 32 = no barrier per item: wrong results)
    const long units = (long)npb * hw4, per = (units + NWS - 1) / NWS;
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (; it < nitems; it += gridDim.x) {
        const float* const xs = cur ? xs1 : xs0;
        const int32_t* const tab = cur ? tab1 : tab0;
        if (loader) {
            if (it + (int)gridDim.x < nitems && !(dbg & 16)) load_item(it + gridDim.x, cur ? xs0 : xs1, cur ? tab0 : tab1);
        } else {
            const int b = it / ngr, n0 = (it % ngr) * npb;
            vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
            const long u0 = wave * per, u1 = min(units, u0 + per);
            long u = u0 + lane;
            int n = (int)(u / hw4);
            int c4 = (int)(u - (long)n * hw4);
            for (; u < u1; u += 64 * UNR) {
                int4 f[UNR];
#pragma unroll
                for (int k = 0; k < UNR; ++k) f[k] = *reinterpret_cast<const int4*>(tab + 4 * (c4 + 64 * k));
                vfloat4 o[UNR];
                int any = 0;
#pragma unroll
                for (int k = 0; k < UNR; ++k) { o[k] = zero4; any |= ~(f[k].x & f[k].y & f[k].z & f[k].w); }
                if (any < 0) {
                    const float* xn = xs + n;
#pragma unroll
                    for (int k = 0; k < UNR; ++k) {
                        const int32_t fi[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float a = xn[max(fi[c], 0) * ld];
                            o[k][c] = fi[c] >= 0 ? a : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < UNR; ++k) __builtin_nontemporal_store(o[k], ob + u + 64 * k);
                c4 += 64 * UNR;
                if (c4 >= hw4) { c4 -= hw4; ++n; }
            }
        }
        if (!(dbg & 32)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!(dbg & 16)) cur ^= 1;
    }
}

// ---- Lu: the shipped kernel with the streaming loop UNROLLED by UNR: the owner quads of UNR consecutive 1 KiB pieces are read
// together, the (few) gathers of all of them are issued before anything is waited for, then UNR stores.  What the in-kernel clocks
// say: a wave of the shipped kernel needs 0.43 us per 1 KiB piece (two dependent LDS round trips + ~40 instructions, eight
// waves per SIMD), a store-only wave 0.13 us; the kernel is bound by the LIFETIME of its workgroups (two 1024-thread slots
// per CU: startup ~14 us + issue ~14 us + drain ~14 us per 512 KB), so a faster loop is a faster kernel.
template <int UNR, bool VEC>
__global__ __launch_bounds__(1024) void lds_cover_unr_kernel(const float* __restrict__ x, float* __restrict__ out, int M, int N, int HW,
                                                             int npb, const int64_t* __restrict__ location, int W) {
    extern __shared__ float s_dyn[];
    PROF(0);
    const int b = blockIdx.y, n0 = blockIdx.x * npb, nn = min(npb, N - n0);
    const int ld = VEC ? npb + 4 : npb + 1;
    float* xs = s_dyn;
    int32_t* s_first = reinterpret_cast<int32_t*>(s_dyn + (((size_t)M * ld + 3) & ~(size_t)3));
    const float* __restrict__ xb = x + (size_t)b * M * N + n0;
    const int64_t* __restrict__ loc = location + (size_t)b * M * 2;
    const int H = HW / W;
    long y0 = -1, x0 = -1;
    if ((int)threadIdx.x < M) { y0 = loc[2 * threadIdx.x]; x0 = loc[2 * threadIdx.x + 1]; }
    {
        const int q = nn >> 2;
        vfloat4 t[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = threadIdx.x + 1024 * r;
            if (e < M * q) { const int m = e / q, j = e - m * q; t[r] = *(reinterpret_cast<const vfloat4*>(xb + (size_t)m * N) + j); }
        }
        const vint4 m1 = {-1, -1, -1, -1};
        for (int c4 = threadIdx.x; c4 < (HW >> 2); c4 += 1024) reinterpret_cast<vint4*>(s_first)[c4] = m1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int e = threadIdx.x + 1024 * r;
            if (e < M * q) {
                const int m = e / q, j = e - m * q;
                float* d = xs + m * ld + 4 * j;
                if (VEC) *reinterpret_cast<vfloat4*>(d) = t[r];
                else { d[0] = t[r].x; d[1] = t[r].y; d[2] = t[r].z; d[3] = t[r].w; }
            }
        }
    }
    __syncthreads();
    {
        const int m = threadIdx.x;
        if (m < M) {
            const int c = (y0 >= 0 && y0 < H && x0 >= 0 && x0 < W) ? (int)(y0 * W + x0) : -1;
            if (c >= 0) atomicMax(s_first + c, m);
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hw4 = HW >> 2;                                  // (hw4 % (64 * UNR) == 0 is checked by the host: a wave's UNR pieces
    const long units = (long)nn * hw4, per = (units + 15) / 16;   //  never straddle a plane)
    const long u0 = wave * per, u1 = min(units, u0 + per);
    vfloat4* __restrict__ ob = reinterpret_cast<vfloat4*>(out + ((size_t)b * N + n0) * HW);
    long u = u0 + lane;
    int n = (int)(u / hw4);
    int c4 = (int)(u - (long)n * hw4);
    const vfloat4 zero4 = {0.f, 0.f, 0.f, 0.f};
    PROF(1);
    for (; u < u1; u += 64 * UNR) {
        int4 f[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) f[k] = *reinterpret_cast<const int4*>(s_first + 4 * (c4 + 64 * k));
        vfloat4 o[UNR];
        int any = 0;
#pragma unroll
        for (int k = 0; k < UNR; ++k) { o[k] = zero4; any |= ~(f[k].x & f[k].y & f[k].z & f[k].w); }
        if (any < 0) {                                        // some cell of the lane's UNR quads has an owner
            const float* xn = xs + n;
#pragma unroll
            for (int k = 0; k < UNR; ++k) {
                const int32_t fi[4] = {f[k].x, f[k].y, f[k].z, f[k].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float a = xn[max(fi[c], 0) * ld];   // branch-free: an unowned cell reads row 0 and drops it
                    o[k][c] = fi[c] >= 0 ? a : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < UNR; ++k) __builtin_nontemporal_store(o[k], ob + u + 64 * k);
        c4 += 64 * UNR;
        if (c4 >= hw4) { c4 -= hw4; ++n; }
    }
    PROF(2);
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> void run(const char* name, double bytes, F&& f) {
        std::vector<float> ms;
        f();
        CK(hipDeviceSynchronize());
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(a));
            f();
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float t;
            CK(hipEventElapsedTime(&t, a, b));
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-64s best %7.3f ms %6.0f GB/s   median %7.3f ms %6.0f GB/s\n", name, ms[0], bytes / ms[0] * 1e-6, ms[3], bytes / ms[3] * 1e-6);
        fflush(stdout);
    }
};

static void prof_run(const char* name, long long* dprof, int B, const std::function<void()>& f) {
    CK(hipMemset(dprof, 0, (size_t)B * 3 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &dprof, sizeof(dprof)));
    f();
    CK(hipDeviceSynchronize());
    long long* none = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), &none, sizeof(none)));
    std::vector<long long> h((size_t)B * 3);
    CK(hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost));
    double st = 0, sm = 0; long long t0 = h[0], t1 = 0;
    for (int b = 0; b < B; ++b) { st += (h[3 * b + 1] - h[3 * b]) * 0.01; sm += (h[3 * b + 2] - h[3 * b + 1]) * 0.01; t0 = std::min(t0, h[3 * b]); t1 = std::max(t1, h[3 * b + 2]); }
    printf("   [in-kernel clock, wave 0 of %d workgroups] %-44s startup %6.2f us   stream %6.2f us   first entry .. last exit %7.1f us\n", B, name, st / B, sm / B, (t1 - t0) * 0.01);
}

int main() {
    const int B = 4096, M = 256, N = 64, H = 64, W = 64, HW = H * W;
    const size_t out_n = (size_t)B * N * HW, x_n = (size_t)B * M * N;
    float *out, *x;
    int64_t* loc;
    CK(hipMalloc(&out, out_n * 4));
    CK(hipMalloc(&x, x_n * 4));
    CK(hipMalloc(&loc, (size_t)B * M * 2 * 8));
    std::vector<float> hx(x_n);
    std::vector<int64_t> hl((size_t)B * M * 2);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& v : hx) v = (float)(rnd() % 2000) / 1000.f - 1.f;
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = (int64_t)(rnd() % 64);
    CK(hipMemcpy(x, hx.data(), x_n * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(loc, hl.data(), hl.size() * 8, hipMemcpyHostToDevice));
    const double bytes = (double)out_n * 4;
    Timer T;
    printf("# ScatterConnection forward at C5: %.2f GB of output; stores only (S*) and cover prototypes (C*)\n", bytes * 1e-9);
    T.run("memset (hipMemsetD32Async)", bytes, [&] { CK(hipMemsetD32Async((hipDeviceptr_t)out, 0, out_n, 0)); });
    {
        const int npb = 32;
        const size_t lds = 50 * 1024;
        T.run("S0 shipped pattern: grid (2, 4096) x 1024, 32 KB per wave", bytes,
              [&] { hipLaunchKernelGGL(s0_kernel, dim3(N / npb, B), dim3(1024), lds, 0, out, N, HW, npb); });
    }

    {
        const int npb = 32;
        const size_t lds = 50 * 1024;
        T.run("S0r stores + x tile read, shipped pattern (128 B halves of 256 B rows)", bytes, [&] { hipLaunchKernelGGL(s0r_kernel<0>, dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb); });
        T.run("S0r stores + x tile read, contiguous 32 KB per workgroup", bytes, [&] { hipLaunchKernelGGL(s0r_kernel<1>, dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb); });
        T.run("S0r stores + x tile read from a cache-resident 4 MB", bytes, [&] { hipLaunchKernelGGL(s0r_kernel<2>, dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb); });
        T.run("S0 again", bytes, [&] { hipLaunchKernelGGL(s0_kernel, dim3(N / npb, B), dim3(1024), lds, 0, out, N, HW, npb); });
    }
    for (int g : {256, 512}) {
        char nm[96];
        snprintf(nm, sizeof nm, "S1 grid-stride 4 KiB, %d x 256", g);
        T.run(nm, bytes, [&] { hipLaunchKernelGGL(s1_kernel, dim3(g), dim3(256), 0, 0, out, (long)(out_n / 4)); });
    }
    for (int g : {256, 512, 1024, 2048, 4096}) {
        char nm[96];
        snprintf(nm, sizeof nm, "S2 one 16 KiB plane per iteration, %d x 256, nt", g);
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((s23_kernel<1, true>), dim3(g), dim3(256), 0, 0, out, HW, (long)B * N); });
        snprintf(nm, sizeof nm, "S2 one 16 KiB plane per iteration, %d x 256, plain", g);
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((s23_kernel<1, false>), dim3(g), dim3(256), 0, 0, out, HW, (long)B * N); });
    }
    for (int g : {256, 512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, sizeof nm, "S3 four planes (64 KiB) per iteration, %d x 256, nt", g);
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((s23_kernel<4, true>), dim3(g), dim3(256), 0, 0, out, HW, (long)B * N / 4); });
    }

    {
        long long* dprof;
        CK(hipMalloc(&dprof, (size_t)B * 3 * 8));
        const int npb = 32;
        const size_t lds = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
#define PROF_L(A, nm) prof_run(nm, dprof, B, [&] { hipLaunchKernelGGL((lds_cover_kernel<A>), dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb, loc, W); });
        PROF_L(0, "L0 shipped") PROF_L(2, "L2 no x staging") PROF_L(4, "L4 nothing staged") PROF_L(6, "La x loaded, 1 word to LDS") PROF_L(1, "L1 loop without LDS reads")
        prof_run("S0r strided x read", dprof, B, [&] { hipLaunchKernelGGL(s0r_kernel<0>, dim3(N / npb, B), dim3(1024), 50 * 1024, 0, x, out, M, N, HW, npb); });
        prof_run("S0r contiguous x read", dprof, B, [&] { hipLaunchKernelGGL(s0r_kernel<1>, dim3(N / npb, B), dim3(1024), 50 * 1024, 0, x, out, M, N, HW, npb); });
    }

    {   // throttled store issue: at most `thr` vector-memory operations outstanding per wave
        const int npb = 32;
        const size_t lds = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        long long* dprof;
        CK(hipMalloc(&dprof, (size_t)B * 3 * 8));
        for (int thr : {0, 16, 8, 4, 2, 1, 0}) {
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_throttle), &thr, sizeof(thr)));
            char nm[96];
            snprintf(nm, sizeof nm, "T%d shipped cover kernel, <= %d operations in flight per wave", thr, thr);
            T.run(nm, bytes, [&] { hipLaunchKernelGGL((lds_cover_kernel<0>), dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb, loc, W); });
            prof_run(nm, dprof, B, [&] { hipLaunchKernelGGL((lds_cover_kernel<0>), dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb, loc, W); });
        }
        int z = 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_throttle), &z, sizeof(z)));
    }

    // cover prototypes
    std::vector<float> ref;   // checked on the first 8 and the last 2 batch elements
    auto check = [&](const char* name) {
        std::vector<float> got((size_t)N * HW);
        long bad = 0;
        for (int b : {0, 1, 2, 3, 4, 5, 6, 7, B - 2, B - 1}) {
            CK(hipMemcpy(got.data(), out + (size_t)b * N * HW, got.size() * 4, hipMemcpyDeviceToHost));
            std::vector<int> owner(HW, -1);
            for (int m = 0; m < M; ++m) owner[hl[((size_t)b * M + m) * 2] * W + hl[((size_t)b * M + m) * 2 + 1]] = m;
            for (int n = 0; n < N; ++n)
                for (int c = 0; c < HW; ++c) {
                    const float want = owner[c] >= 0 ? hx[((size_t)b * M + owner[c]) * N + n] : 0.f;
                    if (got[(size_t)n * HW + c] != want) ++bad;
                }
        }
        printf("   %s: %s\n", name, bad ? "MISMATCH" : "bit-exact on 10 batch elements");
    };

    {   // loader-wave kernels
        for (int cfg = 0; cfg < 6; ++cfg) {
            const int dbg = cfg == 4 ? 16 : cfg == 5 ? 48 : 0;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(g_throttle), &dbg, sizeof(dbg)));
            const int npb = (cfg & 3) < 2 ? 32 : 16;
            const int nt = (cfg & 1) ? 512 : 1024;
            const size_t ldsp = 2 * ((((size_t)M * npb + 3) & ~(size_t)3) * 4 + (size_t)HW * 4);
            const int per_cu = (int)(160 * 1024 / ldsp) < 2048 / nt ? (int)(160 * 1024 / ldsp) : 2048 / nt;
            for (int g : {256 * per_cu}) {
                char nm[128];
                snprintf(nm, sizeof nm, "L9 loader wave, %d threads, npb %d, %d workgroups (%d per CU, %zu KB LDS) dbg %d", nt, npb, g, per_cu, ldsp / 1024, dbg);
                CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
                if (nt == 1024) {
                    CK(hipFuncSetAttribute((const void*)lds_cover_spec_kernel<576>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    T.run("  (the same with 8 streaming waves + loader = 576 threads: aligned 64 KB / 32 KB shares)", bytes, [&] { hipLaunchKernelGGL((lds_cover_spec_kernel<576>), dim3(g), dim3(576), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
                    if (!dbg) check("576");
                    CK(hipFuncSetAttribute((const void*)lds_cover_spec_kernel<576, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    T.run("  (576 threads, streaming loop unrolled x4)", bytes, [&] { hipLaunchKernelGGL((lds_cover_spec_kernel<576, 4>), dim3(g), dim3(576), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
                    if (!dbg) check("576 x4");
                    CK(hipFuncSetAttribute((const void*)lds_cover_spec_kernel<576, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    T.run("  (576 threads, streaming loop unrolled x2)", bytes, [&] { hipLaunchKernelGGL((lds_cover_spec_kernel<576, 2>), dim3(g), dim3(576), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
                    if (!dbg) check("576 x2");
                    CK(hipFuncSetAttribute((const void*)lds_cover_spec_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    T.run(nm, bytes, [&] { hipLaunchKernelGGL((lds_cover_spec_kernel<1024>), dim3(g), dim3(1024), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
                } else {
                    CK(hipFuncSetAttribute((const void*)lds_cover_spec_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    T.run(nm, bytes, [&] { hipLaunchKernelGGL((lds_cover_spec_kernel<512>), dim3(g), dim3(512), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
                }
                if (!dbg) check(nm);
            }
        }
        const int z0 = 0;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_throttle), &z0, sizeof(z0)));
    }

    {   // unrolled streaming loops
        const int npb = 32;
        const size_t lds1 = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        const size_t lds4 = (((size_t)M * (npb + 4) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        long long* dprof;
        CK(hipMalloc(&dprof, (size_t)B * 3 * 8));
#define RUN_U(UNR, VEC, LDS) { char nm[96]; snprintf(nm, sizeof nm, "Lu unrolled x%d, %s tile rows", UNR, VEC ? "16-byte" : "scalar"); \
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((lds_cover_unr_kernel<UNR, VEC>), dim3(N / npb, B), dim3(1024), LDS, 0, x, out, M, N, HW, npb, loc, W); }); check(nm); \
        prof_run(nm, dprof, B, [&] { hipLaunchKernelGGL((lds_cover_unr_kernel<UNR, VEC>), dim3(N / npb, B), dim3(1024), LDS, 0, x, out, M, N, HW, npb, loc, W); }); }
        RUN_U(1, false, lds1) RUN_U(2, false, lds1) RUN_U(4, false, lds1) RUN_U(8, false, lds1) RUN_U(4, true, lds4) RUN_U(2, true, lds4)
    }
    {   // the shipped cover kernel and its ablations (npb = 32: 33.8 KB of x tile + 16 KB owner table per workgroup)
        const int npb = 32;
        const size_t lds = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        const char* names[9] = {"L0 shipped cover kernel (replica)", "L1 stream loop stores zeros, no LDS reads", "L2 no x tile staging",
                                "L3 branch-free gathers", "L4 no staging, no build: empty table", "L5 x tile with plain loads",
                                "La x tile loaded, one LDS word per thread", "Lb x tile read as a contiguous 32 KB", "Lc x staged, no table build"};
#define RUN_L(A) T.run(names[A], bytes, [&] { hipLaunchKernelGGL((lds_cover_kernel<A>), dim3(N / npb, B), dim3(1024), lds, 0, x, out, M, N, HW, npb, loc, W); });
        RUN_L(0) check(names[0]);
        RUN_L(1) RUN_L(2) RUN_L(3) check(names[3]);
        RUN_L(4) RUN_L(5) check(names[5]);
        RUN_L(6) RUN_L(7) RUN_L(8)
        RUN_L(0)
    }

    {   // 512-thread workgroups / vector tile rows / persistent double-buffered staging
        const int npb = 32;
        const size_t lds = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        const char* names[1] = {"L0 shipped cover kernel (replica), again"};
        const size_t lds1 = (((size_t)M * (npb + 1) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        const size_t lds4 = (((size_t)M * (npb + 4) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4;
        CK(hipFuncSetAttribute((const void*)lds_cover_persist_kernel<1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)lds_cover_persist_kernel<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)lds_cover_persist_kernel<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        T.run("L6 512 threads per workgroup (3 per CU), scalar tile rows", bytes, [&] { hipLaunchKernelGGL((lds_cover_nt_kernel<512, false>), dim3(N / npb, B), dim3(512), lds1, 0, x, out, M, N, HW, npb, loc, W); });
        check("L6");
        T.run("L7 1024 threads, 16-byte tile rows, plain loads", bytes, [&] { hipLaunchKernelGGL((lds_cover_nt_kernel<1024, true>), dim3(N / npb, B), dim3(1024), lds4, 0, x, out, M, N, HW, npb, loc, W); });
        check("L7");
        T.run("L7b 512 threads (3 per CU), 16-byte tile rows", bytes, [&] { hipLaunchKernelGGL((lds_cover_nt_kernel<512, true>), dim3(N / npb, B), dim3(512), lds4, 0, x, out, M, N, HW, npb, loc, W); });
        check("L7b");
        const size_t ldsp = 2 * ((((size_t)M * (npb + 4) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4);
        for (int g : {256, 512}) {
            char nm[96];
            snprintf(nm, sizeof nm, "L8 persistent, 1024 threads, npb 32, %d workgroups", g);
            CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
            T.run(nm, bytes, [&] { hipLaunchKernelGGL((lds_cover_persist_kernel<1024, 2>), dim3(g), dim3(1024), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
            check(nm);
        }
        {
            CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
            T.run("L8b persistent, 512 threads, npb 32, 256 workgroups", bytes, [&] { hipLaunchKernelGGL((lds_cover_persist_kernel<512, 4>), dim3(256), dim3(512), ldsp, 0, x, out, M, N, HW, npb, loc, W, B * (N / npb)); });
            check("L8b");
        }
        {
            const int npb2 = 16;
            const size_t ldsq = 2 * ((((size_t)M * (npb2 + 4) + 3) & ~(size_t)3) * 4 + (size_t)HW * 4);
            CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
            T.run("L8c persistent, 512 threads, npb 16, 512 workgroups (2 per CU)", bytes, [&] { hipLaunchKernelGGL((lds_cover_persist_kernel<512, 2>), dim3(512), dim3(512), ldsq, 0, x, out, M, N, HW, npb2, loc, W, B * (N / npb2)); });
            check("L8c");
        }
        RUN_L(0)
    }
    for (int g : {512, 1024, 2048}) {
        char nm[96];
        snprintf(nm, sizeof nm, "C1 cover, one plane per iteration, %d x 256", g);
        CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((cover_sweep_kernel<1>), dim3(g), dim3(256), 2 * HW * 4, 0, x, loc, out, M, N, HW, W, (long)B * N); });
        check(nm);
    }
    for (int g : {256, 512, 1024}) {
        char nm[96];
        snprintf(nm, sizeof nm, "C4 cover, four planes per iteration, %d x 256", g);
        CK(hipMemsetD32Async((hipDeviceptr_t)out, 0x7fc00000, out_n, 0));
        T.run(nm, bytes, [&] { hipLaunchKernelGGL((cover_sweep_kernel<4>), dim3(g), dim3(256), 2 * HW * 4, 0, x, loc, out, M, N, HW, W, (long)B * N / 4); });
        check(nm);
    }
    return 0;
}
