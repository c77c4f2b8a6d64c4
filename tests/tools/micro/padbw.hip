// padbw.hip -- ablation of the packed Pad1D kernel at BASELINE.json configs[4] scale (n = 2^20 rows, len ~ U[32,128),
// L = 127): where do the 350 us go when two plain fills of the outputs take 156?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef int vint4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned div_by(unsigned long o, unsigned L, double inv) {
    unsigned i = (unsigned)((double)o * inv);
    const unsigned long p = (unsigned long)i * L;
    if (p > o) --i; else if (p + L <= o) ++i;
    return i;
}
// MODE 0 full, 1 no source reads, 2 no mask store, 3 plain stores
template <int MODE>
__global__ __launch_bounds__(256) void quad_kernel(const float* __restrict__ flat, const long* __restrict__ off,
                                                   const int* __restrict__ len_, float* __restrict__ x, int* __restrict__ m,
                                                   long n, unsigned L, double inv) {
    const long total4 = n * (long)L / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
        const unsigned long o = (unsigned long)q * 4;
        unsigned i = div_by(o, L, inv);
        unsigned c = (unsigned)(o - (unsigned long)i * L);
        const float* src = flat + off[i];
        unsigned len = (unsigned)len_[i];
        vfloat4 v; vint4 mk;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool in = c < len;
            v[k] = in ? (MODE == 1 ? 1.f : src[c]) : 0.f;
            mk[k] = in ? 1 : 0;
            if (++c == L) { c = 0; ++i; if ((long)i < n) { src = flat + off[i]; len = (unsigned)len_[i]; } }
        }
        if (MODE == 3) { *reinterpret_cast<vfloat4*>(x + o) = v; *reinterpret_cast<vint4*>(m + o) = mk; }
        else {
            __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(x + o));
            if (MODE != 2) __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(m + o));
        }
    }
}
// one wave per row, lanes along the columns, 4-byte accesses
__global__ __launch_bounds__(256) void row_kernel(const float* __restrict__ flat, const long* __restrict__ off,
                                                  const int* __restrict__ len_, float* __restrict__ x, int* __restrict__ m,
                                                  long n, unsigned L) {
    const int lane = threadIdx.x & 63;
    for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < n; r += (long)gridDim.x * 4) {
        const float* src = flat + off[r];
        const unsigned len = (unsigned)len_[r];
        for (unsigned c = lane; c < L; c += 64) {
            const bool in = c < len;
            __builtin_nontemporal_store(in ? src[c] : 0.f, x + r * L + c);
            __builtin_nontemporal_store(in ? 1 : 0, m + r * L + c);
        }
    }
}
// a workgroup owns RB consecutive rows: their source range (contiguous in the packed buffer) goes to LDS with aligned
// 16-byte loads, the outputs are written as 16-byte quads reading LDS
template <int RB>
__global__ __launch_bounds__(256) void lds_kernel(const float* __restrict__ flat, const long* __restrict__ off,
                                                  const int* __restrict__ len_, float* __restrict__ x, int* __restrict__ m,
                                                  long n, unsigned L, double inv) {
    extern __shared__ float tile[];        // RB * L floats + 8
    __shared__ long s_off[RB + 1];
    for (long r0 = (long)blockIdx.x * RB; r0 < n; r0 += (long)gridDim.x * RB) {
        const int nr = (int)(n - r0 < RB ? n - r0 : RB);
        __syncthreads();
        if (threadIdx.x <= nr) s_off[threadIdx.x] = threadIdx.x < nr ? off[r0 + threadIdx.x] : off[r0 + nr - 1] + len_[r0 + nr - 1];
        __syncthreads();
        const long lo = s_off[0], hi = s_off[nr];
        const long lo4 = lo & ~3L;
        for (long p = lo4 + threadIdx.x * 4; p < hi; p += 1024) {
            const vfloat4 t = *reinterpret_cast<const vfloat4*>(flat + p);      // (the buffer is padded: reads past hi are harmless)
            *reinterpret_cast<vfloat4*>(tile + (p - lo4)) = t;
        }
        __syncthreads();
        const unsigned long obase = (unsigned long)r0 * L;
        const unsigned long oend = obase + (unsigned long)nr * L;
        for (unsigned long o = (obase & ~3UL) + threadIdx.x * 4; o < oend; o += 1024) {
            vfloat4 v; vint4 mk;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long oo = o + k;
                bool in = false; float val = 0.f;
                if (oo >= obase && oo < oend) {
                    const unsigned rr = div_by(oo - obase, L, inv);
                    const unsigned c = (unsigned)(oo - obase - (unsigned long)rr * L);
                    const long so = s_off[rr];
                    in = c < (unsigned)(s_off[rr + 1] - so);
                    if (in) val = tile[so - lo4 + c];
                }
                v[k] = val; mk[k] = in ? 1 : 0;
            }
            if (o >= obase && o + 4 <= oend) {
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(x + o));
                __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(m + o));
            } else {
                for (int k = 0; k < 4; ++k) if (o + k >= obase && o + k < oend) { x[o + k] = v[k]; m[o + k] = mk[k]; }
            }
        }
    }
}

// round 4: WAVE-synchronous tiles in OUTPUT space.  A wave owns 1024 consecutive output elements (256 aligned quads) of the
// flat (n*L) output stream -- whatever rows they fall in; the packed source of those elements is ONE contiguous span of at
// most 1024 floats, staged into the wave's own LDS slice with aligned 16-byte loads; no workgroup barrier, no ragged
// quads, one float division per quad (exact: see the product kernel), two row lookups per quad.
template <int DUMMY>
__global__ __launch_bounds__(256) void wave_kernel(const float* __restrict__ flat, const long* __restrict__ off,
                                                   const int* __restrict__ len_, float* __restrict__ x, int* __restrict__ m,
                                                   long n, unsigned L, float invL) {
    __shared__ __attribute__((aligned(16))) float tile_all[4][1032];
    __shared__ int s_rel_all[4][48], s_len_all[4][48];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* tile = tile_all[wv];
    int* s_rel = s_rel_all[wv];
    int* s_len = s_len_all[wv];
    const unsigned long total = (unsigned long)n * L;
    const long ntiles = (long)((total + 1023) / 1024);
    for (long w = (long)blockIdx.x * 4 + wv; w < ntiles; w += (long)gridDim.x * 4) {
        const unsigned long o0 = (unsigned long)w * 1024, o1 = o0 + 1024 < total ? o0 + 1024 : total;
        const long r_lo = (long)(o0 / L), r_hi = (long)((o1 - 1) / L);
        const unsigned c_lo = (unsigned)(o0 - (unsigned long)r_lo * L), c_hi = (unsigned)(o1 - 1 - (unsigned long)r_hi * L);
        const int nr = (int)(r_hi - r_lo + 1);
        long my_off = 0; int my_len = 0;
        if (lane < nr) { my_off = off[r_lo + lane]; my_len = len_[r_lo + lane]; }
        const long off_lo = __shfl(my_off, 0, 64), off_hi = __shfl(my_off, nr - 1, 64);
        const int len_lo = __shfl(my_len, 0, 64), len_hi = __shfl(my_len, nr - 1, 64);
        const long span_lo = off_lo + ((int)c_lo < len_lo ? (int)c_lo : len_lo);
        const long span_hi = off_hi + ((int)c_hi + 1 < len_hi ? (int)c_hi + 1 : len_hi);
        const long lo4 = span_lo & ~3L;
        if (lane < nr) { s_rel[lane] = (int)(my_off - lo4); s_len[lane] = my_len; }
        if (lane == nr) { s_rel[lane] = 0; s_len[lane] = 0; }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const long p = lo4 + 4 * (long)(lane + 64 * j);
            if (p < span_hi) *reinterpret_cast<vfloat4*>(tile + 4 * (lane + 64 * j)) = *reinterpret_cast<const vfloat4*>(flat + p);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = lane + 64 * j;
            const unsigned long o = o0 + 4 * (unsigned long)q;
            if (o >= o1) break;
            const unsigned t = c_lo + 4 * (unsigned)q;
            const unsigned rr = (unsigned)(((float)t + 0.5f) * invL);
            const unsigned c = t - rr * L;
            const int rel0 = s_rel[rr], len0 = s_len[rr], rel1 = s_rel[rr + 1], len1 = s_len[rr + 1];
            vfloat4 v; vint4 mk;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned cc = c + k;
                const bool nxt = cc >= L;
                if (nxt) cc -= L;
                const int ln = nxt ? len1 : len0, rl = nxt ? rel1 : rel0;
                const bool in = (int)cc < ln;
                v[k] = in ? tile[rl + (int)cc] : 0.f;
                mk[k] = in ? 1 : 0;
            }
            if (o + 4 <= o1) {
                __builtin_nontemporal_store(v, reinterpret_cast<vfloat4*>(x + o));
                __builtin_nontemporal_store(mk, reinterpret_cast<vint4*>(m + o));
            } else {
                for (int k = 0; k < 4; ++k) if (o + k < o1) { x[o + k] = v[k]; m[o + k] = mk[k]; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- the product kernel, verbatim (di-hpc_amd/csrc/pad_scatter.hip)
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
        const bool fits = hi - lo4 <= (long)RB * L + 8 && hi >= lo;
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


template <class F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < reps; ++i) f();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3f;
}

int main() {
    const long n = 1 << 20; const unsigned L = 127;
    std::vector<int> len(n); std::vector<long> off(n);
    srand(1); long tot = 0;
    for (long i = 0; i < n; ++i) { len[i] = 32 + rand() % 96; off[i] = tot; tot += len[i]; }
    float *flat, *x; int *m, *dlen; long* doff;
    hipMalloc(&flat, (tot + 64) * 4); hipMalloc(&x, n * L * 4); hipMalloc(&m, n * L * 4);
    hipMalloc(&dlen, n * 4); hipMalloc(&doff, n * 8);
    hipMemset(flat, 0, (tot + 64) * 4);
    hipMemcpy(dlen, len.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(doff, off.data(), n * 8, hipMemcpyHostToDevice);
    const double inv = 1.0 / L; const double by = 4.0 * tot + 8.0 * n * L;
    printf("n=%ld L=%u: read %.0f MB, write %.0f MB\n", n, L, 4.0 * tot / 1e6, 8.0 * n * L / 1e6);
    for (int g : {2048, 4096, 16384, 65536}) {
        float t0 = timeit([&] { hipLaunchKernelGGL(quad_kernel<0>, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, inv); });
        float t1 = timeit([&] { hipLaunchKernelGGL(quad_kernel<1>, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, inv); });
        float t2 = timeit([&] { hipLaunchKernelGGL(quad_kernel<2>, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, inv); });
        float t3 = timeit([&] { hipLaunchKernelGGL(quad_kernel<3>, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, inv); });
        float t4 = timeit([&] { hipLaunchKernelGGL(row_kernel, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L); });
        printf("grid %6d | quad full %.1f us (%.0f GB/s) | no src reads %.1f | no mask store %.1f | plain stores %.1f | wave per row %.1f\n",
               g, t0, by / t0 / 1e3, t1, t2, t3, t4);
    }
    for (int g : {2048, 8192, 32768}) {
        float a = timeit([&] { hipLaunchKernelGGL(lds_kernel<16>, dim3(g), dim3(256), (16 * L + 8) * 4, 0, flat, doff, dlen, x, m, n, L, inv); });
        float b = timeit([&] { hipLaunchKernelGGL(lds_kernel<64>, dim3(g), dim3(256), (64 * L + 8) * 4, 0, flat, doff, dlen, x, m, n, L, inv); });
        printf("grid %6d | LDS-staged 16 rows/wg %.1f us (%.0f GB/s) | 64 rows/wg %.1f us (%.0f GB/s)\n", g, a, by / a / 1e3, b, by / b / 1e3);
    }
    {
        std::vector<long> tab(4 * n);
        for (long i = 0; i < n; ++i) { tab[4 * i] = (long)(flat + off[i]); tab[4 * i + 1] = 1; tab[4 * i + 2] = 1; tab[4 * i + 3] = len[i]; }
        long* dtab; hipMalloc(&dtab, n * 32); hipMemcpy(dtab, tab.data(), n * 32, hipMemcpyHostToDevice);
        for (int RB : {16, 64}) for (int g : {8192, 16384, 65536}) {
            const size_t lds = ((((size_t)RB * L + 8 + 1) & ~(size_t)1)) * 4 + (size_t)(RB + 1) * 8;
            float t = timeit([&] { hipLaunchKernelGGL(pad1d_packed_kernel, dim3(g), dim3(256), lds, 0, flat, (const int64_t*)dtab, x, m, n, L, inv, RB, 0.f, 0); });
            printf("product kernel RB=%d grid %d: %.1f us (%.0f GB/s)\n", RB, g, t, by / t / 1e3);
        }
    }
    {   // round 4: wave tiles in output space; verified against the LDS-staged kernel's output
        std::vector<float> hx((size_t)n * L), hx2((size_t)n * L); std::vector<int> hm((size_t)n * L), hm2((size_t)n * L);
        std::vector<float> hf(tot + 64);
        for (long i = 0; i < tot; ++i) hf[i] = (float)(i % 9973) + 0.5f;
        hipMemcpy(flat, hf.data(), (tot + 64) * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(lds_kernel<16>, dim3(8192), dim3(256), (16 * L + 8) * 4, 0, flat, doff, dlen, x, m, n, L, inv);
        hipMemcpy(hx.data(), x, (size_t)n * L * 4, hipMemcpyDeviceToHost); hipMemcpy(hm.data(), m, (size_t)n * L * 4, hipMemcpyDeviceToHost);
        hipMemset(x, 0xff, (size_t)n * L * 4); hipMemset(m, 0xff, (size_t)n * L * 4);
        hipLaunchKernelGGL(wave_kernel<0>, dim3(8192), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, 1.0f / L);
        hipMemcpy(hx2.data(), x, (size_t)n * L * 4, hipMemcpyDeviceToHost); hipMemcpy(hm2.data(), m, (size_t)n * L * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (size_t i = 0; i < (size_t)n * L; ++i) if (hx[i] != hx2[i] || hm[i] != hm2[i]) ++bad;
        printf("wave kernel vs LDS-staged kernel: %ld mismatching elements of %zu\n", bad, (size_t)n * L);
        for (int g : {2048, 4096, 8192, 16384, 65536}) {
            float t = timeit([&] { hipLaunchKernelGGL(wave_kernel<0>, dim3(g), dim3(256), 0, 0, flat, doff, dlen, x, m, n, L, 1.0f / L); });
            printf("wave tiles in output space, grid %6d: %.1f us (%.0f GB/s)\n", g, t, by / t / 1e3);
        }
    }
    float f = timeit([&] { hipMemsetAsync(x, 0, n * L * 4, 0); hipMemsetAsync(m, 0, n * L * 4, 0); });
    printf("two memsets of the outputs: %.1f us (%.0f GB/s)\n", f, 8.0 * n * L / f / 1e3);
    return 0;
}
