// All-to-all exchange primitive of lstm_persist.hpp in isolation: N workgroups, each publishes M {value,tag} words
// per step (to R replicas) and then gathers all N*M words.  Prints us/step for several variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

__device__ __forceinline__ void put(u64* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 get(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// VAR 0: every thread polls CH words (tagged).  VAR 1: same but only `pollers` threads poll, 8B loads.
// VAR 2: 16-byte loads (two words per load) via inline asm.
template <int VAR, int CH>
__global__ __launch_bounds__(256) void a2a(u64* buf, size_t par_words, size_t rep_words, int nrep, int M, int steps,
                                           u64* out, float* sink) {
    __shared__ float lds[8192];
    const int tid = threadIdx.x, N = gridDim.x, total = N * M;
    const size_t myrep = blockIdx.x % nrep;
    float accum = 0.f;
    const u64 t0 = wall_clock64();
    for (int s = 1; s <= steps; ++s) {
        u64* wbase = buf + (size_t)(s & 1) * par_words;
        if (tid < M)
            for (int r = 0; r < nrep; ++r) put(wbase + r * rep_words + blockIdx.x * M + tid, (float)s, (unsigned)s);
        const u64* rbase = wbase + myrep * rep_words;
        if (VAR == 0) {
            for (int e0 = tid; e0 < total; e0 += 256 * CH) {
                long guard = 0;
                while (true) {
                    u64 w[CH];
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int e = e0 + 256 * i;
                        w[i] = e < total ? get(rbase + e) : ((u64)s << 32);
                    }
#pragma unroll
                    for (int i = 0; i < CH; ++i) ok = ok && (unsigned)(w[i] >> 32) == (unsigned)s;
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) {
                            const int e = e0 + 256 * i;
                            if (e < total) lds[e & 8191] = __uint_as_float((unsigned)w[i]);
                        }
                        break;
                    }
                    if (++guard > (1L << 24)) return;
                }
            }
        } else if (VAR == 2) {   // 16-byte loads: thread handles word pairs
            const int pairs = total / 2;
            for (int e0 = tid; e0 < pairs; e0 += 256 * CH) {
                long guard = 0;
                while (true) {
                    uint4 w[CH];
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int e = e0 + 256 * i;
                        if (e < pairs) {
                            const u64* p = rbase + 2 * e;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w[i]) : "v"(p) : "memory");
                        } else {
                            w[i] = make_uint4(0, s, 0, s);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int i = 0; i < CH; ++i) ok = ok && w[i].y == (unsigned)s && w[i].w == (unsigned)s;
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < CH; ++i) {
                            const int e = e0 + 256 * i;
                            if (e < pairs) { lds[(2 * e) & 8191] = __uint_as_float(w[i].x); lds[(2 * e + 1) & 8191] = __uint_as_float(w[i].z); }
                        }
                        break;
                    }
                    if (++guard > (1L << 24)) return;
                }
            }
        }
        __syncthreads();
        accum += lds[tid];
        __syncthreads();
    }
    if (tid == 0 && blockIdx.x == 0) out[0] = wall_clock64() - t0;
    if (accum == 12345.f) sink[0] = accum;
}

template <int VAR, int CH> void run(const char* name, int N, int M, int nrep) {
    const size_t rep_words = ((size_t)N * M + 15) / 16 * 16, par_words = rep_words * nrep;
    u64 *buf, *out; float* sink;
    hipMalloc(&buf, 2 * par_words * 8); hipMalloc(&out, 8); hipMalloc(&sink, 4);
    hipMemset(buf, 0, 2 * par_words * 8); hipMemset(out, 0, 8);
    const int steps = 500;
    a2a<VAR, CH><<<N, 256>>>(buf, par_words, rep_words, nrep, M, steps, out, sink);
    hipDeviceSynchronize();
    u64 h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-22s N=%3d M=%2d words=%5d rep=%2d CH=%2d : %.2f us/step\n", name, N, M, N * M, nrep, CH, h / 100.0 / steps);
    (void)hipFree(buf); (void)hipFree(out); (void)hipFree(sink);
}

int main() {
    for (int N : {8, 48, 96, 192}) run<0, 8>("tagged 8B", N, 6, 1);
    for (int N : {96, 192}) run<0, 8>("tagged 8B", N, 6, 4);
    run<0, 8>("tagged 8B", 192, 6, 8);
    run<0, 8>("tagged 8B", 192, 2, 1);
    run<0, 8>("tagged 8B", 192, 24, 1);
    run<0, 24>("tagged 8B", 192, 24, 1);
    run<0, 24>("tagged 8B", 192, 24, 4);
    for (int N : {96, 192}) run<2, 4>("tagged 16B loads", N, 6, 1);
    run<2, 4>("tagged 16B loads", 192, 6, 4);
    run<2, 12>("tagged 16B loads", 192, 24, 1);
    run<2, 12>("tagged 16B loads", 192, 24, 4);
    return 0;
}
