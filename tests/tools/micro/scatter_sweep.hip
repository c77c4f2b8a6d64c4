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
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));
typedef int vint4 __attribute__((ext_vector_type(4)));
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
