// STREAM-style ceilings for the traffic mix of the GAE kernels on MI355X, at the headline footprint
// (T=1024, B=65536: three 268 MB arrays).  No scan, no dependence between elements: the fastest these byte
// counts can move through HBM with the same mix of reads and writes.
//   mix 0 ("add",  2 reads : 1 write, the forward's mix):  c = a + b
//   mix 1 ("fan",  1 read : 2 writes, the backward's mix): b = a, c = -a
//   mix 2 ("copy", 1 : 1)
// Patterns: grid-stride with U float4 in flight per thread (nontemporal or plain), and row-tiled like the GAE kernels
// (a workgroup owns a 128- or 256-column strip and walks the rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));

template <int MIX, int U, bool NT>
__global__ __launch_bounds__(256) void gs(const v4* __restrict__ a, v4* __restrict__ b, v4* __restrict__ c, size_t n4) {
    const size_t nt = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += nt * U) {
        v4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * nt;
            if (j < n4) {
                x[u] = NT ? __builtin_nontemporal_load(a + j) : a[j];
                if (MIX == 0) y[u] = NT ? __builtin_nontemporal_load((const v4*)b + j) : b[j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t j = i + (size_t)u * nt;
            if (j < n4) {
                if (MIX == 0) { v4 o = x[u] + y[u]; if (NT) __builtin_nontemporal_store(o, c + j); else c[j] = o; }
                if (MIX == 1) { v4 o = -x[u]; if (NT) { __builtin_nontemporal_store(x[u], b + j); __builtin_nontemporal_store(o, c + j); } else { b[j] = x[u]; c[j] = o; } }
                if (MIX == 2) { if (NT) __builtin_nontemporal_store(x[u], c + j); else c[j] = x[u]; }
            }
        }
    }
}

// row-tiled: workgroup = NW waves, a strip of 128 columns (float2 per lane); wave w takes rows w*LC.. of every
// NW*LC-row span, LC rows in flight (exactly the load/store shape of gae_fwd_kernel<2, LC, NW>)
template <int MIX, int LC, int NW>
__global__ __launch_bounds__(NW * 64) void tiled(const float* __restrict__ a, float* __restrict__ b, float* __restrict__ c, int T, int B) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t col = (size_t)blockIdx.x * 128 + lane * 2;
    for (int t0 = w * LC; t0 < T; t0 += NW * LC) {
        v2 x[LC], y[LC];
#pragma unroll
        for (int j = 0; j < LC; ++j) {
            x[j] = __builtin_nontemporal_load((const v2*)(a + (size_t)(t0 + j) * B + col));
            if (MIX == 0) y[j] = __builtin_nontemporal_load((const v2*)(b + (size_t)(t0 + j) * B + col));
        }
#pragma unroll
        for (int j = 0; j < LC; ++j) {
            if (MIX == 0) __builtin_nontemporal_store(x[j] + y[j], (v2*)(c + (size_t)(t0 + j) * B + col));
            if (MIX == 1) { __builtin_nontemporal_store(x[j], (v2*)(b + (size_t)(t0 + j) * B + col)); __builtin_nontemporal_store(-x[j], (v2*)(c + (size_t)(t0 + j) * B + col)); }
            if (MIX == 2) __builtin_nontemporal_store(x[j], (v2*)(c + (size_t)(t0 + j) * B + col));
        }
    }
}

static float* A; static float* Bf; static float* C;
static const int T = 1024, BB = 65536;
static const size_t N4 = (size_t)T * BB / 4;

template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> v;
    for (int r = 0; r < 9; ++r) {
        hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); v.push_back(ms);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
static void report(const char* name, int mix, float ms) {
    const double bytes = (mix == 2 ? 2.0 : 3.0) * N4 * 16.0;
    printf("%-58s %7.1f us  %6.0f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
}
template <int MIX, int U, bool NT> void run_gs(int grid) {
    char nm[96]; snprintf(nm, sizeof nm, "mix%d grid-stride U=%d %s grid=%d", MIX, U, NT ? "nt" : "plain", grid);
    report(nm, MIX, timeit([&] { gs<MIX, U, NT><<<grid, 256>>>((const v4*)A, (v4*)Bf, (v4*)C, N4); }));
}
template <int MIX, int LC, int NW> void run_tiled() {
    char nm[96]; snprintf(nm, sizeof nm, "mix%d row-tiled 128 cols, LC=%d NW=%d", MIX, LC, NW);
    report(nm, MIX, timeit([&] { tiled<MIX, LC, NW><<<BB / 128, NW * 64>>>(A, Bf, C, T, BB); }));
}
template <int MIX> void all() {
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) { run_gs<MIX, 4, true>(g); }
    run_gs<MIX, 8, true>(256 * 8); run_gs<MIX, 2, true>(256 * 32); run_gs<MIX, 1, true>(256 * 64);
    run_gs<MIX, 4, false>(256 * 8); run_gs<MIX, 4, false>(256 * 32);
    run_tiled<MIX, 8, 2>(); run_tiled<MIX, 4, 4>(); run_tiled<MIX, 8, 4>(); run_tiled<MIX, 16, 2>(); run_tiled<MIX, 8, 8>();
}
int main() {
    hipMalloc(&A, N4 * 16 + 262144); hipMalloc(&Bf, N4 * 16 + 262144); hipMalloc(&C, N4 * 16 + 262144);
    hipMemset(A, 0, N4 * 16); hipMemset(Bf, 0, N4 * 16); hipMemset(C, 0, N4 * 16);
    printf("three arrays of %.1f MB (T=%d, B=%d fp32)\n", N4 * 16 / 1e6, T, BB);
    all<0>(); all<1>(); all<2>();
    // the bench's alternation: add (2R:1W) then fan (1R:2W), best patterns of each
    float ms = timeit([&] {
        gs<0, 4, true><<<256 * 8, 256>>>((const v4*)A, (v4*)Bf, (v4*)C, N4);
        gs<1, 4, true><<<256 * 8, 256>>>((const v4*)C, (v4*)A, (v4*)Bf, N4);
    });
    printf("%-58s %7.1f us  %6.0f GB/s\n", "alternating add + fan (grid-stride U=4 nt, grid=2048)", ms * 1e3, 6.0 * N4 * 16.0 / ms / 1e6);
    return 0;
}
