// Pure-write HBM bandwidth on MI355X for the store patterns used by the scatter output kernel and the fills.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));

// MODE 0: grid-stride float4 nontemporal stores.  MODE 1: plain stores.
// MODE 2: scatter-like: workgroup b (1024 threads) owns 1 MiB = 64 planes of 16 KiB; thread t writes 16 B at offset
//         t*16 of every plane, 16 planes per loop iteration (the pattern of scatter_out4_kernel with all cells empty).
// MODE 3: same region per workgroup, but each wave writes its 64 KiB share contiguously (plane-major per wave).
template <int MODE>
__global__ __launch_bounds__(1024) void wr(v4* __restrict__ y, size_t n4) {
    const v4 z = {1.f, 2.f, 3.f, 4.f};
    if (MODE <= 1) {
        const size_t nt = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += nt) {
            if (MODE == 0) __builtin_nontemporal_store(z, y + i);
            else y[i] = z;
        }
    } else if (MODE == 2) {
        v4* base = y + (size_t)blockIdx.x * 65536 + threadIdx.x;   // 1 MiB = 65536 float4
        for (int n = 0; n < 64; n += 16)
#pragma unroll
            for (int u = 0; u < 16; ++u) __builtin_nontemporal_store(z, base + (size_t)(n + u) * 1024);
    } else {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        v4* base = y + (size_t)blockIdx.x * 65536 + (size_t)wave * 4096 + lane;   // 64 KiB per wave
#pragma unroll 16
        for (int i = 0; i < 64; ++i) __builtin_nontemporal_store(z, base + (size_t)i * 64);
    }
}
template <int MODE> void run(const char* name, v4* y, size_t n4, int grid, int block) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    wr<MODE><<<grid, block>>>(y, n4);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) wr<MODE><<<grid, block>>>(y, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("%-44s grid=%5d block=%4d : %7.1f us  %6.0f GB/s\n", name, grid, block, ms * 1e3, n4 * 16.0 / ms / 1e6);
}
int main() {
    const size_t n4 = (size_t)4096 * 65536;   // 4 GiB
    v4* y; hipMalloc(&y, n4 * 16);
    run<0>("grid-stride nt", y, n4, 256 * 8, 256);
    run<0>("grid-stride nt", y, n4, 256 * 32, 256);
    run<1>("grid-stride plain", y, n4, 256 * 32, 256);
    run<0>("grid-stride nt", y, n4, 256 * 2, 1024);
    run<2>("scatter-like (16 KiB planes, 16 in flight)", y, n4, 4096, 1024);
    run<3>("per-wave contiguous 64 KiB", y, n4, 4096, 1024);
    return 0;
}
