// What does the runtime's fill do that the kernels do not?  hipMemsetD32Async writes 4 GiB at 6.6 TB/s, the library's store
// patterns reach 5.0-5.7 (writebw2.hip).  Grid-stride 16-byte stores by grid size, workgroup size, store flavour and stores in
// flight per thread.  usage: writebw3 [MiB = 4096]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT, int U>
__global__ __launch_bounds__(1024) void wr(v4* __restrict__ y, size_t n4) {
    const v4 z = {1.f, 2.f, 3.f, 4.f};
    const size_t nt = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * nt < n4; i += U * nt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT == 1) __builtin_nontemporal_store(z, y + i + u * nt);
            else y[i + u * nt] = z;
        }
    }
    for (; i < n4; i += nt) y[i] = z;
}
// consecutive 16-byte elements per THREAD (U of them): a wave's store instruction then covers 64 x 16 B at a 16 U byte stride
template <int NT, int U>
__global__ __launch_bounds__(1024) void wr_thread_contig(v4* __restrict__ y, size_t n4) {
    const v4 z = {1.f, 2.f, 3.f, 4.f};
    const size_t nt = (size_t)gridDim.x * blockDim.x;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * U; i + U <= n4; i += U * nt) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT == 1) __builtin_nontemporal_store(z, y + i + u);
            else y[i + u] = z;
        }
    }
}
template <class K> void run(const char* name, K k, v4* y, size_t n4, int grid, int block) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<grid, block>>>(y, n4);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) k<<<grid, block>>>(y, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("  %-40s grid=%6d block=%4d : %8.1f us  %6.0f GB/s\n", name, grid, block, ms * 1e3, n4 * 16.0 / ms / 1e6);
}
int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? atoi(argv[1]) : 4096, n4 = mib * 65536;
    v4* y; hipMalloc(&y, n4 * 16);
    printf("output %zu MiB\n", mib);
    if (argc > 3) {   // third: does the alignment of the output matter?  (offset in bytes, multiple of 16)
        for (int off : {0, 512, 1024, 2048, 4096, 4096 + 512, 65536 + 1536}) {
            char name[64];
            snprintf(name, sizeof name, "plain, 1 in flight, base + %d B", off);
            run(name, wr<0, 1>, y + off / 16, n4 - 65536, 256, 256);
        }
    } else if (argc > 2) {   // second sweep: few waves per CU
        for (int block : {64, 128, 256, 384})
            for (int grid : {128, 256, 304, 512, 768}) {
                run("plain, 1 in flight", wr<0, 1>, y, n4, grid, block);
                run("plain, 2 in flight", wr<0, 2>, y, n4, grid, block);
                run("nt,    1 in flight", wr<1, 1>, y, n4, grid, block);
            }
    } else {
    for (int block : {256, 512, 1024})
        for (int grid : {256, 512, 1024, 2048, 4096, 16384, 65536}) {
            run("plain, 1 in flight", wr<0, 1>, y, n4, grid, block);
            run("plain, 4 in flight", wr<0, 4>, y, n4, grid, block);
            run("nt,    4 in flight", wr<1, 4>, y, n4, grid, block);
        }
    for (int grid : {1024, 4096, 16384}) {
        run("plain, 4 x 16 B per thread contiguous", wr_thread_contig<0, 4>, y, n4, grid, 256);
        run("nt,    4 x 16 B per thread contiguous", wr_thread_contig<1, 4>, y, n4, grid, 256);
    }
    }
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipMemsetD32Async((hipDeviceptr_t)y, 0x3f800000, n4 * 4, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipMemsetD32Async((hipDeviceptr_t)y, 0x3f800000, n4 * 4, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("  %-40s                         : %8.1f us  %6.0f GB/s\n", "hipMemsetD32Async", ms * 1e3, n4 * 16.0 / ms / 1e6);
    }
    return 0;
}
