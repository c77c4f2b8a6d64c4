// Pure-write HBM bandwidth by OUTPUT SIZE and store pattern (follow-up of writebw.hip: a fill runs at 6.8 TB/s up to ~2 GB,
// the 4 GiB outputs of the scatter / one-hot kernels at 4.7-5.5 -- where does the step come from, and does locality of a
// workgroup's stores (pages, channels) move it?).  usage: writebw2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));

// MODE 0: grid-stride float4 nontemporal stores (consecutive workgroups write consecutive 4 KiB)
// MODE 1: the same with plain stores
// MODE 2: every workgroup writes ONE contiguous chunk of n4 / gridDim.x float4 (1024 threads stream through it)
// MODE 3: as 2, but the chunks are dealt so that the workgroups of one XCD (blockIdx % 8) own one contiguous eighth of the output
// MODE 4: every WAVE writes its own contiguous share of the workgroup's chunk
template <int MODE>
__global__ __launch_bounds__(1024) void wr(v4* __restrict__ y, size_t n4) {
    const v4 z = {1.f, 2.f, 3.f, 4.f};
    if (MODE <= 1) {
        const size_t nt = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += nt) {
            if (MODE == 0) __builtin_nontemporal_store(z, y + i);
            else y[i] = z;
        }
        return;
    }
    const size_t per = n4 / gridDim.x;
    size_t chunk = blockIdx.x;
    if (MODE == 3) chunk = (size_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    v4* base = y + chunk * per;
    if (MODE == 4) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
        const size_t pw = per / nw;
        v4* b = base + (size_t)wave * pw + lane;
#pragma unroll 8
        for (size_t i = 0; i < pw; i += 64) __builtin_nontemporal_store(z, b + i);
        return;
    }
#pragma unroll 8
    for (size_t i = threadIdx.x; i < per; i += blockDim.x) __builtin_nontemporal_store(z, base + i);
}
template <int MODE> void run(const char* name, v4* y, size_t n4, int grid, int block) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    wr<MODE><<<grid, block>>>(y, n4);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) wr<MODE><<<grid, block>>>(y, n4);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("  %-58s grid=%5d block=%4d : %8.1f us  %6.0f GB/s\n", name, grid, block, ms * 1e3, n4 * 16.0 / ms / 1e6);
}
int main() {
    v4* y; hipMalloc(&y, (size_t)8 << 30);
    for (size_t mib : {256, 1024, 2048, 3072, 4096, 8192}) {
        const size_t n4 = mib * 65536;
        printf("output %zu MiB\n", mib);
        run<0>("grid-stride nt", y, n4, 8192, 256);
        run<1>("grid-stride plain", y, n4, 8192, 256);
        run<2>("one contiguous chunk per workgroup", y, n4, 2048, 1024);
        run<2>("one contiguous chunk per workgroup", y, n4, 512, 1024);
        run<2>("one contiguous chunk per workgroup", y, n4, 8192, 256);
        run<3>("... chunks of an XCD's workgroups adjacent", y, n4, 2048, 1024);
        run<4>("one contiguous share per wave", y, n4, 2048, 1024);
        run<4>("one contiguous share per wave", y, n4, 8192, 256);
        {   // the runtime's own fill, for reference
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipMemsetD32Async((hipDeviceptr_t)y, 0x3f800000, n4 * 4, 0);
            hipEventRecord(e0);
            for (int i = 0; i < 5; ++i) hipMemsetD32Async((hipDeviceptr_t)y, 0x3f800000, n4 * 4, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("  %-58s                        : %8.1f us  %6.0f GB/s\n", "hipMemsetD32Async", ms * 1e3, n4 * 16.0 / ms / 1e6);
        }
    }
    return 0;
}
