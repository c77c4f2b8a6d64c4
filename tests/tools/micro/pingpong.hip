// Ping-pong latency between two workgroups through global memory (the {value,tag} exchange of lstm_persist.hpp).
// usage: pingpong   -> prints one-way latency in ns for same-XCD / cross-XCD peers and several access flavours
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;

template <int MODE> __device__ __forceinline__ void put(u64* p, u64 v) {
    if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 2) __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int MODE> __device__ __forceinline__ u64 get(u64* p) {
    if (MODE == 0 || MODE == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (MODE == 3) {   // bypass the per-CU L1 only: coherent inside one XCD (L2), NOT across XCDs
        u64 v;
        asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
        return v;
    }
    return 0;
}

template <int MODE>
__global__ void pingpong(u64* a, u64* b, int peer, int iters, u64* out, unsigned* xcc) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x != 0 && blockIdx.x != peer) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x == 0 ? 0 : 1] = id & 0xf;
    const u64 t0 = wall_clock64();
    long guard = 0;
    for (int i = 1; i <= iters; ++i) {
        if (blockIdx.x == 0) {
            put<MODE>(a, (u64)i);
            while (get<MODE>(b) != (u64)i) if (++guard > (1L << 26)) return;
        } else {
            while (get<MODE>(a) != (u64)i) if (++guard > (1L << 26)) return;
            put<MODE>(b, (u64)i);
        }
    }
    if (blockIdx.x == 0) out[0] = wall_clock64() - t0;
}

template <int MODE> void run(const char* name, int peer) {
    u64 *buf, *out; unsigned* xcc;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 8); hipMalloc(&xcc, 8);
    hipMemset(buf, 0, 1 << 20); hipMemset(out, 0, 8);
    const int iters = 2000;
    pingpong<MODE><<<peer + 1, 64>>>(buf, buf + 8192, peer, iters, out, xcc);
    hipDeviceSynchronize();
    u64 h = 0; unsigned x[2];
    hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
    printf("%-28s peer=%3d xcc %u->%u : one-way %.0f ns\n", name, peer, x[0], x[1], h * 10.0 / iters / 2);
    hipFree(buf); hipFree(out); hipFree(xcc);
}

int main() {
    for (int peer : {1, 8, 4, 16}) {
        run<0>("agent store/load (sc1)", peer);
        run<1>("system store/load", peer);
        run<2>("agent xchg / load", peer);
    }
    run<3>("L1-bypass only (sc0), same XCD", 8);
    run<3>("L1-bypass only (sc0), same XCD", 16);
    return 0;
}
