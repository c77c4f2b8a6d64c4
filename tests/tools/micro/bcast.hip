// All-to-all broadcast step of lstm_mid.hpp in isolation: NWG persistent workgroups (1024 threads), per step every workgroup
// writes its own block of a slot (rows x 4 floats), publishes a flag, waits for all flags and reads the WHOLE slot
// (NWG * rows * 16 bytes) the way the kernel's matrix-core waves read their A operand.  Prints microseconds per step for
// several data paths.  usage: bcast [rows=64] [nwg=256] [steps=200] [flag replicas=1; 0 = sweep the flag hop alone]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef float vfloat4 __attribute__((ext_vector_type(4)));

// MODE 0: write-through stores, fresh slot per step, plain loads            (the kernel's protocol)
// MODE 1: the same, loads with sc1 (L2 bypass)
// MODE 2: flags only, no data read                                        (the hop itself)
// MODE 3: mode 0 reading a quarter of the slot
// MODE 4: plain stores + agent release fence / acquire fence, TWO slots reused (the first version's protocol)
// MODE 5: mode 0 with the load order rotated per workgroup
// MODE 6: mode 0, but only ONE workgroup per XCD-sized group of 32 reads first ... (others 2 us later)  [probe of L2 sharing]
template <int MODE>
__global__ __launch_bounds__(1024) void bcast(float* slots, unsigned* flags, int rows, int nwg, int steps, u64* out, float* sink, int rep) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const size_t slot = (size_t)nwg * rows * 4;
    const int mbp = rows / 16, ks = 16 / mbp, mb = wave % mbp, kq = wave / mbp;
    const int qper = nwg / ks;                       // unit quads (= workgroups) per k slice
    const int n16 = qper / 4;
    float acc = 0.f;
    u64 t0 = 0, tw = 0, tr = 0;
    for (int s = 0; s < steps; ++s) {
        float* my = slots + (MODE == 4 ? (size_t)(s & 1) : (size_t)s) * slot;
        // write own block
        if (tid < rows * 4) {
            float* p = my + (size_t)wg * rows * 4 + tid;
            if (MODE == 4) *p = (float)(s + wg);
            else __hip_atomic_store(p, (float)(s + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (MODE != 4) __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid < rep) {   // the flag, to `rep` replicas (workgroup w polls replica w % rep)
            if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(flags + tid * 256 + wg, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const u64 ta = wall_clock64();
        if (tid < 64) {
            while (true) {
                bool ok = true;
                for (int i = 0; i < 4; ++i) {
                    const int w = lane + 64 * i;
                    const unsigned v = __hip_atomic_load(flags + (wg % rep) * 256 + (w < nwg ? w : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (w >= nwg || v >= (unsigned)(s + 1));
                }
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (MODE == 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        const u64 tb = wall_clock64();
        if (MODE == 6 && (wg & 31) != 0) { const u64 t = wall_clock64(); while (wall_clock64() - t < 200) {} }
        if (MODE != 2) {
            const int r = 16 * mb + (lane & 15), j = lane >> 4;
            const float* ap = my + ((size_t)(kq * qper + j) * rows + r) * 4;
            const int lim = MODE == 3 ? (n16 + 3) / 4 : n16;
            const int rot = MODE == 5 ? wg % (lim > 0 ? lim : 1) : 0;
            for (int i0 = 0; i0 < lim; i0 += 16) {
                vfloat4 v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    int i = i0 + u < lim ? i0 + u : i0;
                    i = (i + rot) % lim;
                    const vfloat4* q = reinterpret_cast<const vfloat4*>(ap + (size_t)16 * rows * i);
                    if (MODE == 1) {
                        const u64* q8 = reinterpret_cast<const u64*>(q);
                        const u64 lo = __hip_atomic_load(q8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const u64 hi = __hip_atomic_load(q8 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        v[u] = vfloat4{__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)),
                                       __uint_as_float((unsigned)hi), __uint_as_float((unsigned)(hi >> 32))};
                    } else v[u] = *q;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
            }
        }
        __syncthreads();
        const u64 tc = wall_clock64();
        if (s == 10) t0 = wall_clock64();
        if (s >= 10) { tw += tb - ta; tr += tc - tb; }
    }
    if (wg == 0 && tid == 0) { out[0] = wall_clock64() - t0; out[1] = tw; out[2] = tr; }
    if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE> void run(const char* name, int rows, int nwg, int steps, int rep) {
    float *slots, *sink; unsigned* flags; u64* out;
    const size_t slot = (size_t)nwg * rows * 4;
    hipMalloc(&slots, slot * steps * 4); hipMalloc(&flags, 32 * 1024); hipMalloc(&out, 64); hipMalloc(&sink, 64);
    hipMemset(flags, 0, 32 * 1024); hipMemset(slots, 0, slot * steps * 4);
    bcast<MODE><<<nwg, 1024>>>(slots, flags, rows, nwg, steps, out, sink, rep);
    hipDeviceSynchronize();
    u64 h[3]; hipMemcpy(h, out, 24, hipMemcpyDeviceToHost);
    const double n = steps - 10;
    printf("%-60s rows=%3d nwg=%3d rep=%2d: %.2f us/step  (flag wait %.2f, slot read %.2f)  slot %zu KB\n", name, rows, nwg, rep, h[0] / 100.0 / n,
           h[1] / 100.0 / n, h[2] / 100.0 / n, slot * 4 / 1024);
    hipFree(slots); hipFree(flags); hipFree(out); hipFree(sink);
}

int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 64, nwg = argc > 2 ? atoi(argv[2]) : 256, steps = argc > 3 ? atoi(argv[3]) : 200;
    const int rep = argc > 4 ? atoi(argv[4]) : 1;
    if (rep == 0) {   // the flag hop alone, by the number of replicas
        for (int r : {1, 2, 4, 8, 16, 32}) run<2>("flags only (the hop)", rows, nwg, steps, r);
        return 0;
    }
    run<2>("flags only (the hop)", rows, nwg, steps, rep);
    run<0>("write-through, fresh slot, plain loads", rows, nwg, steps, rep);
    run<3>("  ... a quarter of the slot", rows, nwg, steps, rep);
    run<5>("  ... load order rotated per workgroup", rows, nwg, steps, rep);
    run<6>("  ... one workgroup per 32 reads 2 us ahead", rows, nwg, steps, rep);
    run<1>("write-through, fresh slot, sc1 loads", rows, nwg, steps, rep);
    run<4>("plain stores + release / acquire fences, two slots", rows, nwg, steps, rep);
    return 0;
}
