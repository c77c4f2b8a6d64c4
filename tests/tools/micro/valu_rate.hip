// VALU issue cost per wave64 instruction on gfx950, by instruction kind (round 5): the n-step TD forwards are priced against a
// "VALU-issue" ceiling, and that needs the cycles an instruction occupies its SIMD for -- 4 (a 16-lane pipe), 2 (32 lanes), and
// whether a packed-fp32 instruction costs one slot or two.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rate.hip -o valu_rate.bin && ./valu_rate.bin
// Every wave runs ITER x 64 independent instructions of one kind on 16 register sets (no dependences inside the unrolled body
// closer than 16 instructions); 256 CUs x 4 SIMDs x W waves per SIMD.  cycles/instr/SIMD = time x clock / (W x ITER x 64).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 a[16];
    float s = seed + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v2{s + i, s - i};
    const v2 m = {1.0001f, 0.9999f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
                if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(m.y));
                if (KIND == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
                if (KIND == 7) asm volatile("v_log_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i].x));
                if (KIND == 9) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i].x) : "v"(m.x));
                if (KIND == 10) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 11) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i].x), "v"(m.x) : "vcc");
                if (KIND == 12) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i].x));
                if (KIND == 13) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += a[i].x + a[i].y;
    if (t == 12345.678f) out[0] = t;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, float* out) {
    const int iters = 2000;
    const int blocks = 256 * waves_per_simd;           // 256-thread workgroups = 4 waves: one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double insts = (double)waves_per_simd * iters * 64;
    printf("%-28s waves/SIMD %d: %8.1f us  -> %.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, best * 1e3,
           best * 1e-3 * 2.4e9 / insts);
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, out);
        run<1>("v_pk_fma_f32", w, out);
    }
    run<2>("v_add_f32", 4, out);
    run<3>("v_pk_add_f32", 4, out);
    run<4>("v_med3_f32", 4, out);
    run<5>("v_cndmask_b32 (vcc)", 4, out);
    run<11>("v_cmp_gt_f32 -> vcc", 4, out);
    run<6>("v_mov_b32_dpp quad_perm", 4, out);
    run<12>("v_add_f32_dpp row_shr:1", 4, out);
    run<7>("v_log_f32", 4, out);
    run<8>("v_rcp_f32", 4, out);
    run<9>("v_mul_lo_u32", 4, out);
    run<10>("v_lshl_add_u64", 4, out);
    run<13>("v_fma_f64", 4, out);
    return 0;
}
