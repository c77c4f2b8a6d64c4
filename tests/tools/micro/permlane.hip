// Semantics of v_permlane16_swap / v_permlane32_swap (gfx950): print where each lane's value ends up.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* o) {
    const float v = (float)threadIdx.x, w = 100.f + (float)threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    o[threadIdx.x] = __builtin_bit_cast(float, r[0]);
    o[64 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    o[128 + threadIdx.x] = __builtin_bit_cast(float, q[0]);
    o[192 + threadIdx.x] = __builtin_bit_cast(float, q[1]);
}
int main() {
    float h[256], *o;
    hipMalloc(&o, 1024);
    k<<<1, 64>>>(o);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    const char* names[4] = {"p16.r0", "p16.r1", "p32.r0", "p32.r1"};
    for (int t = 0; t < 4; ++t) {
        printf("%s:", names[t]);
        for (int i = 0; i < 64; i += 8) printf(" [%d]=%g", i, h[t * 64 + i]);
        printf("\n");
    }
    return 0;
}
