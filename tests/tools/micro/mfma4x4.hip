// Layout probe of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction): lane l supplies A = a[l], B = b[l];
// prints, for result register r of lane l, which (a-lane, b-lane) product it holds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    // a[l] = 1000 + l, b[l] = 1 + l / 1024.0  -> product identifies both lanes
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(l + 1) * 100.f, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    k<<<1, 64>>>(d);
    float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            // hypothesis: D[reg r][lane l] = a[4 * (l / 4) + r] * b[l]
            const float want = (float)(4 * (l / 4) + r + 1) * ((float)(l + 1) * 100.f);
            if (h[l * 4 + r] != want) { ok = 0; if (l < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, r, h[l * 4 + r], want); }
        }
    printf("hypothesis D[r][l] = a[4*(l/4)+r] * b[l]: %s\n", ok ? "HOLDS" : "FAILS");
    for (int l = 0; l < 6; ++l) printf("lane %d: %.0f %.0f %.0f %.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
