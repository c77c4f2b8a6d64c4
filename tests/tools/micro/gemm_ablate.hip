// Where does the exact-fp32 MFMA GEMM (di-hpc_amd/csrc/gemm_f32.hpp) lose its time?  Ablation builds of the SAME
// kernel (template parameter ABL, results are wrong by construction) against the full kernel, plus the other tile
// shapes, on the NN products of the C4 LSTM:
//   ABL 1  MFMAs + LDS operand reads only          -> ceiling of the inner loop as written
//   ABL 2  + global prefetch into registers         -> cost of the HBM/L2 stream beside the MFMAs
//   ABL 3  + LDS store + barrier, no global loads   -> cost of the staging / synchronisation structure
//   ABL 0  everything
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I di-hpc_amd/csrc -I include tests/tools/micro/gemm_ablate.hip
//               -o tests/tools/micro/gemm_ablate.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#include "gemm_f32.hpp"

namespace hpc_rll {
int g_gemm_xcd = 1, g_gemm_bk = 0, g_gemm_tile256 = 0, g_gemm_lat_target = 256, g_gemm_thr_ktiles = 8, g_gemm_big_tile128 = 1,
    g_gemm_big_target = 768;
}
using namespace hpc_rll;

template <int BM, int BN, int BK, int WM, int WN, int ABL>
static double run(const char* tag, const float* A, const float* B, float* C, int M, int N, int K, int xcd) {
    GemmArgs g{A, B, C, M, N, K, (long)K, 1, (long)N, 1, (long)N, 0};
    g.xcd_swizzle = xcd;
    const dim3 grid(N / BN, M / BM, 1);
    auto k = gemm_f32_kernel<BM, BN, BK, WM, WN, kContigK, kContigMN, true, ABL>;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, g);
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, grid, dim3(256), 0, 0, g);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms / 5 < best) best = ms / 5;
    }
    const double tf = 2.0 * M * N * K / (best * 1e-3) / 1e12;
    printf("%-34s M=%6d N=%5d K=%5d  %8.3f ms  %6.1f TF  (%4.1f %% of 157.3)\n", tag, M, N, K, best, tf, tf / 157.3 * 100);
    fflush(stdout);
    return tf;
}

int main() {
    const int shapes[3][3] = {{4096, 4096, 4096}, {4096, 4096, 1024}, {65536, 4096, 1024}};
    for (auto& s : shapes) {
        const int M = s[0], N = s[1], K = s[2];
        float *A, *B, *C;
        hipMalloc(&A, (size_t)M * K * 4);
        hipMalloc(&B, (size_t)K * N * 4);
        hipMalloc(&C, (size_t)M * N * 4);
        std::vector<float> h((size_t)(M > N ? M : N) * K);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f - 0.5f;
        hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
        hipMemcpy(B, h.data(), (size_t)K * N * 4, hipMemcpyHostToDevice);
        run<128, 128, 16, 2, 2, 0>("128x128x16 full", A, B, C, M, N, K, 1);
        run<128, 128, 16, 2, 2, 1>("128x128x16 ABL1 mfma+ds_read", A, B, C, M, N, K, 1);
        run<128, 128, 16, 2, 2, 2>("128x128x16 ABL2 +global prefetch", A, B, C, M, N, K, 1);
        run<128, 128, 16, 2, 2, 3>("128x128x16 ABL3 +lds store+barrier", A, B, C, M, N, K, 1);
        run<128, 128, 16, 2, 2, 0>("128x128x16 full, no xcd order", A, B, C, M, N, K, 0);
        run<128, 128, 32, 2, 2, 0>("128x128x32 full", A, B, C, M, N, K, 1);
        run<128, 128, 32, 2, 2, 1>("128x128x32 ABL1", A, B, C, M, N, K, 1);
        run<256, 128, 16, 4, 2, 0>("256x128x16 full", A, B, C, M, N, K, 0);
        run<256, 128, 16, 4, 2, 1>("256x128x16 ABL1", A, B, C, M, N, K, 0);
        run<128, 64, 16, 2, 1, 0>("128x64x16 full", A, B, C, M, N, K, 1);
        hipFree(A);
        hipFree(B);
        hipFree(C);
    }
    return 0;
}
