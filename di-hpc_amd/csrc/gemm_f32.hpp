// gemm_f32.hpp -- exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32) for the LSTM gate GEMMs.
//
// Replaces the reference's cublasSgemm calls (src/torch_utils/network/lstm.cu:121-123,145-147,352-370).
// The f32-input MFMA is bit-for-bit an fmaf chain in k order (no TF32-style truncation exists on gfx950), so the
// results are plain fp32 matmul results at the 157 TFLOP/s matrix rate.
//
//   C (M x N, row stride ldc) (+)= A (M x K) * B (K x N)
// with arbitrary element strides for A and B, so the three layouts the LSTM needs are one kernel:
//   NN  x @ W          A(m,k) = A[m*lda + k]   B(k,n) = B[k*ldb + n]
//   NT  dY @ W^T       A(m,k) = A[m*lda + k]   B(k,n) = W[n*ldb + k]
//   TN  X^T @ dY       A(m,k) = X[k*lda + m]   B(k,n) = B[k*ldb + n]
//
// Tiling: workgroup = 4 waves (2 x 2), block tile 128 x 128 x 16; each wave owns a 64 x 64 quadrant as 2 x 2
// MFMA blocks of 32 x 32 (4 x f32x16 accumulators).  Operand tiles are staged in LDS k-major
// (As[k][m], Bs[k][n]) so the MFMA operand fetch  a = As[k0 + (lane>>5)][m0 + (lane&31)]  is a conflict-free
// ds_read_b32 (the two 32-lane halves are separate LDS lane groups).  The next tile's global loads are issued
// into registers before the current tile's MFMAs and written to the other LDS buffer afterwards (register
// prefetch + LDS double buffer, one barrier per k-tile).  A skinny variant (BM = 32, tile 32 x 256) serves the
// per-timestep recurrent GEMM when the batch is small.
#pragma once
#include <hip/hip_runtime.h>

namespace hpc_rll {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    long a_sm, a_sk;   // A(m,k) = A[m*a_sm + k*a_sk]
    long b_sk, b_sn;   // B(k,n) = B[k*b_sk + n*b_sn]
    long ldc;
    int accumulate;    // C += A*B instead of C = A*B
};

// BM x BN block tile, BK = 16, 256 threads.  WM x WN = MFMA blocks per wave; waves arranged (BM/(32*WM)) x (BN/(32*WN)).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int BK = 16;
    constexpr int WAVES_M = BM / (32 * WM);
    static_assert(WAVES_M * (BN / (32 * WN)) == 4, "4 waves per workgroup");
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ float lds[2 * BK * LDA + 2 * BK * LDB];
    float* const As = lds;                       // [buf][BK][LDA]
    float* const Bs = lds + 2 * BK * LDA;        // [buf][BK][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // ---- global -> register staging.  Element e of the A tile is (row = e % BM, k = e / BM) when A is contiguous
    // along m, (row = e / BK, k = e % BK) when contiguous along k; each thread owns AE = BM*BK/256 elements.
    constexpr int AE = BM * BK / 256, BE = BN * BK / 256;
    const bool a_mc = (g.a_sm == 1);   // contiguous along m (TN)
    const bool b_nc = (g.b_sn == 1);   // contiguous along n (NN, TN)
    float ra[AE], rb[BE];

    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AE; ++i) {
            const int e = tid + i * 256;
            const int row = a_mc ? (e % BM) : (e / BK);
            const int kk = a_mc ? (e / BM) : (e % BK);
            const int m = m0 + row, k = k0 + kk;
            ra[i] = (m < g.M && k < g.K) ? g.A[(long)m * g.a_sm + (long)k * g.a_sk] : 0.f;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < BE; ++i) {
            const int e = tid + i * 256;
            const int col = b_nc ? (e % BN) : (e / BK);
            const int kk = b_nc ? (e / BN) : (e % BK);
            const int n = n0 + col, k = k0 + kk;
            rb[i] = (n < g.N && k < g.K) ? g.B[(long)k * g.b_sk + (long)n * g.b_sn] : 0.f;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < AE; ++i) {
            const int e = tid + i * 256;
            const int row = a_mc ? (e % BM) : (e / BK);
            const int kk = a_mc ? (e / BM) : (e % BK);
            As[(buf * BK + kk) * LDA + row] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BE; ++i) {
            const int e = tid + i * 256;
            const int col = b_nc ? (e % BN) : (e / BK);
            const int kk = b_nc ? (e / BN) : (e % BK);
            Bs[(buf * BK + kk) * LDB + col] = rb[i];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int ktiles = (g.K + BK - 1) / BK;
    load_a(0);
    load_b(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < ktiles; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < ktiles) { load_a((kt + 1) * BK); load_b((kt + 1) * BK); }
        const float* __restrict__ as = As + buf * BK * LDA + wm * 32 * WM + (lane & 31);
        const float* __restrict__ bs = Bs + buf * BK * LDB + wn * 32 * WN + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < BK; ks += 2) {
            const int kr = ks + (lane >> 5);
            float a[WM], b[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a[i] = as[kr * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < WN; ++j) b[j] = bs[kr * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < ktiles) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + wn * 32 * WN + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < g.M && n < g.N) {
                    float* p = g.C + (long)m * g.ldc + n;
                    *p = g.accumulate ? (*p + acc[i][j][r]) : acc[i][j][r];
                }
            }
        }
}

inline void launch_gemm(const GemmArgs& g, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0) return;
    if (g.M <= 32) {   // skinny: 32 x 256 tile, waves side by side along N
        const dim3 grid((g.N + 255) / 256, (g.M + 31) / 32);
        hipLaunchKernelGGL((gemm_f32_kernel<32, 256, 1, 2>), grid, dim3(256), 0, st, g);
    } else {
        const dim3 grid((g.N + 127) / 128, (g.M + 127) / 128);
        hipLaunchKernelGGL((gemm_f32_kernel<128, 128, 2, 2>), grid, dim3(256), 0, st, g);
    }
}

}  // namespace hpc_rll
