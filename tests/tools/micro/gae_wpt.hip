// GAE forward with ONE TRAJECTORY PER WAVEFRONT -- the mapping BASELINE.json's north_star names -- as a stand-alone measurement.
// It was part of the library (flags bit 4 of hpc_rll_gae_forward_ex) in rounds 3-4, measured 2-8x slower than the shipped
// lane-per-column kernels on every shape (profiles/r03_gae_wpt_probe.txt) and moved here in round 5.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I di-hpc_amd/csrc -I include gae_wpt.hip -o gae_wpt.bin && ./gae_wpt.bin [T B]
// Checks the kernel against a host loop on a small shape, then times it (HIP events) at T x B (default 1024 x 65536).
//
// A workgroup of 16 waves owns 64 columns and walks T in tiles of 64 steps: the (65 x 64) value rows and (64 x 64) reward rows of
// a tile are loaded COALESCED along B and staged through LDS, then a wave takes one trajectory (column) at a time -- four per
// wave -- with its 64 LANES ALONG TIME: delta_t from three conflict-free LDS reads (row stride 65), an inclusive suffix scan of
// the affine pairs (c_t, delta_t) in log2(64) = 6 wavefront-shuffle steps, the carry from the later tile applied through the
// product, the result written back through LDS and stored coalesced.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "wave.hpp"
using namespace hpc_rll;

namespace {
template <bool NTL, bool NTS>
__global__ __launch_bounds__(1024) void gae_fwd_wpt_kernel(const float* __restrict__ value,
                                                           const float* __restrict__ reward,
                                                           float* __restrict__ adv, const float* __restrict__ coef,
                                                           int T, int B, float gamma) {
    constexpr int TT = 64, LD = 65;
    __shared__ float sv[(TT + 1) * LD], sr[TT * LD], so[TT * LD];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const long c0 = (long)blockIdx.x * 64;
    const bool col_ok = c0 + lane < (long)B;
    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t1 = T; t1 > 0; t1 -= TT) {
        const int t0 = t1 - TT;                      // may be negative in the last (earliest) tile
        // ---- load phase: wave w takes rows w, w+16, ... (lane <-> column: 256-byte coalesced rows)
        for (int r = w; r <= TT; r += 16) {
            const int t = t0 + r;
            if (t >= 0 && col_ok) sv[r * LD + lane] = ld<NTL>(value + (size_t)t * B + c0 + lane);
        }
        for (int r = w; r < TT; r += 16) {
            const int t = t0 + r;
            if (t >= 0 && col_ok) sr[r * LD + lane] = ld<NTL>(reward + (size_t)t * B + c0 + lane);
        }
        __syncthreads();
        // ---- scan phase: lane <-> time step t0 + lane, wave <-> trajectories 4w .. 4w+3
        const int t = t0 + lane;
        const bool valid = t >= 0;
        float P = valid ? coef[t] : 1.f;
        float b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * w + k;
            const float v0 = sv[lane * LD + c], v1 = sv[(lane + 1) * LD + c], rr = sr[lane * LD + c];
            b[k] = valid ? fmaf(gamma, v1, rr) - v0 : 0.f;
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float an = __shfl_down(P, d, 64);
            const bool ok = lane + d < 64;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float bn = __shfl_down(b[k], d, 64);
                if (ok) b[k] = fmaf(P, bn, b[k]);
            }
            if (ok) P *= an;
        }
        const int first = t0 < 0 ? -t0 : 0;          // the earliest valid lane holds the tile's head
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            b[k] = fmaf(P, carry[k], b[k]);
            carry[k] = __shfl(b[k], first, 64);
            so[lane * LD + 4 * w + k] = b[k];
        }
        __syncthreads();
        // ---- store phase: coalesced rows
        for (int r = w; r < TT; r += 16) {
            const int tt = t0 + r;
            if (tt >= 0 && col_ok) st<NTS>(adv + (size_t)tt * B + c0 + lane, so[r * LD + lane]);
        }
        // the next tile's loads write sv / sr, which this tile's scan phase finished reading before the barrier above;
        // its scan phase writes `so` only after ITS first barrier, behind this store phase
    }
}
}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static void coef_host(std::vector<float>& c, int T, double gamma, double lam) {
    auto S = [&](int k) { return k <= 0 ? 0.0 : (lam == 1.0 ? (double)k : (1.0 - std::pow(lam, k)) / (1.0 - lam)); };
    for (int t = 0; t < T; ++t) c[t] = (float)(gamma * lam * S(T - t - 1) / S(T - t));
}
static double run(int T, int B, bool check) {
    const float gamma = 0.99f, lam = 0.97f;
    std::vector<float> v((size_t)(T + 1) * B), r((size_t)T * B), c(T), out((size_t)T * B);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& x : v) x = rnd();
    for (auto& x : r) x = rnd();
    coef_host(c, T, gamma, lam);
    float *dv, *dr, *dc, *da;
    CK(hipMalloc(&dv, v.size() * 4)); CK(hipMalloc(&dr, r.size() * 4)); CK(hipMalloc(&dc, c.size() * 4)); CK(hipMalloc(&da, out.size() * 4));
    CK(hipMemcpy(dv, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dr, r.data(), r.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    const dim3 grid((unsigned)((B + 63) / 64)), block(1024);
    hipLaunchKernelGGL((gae_fwd_wpt_kernel<false, true>), grid, block, 0, 0, dv, dr, da, dc, T, B, gamma);
    CK(hipDeviceSynchronize());
    double res = 0;
    if (check) {
        CK(hipMemcpy(out.data(), da, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int b = 0; b < B; ++b) {
            float a = 0.f;
            for (int t = T - 1; t >= 0; --t) {
                const float delta = std::fmaf(gamma, v[(size_t)(t + 1) * B + b], r[(size_t)t * B + b]) - v[(size_t)t * B + b];
                a = std::fmaf(c[t], a, delta);
                worst = std::fmax(worst, std::fabs((double)a - out[(size_t)t * B + b]) / std::fmax(1.0, std::fabs((double)a)));
            }
        }
        res = worst;
    } else {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gae_fwd_wpt_kernel<true, true>), grid, block, 0, 0, dv, dr, da, dc, T, B, gamma);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        res = ms / 20 * 1e3;
    }
    CK(hipFree(dv)); CK(hipFree(dr)); CK(hipFree(dc)); CK(hipFree(da));
    return res;
}
int main(int argc, char** argv) {
    const int T = argc > 2 ? atoi(argv[1]) : 1024, B = argc > 2 ? atoi(argv[2]) : 65536;
    printf("wave-per-trajectory GAE forward: max rel. difference to a host loop at T=200 B=200: %.3g\n", run(200, 200, true));
    const double us = run(T, B, false);
    printf("T=%d B=%d: %.1f us per launch = %.0f GB/s of the 12 T B algorithmic bytes\n", T, B, us, 12.0 * T * B / us * 1e-3);
    return 0;
}
