// The C ABI on the GPU with no Python and no torch: hipMalloc'd buffers, hpc_rll_gae_coef/forward/backward on a stream,
// results checked against a host double-precision evaluation of the same recurrences (reference semantics:
// hpc_rll/origin/gae.py:28-37).  Built with hipcc and run by tests/test_gae_gpu.py::test_c_abi_program_on_gpu.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "hpc_rll_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("hip error %d at %s\n", (int)e_, #x); return 10; } } while (0)
#define CHECK_RLL(x) do { int s_ = (x); if (s_ != 0) { std::printf("%s -> %s\n", #x, hpc_rll_status_string(s_)); return 11; } } while (0)

int main() {
    const int T = 37, B = 300;
    const float gamma = 0.99f, lambda = 0.97f;
    std::vector<float> value((size_t)(T + 1) * B), reward((size_t)T * B), gadv((size_t)T * B);
    unsigned z = 12345u;
    auto rnd = [&]() { z = z * 1664525u + 1013904223u; return (float)((z >> 8) & 0xffff) / 32768.f - 1.f; };
    for (auto& v : value) v = rnd();
    for (auto& v : reward) v = rnd();
    for (auto& v : gadv) v = rnd();

    float *d_v, *d_r, *d_a, *d_c, *d_g, *d_gv, *d_gr;
    CHECK_HIP(hipMalloc(&d_v, value.size() * 4)); CHECK_HIP(hipMalloc(&d_r, reward.size() * 4));
    CHECK_HIP(hipMalloc(&d_a, reward.size() * 4)); CHECK_HIP(hipMalloc(&d_c, T * 4));
    CHECK_HIP(hipMalloc(&d_g, reward.size() * 4)); CHECK_HIP(hipMalloc(&d_gv, value.size() * 4));
    CHECK_HIP(hipMalloc(&d_gr, reward.size() * 4));
    hipStream_t st;
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_HIP(hipMemcpyAsync(d_v, value.data(), value.size() * 4, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(d_r, reward.data(), reward.size() * 4, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(d_g, gadv.data(), gadv.size() * 4, hipMemcpyHostToDevice, st));
    CHECK_RLL(hpc_rll_gae_coef(d_c, T, gamma, lambda, st));
    CHECK_RLL(hpc_rll_gae_forward(d_v, d_r, d_a, d_c, T, B, gamma, st));
    CHECK_RLL(hpc_rll_gae_backward(d_g, d_gv, d_gr, d_c, T, B, gamma, st));
    std::vector<float> adv(reward.size()), gv(value.size()), gr(reward.size());
    CHECK_HIP(hipMemcpyAsync(adv.data(), d_a, adv.size() * 4, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(gv.data(), d_gv, gv.size() * 4, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(gr.data(), d_gr, gr.size() * 4, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));

    // host reference: D_t = 1 + lambda D_{t+1}; G_t = D_t delta_t + gamma lambda G_{t+1}; adv_t = G_t / D_t, and its adjoint
    double max_err = 0.0, max_ref = 0.0;
    std::vector<double> D(T + 1, 0.0), c(T, 0.0);
    for (int t = T - 1; t >= 0; --t) D[t] = 1.0 + lambda * D[t + 1];
    for (int t = 0; t < T; ++t) c[t] = (double)gamma * lambda * D[t + 1] / D[t];
    for (int b = 0; b < B; ++b) {
        double a = 0.0;
        for (int t = T - 1; t >= 0; --t) {
            const double delta = reward[(size_t)t * B + b] + (double)gamma * value[(size_t)(t + 1) * B + b] - value[(size_t)t * B + b];
            a = delta + c[t] * a;
            max_err = std::fmax(max_err, std::fabs(a - adv[(size_t)t * B + b]));
            max_ref = std::fmax(max_ref, std::fabs(a));
        }
        double d = 0.0, dprev = 0.0;
        for (int t = 0; t < T; ++t) {
            d = gadv[(size_t)t * B + b] + (t > 0 ? c[t - 1] * dprev : 0.0);
            max_err = std::fmax(max_err, std::fabs(d - gr[(size_t)t * B + b]));
            max_err = std::fmax(max_err, std::fabs((-d + (double)gamma * dprev) - gv[(size_t)t * B + b]));
            dprev = d;
        }
        max_err = std::fmax(max_err, std::fabs((double)gamma * dprev - gv[(size_t)T * B + b]));
    }
    std::printf("max abs error %.3e (scale %.3f)\n", max_err, max_ref);
    if (!(max_err <= 1e-5 * std::fmax(1.0, max_ref))) return 1;
    std::printf("c abi gpu ok\n");
    return 0;
}
