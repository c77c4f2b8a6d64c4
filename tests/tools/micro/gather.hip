// What can a 128-byte-line gather reach?  IQN's forward reads ONE float of every (quantile, sample) row of N = 64 floats (256 B): half of
// the lines of q and next_n_q, chosen by the sample's action.  Variants of the lane -> row map, with the loss arithmetic left out:
//   A  lane = quantile (rows 16 MiB apart within a wave instruction), a 32-lane group per sample -- the shipped kernel's map
//   B  lane = sample (64 consecutive rows per wave instruction), one quantile per wave
//   C  lane = sample, U quantiles per thread in flight
//   S  streaming read of the same number of lines (contiguous), for scale
// usage: gather [B = 65536] [tau = 32] [N = 64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void gather_a(const float* __restrict__ q, const float* __restrict__ nq, const long* __restrict__ act,
                                                const long* __restrict__ nact, float* __restrict__ out, int tau, long B, int N) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long b = t / 32;
    const int gl = (int)(t % 32);
    if (b >= B || gl >= tau) return;
    const long a = act[b], na = nact[b];
    const float v = q[((size_t)gl * B + b) * N + a] + nq[((size_t)gl * B + b) * N + na];
    out[b * tau + gl] = v;
}
// the shipped map plus the other pieces of the forward, one bit each: 1 rq (tau x B, a line per lane), 2 the per-sample scalars
// (5 rewards, done, weight), 4 the pair loop, 8 the buf / td_err stores in their real layout, 16 the workgroup partial + barrier
template <int F>
__global__ __launch_bounds__(256) void gather_f(const float* __restrict__ q, const float* __restrict__ nq, const long* __restrict__ act,
                                                const long* __restrict__ nact, const float* __restrict__ rq, const float* __restrict__ rew,
                                                float* __restrict__ out, float* __restrict__ part, int tau, long B, int N) {
    __shared__ float red[8];
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long b = t / 32;
    const int gl = (int)(t % 32), base = (threadIdx.x & 63) / 32 * 32;
    const long a = act[b], na = nact[b];
    float rho = 0.5f, R = 0.f, vg = 0.99f, w = 1.f;
    if (F & 1) rho = rq[(size_t)gl * B + b];
    if (F & 2) {
        float r[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) r[i] = rew[(size_t)i * B + b];
        const float dn = rew[(size_t)5 * B + b];
        w = rew[(size_t)6 * B + b];
        float g = 1.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) { R = fmaf(g, r[i], R); g *= 0.99f; }
        vg = 0.95f * (1.f - dn);
    }
    const float qi = q[((size_t)gl * B + b) * N + a];
    const float tgt = fmaf(vg, nq[((size_t)gl * B + b) * N + na], R);
    float li = qi + tgt, gi = qi - tgt;
    if (F & 4) {
        const float kappa = 1.f, qneg = fabsf(rho - 1.f) / kappa, qpos = fabsf(rho) / kappa;
        li = 0.f; gi = 0.f;
        for (int j = 0; j < tau; ++j) {
            const float e = __shfl(tgt, base + j, 64) - qi;
            const float dh = __builtin_amdgcn_fmed3f(e, -kappa, kappa);
            const float hub = dh * fmaf(-0.5f, dh, e);
            const float qw = e < 0.f ? qneg : qpos;
            li = fmaf(qw, hub, li);
            gi = fmaf(qw, dh, gi);
        }
    }
    if (F & 8) {
        out[b * tau + gl] = -gi * w;
        float s = li;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (gl == 0) out[(size_t)tau * B + b] = s;
        li = s;
    } else {
        out[b * tau + gl] = li + gi;
    }
    if (F & 16) {
        if (gl == 0) red[threadIdx.x / 32] = li * w;
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int i = 0; i < 8; ++i) s += red[i];
            part[blockIdx.x] = s;
        }
    }
}
template <int U>
__global__ __launch_bounds__(256) void gather_c(const float* __restrict__ q, const float* __restrict__ nq, const long* __restrict__ act,
                                                const long* __restrict__ nact, float* __restrict__ out, int tau, long B, int N) {
    // grid: (B / 256) x (tau / U)
    const long b = (long)blockIdx.x * 256 + threadIdx.x;
    const int g0 = blockIdx.y * U;
    if (b >= B) return;
    const long a = act[b], na = nact[b];
    float v[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v[u] = q[((size_t)(g0 + u) * B + b) * N + a];
        w[u] = nq[((size_t)(g0 + u) * B + b) * N + na];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) out[(size_t)(g0 + u) * B + b] = v[u] + w[u];
}
// lane = sample, the workgroup walks every quantile of its 64-sample... one wave = 64 samples x all quantiles, U in flight
template <int U>
__global__ __launch_bounds__(256) void gather_d(const float* __restrict__ q, const float* __restrict__ nq, const long* __restrict__ act,
                                                const long* __restrict__ nact, float* __restrict__ out, int tau, long B, int N) {
    const long b = (long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int wv = threadIdx.x >> 6;                       // four waves split the quantiles
    if (b >= B) return;
    const long a = act[b], na = nact[b];
    float acc = 0.f;
    for (int g0 = wv * (tau / 4); g0 < (wv + 1) * (tau / 4); g0 += U) {
        float v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = q[((size_t)(g0 + u) * B + b) * N + a];
            w[u] = nq[((size_t)(g0 + u) * B + b) * N + na];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u] + w[u];
    }
    out[(size_t)wv * B + b] = acc;
}
__global__ __launch_bounds__(256) void stream_s(const float4* __restrict__ q, float* __restrict__ out, size_t n4) {
    // each thread reads 16 B; 8 threads per 128-byte line
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = q[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.f) out[0] = acc;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}
int main(int argc, char** argv) {
    const long B = argc > 1 ? atol(argv[1]) : 65536;
    const int tau = argc > 2 ? atoi(argv[2]) : 32, N = argc > 3 ? atoi(argv[3]) : 64;
    const size_t n = (size_t)tau * B * N;
    float *q, *nq, *out; long *act, *nact;
    hipMalloc(&q, n * 4); hipMalloc(&nq, n * 4); hipMalloc(&out, (size_t)(tau + 1) * B * 4 + 1024);
    hipMalloc(&act, B * 8); hipMalloc(&nact, B * 8);
    hipMemset(q, 0, n * 4); hipMemset(nq, 0, n * 4);
    std::vector<long> h(B), h2(B);
    unsigned s = 12345u;
    for (long i = 0; i < B; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s >> 16) % N; s = s * 1664525u + 1013904223u; h2[i] = (s >> 16) % N; }
    hipMemcpy(act, h.data(), B * 8, hipMemcpyHostToDevice); hipMemcpy(nact, h2.data(), B * 8, hipMemcpyHostToDevice);
    const double lines = 2.0 * tau * B, gb = lines * 128 / 1e9;
    printf("B=%ld tau=%d N=%d: %.0f lines = %.1f MB of lines (%.1f MB of rows)\n", B, tau, N, lines, gb * 1e3, 2.0 * n * 4 / 1e6);
    auto rep = [&](const char* name, float ms) { printf("  %-58s %8.1f us  %6.0f GB/s of lines\n", name, ms * 1e3, gb / ms * 1e3); };
    rep("A lane = quantile, 32-lane group per sample (shipped map)", timeit([&] { gather_a<<<(unsigned)((B * 32 + 255) / 256), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    float *rqd, *rew, *part;
    hipMalloc(&rqd, (size_t)tau * B * 4); hipMalloc(&rew, (size_t)7 * B * 4); hipMalloc(&part, (B / 8 + 64) * 4);
    hipMemset(rqd, 0, (size_t)tau * B * 4); hipMemset(rew, 0, (size_t)7 * B * 4);
#define RUNF(F_, name) rep(name, timeit([&] { gather_f<F_><<<(unsigned)(B / 8), 256>>>(q, nq, act, nact, rqd, rew, out, part, tau, B, N); }))
    RUNF(0, "A'  shipped map, gathers only");
    RUNF(1, "A' + rq");
    RUNF(2, "A' + per-sample scalars");
    RUNF(4, "A' + pair loop");
    RUNF(8, "A' + buf / td_err stores");
    RUNF(16, "A' + workgroup partial");
    RUNF(3, "A' + rq + scalars");
    RUNF(7, "A' + rq + scalars + pair loop");
    RUNF(15, "A' + rq + scalars + pair loop + stores");
    RUNF(31, "A' + everything");
    rep("C lane = sample, 1 quantile per thread", timeit([&] { gather_c<1><<<dim3((unsigned)(B / 256), tau), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("C lane = sample, 2 quantiles per thread", timeit([&] { gather_c<2><<<dim3((unsigned)(B / 256), tau / 2), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("C lane = sample, 4 quantiles per thread", timeit([&] { gather_c<4><<<dim3((unsigned)(B / 256), tau / 4), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("C lane = sample, 8 quantiles per thread", timeit([&] { gather_c<8><<<dim3((unsigned)(B / 256), tau / 8), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("D 64 samples per workgroup, 4 waves x tau/4, 2 in flight", timeit([&] { gather_d<2><<<(unsigned)(B / 64), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("D 64 samples per workgroup, 4 waves x tau/4, 4 in flight", timeit([&] { gather_d<4><<<(unsigned)(B / 64), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    rep("D 64 samples per workgroup, 4 waves x tau/4, 8 in flight", timeit([&] { gather_d<8><<<(unsigned)(B / 64), 256>>>(q, nq, act, nact, out, tau, B, N); }));
    {
        const size_t n4 = (size_t)(lines * 8);   // the same number of lines, contiguous
        const float ms = timeit([&] { stream_s<<<8192, 256>>>((const float4*)q, out, n4 < n / 4 ? n4 : n / 4); });
        rep("S contiguous stream of as many lines (q only)", ms);
        const float ms2 = timeit([&] { stream_s<<<8192, 256>>>((const float4*)q, out, n / 4); });
        printf("  %-58s %8.1f us  %6.0f GB/s\n", "S whole q (every line)", ms2 * 1e3, n * 4.0 / ms2 / 1e6);
    }
    return 0;
}
