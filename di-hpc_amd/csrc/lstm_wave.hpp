// lstm_wave.hpp -- layer-wavefront variant of the persistent small-batch LSTM forward (included by lstm.hip after
// lstm_persist.hpp, whose exchange helpers it uses).
//
// lstm_persist.hpp runs the L layers one after another: S*L dependent steps of ~7 us, each dominated by its two
// all-to-all exchanges.  Layer l+1 at time s only needs h_l[s] and h_{l+1}[s-1], so here ONE launch runs all layers
// concurrently as a wavefront -- grid (nwg, L), layer l at time s while layer l+1 is at s-1: S+L-1 dependent steps
// instead of S*L.  Differences to the per-layer kernel:
//   * layers > 0 compute their x-branch in the kernel: the workgroup also keeps its 4*JW columns of Wx in LDS and
//     multiplies the lower layer's h_{l-1}[s] (gathered in the same poll round as h_l[s-1]); the x-branch LayerNorm
//     partials ride in the same exchange as the h-branch ones.  Layer 0's x-branch stays one MFMA GEMM before the launch.
//   * a lower layer may run arbitrarily far ahead of the one above, so exchange slots are never reused: every
//     (layer, step) has its own tagged h and sums slot (a few MB for the shapes this path accepts; larger
//     S*L*B*H falls back to the per-layer kernels), no parity buffers, no flow control.
//   * inter-layer dropout is applied to the gathered h_{l-1}[s] with the same stateless hash the dropout kernel uses;
//     the dropped-out sequences the backward GEMMs need are still materialised by that kernel afterwards.
// Co-residency of all L*nwg workgroups is checked against the occupancy the runtime reports, with a margin.
#pragma once
#include <hip/hip_runtime.h>

namespace hpc_rll {
int g_lstm_wave = 1;   // hpc_rll_tune_set key 8
namespace {

struct WaveFwd {
    const float *xw0 /* layer 0 x-branch (S,B,4H), precomputed */, *wx /* flat, all layers */, *wh, *bias, *gamma, *beta;
    const float *h0, *c0;
    float *xw, *hw, *gates, *c, *hseq, *stats;   // layer 0 pointers; layer l at + l*layer_stride
    size_t layer_stride;
    u64 *hx, *sx;                                 // [L][S][B*H] and [L][S][4*B*nwg] tagged words
    int S, B, I, H, L, nwg;
    uint64_t seed;
    uint32_t drop_threshold;                      // 0: no dropout
    float drop_scale;
};

// Poll words e0 + 256*i (i < CH, those with their `valid` bit set) of the concatenation src1[0,n1) ++ src2[0,n2) until
// every one carries its tag (tag-1 for src1, tag for src2).
// PAR = false: each load sits under `if (valid)`, which compiles to a branch per word with s_waitcnt vmcnt(0) inside: the
// words of a poll are fetched one round trip after the other.  PAR = true: all CH loads are issued unconditionally (an
// out-of-range slot reads word 0 of a valid array, result ignored) before the first check: one round trip per poll.
// Measured in alternating processes on one box (tests/tools/lstm_lib_ab.sh): the backward gather of 2*B*4H words gains
// 6-26 % per launch from PAR (S=64,B=3,H=384,L=3: 1.63 -> 1.24 ms), the forward gather of 2*B*H words gains 6 % at
// 9 words per thread and LOSES 3.5 % at 2 (every waiting thread then hammers the path the producers' stores need), and
// the few-word row-sum exchanges lose 15 % (they stay on xchg_get<.., false>).
template <int CH, bool PAR, class MaskT>
__device__ __forceinline__ void wave_poll(const u64* src1, int n1, const u64* src2, int n2, int e0, MaskT valid,
                                          uint32_t tag, u64 (&w)[CH]) {
    long spins = 0;
    const u64* const safe = n1 > 0 ? src1 : src2;
    while (true) {
        bool ok = true;
        if (PAR) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int e = e0 + 256 * i;
                const u64* pp = ((valid >> i) & 1) ? (e < n1 ? src1 + e : src2 + (e - n1)) : safe;
                w[i] = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const bool first = e0 + 256 * i < n1;
                ok = ok && (!((valid >> i) & 1) || (uint32_t)(w[i] >> 32) == (first ? tag - 1u : tag));
            }
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const int e = e0 + 256 * i;
                if ((valid >> i) & 1) {
                    const bool first = e < n1;
                    w[i] = __hip_atomic_load(first ? src1 + e : src2 + (e - n1), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && ((uint32_t)(w[i] >> 32) == (first ? tag - 1u : tag));
                }
            }
        }
        if (ok) return;
        persist_poll_failed(spins);
    }
}

template <int NB, int JW, bool PARG>   // PARG: the h gather fetches a poll's words together (see wave_poll)
__global__ __launch_bounds__(256) void lstm_wave_fwd_kernel(WaveFwd a) {
    extern __shared__ float smem[];
    constexpr int CW = 4 * JW;
    constexpr int GL = RowGroup<NB>::GL, NI = RowGroup<NB>::NI;
    const int H = a.H, G = 4 * H, B = a.B, nwg = a.nwg, S = a.S;
    const int l = blockIdx.y;
    const bool xin = l > 0;        // x-branch computed here (from the lower layer's h) instead of read from xw0
    float* Wl = smem;              // [CW][H]  recurrent columns
    float* Wxl = Wl + CW * H;      // [CW][H]  input columns (layers > 0)
    float* hs = Wxl + CW * H;      // [NB][H]  h_l[s-1]
    float* xs = hs + NB * H;       // [NB][H]  h_{l-1}[s] (after dropout)
    float* pre = xs + NB * H;      // [NB][CW] slice of h @ Wh
    float* prex = pre + NB * CW;   // [NB][CW] slice of x @ Wx
    float* lnst = prex + NB * CW;  // [NB][4]  mean_x, rstd_x, mean_h, rstd_h
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const int j0 = blockIdx.x * JW;
    const int nvalid = (H - j0) < JW ? (H - j0) : JW;
    const size_t BH = (size_t)B * H;

    const float* wh_l = a.wh + (size_t)l * H * G;
    const float* wx_l = xin ? a.wx + (size_t)a.I * G + (size_t)(l - 1) * H * G : nullptr;
    for (int e = tid; e < CW * H; e += 256) {
        const int jj = e % JW, gg = (e / JW) & 3, k = e / CW;
        const bool ok = jj < nvalid;
        Wl[(gg * JW + jj) * H + k] = ok ? wh_l[(size_t)k * G + gg * H + j0 + jj] : 0.f;
        Wxl[(gg * JW + jj) * H + k] = (ok && xin) ? wx_l[(size_t)k * G + gg * H + j0 + jj] : 0.f;
    }
    for (int e = tid; e < NB * H; e += 256) {
        hs[e] = (e < B * H) ? a.h0[(size_t)l * BH + e] : 0.f;
        xs[e] = 0.f;
    }
    float* const xw_l = a.xw + l * a.layer_stride;
    float* const hw_l = a.hw + l * a.layer_stride;
    float* const gates_l = a.gates + l * a.layer_stride;
    float* const c_l = a.c + l * a.layer_stride;
    float* const hseq_l = a.hseq + l * a.layer_stride;
    float* const stats_l = a.stats + l * a.layer_stride;
    const float* gamma_l = a.gamma + (size_t)l * 2 * G;
    const float* beta_l = a.beta + (size_t)l * 2 * G;
    const float* bias_l = a.bias + (size_t)l * G;

    const int cb = tid / JW, cjj = tid % JW, cj = j0 + cjj;
    const bool cell = tid < NB * JW && cb < B && cjj < nvalid;
    float gx[4], gh[4], bsum[4], creg = 0.f;
    if (cell) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            const int col = gg * H + cj;
            gx[gg] = gamma_l[col];
            gh[gg] = gamma_l[G + col];
            bsum[gg] = (beta_l[col] + beta_l[G + col]) + bias_l[col];
        }
        creg = a.c0[(size_t)l * BH + (size_t)cb * H + cj];
    }
    const float inv_g = 1.f / (float)G;
    const int rb = tid / GL, rpart = tid % GL;
    const bool rgrp = rb < B;
    u64* const hx_l = a.hx + (size_t)l * S * BH;                       // own h slots
    const u64* const hx_lo = xin ? a.hx + (size_t)(l - 1) * S * BH : nullptr;
    const int nq = xin ? 4 : 2;                                        // sums words per batch row and workgroup
    u64* const sx_l = a.sx + (size_t)l * S * (size_t)(4 * B * nwg);
    const uint64_t dseed = a.seed + 0x1000003ull * (uint64_t)l;        // dropout between layer l-1 and l
    __syncthreads();

    for (int s = 0; s < S; ++s) {
        const uint32_t tag = (uint32_t)s + 1u;
        // ---- one poll round: h_l[s-1] (tag s) and, for layers > 0, h_{l-1}[s] (tag s+1)
        {
            const int n1 = s > 0 ? B * H : 0, n2 = xin ? B * H : 0;
            const u64* src1 = hx_l + (size_t)(s > 0 ? s - 1 : 0) * BH;
            const u64* src2 = xin ? hx_lo + (size_t)s * BH : nullptr;
            for (int e0 = tid; e0 < n1 + n2; e0 += 256 * 8) {
                u64 w[8];
                unsigned valid = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (e0 + 256 * i < n1 + n2) valid |= 1u << i;
                wave_poll<8, PARG>(src1, n1, src2, n2, e0, valid, tag, w);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int e = e0 + 256 * i;
                    if ((valid >> i) & 1u) {
                        float v = __uint_as_float((uint32_t)w[i]);
                        if (e < n1) hs[e] = v;
                        else {
                            const int ee = e - n1;
                            if (a.drop_threshold)
                                v = (mix_hash(dseed, (uint64_t)s * BH + (uint64_t)ee) > a.drop_threshold)
                                        ? v * a.drop_scale : 0.f;
                            xs[ee] = v;
                        }
                    }
                }
            }
            __syncthreads();
        }
        float xv[4] = {0.f, 0.f, 0.f, 0.f}, mx = 0.f, rx = 0.f;
        if (cell && !xin) {
            const float* xr = a.xw0 + ((size_t)s * B + cb) * G;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) xv[gg] = xr[gg * H + cj];
            const float* st = stats_l + ((size_t)s * B + cb) * 4;
            mx = st[0];
            rx = st[1];
        }
        // ---- slice products: wave g <-> gate g, lanes split k
        {
            float acc[NB][JW], accx[NB][JW];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) acc[b][jj] = accx[b][jj] = 0.f;
#pragma unroll 2
            for (int k = lane; k < H; k += 64) {
                float wv[JW], wxv[JW];
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    wv[jj] = Wl[(g * JW + jj) * H + k];
                    wxv[jj] = Wxl[(g * JW + jj) * H + k];
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float hv = hs[b * H + k], xvv = xs[b * H + k];
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) {
                        acc[b][jj] = fmaf(hv, wv[jj], acc[b][jj]);
                        accx[b][jj] = fmaf(xvv, wxv[jj], accx[b][jj]);
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    const float t = wave_sum_last(acc[b][jj]);
                    const float tx = wave_sum_last(accx[b][jj]);
                    if (lane == 63) {
                        pre[b * CW + g * JW + jj] = t;
                        prex[b * CW + g * JW + jj] = tx;
                    }
                }
        }
        __syncthreads();
        // ---- LayerNorm partials of the owned columns (h-branch; x-branch for layers > 0) -> exchange
        u64* const sdst = sx_l + (size_t)s * (size_t)(4 * B * nwg);
        if (tid < 2 * B) {
            const int b = tid >> 1, which = tid & 1;   // 0: h-branch, 1: x-branch
            if (which == 0 || xin) {
                const float* src = which ? prex : pre;
                float pv[CW], s1 = 0.f, m2 = 0.f;
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    pv[c] = ((c % JW) < nvalid) ? src[b * CW + c] : 0.f;
                    s1 += pv[c];
                }
                const float m = s1 / (4.f * (float)nvalid);
#pragma unroll
                for (int c = 0; c < CW; ++c) m2 += ((c % JW) < nvalid) ? (pv[c] - m) * (pv[c] - m) : 0.f;
                xchg_put(sdst + (size_t)(b * 4 + which * 2) * nwg + blockIdx.x, s1, tag);
                xchg_put(sdst + (size_t)(b * 4 + which * 2 + 1) * nwg + blockIdx.x, m2, tag);
            }
        }
        {
            int idx[4 * NI];
            unsigned valid = 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int w = rpart + GL * i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    idx[q * NI + i] = (rb * 4 + q) * nwg + w;
                    if (rgrp && w < nwg && q < nq) valid |= 1u << (q * NI + i);
                }
            }
            float v[4 * NI];
            xchg_get<4 * NI>(sdst, idx, valid, tag, v);
#pragma unroll
            for (int br = 0; br < 2; ++br) {   // 0: h-branch (words 0,1), 1: x-branch (words 2,3)
                float s1 = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) s1 += ((valid >> (2 * br * NI + i)) & 1u) ? v[2 * br * NI + i] : 0.f;
                const float mean = group_sum_all<GL>(s1) * inv_g;
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int w = rpart + GL * i;
                    const int units = (H - w * JW) < JW ? (H - w * JW) : JW;
                    const float n = 4.f * (float)units;
                    const float d = v[2 * br * NI + i] / n - mean;
                    t += ((valid >> (2 * br * NI + i)) & 1u) ? v[(2 * br + 1) * NI + i] + n * d * d : 0.f;
                }
                t = group_sum_last<GL>(t);
                if (rgrp && rpart == GL - 1 && (br == 0 || xin)) {
                    const float rstd = rsqrtf(t * inv_g + kLnEps);
                    lnst[rb * 4 + (br ? 0 : 2)] = mean;
                    lnst[rb * 4 + (br ? 1 : 3)] = rstd;
                    if (blockIdx.x == 0) {
                        float* st = stats_l + ((size_t)s * B + rb) * 4;
                        st[br ? 0 : 2] = mean;
                        st[br ? 1 : 3] = rstd;
                    }
                }
            }
        }
        __syncthreads();
        // ---- cell
        if (cell) {
            const float mh = lnst[cb * 4 + 2], rh = lnst[cb * 4 + 3];
            if (xin) { mx = lnst[cb * 4]; rx = lnst[cb * 4 + 1]; }
            float av[4];
            const size_t row = (size_t)s * B + cb;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float p = pre[cb * CW + gg * JW + cjj];
                const float xp = xin ? prex[cb * CW + gg * JW + cjj] : xv[gg];
                av[gg] = ((xp - mx) * rx * gx[gg] + (p - mh) * rh * gh[gg]) + bsum[gg];
                hw_l[row * G + gg * H + cj] = p;
                if (xin) xw_l[row * G + gg * H + cj] = xp;
            }
            const float ig = 1.f / (1.f + expf(-av[0]));
            const float fg = 1.f / (1.f + expf(-av[1]));
            const float og = 1.f / (1.f + expf(-av[2]));
            const float ug = tanhf(av[3]);
            creg = fg * creg + ig * ug;
            const float h = og * tanhf(creg);
            xchg_put(hx_l + (size_t)s * BH + (size_t)cb * H + cj, h, tag);
            float* gr = gates_l + row * G;
            gr[cj] = ig; gr[H + cj] = fg; gr[2 * H + cj] = og; gr[3 * H + cj] = ug;
            c_l[row * H + cj] = creg;
            hseq_l[row * H + cj] = h;
        }
    }
}

struct WaveCfg { int nb, jw, nwg; size_t lds; size_t hx_words, sx_words; };

template <int NB, int JW, bool PARG>
inline int launch_wave_fwd_k(const WaveCfg& c, const WaveFwd& a, hipStream_t st) {
    if (a.S < 0)   // residency query (see wave_fwd_resident): 1 = the runtime's occupancy figure covers the grid
        return persist_resident(lstm_wave_fwd_kernel<NB, JW, PARG>, c.nwg * a.L, c.lds) ? 1 : 0;
    return persist_launch(lstm_wave_fwd_kernel<NB, JW, PARG>, dim3(c.nwg, a.L), c.lds, a, st);
}
// Bulk gathers (2*B*H >= 1024 words, i.e. >= 4 per thread) fetch a poll's words together, small ones one after the
// other.  Only the JW >= 4 kernels (H > 170 at L = 3) get the second instantiation: below that the gather is small.
template <int NB, int JW>
inline int launch_wave_fwd_t(const WaveCfg& c, const WaveFwd& a, hipStream_t st) {
    if (JW >= 4 && 2 * a.B * a.H >= 1024) return launch_wave_fwd_k<NB, JW, (JW >= 4)>(c, a, st);
    return launch_wave_fwd_k<NB, JW, false>(c, a, st);
}
inline int launch_wave_fwd(const WaveCfg& c, const WaveFwd& a, hipStream_t st) {
#define HPC_RLL_WAVE_RUN(JW_)                                               \
    if (c.jw == JW_) {                                                      \
        if (c.nb == 1) return launch_wave_fwd_t<1, JW_>(c, a, st);          \
        if (c.nb == 2) return launch_wave_fwd_t<2, JW_>(c, a, st);          \
        return launch_wave_fwd_t<4, JW_>(c, a, st);                         \
    }
    HPC_RLL_WAVE_RUN(1) HPC_RLL_WAVE_RUN(2) HPC_RLL_WAVE_RUN(4) HPC_RLL_WAVE_RUN(6) HPC_RLL_WAVE_RUN(8)
#undef HPC_RLL_WAVE_RUN
    return HPC_RLL_EUNSUPPORTED;
}

// Eligibility.  Workgroups of different layers that share a CU slow each other's polls down (measured: 3 per CU
// 17 us per wavefront step, 1 per CU 7 us) and more than one workgroup per CU would make co-residency depend on the
// occupancy the runtime reports; so the path is taken only if some JW in {1,2,4,6,8} gives every one of the L*nwg
// workgroups its own CU -- then the launch is co-resident by construction.  `cus` = 256 for workspace sizing.
constexpr size_t kWaveMaxWords = (size_t)8 << 20;   // 64 MB of tagged exchange slots
// `sizing` (workspace layout): the answer must not depend on run-time switches, or forward and backward of one call
// pair could carve the workspace differently.
inline bool wave_shape_ok(int S, int B, int H, int L, int cus, WaveCfg* out, bool sizing = false) {
    if (!sizing && (!g_lstm_wave || !g_lstm_persist)) return false;
    if (L < 2 || S < 1 || B < 1 || B > 4 || H < 1 || H > 1024) return false;
    WaveCfg c;
    c.nb = B <= 1 ? 1 : B <= 2 ? 2 : 4;
    c.jw = 0;
    static const int kJw[5] = {1, 2, 4, 6, 8};
    for (int i = 0; i < 5; ++i)
        if ((long)L * ((H + kJw[i] - 1) / kJw[i]) <= cus) { c.jw = kJw[i]; break; }
    if (!c.jw) return false;
    c.nwg = (H + c.jw - 1) / c.jw;
    c.lds = ((size_t)8 * c.jw * H + (size_t)2 * c.nb * H + (size_t)c.nb * 8 * c.jw + 4 * c.nb + 64) * sizeof(float);
    c.hx_words = (size_t)L * S * B * H;
    c.sx_words = (size_t)L * S * 4 * B * H;   // sized for the largest workgroup count (jw = 1)
    if (c.hx_words + c.sx_words > kWaveMaxWords || c.lds > 144 * 1024) return false;
    *out = c;
    return true;
}
inline bool wave_fwd_ok(int S, int B, int H, int L, WaveCfg* out, hipStream_t st) {
    bool ok = wave_shape_ok(S, B, H, L, persist_cu_count(), out) && persist_runtime_ready(st);
    if (ok) {   // the runtime's own occupancy figure for the chosen instantiation must cover all L*nwg workgroups
        WaveFwd q{};
        q.S = -1; q.B = B; q.H = H; q.L = L;
        ok = launch_wave_fwd(*out, q, st) == 1;
    }
    if (getenv("HPC_RLL_LSTM_PROFILE"))
        fprintf(stderr, "[lstm wave] S=%d B=%d H=%d L=%d -> %s (jw=%d, %d workgroups per layer)\n", S, B, H, L,
                ok ? "wavefront" : "per-layer kernels", ok ? out->jw : 0, ok ? out->nwg : 0);
    return ok;
}

}  // namespace
}  // namespace hpc_rll

// ===================================================================================================== backward
// Layer l walks s = S-1 .. 0 while layer l-1 is one step behind.  Besides its own dHW_l[s+1] (for dh_prev through
// Wh^T) it gathers the upper layer's dXW_{l+1}[s] and multiplies it with the rows of Wx_{l+1} that belong to its own
// hidden units: that is the gradient arriving through the layer output, so no d(xin) GEMM is needed between layers.
// Every (layer, step) has its own tagged slots for dHW, dXW and the four LayerNorm-adjoint sums.
namespace hpc_rll {
namespace {

struct WaveBwd {
    const float *dy /* (S,B,H) or null */, *dhn, *dcn /* (L,B,H) or null */;
    const float *gates, *c, *xw, *hw, *stats;   // layer 0 pointers of the saved tensors; layer l at + l*layer_stride
    size_t layer_stride;
    const float *c0, *gamma, *wx, *wh;
    float *dgate, *dxw, *dhw;                   // layer 0 of the per-layer gradient buffers; layer l at + l*grad_stride
    size_t grad_stride;
    float *dh0, *dc0;
    u64 *xhw, *xxw, *xsum;                      // [L][S][B*4H], [L][S][B*4H], [L][S][4*B*nwg] tagged words
    int S, B, I, H, L, nwg;
    uint64_t seed;
    uint32_t drop_threshold;
    float drop_scale;
};

template <int NB, int JW>
__global__ __launch_bounds__(256) void lstm_wave_bwd_kernel(WaveBwd a) {
    extern __shared__ float smem[];
    constexpr int GL = RowGroup<NB>::GL, NI = RowGroup<NB>::NI;
    const int H = a.H, G = 4 * H, B = a.B, nwg = a.nwg, S = a.S, L = a.L;
    const int l = blockIdx.y;
    const bool top = l == L - 1;        // d_out = dy; otherwise from the upper layer's dXW
    float* Wt = smem;                   // [JW][G]  rows of Wh_l of the owned units
    float* Wxt = Wt + JW * G;           // [JW][G]  rows of Wx_{l+1} of the owned units (not for the top layer)
    float* dl = Wxt + JW * G;           // [NB][G]  dHW_l[s+1]
    float* du = dl + NB * G;            // [NB][G]  dXW_{l+1}[s]
    float* rp = du + NB * G;            // [NB][JW][4]
    float* rtot = rp + NB * JW * 4;     // [NB][4]
    float* gp = rtot + NB * 4;          // [4][2][NB*JW] per-wave partials of the two products
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = blockIdx.x * JW;
    const int nvalid = (H - j0) < JW ? (H - j0) : JW;
    const size_t BH = (size_t)B * H, BG = (size_t)B * G;
    const float* wh_l = a.wh + (size_t)l * H * G;
    const float* wx_up = top ? nullptr : a.wx + (size_t)a.I * G + (size_t)l * H * G;   // Wx of layer l+1: (H, 4H)
    for (int e = tid; e < JW * G; e += 256) {
        const int jj = e / G, cc = e - jj * G;
        const bool ok = jj < nvalid;
        Wt[e] = ok ? wh_l[(size_t)(j0 + jj) * G + cc] : 0.f;
        Wxt[e] = (ok && !top) ? wx_up[(size_t)(j0 + jj) * G + cc] : 0.f;
    }
    for (int e = tid; e < NB * G; e += 256) dl[e] = du[e] = 0.f;
    const float* gates_l = a.gates + l * a.layer_stride;
    const float* c_l = a.c + l * a.layer_stride;
    const float* xw_l = a.xw + l * a.layer_stride;
    const float* hw_l = a.hw + l * a.layer_stride;
    const float* stats_l = a.stats + l * a.layer_stride;
    float* dgate_l = a.dgate + l * a.grad_stride;
    float* dxw_l = a.dxw + l * a.grad_stride;
    float* dhw_l = a.dhw + l * a.grad_stride;
    const float* gamma_l = a.gamma + (size_t)l * 2 * G;

    const int cb = tid / JW, cjj = tid % JW, cj = j0 + cjj;
    const bool cell = tid < NB * JW && cb < B && cjj < nvalid;
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, gh[4] = {0.f, 0.f, 0.f, 0.f};
    float dh_carry = 0.f, dc_carry = 0.f;
    if (cell) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            gx[gg] = gamma_l[gg * H + cj];
            gh[gg] = gamma_l[G + gg * H + cj];
        }
        if (a.dhn) dh_carry = a.dhn[(size_t)l * BH + (size_t)cb * H + cj];
        if (a.dcn) dc_carry = a.dcn[(size_t)l * BH + (size_t)cb * H + cj];
    }
    const float inv_g = 1.f / (float)G;
    const int rb = tid / GL, rpart = tid % GL;
    const bool rgrp = rb < B;
    u64* const xhw_l = a.xhw + (size_t)l * S * BG;
    u64* const xxw_l = a.xxw + (size_t)l * S * BG;
    const u64* const xxw_up = top ? nullptr : a.xxw + (size_t)(l + 1) * S * BG;
    u64* const xsum_l = a.xsum + (size_t)l * S * (size_t)(4 * B * nwg);
    const uint64_t dseed = a.seed + 0x1000003ull * (uint64_t)(l + 1);   // dropout applied to this layer's output
    __syncthreads();

    // step index s runs S-1 .. 0, then one extra pass (s = -1) that only turns dHW_l[0] into dh0
    for (int s = S - 1; s >= -1; --s) {
        const uint32_t tag = (uint32_t)(S - 1 - s) + 1u;   // tag of the words published at step s
        // saved tensors of this step: independent of the exchange, so issued before the poll (their HBM latency
        // overlaps the wait instead of following it)
        float sg[4] = {0.f, 0.f, 0.f, 0.f}, sx[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        float sst[4] = {0.f, 0.f, 0.f, 0.f}, c_new = 0.f, c_prev = 0.f, dyv = 0.f;
        if (cell && s >= 0) {
            const size_t row = (size_t)s * B + cb;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                sg[gg] = gates_l[row * G + gg * H + cj];
                sx[gg] = xw_l[row * G + gg * H + cj];
                sh[gg] = hw_l[row * G + gg * H + cj];
                sst[gg] = stats_l[row * 4 + gg];
            }
            c_new = c_l[row * H + cj];
            c_prev = s == 0 ? a.c0[(size_t)l * BH + (size_t)cb * H + cj] : c_l[(row - B) * H + cj];
            if (top && a.dy) dyv = a.dy[row * H + cj];
        }
        // ---- one poll round: dHW_l[s+1] (published at the previous step, tag-1) and dXW_{l+1}[s] (tag, upper layer)
        const int n1 = s < S - 1 ? B * G : 0, n2 = (!top && s >= 0) ? B * G : 0;
        if (n1 + n2 > 0) {
            const u64* src1 = xhw_l + (size_t)(s + 1 < S ? s + 1 : 0) * BG;
            const u64* src2 = n2 ? xxw_up + (size_t)s * BG : nullptr;
            constexpr int CHB = 16;   // words per thread and poll round (40 = one round at the test shape: +-2 %)
            for (int e0 = tid; e0 < n1 + n2; e0 += 256 * CHB) {
                u64 w[CHB];
                unsigned long long valid = 0;
#pragma unroll
                for (int i = 0; i < CHB; ++i)
                    if (e0 + 256 * i < n1 + n2) valid |= 1ull << i;
                wave_poll<CHB, true>(src1, n1, src2, n2, e0, valid, tag, w);
#pragma unroll
                for (int i = 0; i < CHB; ++i) {
                    const int e = e0 + 256 * i;
                    if ((valid >> i) & 1ull) {
                        const float v = __uint_as_float((uint32_t)w[i]);
                        if (e < n1) dl[e] = v;
                        else du[e - n1] = v;
                    }
                }
            }
            __syncthreads();
        }
        // ---- the two products for the owned units: all 256 threads split the 4H columns
        {
            float acc[NB][JW], accu[NB][JW];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) acc[b][jj] = accu[b][jj] = 0.f;
            if (n1 + n2 > 0) {
#pragma unroll 2
                for (int cc = tid; cc < G; cc += 256) {
                    float w1[JW], w2[JW];
#pragma unroll
                    for (int jj = 0; jj < JW; ++jj) {
                        w1[jj] = Wt[jj * G + cc];
                        w2[jj] = Wxt[jj * G + cc];
                    }
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float d1 = dl[b * G + cc], d2 = du[b * G + cc];
#pragma unroll
                        for (int jj = 0; jj < JW; ++jj) {
                            acc[b][jj] = fmaf(d1, w1[jj], acc[b][jj]);
                            accu[b][jj] = fmaf(d2, w2[jj], accu[b][jj]);
                        }
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int jj = 0; jj < JW; ++jj) {
                    const float t1 = wave_sum_last(acc[b][jj]);
                    const float t2 = wave_sum_last(accu[b][jj]);
                    if (lane == 63) {
                        gp[(wv * 2) * NB * JW + b * JW + jj] = t1;
                        gp[(wv * 2 + 1) * NB * JW + b * JW + jj] = t2;
                    }
                }
        }
        __syncthreads();
        float dout = 0.f;
        if (cell) {
            const int o = cb * JW + cjj, st = NB * JW;
            if (n1) dh_carry = (gp[o] + gp[2 * st + o]) + (gp[4 * st + o] + gp[6 * st + o]);
            if (n2) {
                dout = (gp[st + o] + gp[3 * st + o]) + (gp[5 * st + o] + gp[7 * st + o]);
                if (a.drop_threshold)
                    dout = (mix_hash(dseed, ((uint64_t)s * B + cb) * H + cj) > a.drop_threshold) ? dout * a.drop_scale
                                                                                                   : 0.f;
            }
        }
        if (s < 0) break;
        // ---- gate adjoints of the owned units and their LayerNorm-adjoint partial sums
        float da[4] = {0.f, 0.f, 0.f, 0.f}, xh[4] = {0.f, 0.f, 0.f, 0.f}, hh[4] = {0.f, 0.f, 0.f, 0.f};
        float rx = 0.f, rh = 0.f;
        if (cell) {
            if (top) dout = dyv;
            const float ig = sg[0], fg = sg[1], og = sg[2], ug = sg[3];
            const float dh = dout + dh_carry;
            const float tc = tanhf(c_new);
            const float dc = dc_carry + dh * og * (1.f - tc * tc);
            da[0] = dc * ug * ig * (1.f - ig);
            da[1] = dc * c_prev * fg * (1.f - fg);
            da[2] = dh * tc * og * (1.f - og);
            da[3] = dc * ig * (1.f - ug * ug);
            dc_carry = dc * fg;
            rx = sst[1];
            rh = sst[3];
            float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                xh[gg] = (sx[gg] - sst[0]) * sst[1];
                hh[gg] = (sh[gg] - sst[2]) * sst[3];
                const float dyx = da[gg] * gx[gg], dyh = da[gg] * gh[gg];
                r[0] += dyx; r[1] += dyx * xh[gg];
                r[2] += dyh; r[3] += dyh * hh[gg];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) rp[(cb * JW + cjj) * 4 + q] = r[q];
        }
        __syncthreads();
        u64* const sdst = xsum_l + (size_t)s * (size_t)(4 * B * nwg);
        if (tid < 4 * B) {
            const int b = tid >> 2, q = tid & 3;
            float t = 0.f;
            for (int jj = 0; jj < nvalid; ++jj) t += rp[(b * JW + jj) * 4 + q];
            xchg_put(sdst + (size_t)(b * 4 + q) * nwg + blockIdx.x, t, tag);
        }
        {
            int idx[4 * NI];
            unsigned valid = 0;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int w = rpart + GL * i;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    idx[q * NI + i] = (rb * 4 + q) * nwg + w;
                    if (rgrp && w < nwg) valid |= 1u << (q * NI + i);
                }
            }
            float v[4 * NI];
            xchg_get<4 * NI>(sdst, idx, valid, tag, v);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < NI; ++i) t += ((valid >> (q * NI + i)) & 1u) ? v[q * NI + i] : 0.f;
                t = group_sum_last<GL>(t);
                if (rgrp && rpart == GL - 1) rtot[rb * 4 + q] = t * inv_g;
            }
        }
        __syncthreads();
        // ---- dXW, dHW of the owned columns: publish (dHW for this layer, dXW for the layer below), store
        if (cell) {
            const float r0 = rtot[cb * 4], r1 = rtot[cb * 4 + 1], r2 = rtot[cb * 4 + 2], r3 = rtot[cb * 4 + 3];
            const size_t row = (size_t)s * B + cb;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int col = gg * H + cj;
                const float dyx = da[gg] * gx[gg], dyh = da[gg] * gh[gg];
                const float dhw = rh * (dyh - r2 - hh[gg] * r3);
                const float dxw = rx * (dyx - r0 - xh[gg] * r1);
                xchg_put(xhw_l + (size_t)s * BG + (size_t)cb * G + col, dhw, tag);
                if (l > 0) xchg_put(xxw_l + (size_t)s * BG + (size_t)cb * G + col, dxw, tag);
                dhw_l[row * G + col] = dhw;
                dxw_l[row * G + col] = dxw;
                dgate_l[row * G + col] = da[gg];
            }
        }
    }
    if (cell) {
        a.dh0[(size_t)l * BH + (size_t)cb * H + cj] = dh_carry;
        a.dc0[(size_t)l * BH + (size_t)cb * H + cj] = dc_carry;
    }
}

template <int NB, int JW>
inline int launch_wave_bwd_t(const WaveCfg& c, const WaveBwd& a, hipStream_t st) {
    if (a.S < 0)   // residency query (see wave_bwd_resident)
        return persist_resident(lstm_wave_bwd_kernel<NB, JW>, c.nwg * a.L, c.lds) ? 1 : 0;
    return persist_launch(lstm_wave_bwd_kernel<NB, JW>, dim3(c.nwg, a.L), c.lds, a, st);
}
inline int launch_wave_bwd(const WaveCfg& c, const WaveBwd& a, hipStream_t st) {
#define HPC_RLL_WAVE_RUN(JW_)                                               \
    if (c.jw == JW_) {                                                      \
        if (c.nb == 1) return launch_wave_bwd_t<1, JW_>(c, a, st);          \
        if (c.nb == 2) return launch_wave_bwd_t<2, JW_>(c, a, st);          \
        return launch_wave_bwd_t<4, JW_>(c, a, st);                         \
    }
    HPC_RLL_WAVE_RUN(1) HPC_RLL_WAVE_RUN(2) HPC_RLL_WAVE_RUN(4) HPC_RLL_WAVE_RUN(6) HPC_RLL_WAVE_RUN(8)
#undef HPC_RLL_WAVE_RUN
    return HPC_RLL_EUNSUPPORTED;
}

// backward eligibility: the forward rule plus the backward's LDS footprint and exchange storage
inline bool wave_bwd_shape_ok(int S, int B, int H, int L, int cus, WaveCfg* out, bool sizing = false) {
    WaveCfg c;
    if (!wave_shape_ok(S, B, H, L, cus, &c, sizing)) return false;
    // two layers of a wide LSTM: the doubled gather (dHW_l and dXW_{l+1}, 2*B*4H words per step) costs more than
    // halving the number of dependent steps saves (measured L=2, H=512: 2.0 ms per-layer vs 2.2 ms wavefront)
    if (L < 3 && H > 256) return false;
    const size_t G = 4 * (size_t)H;
    c.lds = ((size_t)2 * c.jw * G + (size_t)2 * c.nb * G + (size_t)c.nb * c.jw * 12 + 4 * c.nb + 64) * sizeof(float);
    c.hx_words = (size_t)L * S * B * G;        // dHW slots (and as many dXW slots)
    c.sx_words = (size_t)L * S * 4 * B * H;    // sums slots, sized for jw = 1
    if (2 * c.hx_words + c.sx_words > kWaveMaxWords || c.lds > 144 * 1024) return false;
    *out = c;
    return true;
}

inline bool wave_bwd_ok(int S, int B, int H, int L, WaveCfg* out, hipStream_t st) {
    if (!wave_bwd_shape_ok(S, B, H, L, persist_cu_count(), out) || !persist_runtime_ready(st)) return false;
    WaveBwd q{};
    q.S = -1; q.B = B; q.H = H; q.L = L;
    return launch_wave_bwd(*out, q, st) == 1;
}

}  // namespace
}  // namespace hpc_rll
