// lstm_block.hpp -- large-batch persistent ROW-BLOCK kernels of the LayerNorm-LSTM (included by lstm.hip only).
//
// What one step costs in the reference: cublasSgemm + layernorm kernel + activation kernel
// (src/torch_utils/network/lstm.cu:145-161).  The step kernels of lstm.hip made that one product + one fused cell launch;
// at the C4 shape (S=128, B=4096, H=1024) the 128 cell launches are HBM-bound work (42 us each, 8 % of the forward) that
// cannot overlap the MFMA-bound products through streams (measured).  Here ONE kernel walks all S steps
// and the cell runs in the product's own epilogue:
//   * tile = 256 batch rows x 256 gate columns, 8 waves, one workgroup per CU -- the loop of gemm_f32_nn_dma_kernel
//     (LDS-DMA staged operands, v_mfma_f32_32x32x2_f32, exact fp32).  The columns of the weight copy are
//     GATE-INTERLEAVED (column 4u + g = gate g of hidden unit u, lstm.hip: lstm_perm_shape): a lane of that kernel owns four
//     consecutive output columns, i.e. the four gates of ONE unit for its 32 rows, so everything after the LayerNorm
//     statistics is lane-local: no transposition, no LDS;
//   * a ROW BLOCK (256 batch rows) is covered by nct = 4H/256 workgroups that need each other twice per step:
//       (1) LayerNorm statistics of h@Wh rows: every workgroup publishes (mean, M2) of its 256 columns per row, all
//           combine the nct partials with Chan's formula (exact; no E[x^2] - E[x]^2 cancellation);
//       (2) h_s: the next step's A operand is the row block's (256, H) slice of h_s, written by all nct workgroups.
//     Both are counter barriers over the nct workgroups of the row block only (release fence + agent-scope atomic add;
//     one thread polls, acquire fence, workgroup barrier -- the cooperative-groups grid-sync pattern restricted to a
//     row block).  Row blocks never wait for each other: their epilogues (HBM-bound) drift apart and overlap other
//     row blocks' MFMA phases, which is where the time comes from (a start skew, tune key 27, seeds the drift);
//   * x-branch statistics are per (s, b) row and independent of the recurrence: the x-branch product's epilogue emits
//     per-row partials (gemm_f32_nn_dma_kernel<.., ROWSTATS>), one small pass combines them (lstm_xstats_kernel);
//   * saved for the backward exactly what the step path saves (hw pre-LayerNorm, c, h, row statistics), in the same
//     gate-interleaved layout the step kernels use at these shapes: forward / backward paths can be mixed.
// Residency is requested, not assumed (same protocol as lstm_persist.hpp): the runtime's occupancy answer must cover the
// grid (LDS is sized so that exactly one workgroup fits a CU), launches are chained per device, waits are bounded and
// end in HPC_RLL_ETIMEOUT instead of a hang.  Workgroups of a row block are consecutive in dispatch order.
#pragma once
#include <hip/hip_runtime.h>

#include "gemm_f32.hpp"
#include "hpc_rll_hip.h"
#include "wave.hpp"

namespace hpc_rll {
int g_lstm_block = 1;        // hpc_rll_tune_set key 26: 0 = step kernels (same layout), 1 = persistent row-block forward
int g_lstm_block_skew = 0;   // hpc_rll_tune_set key 27: microseconds between the starts of consecutive row blocks
namespace {

// ---- counter barrier over the workgroups of one row block -----------------------------------------------------------
__device__ __forceinline__ void block_arrive(unsigned* flag) {
    __syncthreads();   // workgroup-scope release of every thread's stores, then the barrier
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void block_wait(unsigned* flag, unsigned target) {
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) persist_poll_failed(spins);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// x-branch row statistics from the per-row partials the x-branch product's epilogue wrote: rowpart[cb][rows][2] = (mean,
// M2) over 128 columns each, cb < ncb -> stats[row*4 + {0,1}] = mean, rstd over all 128*ncb columns (Chan's combination).
__global__ __launch_bounds__(256) void lstm_xstats_kernel(const float* __restrict__ rowpart, int ncb, long rows,
                                                          float* __restrict__ stats) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    float ms = 0.f, m2 = 0.f;
    for (int c = 0; c < ncb; ++c) {
        const vfloat2 p = *reinterpret_cast<const vfloat2*>(rowpart + ((size_t)c * rows + row) * 2);
        ms += p.x;
        m2 += p.y;
    }
    const float mean = ms / (float)ncb;
    float dev = 0.f;
    for (int c = 0; c < ncb; ++c) {
        const float d = rowpart[((size_t)c * rows + row) * 2] - mean;
        dev += d * d;
    }
    const float var = (m2 + 128.f * dev) / (128.f * (float)ncb);
    stats[row * 4] = mean;
    stats[row * 4 + 1] = rsqrtf(var + kLnEps);
}

struct BlockFwd {
    const float* xw;                    // (S, B, 4H) x-branch pre-activations, gate-interleaved columns
    const float* whp;                   // (H, 4H)   Wh with gate-interleaved columns
    const float *bias, *gamma, *beta;   // standard layouts: (4H), (2, 4H), (2, 4H)
    const float *h0, *c0;               // (B, H)
    float *hw, *c, *hseq, *stats;       // (S,B,4H) interleaved, (S,B,H), (S,B,H), (S,B,4): stats[.,0..1] are inputs
    float* part;                        // [row blocks of this launch][2][nct][256][2]
    unsigned* flags;                    // [row blocks of this launch][2]: arrivals (statistics, h); zero at launch
    int S, B, H, rb0, nct;
    int skew_ticks;                     // wall-clock ticks (100 MHz) between the starts of consecutive row blocks
};

constexpr int kRC = 4;   // rows of a lane whose loads are in flight together in the cell epilogue (register budget)
constexpr int kBlkLdsFloats = 2 * 16 * 256 + 2 * 16 * 256 + 2 * 256 * 2 + 256 * 4;
constexpr size_t kBlkLdsBytes = 96 * 1024;   // > half a CU's 160 KB: exactly one workgroup per CU
static_assert(kBlkLdsFloats * sizeof(float) <= kBlkLdsBytes, "row-block kernel LDS");

__global__ __launch_bounds__(512, 2) void lstm_block_fwd_kernel(const BlockFwd a) {
    constexpr int BM = 256, BN = 256, BK = 16, NTH = 512;
    extern __shared__ __attribute__((aligned(16))) float blk_lds[];
    float* const As = blk_lds;                    // [2][BM rows][BK]  (swizzled chunks, lds_pos)
    float* const Bs = As + 2 * BK * BM;           // [2][BK][BN]       (k-major)
    float* const pl = Bs + 2 * BK * BN;           // [2 (wn)][256 rows][2]: (mean, M2) of a wave's 128 columns
    float* const sl = pl + 2 * 256 * 2;           // [256 rows][4]: mean_x, rstd_x, mean_h, rstd_h
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values live in scalar registers: row
    const int wm = wave & 3, wn = wave >> 2, h = lane >> 5, i32 = lane & 31;   // addresses = scalar base + per-lane offset
    const int rbl = (int)blockIdx.x / a.nct, ct = (int)blockIdx.x % a.nct;
    const int H = a.H, G = 4 * H, nct = a.nct;
    const long row0 = (long)(a.rb0 + rbl) * BM;   // first batch row of the row block
    const int n0 = ct * BN;                       // first (interleaved) gate column of the tile
    const int unit = ct * 64 + wn * 32 + i32;     // the hidden unit whose four gates this lane holds
    unsigned* const flag_p = a.flags + 2 * rbl;
    unsigned* const flag_h = flag_p + 1;
    float* const part = a.part + (size_t)rbl * 2 * nct * 256 * 2;

    float gx[4], gh[4], bx[4], bh[4], bb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int col = g * H + unit;
        gx[g] = a.gamma[col];
        gh[g] = a.gamma[G + col];
        bx[g] = a.beta[col];
        bh[g] = a.beta[G + col];
        bb[g] = a.bias[col];
    }
    // per-lane parts of the epilogue's addresses (a lane's rows are 4 h + a wave-uniform row, its columns those of `unit`)
    const unsigned xoff = (unsigned)(4 * h) * (unsigned)G + 4u * (unsigned)unit;   // in a (rows, 4H) tensor
    const unsigned uoff = (unsigned)(4 * h) * (unsigned)H + (unsigned)unit;        // in a (rows, H) tensor
    if (a.skew_ticks > 0 && rbl > 0) {   // seed the phase drift between row blocks (see the header)
        const long long t0 = wall_clock64(), want = (long long)rbl * a.skew_ticks;
        while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(32);
    }

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gl_ptr;
    const int ktiles = H / BK;
    int a_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) a_off[q] = lds_pos<BK>(wm * 64 + i32, 8 * h + 4 * q);
    const int b_off = (8 * h) * BN + wn * 128 + 4 * i32;
    const long b8 = 8L * G, b16 = 16L * G;

    for (int s = 0; s < a.S; ++s) {
        const size_t srow = (size_t)s * a.B + row0;   // this row block's first row in the (S*B, .) tensors
        const float* hprev = s == 0 ? a.h0 : a.hseq + (size_t)(s - 1) * a.B * H;
        const float* cprev = s == 0 ? a.c0 : a.c + (size_t)(s - 1) * a.B * H;
        if (s > 0) block_wait(flag_h, (unsigned)(nct * s));   // h_{s-1} of this row block is complete

        // ---- acc = hprev[row block] @ whp[:, tile]   (the loop of gemm_f32_nn_dma_kernel)
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        DmaStage<BM, BK, NTH> da;
        da.init(hprev + row0 * H, H, 0, 0);
        const float* pb = a.whp + (long)wave * G + n0 + 4 * lane;   // wave w: k-rows w and w + 8
        auto issue = [&](int buf) __attribute__((always_inline)) {
            da.issue(As + buf * BK * BM);
            float* bt = Bs + buf * BK * BN + wave * BN;
            __builtin_amdgcn_global_load_lds((gl_ptr)pb, (lds_ptr)bt, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gl_ptr)(pb + b8), (lds_ptr)(bt + 8 * BN), 16, 0, 0);
            pb += b16;
        };
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) issue(buf ^ 1);
            const float* __restrict__ as = As + buf * BK * BM;
            const float* __restrict__ bs = Bs + buf * BK * BN + b_off;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                gf4 av[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const gf4*>(as + a_off[q] + i * 32 * BK);
                gf4 b = *reinterpret_cast<const gf4*>(bs + 4 * q * BN);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    gf4 bnx = b;
                    if (t < 3) bnx = *reinterpret_cast<const gf4*>(bs + (4 * q + t + 1) * BN);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], b[j], acc[i][j], 0, 0, 0);
                    b = bnx;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }

        // ---- (1) LayerNorm statistics of the h-branch rows.  A wave holds 64 rows x 128 columns: per row the 32 lanes of a
        // half wave sum their four columns (two passes: mean, then M2 around it); lane i32 = 16 i + r keeps row (i, r).
        {
            float my_m = 0.f, my_d = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float sm = half_sum_all((acc[i][0][r] + acc[i][1][r]) + (acc[i][2][r] + acc[i][3][r]));
                    const float m = sm * (1.f / 128.f);
                    float d = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) d += (acc[i][j][r] - m) * (acc[i][j][r] - m);
                    d = half_sum_all(d);
                    if (i32 == 16 * i + r) { my_m = m; my_d = d; }
                }
            const int rr = i32 & 15;
            const int row = wm * 64 + (i32 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            *reinterpret_cast<vfloat2*>(pl + (wn * 256 + row) * 2) = vfloat2{my_m, my_d};
        }
        __syncthreads();
        float* const pslot = part + (size_t)(s & 1) * nct * 256 * 2;
        if (tid < 256) {   // the two waves of a row (128 columns each) -> the workgroup's (mean, M2) over 256 columns
            const vfloat2 p0 = *reinterpret_cast<const vfloat2*>(pl + tid * 2);
            const vfloat2 p1 = *reinterpret_cast<const vfloat2*>(pl + (256 + tid) * 2);
            const float dm = p0.x - p1.x;
            *reinterpret_cast<vfloat2*>(pslot + ((size_t)ct * 256 + tid) * 2) =
                vfloat2{0.5f * (p0.x + p1.x), p0.y + p1.y + 64.f * dm * dm};
        }
        block_arrive(flag_p);
        block_wait(flag_p, (unsigned)(nct * (s + 1)));
        if (tid < 256) {
            float ms = 0.f, m2 = 0.f;
            float pm[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int cc = c < nct ? c : 0;   // nct <= 16 (H <= 1024); idle slots re-read tile 0, unused
                const vfloat2 p = *reinterpret_cast<const vfloat2*>(pslot + ((size_t)cc * 256 + tid) * 2);
                pm[c] = p.x;
                if (c < nct) { ms += p.x; m2 += p.y; }
            }
            const float mean = ms / (float)nct;
            float dev = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c < nct) dev += (pm[c] - mean) * (pm[c] - mean);
            const float rstd = rsqrtf((m2 + 256.f * dev) / (float)G + kLnEps);
            float* st = a.stats + (srow + tid) * 4;
            const vfloat2 xs = *reinterpret_cast<const vfloat2*>(st);
            *reinterpret_cast<vfloat4*>(sl + tid * 4) = vfloat4{xs.x, xs.y, mean, rstd};
            if (ct == 0) *reinterpret_cast<vfloat2*>(st + 2) = vfloat2{mean, rstd};
        }
        __syncthreads();

        // ---- (2) gates, state update, saved tensors: lane-local (four gates of `unit` for 32 rows)
        {
            const float* const xw_s = a.xw + srow * G;                        // scalar bases of this step and row block
            const float* const cp_s = cprev + (size_t)row0 * H;
            float* const hw_s = a.hw + srow * G;
            float* const c_s = a.c + srow * H;
            float* const h_s = a.hseq + srow * H;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r8 = 0; r8 < 16; r8 += kRC) {
                    vfloat4 xv[kRC];
                    float cp[kRC];
#pragma unroll
                    for (int q = 0; q < kRC; ++q) {
                        const int r = r8 + q;
                        const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);   // wave-uniform part of the row
                        xv[q] = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(xw_s + (size_t)R * G + xoff));
                        cp[q] = (cp_s + (size_t)R * H)[uoff];
                    }
#pragma unroll
                    for (int q = 0; q < kRC; ++q) {
                        const int r = r8 + q;
                        const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
                        const vfloat4 st = *reinterpret_cast<const vfloat4*>(sl + (R + 4 * h) * 4);
                        float pre[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            pre[j] = gate_pre(xv[q][j], st.x, st.y, gx[j], bx[j], acc[i][j][r], st.z, st.w, gh[j], bh[j], bb[j]);
                        const float ig = gate_sigmoid(pre[0]), fg = gate_sigmoid(pre[1]), og = gate_sigmoid(pre[2]);
                        const float ug = tanhf(pre[3]);
                        const float cn = fg * cp[q] + ig * ug;
                        (h_s + (size_t)R * H)[uoff] = og * tanhf(cn);
                        (c_s + (size_t)R * H)[uoff] = cn;
                        __builtin_nontemporal_store(vfloat4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]},
                                                    reinterpret_cast<vfloat4*>(hw_s + (size_t)R * G + xoff));
                    }
                }
        }
        block_arrive(flag_h);
    }
}

// Row blocks one launch can hold: every workgroup needs its own CU (co-residency).
inline int block_rows_per_launch(int H) {
    const int nct = 4 * H / 256;
    const int cus = persist_cu_count();
    return nct > 0 ? cus / nct : 0;
}
inline size_t block_part_floats(int B, int H) { return (size_t)(B / 256) * 2 * (4 * H / 256) * 256 * 2; }
inline size_t block_flag_words(int B) { return (size_t)(B / 256) * 2; }

inline bool block_fwd_ok(int B, int H, hipStream_t st) {
    if (!g_lstm_block || !g_lstm_persist || !lstm_perm_shape(B, H) || !persist_runtime_ready(st)) return false;
    if (block_rows_per_launch(H) < 1) return false;
    return persist_resident_t(lstm_block_fwd_kernel, 512, (4 * H / 256) * block_rows_per_launch(H), kBlkLdsBytes);
}

// All row blocks of one layer, in as many launches as the CU count asks for (C4: 16 row blocks x 16 column tiles = one).
inline int launch_block_fwd(BlockFwd a, float* part, unsigned* flags, hipStream_t st) {
    const int nrb = a.B / 256, per = block_rows_per_launch(a.H);
    if (hipMemsetAsync(flags, 0, block_flag_words(a.B) * sizeof(unsigned), st) != hipSuccess) return last_error();
    const hipError_t e = hipFuncSetAttribute((const void*)lstm_block_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)kBlkLdsBytes);
    if (e != hipSuccess) return (int)e;
    a.nct = 4 * a.H / 256;
    a.skew_ticks = g_lstm_block_skew * 100;
    for (int rb = 0; rb < nrb; rb += per) {
        const int n = nrb - rb < per ? nrb - rb : per;
        a.rb0 = rb;
        a.part = part + (size_t)rb * 2 * a.nct * 256 * 2;
        a.flags = flags + 2 * rb;
        persist_chain_before(st);
        hipLaunchKernelGGL(lstm_block_fwd_kernel, dim3(n * a.nct), dim3(512), kBlkLdsBytes, st, a);
        persist_chain_after(st);
    }
    return last_error();
}

}  // namespace
}  // namespace hpc_rll
