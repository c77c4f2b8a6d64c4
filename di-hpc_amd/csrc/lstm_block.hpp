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
int g_lstm_block = 9;        // hpc_rll_tune_set key 26: 0 = step kernels (same layout); bit 0 = persistent row-block forward,
                             // bit 3 = persistent row-block backward, bit 7 = the forward's exchanges without cache-wide fences
                             // (the backward's never have them)
int g_lstm_block_skew = 10;  // hpc_rll_tune_set key 27: microseconds between the starts of consecutive row blocks (C4: 69.9 -> 65.9 ms)
namespace {

// ---- counter barrier over the workgroups of one row block -----------------------------------------------------------
// nofence (round 4, tune key 26 bit 7; what lstm_mid.hpp measured: an agent-scope release writes back EVERY dirty line of the
// XCD's L2 -- here the hw / c / h streams of 32 workgroups -- and an acquire invalidates that L2, Wh included): the exchanged data
// is stored write-through (sc1) and either lives at addresses written once per launch (h_s, dHW_s: read with ordinary loads,
// nothing stale can be cached) or is read with agent-scope loads (the LayerNorm partials, whose slots are reused); a workgroup's
// waves wait for their own stores, barrier, one counter increment -- no cache-wide operation on either side.
__device__ __forceinline__ void block_arrive(unsigned* flag, bool nofence = false) {
    if (nofence) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): this wave's write-through stores are acknowledged
    __syncthreads();   // workgroup-scope release of every thread's stores, then the barrier
    if (threadIdx.x == 0) {
        if (!nofence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void block_wait(unsigned* flag, unsigned target, bool nofence = false) {
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) persist_poll_failed(spins);
        if (!nofence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}
__device__ __forceinline__ void blk_put2(float* p, float x, float y, bool nofence) {
    if (nofence) __hip_atomic_store(reinterpret_cast<u64*>(p), ((u64)__float_as_uint(y) << 32) | (u64)__float_as_uint(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *reinterpret_cast<vfloat2*>(p) = vfloat2{x, y};
}
__device__ __forceinline__ vfloat2 blk_get2(const float* p, bool nofence) {
    if (!nofence) return *reinterpret_cast<const vfloat2*>(p);
    const u64 w = __hip_atomic_load(reinterpret_cast<const u64*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return vfloat2{__uint_as_float((uint32_t)w), __uint_as_float((uint32_t)(w >> 32))};
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
    float* part;                        // [row blocks of this launch][2][2 nct][rows of a block][2]
    unsigned* flags;                    // [row blocks of this launch][2]: arrivals (statistics, h); zero at launch
    int S, B, H, rb0, nct;
    int skew_ticks;                     // wall-clock ticks (100 MHz) between the starts of consecutive row blocks
    u64* prof;                          // optional (HPC_RLL_LSTM_PROFILE=1): 8 phase accumulators of workgroup gridDim.x / 2
    int nofence;                        // exchange without cache-wide fences (see block_arrive)
};
#define HPC_RLL_BLK_TICK(i)                                   \
    if (prof_on) {                                            \
        const u64 now_ = wall_clock64();                      \
        a.prof[i] += now_ - tprev_;                           \
        tprev_ = now_;                                        \
    }

constexpr int kRC = 4;   // rows of a lane whose loads are in flight together in the cell epilogue (register budget)

// Gate activations of the row-block epilogue.  FAST: hardware exp2 / reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of
// libm's expf / tanhf / IEEE division -- ~55 instead of ~170 vector instructions per hidden unit.  In the step path the cell
// is a kernel of its own and HBM-bound, the arithmetic is free there (0.3 %); in the product's epilogue it
// is on the critical path of the CU (measured 25 us of the 60 us epilogue at C4).  Absolute error of a gate <= 2e-7; the
// backward recomputes the gates with the precise forms from the saved pre-activations (difference <= 2e-7, far inside the
// 2e-5 / 2e-4 bounds of the parity tests).
template <bool FAST> __device__ __forceinline__ float blk_sigmoid(float a) {
    if (FAST) return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504088896341f * a));
    return gate_sigmoid(a);
}
template <bool FAST> __device__ __forceinline__ float blk_tanh(float a) {
    if (FAST) return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.88539008177792681f * a));
    return tanhf(a);
}

// MW = waves along the rows: 4 -> tile 256 x 256, 8 waves, ONE workgroup per CU; 2 -> tile 128 x 256, 4 waves, TWO
// workgroups per CU (two independent row blocks: while one sits in its exchange / epilogue the other's product has the
// matrix pipe).  A row block = 64 * MW batch rows.
template <int MW> struct BlkCfg {
    static constexpr int BM = 64 * MW, NW = 2 * MW, NTH = 64 * NW;
    static constexpr int lds_floats = 2 * 16 * BM + 2 * 16 * 256 + BM * 4;
    // 1 per CU: more than half of the 160 KB; 2 per CU: 64 KB (a third would not fit; the 256 registers per wave cap it too)
    static constexpr size_t lds_bytes = MW == 4 ? 96 * 1024 : 64 * 1024;
    static constexpr int wgs_per_cu = MW == 4 ? 1 : 2;
    static_assert(lds_floats * sizeof(float) <= lds_bytes, "row-block kernel LDS");
};

template <int MW, bool FAST>
__global__ __launch_bounds__(128 * MW, 2) void lstm_block_fwd_kernel(const BlockFwd a) {
    typedef BlkCfg<MW> C;
    constexpr int BM = C::BM, BN = 256, BK = 16, NTH = C::NTH, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) float blk_lds[];
    float* const As = blk_lds;                    // [2][BM rows][BK]  (swizzled chunks, lds_pos)
    float* const Bs = As + 2 * BK * BM;           // [2][BK][BN]       (k-major)
    float* const sl = Bs + 2 * BK * BN;           // [BM rows][4]: mean_x, rstd_x, mean_h, rstd_h
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values live in scalar registers: row
    const int wm = wave % MW, wn = wave / MW, h = lane >> 5, i32 = lane & 31;   // addresses = scalar base + per-lane offset
    const float* const slp = sl + (wm * 64 + 4 * h) * 4;   // the lane's row records: one address + compile-time offsets
    const int rbl = (int)blockIdx.x / a.nct, ct = (int)blockIdx.x % a.nct;
    const int H = a.H, G = 4 * H, nct = a.nct;
    const long row0 = (long)(a.rb0 + rbl) * BM;   // first batch row of the row block
    const int n0 = ct * BN;                       // first (interleaved) gate column of the tile
    const int unit = ct * 64 + wn * 32 + i32;     // the hidden unit whose four gates this lane holds
    unsigned* const flag_p = a.flags + 2 * rbl;
    unsigned* const flag_h = flag_p + 1;
    float* const part = a.part + (size_t)rbl * 2 * 2 * nct * BM * 2;
    const bool nf = a.nofence != 0;

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
    const long bstep = (long)NW * G, b16 = 16L * G;

    const bool prof_on = a.prof && blockIdx.x == gridDim.x / 2 && tid == 0;
    u64 tprev_ = prof_on ? wall_clock64() : 0;
    f32x16 acc[2][4];
    for (int s = 0; s < a.S; ++s) {
        const size_t srow = (size_t)s * a.B + row0;   // this row block's first row in the (S*B, .) tensors
        const float* hprev = s == 0 ? a.h0 : a.hseq + (size_t)(s - 1) * a.B * H;
        const float* cprev = s == 0 ? a.c0 : a.c + (size_t)(s - 1) * a.B * H;
        // the previous step's pre-LayerNorm product goes out now (it is only read by the backward): the stores drain under
        // the wait for h and the first tiles of this step's product
        if (s > 0) {
            float* const hw_p = a.hw + (srow - a.B) * G;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
                    __builtin_nontemporal_store(vfloat4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]},
                                                reinterpret_cast<vfloat4*>(hw_p + (size_t)R * G + xoff));
                }
            block_wait(flag_h, (unsigned)(nct * s), nf);   // h_{s-1} of this row block is complete
        }
        HPC_RLL_BLK_TICK(0)   // hw stores issued, wait for h

        // ---- acc = hprev[row block] @ whp[:, tile]   (the loop of gemm_f32_nn_dma_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        DmaStage<BM, BK, NTH> da;
        da.init(hprev + row0 * H, H, 0, 0);
        const float* pb = a.whp + (long)wave * G + n0 + 4 * lane;   // wave w: k-rows w, w + NW, ...
        auto issue = [&](int buf) __attribute__((always_inline)) {
            da.issue(As + buf * BK * BM);
            float* bt = Bs + buf * BK * BN + wave * BN;
#pragma unroll
            for (int j = 0; j < 16 / NW; ++j)
                __builtin_amdgcn_global_load_lds((gl_ptr)(pb + j * bstep), (lds_ptr)(bt + j * NW * BN), 16, 0, 0);
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
        HPC_RLL_BLK_TICK(1)   // product

        // ---- the cell's own inputs (x-branch pre-activations, c_{s-1}: both independent of the exchange below) are
        // requested kRC rows at a time, one chunk ahead of the arithmetic; the first chunk already before the exchange
        const float* const xw_s = a.xw + srow * G;   // scalar bases of this step and row block
        const float* const cp_s = cprev + (size_t)row0 * H;
        vfloat4 xv[2][kRC];
        float cp[2][kRC];
        auto load_chunk = [&](int ci, vfloat4 (&x4)[kRC], float (&c1)[kRC]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < kRC; ++q) {
                const int r = (ci * kRC + q) & 15, i = (ci * kRC) >> 4;
                const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);   // wave-uniform part of the row
                x4[q] = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(xw_s + (size_t)R * G + xoff));
                c1[q] = (cp_s + (size_t)R * H)[uoff];
            }
        };
        load_chunk(0, xv[0], cp[0]);

        // ---- (1) LayerNorm statistics of the h-branch rows.  A wave holds 64 rows x 128 columns: per row the 32 lanes of a
        // half wave sum their four columns (two passes: mean, then M2 around it); lane i32 = 16 i + r keeps row (i, r).
        float* const pslot = part + (size_t)(s & 1) * 2 * nct * BM * 2;
        vfloat2 xs = {0.f, 0.f};   // x-branch statistics of row `tid` (written before this launch): requested now, used after the exchange
        if (tid < BM) xs = *reinterpret_cast<const vfloat2*>(a.stats + (srow + tid) * 4);
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
            // every wave publishes its own (mean, M2) over 128 columns: 2 nct partials per row
            blk_put2(pslot + ((size_t)(2 * ct + wn) * BM + row) * 2, my_m, my_d, nf);
        }
        block_arrive(flag_p, nf);
        HPC_RLL_BLK_TICK(2)   // row partials + publish
        block_wait(flag_p, (unsigned)(nct * (s + 1)), nf);
        HPC_RLL_BLK_TICK(3)   // wait for the row block's partials
        if (tid < BM) {
            // Chan's update, one partial (128 columns) at a time, eight loads in flight: mean and M2 of the whole row
            float mean = 0.f, m2 = 0.f, cnt = 0.f;
            const float* pp = pslot + (size_t)tid * 2;
            for (int c0 = 0; c0 < 2 * nct; c0 += 8) {
                vfloat2 p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    p[k] = blk_get2(pp + (size_t)(c0 + k < 2 * nct ? c0 + k : 0) * BM * 2, nf);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < 2 * nct) {
                        const float tot = cnt + 128.f, d = p[k].x - mean;
                        mean += d * (128.f / tot);
                        m2 += p[k].y + d * d * (cnt * 128.f / tot);
                        cnt = tot;
                    }
            }
            const float rstd = rsqrtf(m2 / (float)G + kLnEps);
            *reinterpret_cast<vfloat4*>(sl + tid * 4) = vfloat4{xs.x, xs.y, mean, rstd};
            if (ct == 0) *reinterpret_cast<vfloat2*>(a.stats + (srow + tid) * 4 + 2) = vfloat2{mean, rstd};
        }
        __syncthreads();
        HPC_RLL_BLK_TICK(4)   // combine

        // ---- (2) gates and state update: lane-local (four gates of `unit` for 32 rows)
        {
            float* const c_s = a.c + srow * H;
            float* const h_s = a.hseq + srow * H;
            constexpr int NCH = 32 / kRC;
#pragma unroll
            for (int ci = 0; ci < NCH; ++ci) {
                if (ci + 1 < NCH) load_chunk(ci + 1, xv[(ci + 1) & 1], cp[(ci + 1) & 1]);
#pragma unroll
                for (int q = 0; q < kRC; ++q) {
                    const int r = (ci * kRC + q) & 15, i = (ci * kRC) >> 4;
                    const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
                    const vfloat4 st = *reinterpret_cast<const vfloat4*>(slp + (i * 32 + (r & 3) + 8 * (r >> 2)) * 4);
                    float pre[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        pre[j] = gate_pre(xv[ci & 1][q][j], st.x, st.y, gx[j], bx[j], acc[i][j][r], st.z, st.w, gh[j], bh[j], bb[j]);
                    const float ig = blk_sigmoid<FAST>(pre[0]), fg = blk_sigmoid<FAST>(pre[1]), og = blk_sigmoid<FAST>(pre[2]);
                    const float ug = blk_tanh<FAST>(pre[3]);
                    const float cn = fg * cp[ci & 1][q] + ig * ug;
                    const float hv = og * blk_tanh<FAST>(cn);
                    if (nf) __hip_atomic_store((h_s + (size_t)R * H) + uoff, hv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else (h_s + (size_t)R * H)[uoff] = hv;
                    (c_s + (size_t)R * H)[uoff] = cn;
                }
            }
        }
        HPC_RLL_BLK_TICK(5)   // gates
        block_arrive(flag_h, nf);
        HPC_RLL_BLK_TICK(6)   // h / c stores drained, release fence, arrival
    }
    {   // the last step's pre-LayerNorm product
        float* const hw_p = a.hw + ((size_t)(a.S - 1) * a.B + row0) * G;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
                __builtin_nontemporal_store(vfloat4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]},
                                            reinterpret_cast<vfloat4*>(hw_p + (size_t)R * G + xoff));
            }
    }
}

// Row blocks one launch can hold: every workgroup needs its slot on a CU (co-residency).
template <int MW> inline int block_rows_per_launch(int H) {
    const int nct = 4 * H / 256;
    const int cus = persist_cu_count();
    return nct > 0 ? cus * BlkCfg<MW>::wgs_per_cu / nct : 0;
}
// (sized for the 128-row blocks: the larger of the two tilings)
inline size_t block_part_floats(int B, int H) {   // the larger of the forward's and the backward's exchange buffers
    const size_t f = (size_t)(B / 128) * 2 * 2 * (4 * H / 256) * 128 * 2, b = (size_t)(B / 128) * 2 * 4 * ((H + 127) / 128) * 128 * 4;
    return f > b ? f : b;
}
inline size_t block_flag_words(int B) { return (size_t)(B / 128) * 2; }

// 256-row blocks, one workgroup per CU, hardware exp2 / reciprocal gates.  (Measured in round 4 and removed in round 5: 128-row
// blocks with two workgroups per CU -- 67.8-70.0 against 65.1-66.4 ms at C4 -- and libm gate functions in the epilogue, +0.8 ms.)
constexpr int block_mw() { return 4; }
constexpr bool block_fast() { return true; }

// The row blocks of a layer run in launches of `per` (co-residency); a last launch that fills less than 70 % of the CUs
// (e.g. B = 4352: 16 + 1 row blocks of 256) would cost a whole recurrence for a few rows: such shapes keep the step kernels.
inline bool block_launches_fill(int nrb, int per, int wgs_per_rb) {
    const int cus = persist_cu_count();
    if (per < 1 || cus < 1) return false;
    const int tail = nrb % per;
    return tail == 0 || (long)tail * wgs_per_rb * 10 >= (long)cus * 7;
}
template <int MW, bool FAST> inline bool block_resident(int H) {
    const int per = block_rows_per_launch<MW>(H);
    return per >= 1 && persist_resident_t(lstm_block_fwd_kernel<MW, FAST>, BlkCfg<MW>::NTH, (4 * H / 256) * per, BlkCfg<MW>::lds_bytes);
}
inline bool block_fwd_ok(int B, int H, hipStream_t st) {
    if (!(g_lstm_block & 1) || !g_lstm_persist || !lstm_perm_shape(B, H) || !persist_runtime_ready(st)) return false;
    const int mw = block_mw(), nrb = B / (64 * mw), per = mw == 4 ? block_rows_per_launch<4>(H) : block_rows_per_launch<2>(H);
    if (!block_launches_fill(nrb, per, mw == 4 ? 4 * H / 256 : 2 * H / 256)) return false;   // (128-row blocks share a CU in pairs)
    return block_resident<4, true>(H);
}

// All row blocks of one layer, in as many launches as the CU count asks for (C4: 16 row blocks x 16 column tiles = one).
template <int MW, bool FAST>
inline int launch_block_fwd_t(BlockFwd a, float* part, unsigned* flags, hipStream_t st) {
    typedef BlkCfg<MW> C;
    const int nrb = a.B / C::BM, per = block_rows_per_launch<MW>(a.H);
    if (hipMemsetAsync(flags, 0, block_flag_words(a.B) * sizeof(unsigned), st) != hipSuccess) return last_error();
    const hipError_t e = hipFuncSetAttribute((const void*)lstm_block_fwd_kernel<MW, FAST>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
    if (e != hipSuccess) return (int)e;
    a.nct = 4 * a.H / 256;
    a.skew_ticks = g_lstm_block_skew * 100;
    a.prof = persist_prof();
    a.nofence = (g_lstm_block & 128) ? 1 : 0;
    for (int rb = 0; rb < nrb; rb += per) {
        const int n = nrb - rb < per ? nrb - rb : per;
        a.rb0 = rb;
        a.part = part + (size_t)rb * 2 * 2 * a.nct * C::BM * 2;
        a.flags = flags + 2 * rb;
        persist_chain_before(st);
        hipLaunchKernelGGL((lstm_block_fwd_kernel<MW, FAST>), dim3(n * a.nct), dim3(C::NTH), C::lds_bytes, st, a);
        persist_chain_after(st);
    }
    persist_prof_report("row-block fwd: wait_h product partials wait_p combine gates arrive_h", 0, a.S, st);
    return last_error();
}
inline int launch_block_fwd(const BlockFwd& a, float* part, unsigned* flags, hipStream_t st) {
    return launch_block_fwd_t<4, true>(a, part, flags, st);
}

// ---- epilogue memory operations as BUFFER instructions: address = descriptor (4 scalar registers, the array's base for this
// step and row block) + a per-lane byte offset that never changes (the lane's unit / row-half) + a scalar byte offset (the
// wave-uniform row).  With flat global addressing the compiler hoisted the 64-bit address of every (array, row) pair of the
// unrolled epilogue out of the step loop -- 11 arrays x 32 rows x 2 registers -- and the backward kernel spilled 800
// registers; in this form no vector register holds an address at all.  AUX 2 = streaming (nontemporal) hint.
typedef unsigned int blk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t blk_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
template <int AUX> __device__ __forceinline__ vfloat4 blk_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(vfloat4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX));
}
template <int AUX> __device__ __forceinline__ float blk_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, AUX));
}
template <int AUX> __device__ __forceinline__ void blk_st4(vfloat4 v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(blk_u32x4, v), r, (int)voff, (int)soff, AUX);
}
template <int AUX> __device__ __forceinline__ void blk_st1(float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, AUX);
}

// ================================================================================================== backward
// The backward recurrence of a layer in ONE kernel (tune key 26 bit 3).  Per step (time runs backwards):
//     dh_s = dy_s + dHW_{s+1} @ Wh^T                 (M = B, N = H, K = 4H: the product)
//     cell adjoint of (dh_s, dc) -> da (4 gates), both LayerNorm adjoints -> dXW_s, dHW_s, dc
// which the step path runs as one product launch (256 us at C4) + one HBM-bound cell launch (103 us) that cannot overlap.
// Here: tile = 128 batch rows x 128 hidden units with the FULL K (no split-K partials to exchange), both operands
// k-contiguous (dHW rows; the rows of the gate-interleaved Wh copy as they lie) = the NT LDS-DMA loop of gemm_f32_kernel
// (raw rows, k convention of DmaStage), 8 waves of 64 x 32, one workgroup per CU; a lane ends up with dh of ONE hidden unit for
// 32 rows, so the cell adjoint is lane-local: its inputs are that unit's four gates of xw / hw (one float4 each).  A row
// block (128 rows) is covered by nnt = H / 128 workgroups, which meet twice per step:
//   (1) the four LayerNorm-adjoint row sums (sum dy_g, sum dy_g xhat, for both branches) over all 4H columns: every wave
//       publishes its partial over its 32 units, all combine;
//   (2) dHW_s complete: it is the next step's A operand (all 4H columns of the row block).
// The gate adjoints of a lane's 32 (row, unit) pairs cross exchange (1) through a scratch buffer (written before, read back
// after, together with xw / hw: an L2 round trip -- held in registers they cost 128 of the 256 and the kernel spilled
// 900); dc travels through the (B, H) buffer the step path uses (read-modify-write by the same lane every step); the bias / gamma / beta column sums are accumulated in registers over ALL steps (a lane owns its unit's
// four gate columns for the whole sequence) and reduced once at the end -> colacc[row block][3][4H] for
// lstm_colfinal_kernel.  Gates are recomputed with the epilogue's own forms (blk_sigmoid / blk_tanh).
struct BlockBwd {
    const float *dy, *dhn, *dcn;        // (S,B,H) / (B,H) / (B,H); each may be null (= zero)
    const float *xw, *hw;               // (S,B,4H) gate-interleaved, as the forward saved them
    const float *c, *c0;                // (S,B,H), (B,H)
    const float* stats;                 // (S,B,4)
    const float* pp;                    // [5][4H] interleaved parameters: gamma_x, gamma_h, beta_x, beta_h, bias
    const float* whp;                   // (H, 4H) Wh with gate-interleaved columns
    float *dxw, *dhw;                   // (S,B,4H) gate-interleaved
    float* dgate;                       // (S,B,4H) scratch: the gate adjoints between the two passes of a step (L2 round trip)
    float *dc, *dh0, *dc0;              // (B,H) each; dc: scratch carried from step to step
    float* colacc;                      // [row blocks of this launch][3][4H] interleaved column sums
    float* part;                        // [row blocks of this launch][2][4 nnt][128][4]
    unsigned* flags;                    // [row blocks of this launch][2]
    int S, B, H, rb0, nnt;
    int skew_ticks;
    u64* prof;
    int xcd_map;                        // the 8-row-block groups of the launch are laid out XCD-major (see the kernel)
};

// Round 5 on this kernel (profiles/r05_lstm_block_bwd_ab.json, _phases.txt, _abl.txt; C4, one process, interleaved):
//   * exchanges WITHOUT cache-wide fences (what the forward's bit 7 does; shipped): the row-sum partials and dHW_s are stored
//     write-through (sc1); dHW_s lives at addresses written once per launch (ordinary loads / LDS-DMA cannot find a stale copy),
//     the partials' slots are reused every other step and are read with sc1 loads: wait + arrive phases 17 -> 11 us per step,
//     backward 135.1 -> 133.9 ms;
//   * THREE operand buffers with a counted vmcnt and a bare s_barrier (the DMA two k-tiles ahead): product 265 -> 265 us, total
//     +0.5 ms -- the stream's latency is already covered; what the operand stream costs is its VOLUME: with the A / the B / both
//     requests removed after the first tile (wrong results) the product phase reads 250 / 246 / 237 us against 259, the whole
//     backward 133.5 / 131.7 / 130.5 ms -- 237 us is the bare matrix loop of this tile shape (0.94 of the pipe), 22 us per step
//     are the 4 MB of operands a 128 x 128 tile pulls through L2 per step (twice the forward tile's bytes per flop);
//   * two-row epilogue chunks in four register sets (three chunks of loads in flight instead of one): 157 spill instructions
//     inside pass A, pass A 33 -> 96 us; start skew by XCD instead of by row block (the four row blocks of an XCD in step, Wh
//     tiles shared through its L2): 133.6 against 133.2 ms.  Neither shipped;
//   * 128 x 64 tiles with TWO 4-wave workgroups per CU (one's epilogue under the other's product; commit 7e6db35): 136.2 against
//     134.5 ms -- a lone wave per SIMD does not drive the matrix pipe at twice its shared rate, the operand bytes per flop grow
//     1.5 x, the epilogue passes slow down beside the partner's product (profiles/r05_lstm_bwd_bn_probe.txt).
// Round 6 (profiles/r06_lstm_block_bwd.txt; the code is commit "experiment: row-block LSTM backward variants"): the product
// starting on the workgroup's own k-range of dHW before the row block's flag (+10 ms: the eight workgroups of a row block stop
// sharing A tiles in L2), pass A with every other chunk's xw / hw by LDS-DMA into per-wave LDS buffers or with register slots
// reloaded the moment their pair is consumed (170-260 spills: 256 registers per wave leave no room for a second chunk in flight),
// pass A's first chunk under the peeled last k-tile (pass A -3.8 us, product +12 us), the row-sum combine by all threads (neutral).
struct BlkBwdCfg {
    static constexpr int NBUF = 2, BK = 32, BM = 128, BN = 128, NTH = 512;
    static constexpr int tile_floats = NBUF * BK * (BM + BN);   // operand tiles; their first 32 * NTH floats are also the
                                                                // accumulator dump [32][NTH] of the epilogue
    static_assert(tile_floats >= 32 * NTH, "accumulator dump");
    static constexpr int lds_floats = tile_floats + BM * 4 + BM * 4 + 3 * 4 * BN;   // + stats, sums, column sums
    static constexpr size_t lds_bytes = 96 * 1024;   // more than half of the 160 KB: one workgroup per CU
    static_assert(lds_floats * sizeof(float) <= lds_bytes, "row-block backward LDS");
};

// RC rows per load chunk of the two epilogue passes: 32 / RC chunks in NS = 8 / RC statically indexed register sets, NS - 1
// chunks of loads in flight ahead of the arithmetic.
__global__ __launch_bounds__(512, 2) void lstm_block_bwd_kernel(const BlockBwd a) {
    typedef BlkBwdCfg C;
    constexpr int NBUF = C::NBUF, RC = 4, BK = C::BK, BM = C::BM, BN = C::BN, NTH = C::NTH, NQ = BK / 8, NS = 8 / RC, NCH = 32 / RC;
    constexpr bool FAST = true, NF = true;
    constexpr int SC = 16;   // sc1 (write-through / L1-bypassing) on the exchanged stores and loads
    extern __shared__ __attribute__((aligned(16))) float blk_lds[];
    float* const As = blk_lds;                    // [NBUF][BM rows][BK]
    float* const Bs = As + NBUF * BK * BM;        // [NBUF][BN rows (units)][BK]
    float* const sl = blk_lds + C::tile_floats;   // [BM][4] mean_x, rstd_x, mean_h, rstd_h of this step's rows
    float* const sa = sl + BM * 4;                // [BM][4] the four LayerNorm-adjoint row sums / 4H
    float* const cl = sa + BM * 4;                // [3][4 BN] column sums of the workgroup (final reduction only)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1, h = lane >> 5, i32 = lane & 31;
    // the lane's row records in sl / sa: ONE address each + compile-time offsets (written as sl + (row) * 4 the compiler
    // materialised all 32 addresses per array as loop invariants and spilled them)
    const float* const slp = sl + (wm * 64 + 4 * h) * 4;
    const float* const sap = sa + (wm * 64 + 4 * h) * 4;
    // Workgroup id -> (row block, unit tile).  Workgroups are dealt round-robin over the 8 XCDs (id % 8), each with its own
    // L2.  The A operand of a row block (128 rows x 4H of dHW, 2 MB) is read by all nnt of its workgroups and by nobody else:
    // put them on ONE XCD (ids x, x + 8, x + 16, ...), so that it crosses the fabric once instead of nnt times; every XCD then
    // streams all of Wh (16 MB per step, shared by its 32 workgroups) instead of one tile of it.  (Linear order otherwise.)
    int rbl = (int)blockIdx.x / a.nnt, nt = (int)blockIdx.x % a.nnt;
    if (a.xcd_map) {
        const int id = (int)blockIdx.x, grp = 8 * a.nnt;
        rbl = (id / grp) * 8 + (id & 7);
        nt = (id % grp) >> 3;
    }
    const int H = a.H, G = 4 * H, nnt = a.nnt;
    const long row0 = (long)(a.rb0 + rbl) * BM;
    const int unit = nt * BN + wn * 32 + i32;
    unsigned* const flag_s = a.flags + 2 * rbl;
    unsigned* const flag_h = flag_s + 1;
    float* const part = a.part + (size_t)rbl * 2 * 4 * nnt * BM * 4;
    const vfloat4 gx = *reinterpret_cast<const vfloat4*>(a.pp + 4 * unit);
    const vfloat4 gh = *reinterpret_cast<const vfloat4*>(a.pp + G + 4 * unit);
    const vfloat4 bx = *reinterpret_cast<const vfloat4*>(a.pp + 2 * G + 4 * unit);
    const vfloat4 bh = *reinterpret_cast<const vfloat4*>(a.pp + 3 * G + 4 * unit);
    const vfloat4 bb = *reinterpret_cast<const vfloat4*>(a.pp + 4 * G + 4 * unit);
    const unsigned xoff = (unsigned)(4 * h) * (unsigned)G + 4u * (unsigned)unit;
    const unsigned uoff = (unsigned)(4 * h) * (unsigned)H + (unsigned)unit;
    if (a.skew_ticks > 0 && rbl > 0) {
        const long long t0 = wall_clock64(), want = (long long)rbl * a.skew_ticks;
        while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(32);
    }
    const int ktiles = G / BK;
    int a_off[NQ], b_off[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        a_off[q] = lds_pos<BK>(wm * 64 + i32, h * (BK / 2) + 4 * q);
        b_off[q] = lds_pos<BK>(wn * 32 + i32, h * (BK / 2) + 4 * q);
    }
    const float inv_g = 1.f / (float)G;
    vfloat4 cs0 = {0.f, 0.f, 0.f, 0.f}, cs1 = cs0, cs2 = cs0;   // column sums of this lane's four gate columns, all steps
    const bool prof_on = a.prof && blockIdx.x == gridDim.x / 2 && tid == 0;
    u64 tprev_ = prof_on ? wall_clock64() : 0;

    // the product: acc = A[row block rows, all 4H] @ whp[units of the tile, all 4H]^T
    f32x16 acc[2];
    auto product = [&](const float* arows) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        DmaStage<BM, BK, NTH> da;
        DmaStage<BN, BK, NTH> db;
        da.init(arows, G, 0, 0);
        db.init(a.whp + (size_t)(nt * BN) * G, G, 0, 0);
        da.issue(As);
        db.issue(Bs);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < ktiles; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < ktiles) {
                da.issue(As + (buf ^ 1) * BK * BM);
                db.issue(Bs + (buf ^ 1) * BK * BN);
            }
            const float* __restrict__ as = As + buf * BK * BM;
            const float* __restrict__ bs = Bs + buf * BK * BN;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                gf4 av[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const gf4*>(as + a_off[q] + i * 32 * BK);
                const gf4 bv = *reinterpret_cast<const gf4*>(bs + b_off[q]);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][t], bv[t], acc[i], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };

    // pair k (0 .. 31) of a lane = accumulator register (k >> 4, k & 15): its row inside the wave's 64 (without the lane's 4 h)
    auto pair_row = [&](int k) __attribute__((always_inline)) { return (k >> 4) * 32 + 8 * ((k & 15) >> 2) + (k & 3); };

    for (int s = a.S - 1; s >= 0; --s) {
        const size_t srow = (size_t)s * a.B + row0;
        const bool last = s == a.S - 1;
        if (!last) {
            block_wait(flag_h, (unsigned)(nnt * (a.S - 1 - s)), NF);   // dHW_{s+1} of this row block is complete
            HPC_RLL_BLK_TICK(0)
            product(a.dhw + (srow + a.B) * G);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        }
        HPC_RLL_BLK_TICK(1)
        if (tid < BM) *reinterpret_cast<vfloat4*>(sl + tid * 4) = *reinterpret_cast<const vfloat4*>(a.stats + (srow + tid) * 4);
        __syncthreads();

        // ---- pass A: gate adjoints of the lane's 32 (row, unit) pairs, LayerNorm-adjoint row sums, column sums.
        // The product's accumulators go through LDS (the operand tiles are dead now): pair k = 16 i + r of every lane at
        // accl[k][tid].  That turns the epilogue into ROLLED loops over chunks of consecutive pairs -- fully unrolled
        // (the accumulator registers can only be indexed statically) it was 10 k instructions and spilled 500 registers.
        float* const accl = blk_lds;   // [32][NTH] floats = the first 64 KB of the operand tiles
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) accl[(16 * i + r) * NTH + tid] = acc[i][r];
        const __amdgpu_buffer_rsrc_t r_xw = blk_rsrc(a.xw + srow * G), r_hw = blk_rsrc(a.hw + srow * G);
        const __amdgpu_buffer_rsrc_t r_dg = blk_rsrc(a.dgate + srow * G);
        const __amdgpu_buffer_rsrc_t r_cn = blk_rsrc(a.c + srow * H);
        const __amdgpu_buffer_rsrc_t r_cp = blk_rsrc(s == 0 ? a.c0 + (size_t)row0 * H : a.c + (srow - a.B) * H);
        const bool has_dy = a.dy != nullptr, has_dhn = last && a.dhn != nullptr, has_dci = !last || a.dcn != nullptr;
        const __amdgpu_buffer_rsrc_t r_dy = blk_rsrc(has_dy ? a.dy + srow * H : a.c);
        const __amdgpu_buffer_rsrc_t r_dhn = blk_rsrc(has_dhn ? a.dhn + (size_t)row0 * H : a.c);
        const __amdgpu_buffer_rsrc_t r_dci = blk_rsrc(last ? (a.dcn ? a.dcn + (size_t)row0 * H : a.c) : a.dc + (size_t)row0 * H);
        const __amdgpu_buffer_rsrc_t r_dco = blk_rsrc((s == 0 ? a.dc0 : a.dc) + (size_t)row0 * H);
        const unsigned xob = 4u * xoff, uob = 4u * uoff;             // the lane's byte offsets
        const unsigned gb = 4u * (unsigned)G, hb = 4u * (unsigned)H;   // bytes per row
        const unsigned wrow = (unsigned)(wm * 64);
        vfloat4 my = {0.f, 0.f, 0.f, 0.f};
        {
            struct In { vfloat4 x[RC], hh[RC]; float cn[RC], cp[RC], dci[RC], dy[RC]; };
            auto load_chunk = [&](int c, In& v) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const unsigned R = wrow + (unsigned)pair_row(RC * c + q);
                    v.x[q] = blk_ld4<0>(r_xw, xob, R * gb);   // (read again in pass B: no streaming hint)
                    v.hh[q] = blk_ld4<0>(r_hw, xob, R * gb);
                    v.cn[q] = blk_ld1<0>(r_cn, uob, R * hb);
                    v.cp[q] = blk_ld1<0>(r_cp, uob, R * hb);
                    v.dci[q] = has_dci ? blk_ld1<0>(r_dci, uob, R * hb) : 0.f;
                    float d = has_dy ? blk_ld1<0>(r_dy, uob, R * hb) : 0.f;
                    if (has_dhn) d += blk_ld1<0>(r_dhn, uob, R * hb);
                    v.dy[q] = d;
                }
            };
            auto do_chunk = [&](int c, const In& v) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int k = RC * c + q;
                    const unsigned R = wrow + (unsigned)pair_row(k);
                    const vfloat4 st = *reinterpret_cast<const vfloat4*>(slp + pair_row(k) * 4);
                    const vfloat4 x4 = v.x[q], h4 = v.hh[q];
                    float pre[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) pre[g] = gate_pre(x4[g], st.x, st.y, gx[g], bx[g], h4[g], st.z, st.w, gh[g], bh[g], bb[g]);
                    const float ig = blk_sigmoid<FAST>(pre[0]), fg = blk_sigmoid<FAST>(pre[1]), og = blk_sigmoid<FAST>(pre[2]);
                    const float ug = blk_tanh<FAST>(pre[3]);
                    const float dh = accl[k * NTH + tid] + v.dy[q];
                    const float tc = blk_tanh<FAST>(v.cn[q]);
                    const float dc = v.dci[q] + dh * og * (1.f - tc * tc);
                    const vfloat4 d4 = {dc * ug * ig * (1.f - ig), dc * v.cp[q] * fg * (1.f - fg), dh * tc * og * (1.f - og),
                                        dc * ig * (1.f - ug * ug)};
                    blk_st4<0>(d4, r_dg, xob, R * gb);
                    blk_st1<0>(dc * fg, r_dco, uob, R * hb);
                    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
                    vfloat4 xh, hh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        xh[g] = (x4[g] - st.x) * st.y;
                        hh[g] = (h4[g] - st.z) * st.w;
                        const float dyx = d4[g] * gx[g], dyh = d4[g] * gh[g];
                        p0 += dyx; p1 += dyx * xh[g];
                        p2 += dyh; p3 += dyh * hh[g];
                    }
                    cs0 += d4;
                    cs1 += d4 * xh;
                    cs2 += d4 * hh;
                    p0 = half_sum_all(p0); p1 = half_sum_all(p1); p2 = half_sum_all(p2); p3 = half_sum_all(p3);
                    if (i32 == k) my = vfloat4{p0, p1, p2, p3};   // lane i32 = 16 i + r = k keeps row k of its half
                }
            };
            In v[NS];
#pragma unroll
            for (int j = 0; j < NS - 1; ++j) load_chunk(j, v[j]);
#pragma unroll 1
            for (int c = 0; c < NCH; c += NS) {   // NS chunks per trip: the input sets rotate statically
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    if (c + j + NS - 1 < NCH) load_chunk(c + j + NS - 1, v[(j + NS - 1) % NS]);
                    do_chunk(c + j, v[j]);
                }
            }
        }
        float* const pslot = part + (size_t)(s & 1) * 4 * nnt * BM * 4;
        const __amdgpu_buffer_rsrc_t r_ps = blk_rsrc(pslot);
        {
            const int rr = i32 & 15;
            const int row = wm * 64 + (i32 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
            blk_st4<SC>(my, r_ps, (unsigned)(((4 * nt + wn) * BM + row) * 16), 0u);
        }
        block_arrive(flag_s, NF);
        HPC_RLL_BLK_TICK(2)
        block_wait(flag_s, (unsigned)(nnt * (a.S - s)), NF);
        HPC_RLL_BLK_TICK(3)
        if (tid < BM) {
            vfloat4 t = {0.f, 0.f, 0.f, 0.f};
            for (int c0 = 0; c0 < 4 * nnt; c0 += 8) {
                vfloat4 p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    p[k] = blk_ld4<SC>(r_ps, (unsigned)(tid * 16), (unsigned)((c0 + k < 4 * nnt ? c0 + k : 0) * BM * 16));
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (c0 + k < 4 * nnt) t += p[k];
            }
            *reinterpret_cast<vfloat4*>(sa + tid * 4) = t * inv_g;
        }
        __syncthreads();
        HPC_RLL_BLK_TICK(4)

        // ---- pass B: dXW_s, dHW_s of the lane's pairs (gate adjoints, xw, hw read back: this lane's own lines, from L2)
        {
            const __amdgpu_buffer_rsrc_t r_dxw = blk_rsrc(a.dxw + srow * G), r_dhw = blk_rsrc(a.dhw + srow * G);
            struct In { vfloat4 x[RC], hh[RC], d[RC]; };
            auto load_chunk = [&](int c, In& v) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const unsigned R = wrow + (unsigned)pair_row(RC * c + q);
                    v.x[q] = blk_ld4<2>(r_xw, xob, R * gb);
                    v.hh[q] = blk_ld4<2>(r_hw, xob, R * gb);
                    v.d[q] = blk_ld4<2>(r_dg, xob, R * gb);
                }
            };
            auto do_chunk = [&](int c, const In& v) __attribute__((always_inline)) {
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int pr = pair_row(RC * c + q);
                    const unsigned R = wrow + (unsigned)pr;
                    const vfloat4 st = *reinterpret_cast<const vfloat4*>(slp + pr * 4);
                    const vfloat4 av = *reinterpret_cast<const vfloat4*>(sap + pr * 4);
                    vfloat4 ox, oh;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float dyx = v.d[q][g] * gx[g], dyh = v.d[q][g] * gh[g];
                        ox[g] = st.y * (dyx - av.x - (v.x[q][g] - st.x) * st.y * av.y);
                        oh[g] = st.w * (dyh - av.z - (v.hh[q][g] - st.z) * st.w * av.w);
                    }
                    blk_st4<2>(ox, r_dxw, xob, R * gb);
                    blk_st4<SC>(oh, r_dhw, xob, R * gb);   // the next product's operand
                }
            };
            In v[NS];
#pragma unroll
            for (int j = 0; j < NS - 1; ++j) load_chunk(j, v[j]);
#pragma unroll 1
            for (int c = 0; c < NCH; c += NS) {
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    if (c + j + NS - 1 < NCH) load_chunk(c + j + NS - 1, v[(j + NS - 1) % NS]);
                    do_chunk(c + j, v[j]);
                }
            }
        }
        HPC_RLL_BLK_TICK(5)
        block_arrive(flag_h, NF);
        HPC_RLL_BLK_TICK(6)
    }
    // ---- dh0 = dHW_0 @ Wh^T, dc0 is already in place
    block_wait(flag_h, (unsigned)(nnt * a.S), NF);
    product(a.dhw + (size_t)row0 * G);
    {
        float* const dh0 = a.dh0 + (size_t)row0 * H;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int R = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);
                (dh0 + (size_t)R * H)[uoff] = acc[i][r];
            }
    }
    // ---- column sums: the 4 lanes-sets that share a unit (two halves x two row waves) meet in LDS; fixed order
    __syncthreads();
    for (int e = tid; e < 3 * 4 * BN; e += NTH) cl[e] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {   // (wm, h) = turn: each column gets exactly one writer per turn
        if (wm * 2 + h == turn) {
            float* c = cl + 4 * (wn * 32 + i32);
            *reinterpret_cast<vfloat4*>(c) += cs0;
            *reinterpret_cast<vfloat4*>(c + 4 * BN) += cs1;
            *reinterpret_cast<vfloat4*>(c + 2 * 4 * BN) += cs2;
        }
        __syncthreads();
    }
    float* const mine = a.colacc + (size_t)rbl * 3 * G;
    for (int e = tid; e < 3 * 4 * BN; e += NTH) {
        const int k3 = e / (4 * BN), c = e - k3 * 4 * BN;
        mine[(size_t)k3 * G + 4 * nt * BN + c] = cl[e];
    }
}

// One column-sum row of 3 * 4H floats per 128-row block goes to the workspace's `colpart` (launch_block_bwd_t), which carve()
// sizes for kBlockBwdMaxRowBlocks (= kCellRowsMaxWgs) rows: larger batches (B > 131072) keep the step kernels.
constexpr int kBlockBwdMaxRowBlocks = kCellRowsMaxWgs;   // (lstm.hip, above this include: the rows carve() gives colpart)
inline bool lstm_block_bwd_shape(int B, int H) {
    return lstm_perm_shape(B, H) && H % 128 == 0 && B % 128 == 0 && B / 128 <= kBlockBwdMaxRowBlocks;
}
inline int block_bwd_rows_per_launch(int H) {
    const int nnt = H / 128;
    return nnt > 0 ? persist_cu_count() / nnt : 0;
}
inline size_t block_bwd_part_floats(int B, int H) { return (size_t)(B / 128) * 2 * 4 * (H / 128) * 128 * 4; }
inline bool block_bwd_resident(int H) {
    const int per = block_bwd_rows_per_launch(H);
    return per >= 1 && persist_resident_t(lstm_block_bwd_kernel, 512, (H / 128) * per, BlkBwdCfg::lds_bytes);
}
inline bool block_bwd_ok(int B, int H, hipStream_t st) {
    if (!(g_lstm_block & 8) || !g_lstm_persist || !lstm_block_bwd_shape(B, H) || !persist_runtime_ready(st)) return false;
    if (!block_launches_fill(B / 128, block_bwd_rows_per_launch(H), H / 128)) return false;
    return block_bwd_resident(H);
}
inline int launch_block_bwd(BlockBwd a, float* part, unsigned* flags, float* colacc, hipStream_t st) {
    typedef BlkBwdCfg C;
    const int nrb = a.B / 128, per = block_bwd_rows_per_launch(a.H);
    if (hipMemsetAsync(flags, 0, block_flag_words(a.B) * sizeof(unsigned), st) != hipSuccess) return last_error();
    const hipError_t e = hipFuncSetAttribute((const void*)lstm_block_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
    if (e != hipSuccess) return (int)e;
    a.nnt = a.H / 128;
    a.skew_ticks = g_lstm_block_skew * 100;
    a.prof = persist_prof();
    for (int rb = 0; rb < nrb; rb += per) {
        const int n = nrb - rb < per ? nrb - rb : per;
        a.rb0 = rb;
        a.part = part + (size_t)rb * 2 * 4 * a.nnt * 128 * 4;
        a.flags = flags + 2 * rb;
        a.colacc = colacc + (size_t)rb * 3 * 4 * a.H;
        a.xcd_map = n % 8 == 0 ? 1 : 0;   // (linear order: neutral end to end, round 4)
        persist_chain_before(st);
        hipLaunchKernelGGL(lstm_block_bwd_kernel, dim3(n * a.nnt), dim3(512), C::lds_bytes, st, a);
        persist_chain_after(st);
    }
    persist_prof_report("row-block bwd: wait_h product passA+publish wait_s combine passB arrive_h", 0, a.S, st);
    return last_error();
}

}  // namespace
}  // namespace hpc_rll
