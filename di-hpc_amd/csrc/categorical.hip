// categorical.hip -- fused categorical-head row kernels shared by V-trace, UPGO and PPO (gfx950).
//
// Replaces the reference's categoricalTarget / categoricalBehaviour (vtrace_kernel.h:11-151),
// crossEntropyKernel (upgo_kernel.h:40-81), categoricalProbEntropy / categoricalProb (ppo_kernel.h:12-150)
// and the three backward kernels that consume their saved (rows,N) buffers (vtrace_kernel.h:235-273,
// upgo_kernel.h:96-108, ppo_kernel.h:252-283).
//
// Reference design: one 256-thread block per row, 5-6 passes over the N logits through 5 block reductions,
// and three N-wide gradient buffers written in forward and re-read in backward.  Here:
//   * a row is owned by a GROUP of G lanes (G = 1..64, a power of two) that keeps the whole row in VGPRs:
//     logits are read from HBM exactly once per kernel, 16 B per lane when N % 4 == 0;
//   * max / sum-exp / sum p*log p are G-lane butterflies (no LDS, no barrier);
//   * forward emits only per-row scalars (log pi(a), entropy); backward RECOMPUTES the softmax from the
//     logits instead of loading saved buffers:  forward writes 8 B/row instead of 12*N B/row.
//
//   * R rows are processed per group per iteration (independent reduction chains interleave, hiding the
//     cross-lane latency), the 16-lane part of every butterfly is DPP (no LDS traffic), exp/log are the
//     hardware v_exp_f32 / v_log_f32 forms: first version (1 row at a time, ds_bpermute butterflies, libm expf)
//     was row-rate bound at 3.3 TB/s;
//   * rows of 2048 < N <= 16384 (N % 4 == 0) are held by a whole WORKGROUP in registers (categorical_blockrow_kernel):
//     forward 6.8 TB/s, backward 5.6-5.9 TB/s at N = 4096 / 8192 / 16384 (the LDS-row kernel they replace: 1-3 TB/s).
//
// Algorithmic HBM bytes: forward 4*N + 8 (+8 for the int64 action) per row; backward 4*N read + 4*N write.
#include <hip/hip_runtime.h>

#include "hpc_rll_hip.h"
#include "wave.hpp"
#include "colscan.hpp"
#include "ppo_op.hpp"

namespace hpc_rll {
namespace {

constexpr float kNegInf = -3.0e38f;
constexpr float kFltMax = 3.402823466e38f;

// Masked actions arrive as logits = -inf.  torch.distributions.Categorical (what hpc_rll.origin uses) clamps
// the normalised logits to the most negative finite float so that p*log p is 0 instead of 0*(-inf) = NaN; clamping
// the raw logit on load has the same effect (its probability underflows to exactly 0).  NaN passes through.
__device__ __forceinline__ float clamp_logit(float x) { return fmaxf(x, -kFltMax); }   // -inf (masked) -> finite

// ---- all-reduce butterflies over aligned groups of G lanes: DPP inside a 16-lane row, ds_bpermute above it.
template <int CTRL> __device__ __forceinline__ float dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
struct SumOp { static __device__ __forceinline__ float f(float a, float b) { return a + b; } };
// max of two values that are never NaN here (finish() removed NaNs): med3(a, b, +inf) is ONE v_med3_f32, whereas fmaxf
// first canonicalises an operand the compiler cannot prove quiet (every DPP move): 3 instructions per butterfly step -> 2
// (+inf comes out of an asm so that the compiler cannot fold med3(a, b, inf) back into maxnum(a, b) + canonicalise)
__device__ __forceinline__ float opaque_inf() { float v; asm("s_mov_b32 %0, 0x7f800000" : "=s"(v)); return v; }
struct MaxOp { static __device__ __forceinline__ float f(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, opaque_inf()); } };
// log of a partition sum relative to the row maximum: s is in [1, N], so the bare v_log_f32 (log2) needs neither the
// denormal pre-scaling nor the two-word ln2 product of __logf (14 instructions per row -> 2)
__device__ __forceinline__ float log_sum(float s) { return __builtin_amdgcn_logf(s) * 0.69314718055994530942f; }
template <int G, class Op> __device__ __forceinline__ float group_all(float x) {
    if (G >= 2) x = Op::f(x, dpp<0xB1>(x));    // quad_perm [1,0,3,2]
    if (G >= 4) x = Op::f(x, dpp<0x4E>(x));    // quad_perm [2,3,0,1]
    if (G >= 8) x = Op::f(x, dpp<0x141>(x));   // row_half_mirror: lane i <-> 7-i
    if (G >= 16) x = Op::f(x, dpp<0x140>(x));  // row_mirror: lane i <-> 15-i
    if (G >= 32) x = Op::f(x, __shfl_xor(x, 16, 64));
    if (G >= 64) x = Op::f(x, __shfl_xor(x, 32, 64));
    return x;
}

// Per-lane slice of one row: E pieces of VEC consecutive floats, piece e at column (e*G + gl)*VEC.
// Two phases.  load() only ISSUES the (nontemporal: logits are read exactly once) loads: every lane loads
// unconditionally -- padding lanes re-read column 0 -- so there is no divergent branch around a load and all E loads of
// all R rows of an iteration are in flight before the first use.  finish() then clamps with ONE v_med3_f32 per
// element: -inf (masked action) becomes the most negative finite float, padding (hi = -FLT_MAX there) becomes that
// same value, a NaN logit becomes -FLT_MAX exactly as fmaxf(x, -FLT_MAX) made it.  (The first version clamped inside
// `if (c < N)`, which compiled to a branch and s_waitcnt vmcnt(0) per load plus two v_max per element; in an in-process
// A/B the two builds time the same to 1 % at every N (tests/tools/cat_ab_probe.py: the kernels are VALU-bound and
// eight waves per SIMD hid the serialised loads) -- this form is kept for being branch-free and 40 instructions shorter.)
template <int G, int VEC, int E>
struct RowSlice {
    float x[E * VEC];
    __device__ __forceinline__ void load(const float* __restrict__ row, int N, int gl) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = (e * G + gl) * VEC;
            const int cc = (c < N) ? c : 0;
            if (VEC == 4) {
                const vfloat4 t = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(row + cc));
                x[e * 4 + 0] = t.x; x[e * 4 + 1] = t.y; x[e * 4 + 2] = t.z; x[e * 4 + 3] = t.w;
            } else {
                x[e] = __builtin_nontemporal_load(row + cc);
            }
        }
    }
    __device__ __forceinline__ void finish(int N, int gl) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = (e * G + gl) * VEC;
            const float hi = (c < N) ? __builtin_inff() : -kFltMax;   // padding: finite, so that merges stay NaN-free
#pragma unroll
            for (int k = 0; k < VEC; ++k) x[e * VEC + k] = __builtin_amdgcn_fmed3f(x[e * VEC + k], -kFltMax, hi);
        }
    }
};

// ---- Softmax statistics of the row held by a G-lane group.
// Instruction count matters here: at N = 128 the first version issued ~130 VALU instructions per row pair and the
// forward kernel was VALU-bound (PMC: SQ_ACTIVE_INST_VALU = 0.47 ms of a 0.52 ms kernel, 4.2 TB/s) although a pure
// read streams at 7.0 TB/s (tests/tools/micro/readbw.hip).  Hence:
//   * entropy from the same pass as the partition sum: H = log s - (sum e_i d_i) / s with d_i = x_i - m, e_i = exp d_i
//     (no second loop over the row, no per-element division);
//   * the action compare in 32 bits against an index that is -1 when out of range;
//   * padding checks only when the row does not fill the group exactly (uniform branch);
//   * forward: NO cross-row all-reduce.  Each 16-lane DPP row reduces (max, s, t, x_a) relative to its OWN maximum
//     with DPP butterflies only; rows are then merged pairwise with the log-sum-exp merge rule through row_bcast
//     moves (VALU, no ds_bpermute), leaving the result in the LAST lane of the group, which writes the outputs;
//   * backward needs p_i in every lane: three all-reduces (max, s, t) instead of four.
template <int G, class Op> __device__ __forceinline__ float row_all(float x) {   // all-reduce over min(G,16) lanes
    if (G >= 2) x = Op::f(x, dpp<0xB1>(x));
    if (G >= 4) x = Op::f(x, dpp<0x4E>(x));
    if (G >= 8) x = Op::f(x, dpp<0x141>(x));
    if (G >= 16) x = Op::f(x, dpp<0x140>(x));
    return x;
}
// rows 1,3 <- lane 15 of rows 0,2 (CTRL 0x142, mask 0xA); rows 2,3 <- lane 31 (CTRL 0x143, mask 0xC); others keep x
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float row_fetch(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x),
                                                                 CTRL, ROW_MASK, 0xF, false));
}
struct RowAcc { float m, s, t, xa; };
// merge the statistics of two disjoint parts of a row (each relative to its own maximum)
__device__ __forceinline__ RowAcc merge(const RowAcc& a, const RowAcc& b) {
    RowAcc r;
    r.m = MaxOp::f(a.m, b.m);
    const float da = a.m - r.m, db = b.m - r.m;
    const float wa = __expf(da), wb = __expf(db);
    r.s = a.s * wa + b.s * wb;
    r.t = wa * fmaf(da, a.s, a.t) + wb * fmaf(db, b.s, b.t);
    r.xa = a.xa + b.xa;
    return r;
}

template <int G, int VEC, int E, bool PACK>
__device__ __forceinline__ RowAcc lane_stats(const RowSlice<G, VEC, E>& r, int N, int gl, int ai, bool full,
                                             float (&ex)[E * VEC], float m) {
    RowAcc a;
    a.m = m;
    a.s = 0.f; a.t = 0.f; a.xa = 0.f;
    if (PACK && full && VEC == 4) {   // forward only: measured -4 % there, +7 % in the backward kernel
        // packed fp32 (v_pk_add/mul/fma_f32: two elements per instruction) for everything but exp and the compare
        typedef float f2 __attribute__((ext_vector_type(2)));
        const f2 m2 = {m, m};
        const f2 l2e = {1.44269504088896340736f, 1.44269504088896340736f};
        f2 s2 = {0.f, 0.f}, t2 = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < VEC; k += 2) {
                const int i = e * VEC + k;
                const f2 x2 = {r.x[i], r.x[i + 1]};
                const f2 d2 = x2 - m2;
                const f2 a2 = d2 * l2e;
                f2 e2;
                e2.x = __builtin_amdgcn_exp2f(a2.x);
                e2.y = __builtin_amdgcn_exp2f(a2.y);
                ex[i] = e2.x;
                ex[i + 1] = e2.y;
                s2 += e2;
                t2 = __builtin_elementwise_fma(e2, d2, t2);
                const int c = (e * G + gl) * VEC + k;
                a.xa = (c == ai) ? r.x[i] : a.xa;          // at most one column matches: a select, not a masked add
                a.xa = (c + 1 == ai) ? r.x[i + 1] : a.xa;
            }
        a.s = s2.x + s2.y;
        a.t = t2.x + t2.y;
    } else if (full) {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int i = e * VEC + k;
                const float d = r.x[i] - m;
                ex[i] = __expf(d);
                a.s += ex[i];
                a.t = fmaf(ex[i], d, a.t);
                a.xa = ((e * G + gl) * VEC + k == ai) ? r.x[i] : a.xa;
            }
    } else {
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const int i = e * VEC + k;
                const int c = (e * G + gl) * VEC + k;
                const float d = r.x[i] - m;
                ex[i] = (c < N) ? __expf(d) : 0.f;
                a.s += ex[i];
                a.t = (c < N) ? fmaf(ex[i], d, a.t) : a.t;
                a.xa = (c == ai) ? r.x[i] : a.xa;
            }
    }
    return a;
}

// forward: log p(action) and entropy, valid in the LAST lane of the group
template <int G, int VEC, int E>
__device__ __forceinline__ void row_stats_fwd(const RowSlice<G, VEC, E>& r, int N, int gl, int ai, bool full,
                                              float& logp_a, float& ent) {
    float m = r.x[0];
#pragma unroll
    for (int i = 1; i < E * VEC; ++i) m = fmaxf(m, r.x[i]);
    m = row_all<G, MaxOp>(m);
    float ex[E * VEC];
    RowAcc a = lane_stats<G, VEC, E, true>(r, N, gl, ai, full, ex, m);
    a.s = row_all<G, SumOp>(a.s);
    a.t = row_all<G, SumOp>(a.t);
    a.xa = row_all<G, SumOp>(a.xa);
    if (G >= 32) {   // rows 1 and 3 merge the row before them
        RowAcc b;
        b.m = row_fetch<0x142, 0xA>(a.m); b.s = row_fetch<0x142, 0xA>(a.s);
        b.t = row_fetch<0x142, 0xA>(a.t); b.xa = row_fetch<0x142, 0xA>(a.xa);
        const bool odd = (threadIdx.x & 16) != 0;
        const RowAcc mm = merge(a, b);
        a.m = odd ? mm.m : a.m; a.s = odd ? mm.s : a.s; a.t = odd ? mm.t : a.t; a.xa = odd ? mm.xa : a.xa;
    }
    if (G >= 64) {   // row 3 merges lanes 0..31 (held by lane 31)
        RowAcc b;
        b.m = row_fetch<0x143, 0xC>(a.m); b.s = row_fetch<0x143, 0xC>(a.s);
        b.t = row_fetch<0x143, 0xC>(a.t); b.xa = row_fetch<0x143, 0xC>(a.xa);
        const bool hi = (threadIdx.x & 32) != 0;
        const RowAcc mm = merge(a, b);
        a.m = hi ? mm.m : a.m; a.s = hi ? mm.s : a.s; a.t = hi ? mm.t : a.t; a.xa = hi ? mm.xa : a.xa;
    }
    const float ls = log_sum(a.s);
    logp_a = a.xa - (a.m + ls);
    ent = ls - a.t * __builtin_amdgcn_rcpf(a.s);
}

// backward: ex[i] = exp(x_i - max) and the row scalars in EVERY lane of the group
template <int G, int VEC, int E>
__device__ __forceinline__ void row_stats(const RowSlice<G, VEC, E>& r, int N, int gl, int ai, bool full,
                                          float (&ex)[E * VEC], float& lse, float& inv_sum, float& ent) {
    float m = r.x[0];
#pragma unroll
    for (int i = 1; i < E * VEC; ++i) m = fmaxf(m, r.x[i]);
    m = group_all<G, MaxOp>(m);
    const RowAcc a = lane_stats<G, VEC, E, false>(r, N, gl, ai, full, ex, m);
    const float s = group_all<G, SumOp>(a.s);
    const float t = group_all<G, SumOp>(a.t);
    const float ls = log_sum(s);
    inv_sum = __builtin_amdgcn_rcpf(s);
    lse = m + ls;
    ent = ls - t * inv_sum;
}

// R rows per group per iteration: R independent load + reduction chains in flight.
template <int G, int VEC, int E> struct RowsPerIter { static constexpr int value = (E * VEC <= 4) ? 4 : ((E * VEC <= 8) ? 2 : 1); };

// ENT = false (no entropy output: V-trace's behaviour head, PPO's old policy, UPGO): the entropy accumulation, its
// reduction and its part of the row merges are dead code and disappear (~10 % of the instructions of a VALU-bound kernel)
template <int G, int VEC, int E, bool ENT>
__device__ __forceinline__ void categorical_fwd_body(const float* __restrict__ logits,
                                                     const int64_t* __restrict__ action,
                                                     float* __restrict__ logp_out,
                                                     float* __restrict__ ent_out, long rows, int N) {
    constexpr int GPB = 256 / G;  // groups per block
    constexpr int R = RowsPerIter<G, VEC, E>::value;
    const int gl = threadIdx.x % G;
    const int gi = threadIdx.x / G;
    const bool full = N == G * VEC * E;   // uniform: no padding lanes
    // row of (iteration block bb, slot k, group gi) = bb + k*GPB + gi: for a fixed k the groups of the whole workgroup
    // read consecutive rows, i.e. one contiguous span per load instruction whatever G is.
    // Row addresses advance by a uniform stride (one 64-bit add per iteration): row * N as a per-row 64-bit product
    // cost two quarter-rate v_mul_lo_u32 and a v_mad_u64_u32 per row in a VALU-bound kernel.  Rows past the end
    // (last iteration only) re-read the last row; their stores are guarded.
    const long stride = (long)gridDim.x * GPB * R;
    const float* const last_row = logits + (rows - 1) * (long)N;
    long w0 = (long)blockIdx.x * GPB * R + gi;                 // this group's row in slot 0
    const float* p0 = logits + w0 * (long)N;
    for (long bb = (long)blockIdx.x * GPB * R; bb < rows; bb += stride, w0 += stride, p0 += stride * N) {
        RowSlice<G, VEC, E> r[R];
        long a[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool ok = w0 + (long)k * GPB < rows;
            r[k].load(ok ? p0 + (long)k * GPB * N : last_row, N, gl);
            a[k] = action[ok ? w0 + (long)k * GPB : rows - 1];
        }
        float lp[R], h[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ai = (a[k] >= 0 && a[k] < (long)N) ? (int)a[k] : -1;
            r[k].finish(N, gl);
            row_stats_fwd<G, VEC, E>(r[k], N, gl, ai, full, lp[k], h[k]);
        }
        if (gl == G - 1) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const long row = bb + (long)k * GPB + gi;
                if (row < rows) {
                    logp_out[row] = lp[k];
                    if (ENT) ent_out[row] = h[k];
                }
            }
        }
    }
}
template <int G, int VEC, int E>
__global__ __launch_bounds__(256) void categorical_fwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ action,
                                                              float* __restrict__ logp_out,
                                                              float* __restrict__ ent_out, long rows, int N) {
    categorical_fwd_body<G, VEC, E, true>(logits, action, logp_out, ent_out, rows, N);
}
template <int G, int VEC, int E>
__global__ __launch_bounds__(256) void categorical_fwd_noent_kernel(const float* __restrict__ logits,
                                                                    const int64_t* __restrict__ action,
                                                                    float* __restrict__ logp_out,
                                                                    float* __restrict__ ent_out, long rows, int N) {
    categorical_fwd_body<G, VEC, E, false>(logits, action, logp_out, ent_out, rows, N);
}

// PPO forward in ONE launch: a group reads the SAME row of both policy heads (new: log p and entropy, old: log p), its last
// lane applies the per-sample loss arithmetic (ppo_op.hpp) and keeps five running sums; workgroup sums -> partials -> the
// last workgroup folds them (colscan.hpp).  Replaces two categorical launches + the sample launch: at B = 65536, N = 128 the
// forward is 67 MB of reads, i.e. launch boundaries, not bytes, were most of its 24 us.  Per-sample outputs are the bits of
// the three-launch path (same functions); the five sums group the rows differently (rounding).
template <int G, int VEC, int E>
__global__ __launch_bounds__(256) void ppo_fwd_fused_kernel(const float* __restrict__ logits_new,
                                                            const float* __restrict__ logits_old,
                                                            const int64_t* __restrict__ action, const PpoOp op, long rows,
                                                            int N, float* __restrict__ partials, const ScanFold fold) {
    constexpr int GPB = 256 / G;
    constexpr int R = 4;   // rows per group and iteration: 4 rows x 2 heads x E loads in flight per lane (the grid is at most 512 workgroups)
    __shared__ float red[PpoOp::NACC * 4];
    const int gl = threadIdx.x % G;
    const int gi = threadIdx.x / G;
    const bool full = N == G * VEC * E;
    const long stride = (long)gridDim.x * GPB * R;
    float acc[PpoOp::NACC];
#pragma unroll
    for (int k = 0; k < PpoOp::NACC; ++k) acc[k] = 0.f;
    for (long bb = (long)blockIdx.x * GPB * R; bb < rows; bb += stride) {
        RowSlice<G, VEC, E> rn[R], ro[R];
        long a[R];
        PpoOp::In in[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            long row = bb + (long)k * GPB + gi;
            if (row >= rows) row = rows - 1;             // (re-reads the last row; the sample op below is guarded)
            rn[k].load(logits_new + row * (long)N, N, gl);
            ro[k].load(logits_old + row * (long)N, N, gl);
            a[k] = action[row];
            in[k] = op.load(row);                        // (every lane: the same address per group, one request)
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int ai = (a[k] >= 0 && a[k] < (long)N) ? (int)a[k] : -1;
            float lpn, h, lpo, h_unused;
            rn[k].finish(N, gl);
            ro[k].finish(N, gl);
            row_stats_fwd<G, VEC, E>(rn[k], N, gl, ai, full, lpn, h);
            row_stats_fwd<G, VEC, E>(ro[k], N, gl, ai, full, lpo, h_unused);
            const long row = bb + (long)k * GPB + gi;
            if (gl == G - 1 && row < rows) op.apply(row, in[k], lpn, h, lpo, acc);
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < PpoOp::NACC; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[k * 4 + w] = s;
    }
    __syncthreads();
    float sum = 0.f;
    if (threadIdx.x < PpoOp::NACC)
        sum = (red[threadIdx.x * 4] + red[threadIdx.x * 4 + 1]) + (red[threadIdx.x * 4 + 2] + red[threadIdx.x * 4 + 3]);
    publish_sums<PpoOp::NACC, 256>(sum, partials, fold);
}

// grad[row,i] = g1*c1[row]*(1[i==a] - p_i) + g2*c2[row]*(-p_i*(log p_i + H))
template <int G, int VEC, int E>
__global__ __launch_bounds__(256) void categorical_bwd_kernel(const float* __restrict__ logits,
                                                              const int64_t* __restrict__ action,
                                                              const float* __restrict__ c1,
                                                              const float* __restrict__ g1,
                                                              const float* __restrict__ c2,
                                                              const float* __restrict__ g2,
                                                              float* __restrict__ grad, long rows, int N) {
    constexpr int GPB = 256 / G;
    constexpr int R = RowsPerIter<G, VEC, E>::value;
    const int gl = threadIdx.x % G;
    const int gi = threadIdx.x / G;
    const float u1 = g1 ? g1[0] : 1.f;
    const float u2 = (c2 != nullptr) ? (g2 ? g2[0] : 1.f) : 0.f;
    const bool full = N == G * VEC * E;
    const long stride = (long)gridDim.x * GPB * R;            // uniform row stride per iteration (see the forward body)
    const long last_off = (rows - 1) * (long)N;
    long w0 = (long)blockIdx.x * GPB * R + gi;
    long off0 = w0 * (long)N;                                  // element offset of this group's slot-0 row
    for (long bb = (long)blockIdx.x * GPB * R; bb < rows; bb += stride, w0 += stride, off0 += stride * N) {
        RowSlice<G, VEC, E> r[R];
        long a[R];
        float k1[R], k2[R];
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const bool ok = w0 + (long)k * GPB < rows;
            const long row = ok ? w0 + (long)k * GPB : rows - 1;
            r[k].load(logits + (ok ? off0 + (long)k * GPB * N : last_off), N, gl);
            a[k] = action[row];
            k1[k] = u1 * c1[row];
            k2[k] = (c2 != nullptr) ? u2 * c2[row] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < R; ++k) {
            float ex[E * VEC], lse, inv, h;
            const int ai = (a[k] >= 0 && a[k] < (long)N) ? (int)a[k] : -1;
            r[k].finish(N, gl);
            row_stats<G, VEC, E>(r[k], N, gl, ai, full, ex, lse, inv, h);
            if (w0 + (long)k * GPB >= rows) continue;
            float* __restrict__ out = grad + (off0 + (long)k * GPB * N);
            // grad_i = k1 1[i==a] - p_i (k1 + k2 (x_i - c)),  c = lse - H.  Evaluated as fma(u, x_i - c, fma(q, k1, k1 1[i==a]))
            // with q = -p_i, u = q k2: seven instructions per element instead of ten, and a masked action (p = 0,
            // x = -FLT_MAX) multiplies 0 by a FINITE number whatever k2 is.
            const float ninv = -inv, cc = lse - h, kk1 = k1[k], kk2 = k2[k];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int c0 = (e * G + gl) * VEC;
                float o[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) {
                    const int i = e * VEC + q;
                    const float qq = ex[i] * ninv;
                    const float hot = (c0 + q == ai) ? kk1 : 0.f;
                    o[q] = fmaf(qq * kk2, r[k].x[i] - cc, fmaf(qq, kk1, hot));
                }
                if (c0 < N) {
                    if (VEC == 4) {
                        vfloat4 t; t.x = o[0]; t.y = o[1]; t.z = o[2]; t.w = o[3];
                        __builtin_nontemporal_store(t, reinterpret_cast<vfloat4*>(out + c0));
                    } else {
                        __builtin_nontemporal_store(o[0], out + c0);
                    }
                }
            }
        }
    }
}

// ---- small action spaces whose rows are not 16-byte multiples (N % 4 != 0, N <= 32: Atari's 6 / 9 / 18, ...).
// A row-per-group mapping would issue 4-byte loads at odd offsets (measured 1.1-1.9 TB/s).  Instead a workgroup
// moves 256 consecutive rows as ONE flat, 16-byte aligned stream through LDS (coalesced float4 both ways), and each
// thread owns one row in LDS (stride N words: conflict free for odd N, 2-way for N = 2 mod 4).
constexpr int kSmallRows = 256;
constexpr int kSmallMaxN = 32;
template <bool BWD>
__global__ __launch_bounds__(256) void categorical_small_kernel(const float* __restrict__ logits,
                                                                const int64_t* __restrict__ action,
                                                                float* __restrict__ logp_out,
                                                                float* __restrict__ ent_out,
                                                                const float* __restrict__ c1,
                                                                const float* __restrict__ g1,
                                                                const float* __restrict__ c2,
                                                                const float* __restrict__ g2,
                                                                float* __restrict__ grad, long rows, int N) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // kSmallRows * N floats (6 KB at N = 6: 8 workgroups per CU)
    const float u1 = (BWD && g1) ? g1[0] : 1.f;
    const float u2 = (BWD && c2 != nullptr) ? (g2 ? g2[0] : 1.f) : 0.f;
    for (long row0 = (long)blockIdx.x * kSmallRows; row0 < rows; row0 += (long)gridDim.x * kSmallRows) {
        const int nr = (int)((rows - row0 < kSmallRows) ? rows - row0 : kSmallRows);
        const int total = nr * N;
        const float* __restrict__ src = logits + row0 * (long)N;   // row0 % 256 == 0 -> 16-byte aligned
        for (int i = threadIdx.x * 4; i + 3 < total; i += 1024)
            *reinterpret_cast<vfloat4*>(tile + i) = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(src + i));
        for (int i = (total & ~3) + threadIdx.x; i < total; i += 256) tile[i] = src[i];
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            float* __restrict__ x = tile + threadIdx.x * N;
            const long row = row0 + threadIdx.x;
            const long a = action[row];
            float m = kNegInf;
            for (int i = 0; i < N; ++i) m = fmaxf(m, clamp_logit(x[i]));
            // partition sum and entropy from ONE exp per element: H = log s - (sum e_i d_i) / s, d_i = x_i - m
            float sum = 0.f, t = 0.f;
            for (int i = 0; i < N; ++i) {
                const float d = clamp_logit(x[i]) - m;
                const float e = __expf(d);
                sum += e;
                t = fmaf(e, d, t);
            }
            const float ls = log_sum(sum);
            const float lse = m + ls;
            const float h = ls - t * __builtin_amdgcn_rcpf(sum);
            if (!BWD) {
                logp_out[row] = ((a >= 0 && a < N) ? clamp_logit(x[a]) : 0.f) - lse;
                if (ent_out) ent_out[row] = h;
            } else {
                const float k1 = u1 * c1[row];
                const float k2 = (c2 != nullptr) ? u2 * c2[row] : 0.f;
                for (int i = 0; i < N; ++i) {
                    const float lp = clamp_logit(x[i]) - lse;
                    const float p = __expf(lp);
                    x[i] = k1 * (((long)i == a ? 1.f : 0.f) - p) - k2 * p * (lp + h);
                }
            }
        }
        if (BWD) {
            __syncthreads();
            float* __restrict__ dst = grad + row0 * (long)N;
            for (int i = threadIdx.x * 4; i + 3 < total; i += 1024)
                __builtin_nontemporal_store(*reinterpret_cast<const vfloat4*>(tile + i), reinterpret_cast<vfloat4*>(dst + i));
            for (int i = (total & ~3) + threadIdx.x; i < total; i += 256) dst[i] = tile[i];
        }
        __syncthreads();
    }
}

// ---- wide rows, 2048 < N <= 16384 with N % 4 == 0: one WORKGROUP per row, the row lives in registers (E float4 per
// thread, all loads in flight at once), statistics in one pass exactly like the G-lane kernels (max, then s = sum e_i and
// t = sum e_i d_i from the same exp), two barriers per row through double-buffered LDS words.  Replaces the LDS-row
// kernel on these shapes (4-byte loads, three passes and two exps per element: 1.8 TB/s at N = 8192).
template <int E, bool BWD>
__global__ __launch_bounds__(256) void categorical_blockrow_kernel(const float* __restrict__ logits,
                                                                   const int64_t* __restrict__ action,
                                                                   float* __restrict__ logp_out,
                                                                   float* __restrict__ ent_out,
                                                                   const float* __restrict__ c1,
                                                                   const float* __restrict__ g1,
                                                                   const float* __restrict__ c2,
                                                                   const float* __restrict__ g2,
                                                                   float* __restrict__ grad, long rows, int N) {
    __shared__ float red_m[2][4], red_s[2][4], red_t[2][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float u1 = (BWD && g1) ? g1[0] : 1.f;
    const float u2 = (BWD && c2 != nullptr) ? (g2 ? g2[0] : 1.f) : 0.f;
    int buf = 0;
    for (long row = blockIdx.x; row < rows; row += gridDim.x, buf ^= 1) {
        const float* __restrict__ x = logits + row * (long)N;
        float v[E * 4];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int c = (e * 256 + (int)threadIdx.x) * 4;
            const vfloat4 t = __builtin_nontemporal_load(reinterpret_cast<const vfloat4*>(x + (c < N ? c : 0)));
            v[e * 4 + 0] = t.x; v[e * 4 + 1] = t.y; v[e * 4 + 2] = t.z; v[e * 4 + 3] = t.w;
        }
        const long a = action[row];
        const bool a_ok = a >= 0 && a < (long)N;
        float xa = 0.f;
        if (!BWD) xa = x[a_ok ? a : 0];                      // one 4-byte load, in flight with the row
        float k1 = 0.f, k2 = 0.f;
        if (BWD) {
            k1 = u1 * c1[row];
            k2 = (c2 != nullptr) ? u2 * c2[row] : 0.f;
        }
        float m = -kFltMax;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float hi = ((e * 256 + (int)threadIdx.x) * 4 < N) ? __builtin_inff() : -kFltMax;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[e * 4 + k] = __builtin_amdgcn_fmed3f(v[e * 4 + k], -kFltMax, hi);
                m = fmaxf(m, v[e * 4 + k]);
            }
        }
        m = group_all<64, MaxOp>(m);
        if (lane == 0) red_m[buf][w] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red_m[buf][0], red_m[buf][1]), fmaxf(red_m[buf][2], red_m[buf][3]));
        float s = 0.f, t = 0.f;
#pragma unroll
        for (int i = 0; i < E * 4; ++i) {
            const float d = v[i] - m;
            const float ex = __expf(d);
            s += ex;
            t = fmaf(ex, d, t);
            if (BWD) v[i] = d;          // keep d; exp(d) is recomputed below only when registers are short (E = 16)
        }
        s = group_all<64, SumOp>(s);
        t = group_all<64, SumOp>(t);
        if (lane == 0) { red_s[buf][w] = s; red_t[buf][w] = t; }
        __syncthreads();
        s = (red_s[buf][0] + red_s[buf][1]) + (red_s[buf][2] + red_s[buf][3]);
        t = (red_t[buf][0] + red_t[buf][1]) + (red_t[buf][2] + red_t[buf][3]);
        const float ls = log_sum(s);
        const float inv = __builtin_amdgcn_rcpf(s);
        const float h = ls - t * inv;
        if (!BWD) {
            if (threadIdx.x == 0) {
                logp_out[row] = (a_ok ? clamp_logit(xa) : 0.f) - (m + ls);
                if (ent_out) ent_out[row] = h;
            }
        } else {
            // grad_i = k1 (1[i==a] - p_i) - k2 p_i (log p_i + H),  log p_i = d_i - ls
            float* __restrict__ out = grad + row * (long)N;
            const int ai = a_ok ? (int)a : -1;
            const float hl = h - ls;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int c = (e * 256 + (int)threadIdx.x) * 4;
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = v[e * 4 + k];
                    const float pr = __expf(d) * inv;
                    const float onehot = (c + k == ai) ? 1.f : 0.f;
                    o[k] = k1 * (onehot - pr) - k2 * pr * (d + hl);
                }
                if (c < N) {
                    vfloat4 q; q.x = o[0]; q.y = o[1]; q.z = o[2]; q.w = o[3];
                    __builtin_nontemporal_store(q, reinterpret_cast<vfloat4*>(out + c));
                }
            }
        }
    }
}

// ---- long rows that cannot take 16-byte loads (N % 4 != 0 or unaligned, up to 16384): one workgroup per row, the row is read from HBM ONCE into
// LDS with coalesced loads and the three passes (max, sum-exp, entropy / gradient) run out of LDS.
template <bool BWD>
__global__ __launch_bounds__(256) void categorical_ldsrow_kernel(const float* __restrict__ logits,
                                                                 const int64_t* __restrict__ action,
                                                                 float* __restrict__ logp_out,
                                                                 float* __restrict__ ent_out,
                                                                 const float* __restrict__ c1,
                                                                 const float* __restrict__ g1,
                                                                 const float* __restrict__ c2,
                                                                 const float* __restrict__ g2,
                                                                 float* __restrict__ grad, long rows, int N) {
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    __shared__ float red[8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float u1 = (BWD && g1) ? g1[0] : 1.f;
    const float u2 = (BWD && c2 != nullptr) ? (g2 ? g2[0] : 1.f) : 0.f;
    auto block_reduce = [&](float v, bool is_max) -> float {
        v = is_max ? wave_max(v) : wave_sum(v);
        __syncthreads();
        if (lane == 0) red[w] = v;
        __syncthreads();
        return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
    };
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* __restrict__ x = logits + row * (long)N;
        float m = kNegInf;
        for (int c = threadIdx.x; c < N; c += 256) {
            const float v = clamp_logit(__builtin_nontemporal_load(x + c));
            rowbuf[c] = v;
            m = fmaxf(m, v);
        }
        m = block_reduce(m, true);
        float s = 0.f;
        for (int c = threadIdx.x; c < N; c += 256) s += __expf(rowbuf[c] - m);
        s = block_reduce(s, false);
        const float lse = m + __logf(s);
        float h = 0.f;
        for (int c = threadIdx.x; c < N; c += 256) {
            const float lp = rowbuf[c] - lse;
            h -= __expf(lp) * lp;
        }
        h = block_reduce(h, false);
        const long a = action[row];
        if (!BWD) {
            if (threadIdx.x == 0) {
                logp_out[row] = ((a >= 0 && a < N) ? rowbuf[a] : 0.f) - lse;
                if (ent_out) ent_out[row] = h;
            }
        } else {
            const float k1 = u1 * c1[row];
            const float k2 = (c2 != nullptr) ? u2 * c2[row] : 0.f;
            float* __restrict__ out = grad + row * (long)N;
            for (int c = threadIdx.x; c < N; c += 256) {
                const float lp = rowbuf[c] - lse;
                const float p = __expf(lp);
                __builtin_nontemporal_store(k1 * (((long)c == a ? 1.f : 0.f) - p) - k2 * p * (lp + h), out + c);
            }
        }
        __syncthreads();
    }
}

// ---- generic fallback for rows too long for registers (N > 64*VEC*8): three L2-friendly passes, one wave per row
__global__ __launch_bounds__(256) void categorical_fwd_long_kernel(const float* __restrict__ logits,
                                                                   const int64_t* __restrict__ action,
                                                                   float* __restrict__ logp_out,
                                                                   float* __restrict__ ent_out, long rows, int N) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wv; row < rows; row += (long)gridDim.x * 4) {
        const float* __restrict__ x = logits + row * (long)N;
        float m = kNegInf;
        for (int c = lane; c < N; c += 64) m = fmaxf(m, clamp_logit(x[c]));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < N; c += 64) s += expf(clamp_logit(x[c]) - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        float h = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float lp = clamp_logit(x[c]) - lse;
            h -= expf(lp) * lp;
        }
        h = wave_sum(h);
        if (lane == 0) {
            const long a = action[row];
            logp_out[row] = ((a >= 0 && a < N) ? clamp_logit(x[a]) : 0.f) - lse;
            if (ent_out) ent_out[row] = h;
        }
    }
}

__global__ __launch_bounds__(256) void categorical_bwd_long_kernel(const float* __restrict__ logits,
                                                                   const int64_t* __restrict__ action,
                                                                   const float* __restrict__ c1,
                                                                   const float* __restrict__ g1,
                                                                   const float* __restrict__ c2,
                                                                   const float* __restrict__ g2,
                                                                   float* __restrict__ grad, long rows, int N) {
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const float u1 = g1 ? g1[0] : 1.f;
    const float u2 = (c2 != nullptr) ? (g2 ? g2[0] : 1.f) : 0.f;
    for (long row = (long)blockIdx.x * 4 + wv; row < rows; row += (long)gridDim.x * 4) {
        const float* __restrict__ x = logits + row * (long)N;
        float m = kNegInf;
        for (int c = lane; c < N; c += 64) m = fmaxf(m, clamp_logit(x[c]));
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < N; c += 64) s += expf(clamp_logit(x[c]) - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        float h = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float lp = clamp_logit(x[c]) - lse;
            h -= expf(lp) * lp;
        }
        h = wave_sum(h);
        const long a = action[row];
        const float k1 = u1 * c1[row];
        const float k2 = (c2 != nullptr) ? u2 * c2[row] : 0.f;
        float* __restrict__ out = grad + row * (long)N;
        for (int c = lane; c < N; c += 64) {
            const float lp = clamp_logit(x[c]) - lse;
            const float p = expf(lp);
            out[c] = k1 * (((long)c == a ? 1.f : 0.f) - p) - k2 * p * (lp + h);
        }
    }
}

}  // namespace
constexpr int g_blocks_per_cu = 1024;  // cap of the row kernels' grids, in workgroups per CU.  Rounds 1-3 shipped 24 (resident
// workgroups looping over ~20 slices of the rows: "12: 364 / 736 us, 24: 335 / 724").  Round 4 (profiles/r04_writebw.txt: a stream of
// short-lived workgroups that each write one aligned block beats long-lived ones) lifted the cap -- at the C3 shape every workgroup
// now takes ONE slice of 32 rows and retires: V-trace 0.755 / 0.784 -> 0.72 / 0.68 ms, UPGO 0.366 / 0.736 -> 0.323 / 0.680
// (24 -> 64 -> 256 -> 1024 workgroups per CU: backward 0.784, 0.714, 0.686, 0.678 ms)
namespace {
inline unsigned grid_for(long rows, int rows_per_block) {
    long g = (rows + rows_per_block - 1) / rows_per_block;
    const long cap = 256L * g_blocks_per_cu;  // 256 CUs x workgroups per CU (above it the workgroups loop)
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

struct RowCfg { int g, vec, e; };

inline RowCfg row_cfg(int N, bool can_vec4) {
    RowCfg c;
    c.vec = (can_vec4 && (N % 4) == 0) ? 4 : 1;
    const int pieces = (N + c.vec - 1) / c.vec;
    // A row is held by at most 16 lanes (one DPP row) whenever 8 pieces per lane suffice: the per-row reductions are
    // then four DPP steps with no cross-row merge, and their cost is amortised over more elements per lane (the
    // kernels are VALU-bound, not bandwidth-bound, at one float4 per lane).  Longer rows take the whole wave.
    const int gmax = pieces <= 16 * 8 ? 16 : 64;   // (8 lanes per row measured no better at N = 64..128, worse at 256)
    c.g = 1;
    while (c.g < gmax && c.g < pieces) c.g <<= 1;
    const int e = (pieces + c.g - 1) / c.g;
    c.e = 1;
    while (c.e < e) c.e <<= 1;
    return c;  // e > 8 means "use the long-row fallback"
}

#define HPC_RLL_ROW_CASE(G_, V_, E_, KERNEL, ...)                                                         \
    if (cfg.g == G_ && cfg.vec == V_ && cfg.e == E_) {                                                    \
        hipLaunchKernelGGL((KERNEL<G_, V_, E_>),                                                           \
                           dim3(grid_for(rows, (256 / G_) * RowsPerIter<G_, V_, E_>::value)), dim3(256), 0, st, \
                           __VA_ARGS__);                                                                  \
        return true;                                                                                      \
    }
#define HPC_RLL_ROW_DISPATCH(KERNEL, ...)                                                                  \
    HPC_RLL_ROW_CASE(1, 1, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(2, 1, 1, KERNEL, __VA_ARGS__)          \
    HPC_RLL_ROW_CASE(4, 1, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(8, 1, 1, KERNEL, __VA_ARGS__)          \
    HPC_RLL_ROW_CASE(16, 1, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(16, 1, 2, KERNEL, __VA_ARGS__)        \
    HPC_RLL_ROW_CASE(16, 1, 4, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(16, 1, 8, KERNEL, __VA_ARGS__)        \
    HPC_RLL_ROW_CASE(64, 1, 4, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(64, 1, 8, KERNEL, __VA_ARGS__)        \
    HPC_RLL_ROW_CASE(1, 4, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(2, 4, 1, KERNEL, __VA_ARGS__)          \
    HPC_RLL_ROW_CASE(4, 4, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(8, 4, 1, KERNEL, __VA_ARGS__)          \
    HPC_RLL_ROW_CASE(16, 4, 1, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(16, 4, 2, KERNEL, __VA_ARGS__)        \
    HPC_RLL_ROW_CASE(16, 4, 4, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(16, 4, 8, KERNEL, __VA_ARGS__)        \
    HPC_RLL_ROW_CASE(64, 4, 4, KERNEL, __VA_ARGS__) HPC_RLL_ROW_CASE(64, 4, 8, KERNEL, __VA_ARGS__)        \
    return false;

bool launch_fwd(const RowCfg& cfg, hipStream_t st, const float* logits, const int64_t* action, float* logp,
                float* ent, long rows, int N) {
    if (ent) {
        HPC_RLL_ROW_DISPATCH(categorical_fwd_kernel, logits, action, logp, ent, rows, N)
    }
    HPC_RLL_ROW_DISPATCH(categorical_fwd_noent_kernel, logits, action, logp, ent, rows, N)
}
bool launch_bwd(const RowCfg& cfg, hipStream_t st, const float* logits, const int64_t* action, const float* c1,
                const float* g1, const float* c2, const float* g2, float* grad, long rows, int N) {
    HPC_RLL_ROW_DISPATCH(categorical_bwd_kernel, logits, action, c1, g1, c2, g2, grad, rows, N)
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define HPC_RLL_PPO_CASE(G_, V_, E_)                                                                                      \
    if (cfg.g == G_ && cfg.vec == V_ && cfg.e == E_) {                                                                    \
        const long per = (256 / G_) * 4;                                                                                  \
        long grid = (rows + per - 1) / per;                                                                               \
        const long gmax = kFoldMaxGrid;   /* (2048 / 4096 workgroups + a finalize launch: slower, round 4; 1024-thread */ \
                                          /*  workgroups instead of looping 256-thread ones: 22 -> 27.5 us, round 5)   */ \
        if (grid > gmax) grid = gmax;                                                                                     \
        const ScanFold fold = make_fold(st, PpoOp::NACC, scales, out5, grid);                                             \
        hipLaunchKernelGGL((ppo_fwd_fused_kernel<G_, V_, E_>), dim3((unsigned)grid), dim3(256), 0, st, logits_new,        \
                           logits_old, action, op, rows, N, partials, fold);                                              \
        const hipError_t e = hipGetLastError();                                                                           \
        *rc = e == hipSuccess ? HPC_RLL_OK : (int)e;                                                                      \
        if (*rc == HPC_RLL_OK && !fold.out) *rc = finalize_sums(partials, (int)grid, PpoOp::NACC, scales, out5, st);      \
        return true;                                                                                                      \
    }

}  // namespace

int g_ppo_fused = 1;   // hpc_rll_tune_set key 32: PPO forward in ONE launch (at most 512 workgroups, sums folded in the launch); 0 = three launches

// Rows of at most 16 pieces per lane group (N <= 512 with 16-byte loads): the configurations of the row kernels that keep a
// row in one DPP row.  Anything else: false, the caller runs the three launches.
bool ppo_forward_fused(const float* logits_new, const float* logits_old, const int64_t* action, const PpoOp& op, long rows,
                       int N, float* partials, const float* scales, float* out5, hipStream_t st, int* rc) {
    if (rows <= 0 || N <= 0) return false;
    if ((N % 4) != 0 && N <= kSmallMaxN) return false;          // (the small-N forward has its own kernel)
    const RowCfg cfg = row_cfg(N, al16(logits_new) && al16(logits_old));
    if (cfg.e > 2 || cfg.g > 16) return false;                  // registers: two heads x R rows x E pieces per lane
    HPC_RLL_PPO_CASE(1, 4, 1) HPC_RLL_PPO_CASE(2, 4, 1) HPC_RLL_PPO_CASE(4, 4, 1) HPC_RLL_PPO_CASE(8, 4, 1)
    HPC_RLL_PPO_CASE(16, 4, 1) HPC_RLL_PPO_CASE(16, 4, 2)
    HPC_RLL_PPO_CASE(1, 1, 1) HPC_RLL_PPO_CASE(2, 1, 1) HPC_RLL_PPO_CASE(4, 1, 1) HPC_RLL_PPO_CASE(8, 1, 1)
    HPC_RLL_PPO_CASE(16, 1, 1) HPC_RLL_PPO_CASE(16, 1, 2)
    return false;
}

// Internal C++ entry points used by vtrace.hip / upgo.hip / ppo.hip (same library).
int categorical_forward(const float* logits, const int64_t* action, float* logp, float* ent, long rows, int N,
                        hipStream_t st) {
    if (rows < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (rows == 0) return HPC_RLL_OK;
    if (!logits || !action || !logp) return HPC_RLL_EINVAL;
    if ((N % 4) != 0 && N <= kSmallMaxN && al16(logits)) {
        hipLaunchKernelGGL(categorical_small_kernel<false>, dim3(grid_for(rows, kSmallRows)), dim3(256),
                           (size_t)kSmallRows * N * sizeof(float), st, logits,
                           action, logp, ent, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (float*)nullptr, rows, N);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? HPC_RLL_OK : (int)e;
    }
    const RowCfg cfg = row_cfg(N, al16(logits));
    if (cfg.e > 8 && cfg.vec == 4 && N <= 16384) {
        const dim3 grid(grid_for(rows, 1));
#define HPC_RLL_BLOCKROW(E_)                                                                                          \
        hipLaunchKernelGGL((categorical_blockrow_kernel<E_, false>), grid, dim3(256), 0, st, logits, action, logp, ent,   \
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,   \
                           (float*)nullptr, rows, N)
        if (N <= 4096) HPC_RLL_BLOCKROW(4); else if (N <= 8192) HPC_RLL_BLOCKROW(8); else HPC_RLL_BLOCKROW(16);
#undef HPC_RLL_BLOCKROW
    } else if (cfg.e > 8 && N <= 16384) {
        hipLaunchKernelGGL(categorical_ldsrow_kernel<false>, dim3(grid_for(rows, 1)), dim3(256), (size_t)N * sizeof(float),
                           st, logits, action, logp, ent, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (float*)nullptr, rows, N);
    } else if (cfg.e > 8 || !launch_fwd(cfg, st, logits, action, logp, ent, rows, N)) {
        hipLaunchKernelGGL(categorical_fwd_long_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, st, logits, action,
                           logp, ent, rows, N);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

int categorical_backward(const float* logits, const int64_t* action, const float* c1, const float* g1,
                         const float* c2, const float* g2, float* grad, long rows, int N, hipStream_t st) {
    if (rows < 0 || N <= 0) return HPC_RLL_EINVAL;
    if (rows == 0) return HPC_RLL_OK;
    if (!logits || !action || !c1 || !grad) return HPC_RLL_EINVAL;
    if ((N % 4) != 0 && N <= kSmallMaxN && al16(logits) && al16(grad)) {
        hipLaunchKernelGGL(categorical_small_kernel<true>, dim3(grid_for(rows, kSmallRows)), dim3(256),
                           (size_t)kSmallRows * N * sizeof(float), st, logits,
                           action, (float*)nullptr, (float*)nullptr, c1, g1, c2, g2, grad, rows, N);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? HPC_RLL_OK : (int)e;
    }
    const RowCfg cfg = row_cfg(N, al16(logits) && al16(grad));
    if (cfg.e > 8 && cfg.vec == 4 && N <= 16384) {
        const dim3 grid(grid_for(rows, 1));
#define HPC_RLL_BLOCKROW(E_)                                                                                          \
        hipLaunchKernelGGL((categorical_blockrow_kernel<E_, true>), grid, dim3(256), 0, st, logits, action,              \
                           (float*)nullptr, (float*)nullptr, c1, g1, c2, g2, grad, rows, N)
        if (N <= 4096) HPC_RLL_BLOCKROW(4); else if (N <= 8192) HPC_RLL_BLOCKROW(8); else HPC_RLL_BLOCKROW(16);
#undef HPC_RLL_BLOCKROW
    } else if (cfg.e > 8 && N <= 16384) {
        hipLaunchKernelGGL(categorical_ldsrow_kernel<true>, dim3(grid_for(rows, 1)), dim3(256), (size_t)N * sizeof(float),
                           st, logits, action, (float*)nullptr, (float*)nullptr, c1, g1, c2, g2, grad, rows, N);
    } else if (cfg.e > 8 || !launch_bwd(cfg, st, logits, action, c1, g1, c2, g2, grad, rows, N)) {
        hipLaunchKernelGGL(categorical_bwd_long_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, st, logits, action,
                           c1, g1, c2, g2, grad, rows, N);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}

}  // namespace hpc_rll

extern "C" int hpc_rll_categorical_forward(const float* logits, const int64_t* action, float* logp, float* entropy,
                                           int64_t rows, int N, void* stream) {
    return hpc_rll::categorical_forward(logits, action, logp, entropy, (long)rows, N, (hipStream_t)stream);
}

extern "C" int hpc_rll_categorical_backward(const float* logits, const int64_t* action, const float* coef_logp,
                                            const float* g_logp, const float* coef_ent, const float* g_ent,
                                            float* grad_logits, int64_t rows, int N, void* stream) {
    return hpc_rll::categorical_backward(logits, action, coef_logp, g_logp, coef_ent, g_ent, grad_logits, (long)rows,
                                         N, (hipStream_t)stream);
}
