// scan_ops.hip -- TD(lambda), V-trace and UPGO on the generic reverse column scan (colscan.hpp) for gfx950.
//
// Replaces (all under /root/reference):
//   TdLambdaForward/Backward  src/rl_utils/td_lambda.cu:8-52,  td_lambda_kernel.h:11-51
//   VTraceForward/Backward    src/rl_utils/vtrace.cu:8-130,    vtrace_kernel.h:11-273
//   UpgoForward/Backward      src/rl_utils/upgo.cu:8-70,       upgo_kernel.h:11-108
// Semantics follow hpc_rll/origin/{td.py:148-244, vtrace.py:5-111, upgo.py:7-70}.
//
// Structure of every op:   [categorical row kernel(s)]  ->  column scan (+ per-workgroup loss partials, + the
// per-sample "unit gradient" coefficients for backward)  ->  finalize (fixed-order sum * scale).
// Backward = upstream scalar x saved per-sample coefficient, with the softmax RECOMPUTED from the logits
// (categorical.hip) instead of three saved (T,B,N) buffers.
#include <hip/hip_runtime.h>

#include <mutex>
#include <unordered_map>

#include "colscan.hpp"
#include "hpc_rll_hip.h"

namespace hpc_rll {

int g_scan_fold = 1;   // fold the loss finalisation into the launch (tune key 21) for grids up to kFoldMaxGrid; 0 = launch + finalize kernel

// Arrival tickets of the folded finalisation (colscan.hpp: ScanFold).  Zero at module load; every launch leaves its
// ticket at zero.  A ticket must never be shared by two launches that can run concurrently:
//   * eager launches: one ticket per (device, stream) -- launches on one stream run one after the other;
//   * launches recorded into a hipGraph (a captured kernel keeps its ticket address, and graphs can be replayed on
//     any stream): a ticket of their own each, never handed out again.
// When a pool is exhausted the caller gets nullptr and runs scan + finalize as two launches.
constexpr int kStreamTickets = 1024, kGraphTickets = 3072;
__device__ unsigned g_scan_tickets[kStreamTickets + kGraphTickets];

int categorical_forward(const float* logits, const int64_t* action, float* logp, float* ent, long rows, int N,
                        hipStream_t st);
int categorical_backward(const float* logits, const int64_t* action, const float* c1, const float* g1,
                         const float* c2, const float* g2, float* grad, long rows, int N, hipStream_t st);

namespace {

inline int last_error() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? HPC_RLL_OK : (int)e;
}
inline bool al8(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

struct TicketPool {
    std::mutex mu;
    struct Dev {
        unsigned* base = nullptr;
        bool tried = false;
        int next_stream = 0, next_graph = 0;
        std::unordered_map<hipStream_t, int> by_stream;
    } dev[64];
};
inline unsigned* scan_ticket(hipStream_t st) {
    if (!g_scan_fold) return nullptr;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    static TicketPool pool;
    std::lock_guard<std::mutex> lk(pool.mu);
    TicketPool::Dev& dv = pool.dev[d];
    if (!dv.base && cs != hipStreamCaptureStatusNone) return nullptr;   // no symbol lookup (a module load) inside a capture
    if (!dv.tried) {
        dv.tried = true;
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_scan_tickets)) == hipSuccess) dv.base = (unsigned*)p;
        else (void)hipGetLastError();
    }
    if (!dv.base) return nullptr;
    int idx = -1;
    if (cs != hipStreamCaptureStatusNone) {
        if (dv.next_graph < kGraphTickets) idx = kStreamTickets + dv.next_graph++;
    } else {
        auto it = dv.by_stream.find(st);
        if (it != dv.by_stream.end()) idx = it->second;
        else if (dv.next_stream < kStreamTickets) {
            dv.by_stream.emplace(st, dv.next_stream);
            idx = dv.next_stream++;
        }
    }
    if (idx < 0) return nullptr;
    return dv.base + idx;
}
// scan launch + finalisation of its NACC sums into `out` (x scale[k]): one launch when a ticket is available
template <class Op, bool ALLOW_V2, bool ALLOW_LC16 = false>
inline int scan_and_finalize(const Op& op, const ScanCfg& c, int T, int B, float* partials, int nacc, const float* scale,
                             float* out, hipStream_t st) {
    const ScanFold fold = make_fold(st, nacc, scale, out, (long)scan_grid(c, B));
    launch_colscan<Op, ALLOW_V2, ALLOW_LC16>(op, c, T, B, partials, st, fold);
    const int rc = last_error();
    if (rc || fold.out) return rc;
    return finalize_sums(partials, (int)scan_grid(c, B), nacc, scale, out, st);
}

template <int V> __device__ __forceinline__ Pack<V> ldz(const float* p, bool ok) {
    if (ok) return load_pack<V>(p);
    Pack<V> z;
#pragma unroll
    for (int k = 0; k < V; ++k) z.v[k] = 0.f;
    return z;
}

// ================================================================================================
// TD(lambda):  ret_t = r_t + (gamma - gamma*lambda) * V_{t+1} + gamma*lambda * ret_{t+1},  ret_T := V_T
//              loss = 0.5 * scale * sum w (ret - V_t)^2 ;  dloss/dV_t = w (V_t - ret_t) * scale   (t < T)
// ================================================================================================
struct TdLambdaOp {
    static constexpr int NACC = 1;
    const float* value; const float* reward; const float* weight; int weight_mode;  // 0 none, 1 (B,), 2 (T,B)
    float* grad_buf; int T, B; float disc, rest, scale;                            // disc = gamma*lambda
    template <int V> struct Row { Pack<V> v0, v1, r, w; };

    template <int V> __device__ void init(long col, bool ok, float (&carry)[V]) const {
        const Pack<V> vt = ldz<V>(value + (size_t)T * B + col, ok);
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = vt.v[k];
    }
    template <int V> __device__ void load(Row<V>& row, int t, long col, bool ok, bool next_in_regs) const {
        row.v0 = ldz<V>(value + (size_t)t * B + col, ok);
        if (!next_in_regs) row.v1 = ldz<V>(value + (size_t)(t + 1) * B + col, ok);
        row.r = ldz<V>(reward + (size_t)t * B + col, ok);
        if (weight_mode == 2) row.w = ldz<V>(weight + (size_t)t * B + col, ok);
        else if (weight_mode == 1) row.w = ldz<V>(weight + col, ok);
        else {
#pragma unroll
            for (int k = 0; k < V; ++k) row.w.v[k] = 1.f;
        }
    }
    template <int V> __device__ void link(Row<V>& row, const Row<V>& nxt) const { row.v1 = nxt.v0; }
    template <int V> __device__ void coeffs(const Row<V>& row, int, float (&a)[V], float (&b)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) { a[k] = disc; b[k] = fmaf(rest, row.v1.v[k], row.r.v[k]); }
    }
    template <int V> __device__ void finish(const Row<V>& row, int t, long col, bool ok, const float (&s)[V],
                                            const float (&)[V], float (&acc)[NACC]) const {
        if (!ok) return;
        Pack<V> g;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = s[k] - row.v0.v[k];
            acc[0] = fmaf(row.w.v[k] * d, d, acc[0]);
            g.v[k] = -row.w.v[k] * d * scale;
        }
        store_pack<V, true>(grad_buf + (size_t)t * B + col, g);
    }
};

// ================================================================================================
// V-trace (origin/vtrace.py:63-79):
//   IS = exp(logp_t - logp_b); item_t = min(IS,rho)*(r + g V_{t+1} - V_t) + g*lam*min(IS,c) * item_{t+1}
//   vs_t = V_t + item_t (vs_T = V_T);  adv_t = min(IS,rho_pg) * (r + g vs_{t+1} - V_t)
//   pg = -scale sum logp_t adv w ; value = scale sum w (V_t - vs_t)^2 ; ent = scale sum w H
// saved for backward: coef_pg = -w adv scale, coef_ent = w scale, gv_unit = 2 w (V_t - vs_t) scale
// ================================================================================================
struct VtraceOp {
    static constexpr int NACC = 3;
    const float* value; const float* reward; const float* weight; const float* logp_t; const float* logp_b;
    const float* ent; float* coef_pg; float* coef_ent; float* gv_unit; int T, B;
    float gamma, disc, rho_clip, c_clip, pg_clip, scale;
    template <int V> struct Row { Pack<V> v0, v1, r, w, is, lp, h; };

    template <int V> __device__ void init(long, bool, float (&carry)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = 0.f;
    }
    template <int V> __device__ void link(Row<V>& row, const Row<V>& nxt) const { row.v1 = nxt.v0; }
    template <int V> __device__ void load(Row<V>& row, int t, long col, bool ok, bool next_in_regs) const {
        const size_t o = (size_t)t * B + col;
        row.v0 = ldz<V>(value + o, ok);
        if (!next_in_regs) row.v1 = ldz<V>(value + o + B, ok);
        row.r = ldz<V>(reward + o, ok);
        row.lp = ldz<V>(logp_t + o, ok);
        row.h = ldz<V>(ent + o, ok);
        const Pack<V> lb = ldz<V>(logp_b + o, ok);
        if (weight) row.w = ldz<V>(weight + o, ok);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            row.is.v[k] = expf(row.lp.v[k] - lb.v[k]);
            if (!weight) row.w.v[k] = 1.f;
        }
    }
    template <int V> __device__ void coeffs(const Row<V>& row, int, float (&a)[V], float (&b)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            a[k] = disc * fminf(row.is.v[k], c_clip);
            b[k] = fminf(row.is.v[k], rho_clip) * (fmaf(gamma, row.v1.v[k], row.r.v[k]) - row.v0.v[k]);
        }
    }
    template <int V> __device__ void finish(const Row<V>& row, int t, long col, bool ok, const float (&s)[V],
                                            const float (&s_next)[V], float (&acc)[NACC]) const {
        if (!ok) return;
        Pack<V> cp, ce, gv;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float w = row.w.v[k];
            const float vs_next = row.v1.v[k] + s_next[k];
            const float adv = fminf(row.is.v[k], pg_clip) * (fmaf(gamma, vs_next, row.r.v[k]) - row.v0.v[k]);
            acc[0] -= row.lp.v[k] * adv * w;
            acc[1] = fmaf(w * s[k], s[k], acc[1]);   // (V_t - vs_t)^2 = item_t^2
            acc[2] = fmaf(w, row.h.v[k], acc[2]);
            cp.v[k] = -w * adv * scale;
            ce.v[k] = w * scale;
            gv.v[k] = -2.f * w * s[k] * scale;       // 2 w (V_t - vs_t) scale
        }
        const size_t o = (size_t)t * B + col;
        store_pack<V, true>(coef_pg + o, cp);
        store_pack<V, true>(coef_ent + o, ce);
        store_pack<V, true>(gv_unit + o, gv);
    }
};

// ================================================================================================
// UPGO (origin/upgo.py:36-38,64-70): ret_t = r_t + (lam_t ? ret_{t+1} : V_{t+1}), ret_T := V_T,
//   lam_t = [r_{t+1} + V_{t+2} >= V_{t+1}] for t < T-1, lam_{T-1} = 1.
//   loss = -scale sum rho (ret - V_t) logp ;  saved coef = -rho (ret - V_t) scale
// ================================================================================================
struct UpgoOp {
    static constexpr int NACC = 1;
    const float* value; const float* reward; const float* rho; const float* logp; float* coef; int T, B; float scale;
    template <int V> struct Row { Pack<V> v0, v1, r, rho, lp; float lam[V]; };

    template <int V> __device__ void init(long col, bool ok, float (&carry)[V]) const {
        const Pack<V> vt = ldz<V>(value + (size_t)T * B + col, ok);
#pragma unroll
        for (int k = 0; k < V; ++k) carry[k] = vt.v[k];
    }
    // row t+1 in registers: V_{t+1} = its v0, r_{t+1} = its r, V_{t+2} = its v1 (already linked: rows are linked from the
    // end of the chunk backwards); t+1 <= T-1 there, so lam_t is the comparison
    template <int V> __device__ void link(Row<V>& row, const Row<V>& nxt) const {
        row.v1 = nxt.v0;
#pragma unroll
        for (int k = 0; k < V; ++k) row.lam[k] = (nxt.r.v[k] + nxt.v1.v[k] >= row.v1.v[k]) ? 1.f : 0.f;
    }
    template <int V> __device__ void load(Row<V>& row, int t, long col, bool ok, bool next_in_regs) const {
        const size_t o = (size_t)t * B + col;
        row.v0 = ldz<V>(value + o, ok);
        row.r = ldz<V>(reward + o, ok);
        row.rho = ldz<V>(rho + o, ok);
        row.lp = ldz<V>(logp + o, ok);
        if (next_in_regs) return;
        row.v1 = ldz<V>(value + o + B, ok);
        if (t < T - 1) {
            const Pack<V> r1 = ldz<V>(reward + o + B, ok);
            const Pack<V> v2 = ldz<V>(value + o + 2 * (size_t)B, ok);
#pragma unroll
            for (int k = 0; k < V; ++k) row.lam[k] = (r1.v[k] + v2.v[k] >= row.v1.v[k]) ? 1.f : 0.f;
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) row.lam[k] = 1.f;
        }
    }
    template <int V> __device__ void coeffs(const Row<V>& row, int, float (&a)[V], float (&b)[V]) const {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            a[k] = row.lam[k];
            b[k] = row.r.v[k] + (1.f - row.lam[k]) * row.v1.v[k];
        }
    }
    template <int V> __device__ void finish(const Row<V>& row, int t, long col, bool ok, const float (&s)[V],
                                            const float (&)[V], float (&acc)[NACC]) const {
        if (!ok) return;
        Pack<V> c;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float adv = row.rho.v[k] * (s[k] - row.v0.v[k]);
            acc[0] -= adv * row.lp.v[k];
            c.v[k] = -adv * scale;
        }
        store_pack<V, true>(coef + (size_t)t * B + col, c);
    }
};

}  // namespace

ScanFold make_fold(hipStream_t st, int nacc, const float* scale, float* out, long grid) {
    ScanFold fold{nullptr, nullptr, {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if (nacc < 1 || nacc > 8 || grid > kFoldMaxGrid) return fold;   // larger grids: partials + the finalize launch
    fold.ticket = scan_ticket(st);
    if (!fold.ticket) return fold;
    fold.out = out;
    for (int k = 0; k < nacc; ++k) fold.scale[k] = scale[k];
    return fold;
}

}  // namespace hpc_rll

using namespace hpc_rll;

// ------------------------------------------------------------------------------------------------ TD(lambda)
extern "C" int hpc_rll_td_lambda_forward(const float* value, const float* reward, const float* weight,
                                         int weight_mode, float* loss, float* grad_buf, float* partials, int T,
                                         int B, float gamma, float lambda, float scale, void* stream) {
    if (T < 0 || B < 0 || weight_mode < 0 || weight_mode > 2) return HPC_RLL_EINVAL;
    if (!loss) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (T == 0 || B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!value || !reward || !grad_buf || !partials || (weight_mode != 0 && !weight)) return HPC_RLL_EINVAL;
    const bool v2 = (B % 2 == 0) && al8(value) && al8(reward) && al8(weight) && al8(grad_buf);
    const ScanCfg c = scan_cfg(T, B, v2, true);
    // oracle arithmetic (origin/td.py:239-243): discounts = gamma*lambda ; (gammas - discounts) * V_{t+1}
    const float disc = gamma * lambda;
    TdLambdaOp op{value, reward, weight, weight_mode, grad_buf, T, B, disc, gamma - disc, scale};
    const float sc = 0.5f * scale;
    return scan_and_finalize<TdLambdaOp, true, true>(op, c, T, B, partials, 1, &sc, loss, st);
}

extern "C" int hpc_rll_td_lambda_backward(const float* grad_loss, const float* grad_buf, float* grad_value, int T,
                                          int B, void* stream) {
    if (T < 0 || B < 0) return HPC_RLL_EINVAL;
    return scale_rows(grad_loss, grad_buf, grad_value, (long)T * B, (long)(T + 1) * B, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ V-trace
// ws layout (floats): [coef_pg TB | coef_ent TB | gv_unit TB | logp_t TB | ent TB | logp_b TB | partials]
extern "C" int64_t hpc_rll_vtrace_workspace_floats(int T, int B) {
    return 6 * (int64_t)T * B + 8 * (((int64_t)B + 15) / 16 + 1);   // partials: 3 sums x up to ceil(B/8) workgroups
}

extern "C" int hpc_rll_vtrace_forward(const float* target_output, const float* behaviour_output,
                                      const int64_t* action, const float* value, const float* reward,
                                      const float* weight, float* losses, float* ws, int T, int B, int N,
                                      float gamma, float lambda, float rho_clip, float c_clip, float rho_pg_clip,
                                      float scale, void* stream) {
    if (T < 0 || B < 0 || N <= 0 || !losses) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (T == 0 || B == 0) return (int)hipMemsetAsync(losses, 0, 3 * sizeof(float), st);
    if (!target_output || !behaviour_output || !action || !value || !reward || !ws) return HPC_RLL_EINVAL;
    const size_t TB = (size_t)T * B;
    float *coef_pg = ws, *coef_ent = ws + TB, *gv_unit = ws + 2 * TB, *logp_t = ws + 3 * TB, *ent = ws + 4 * TB,
          *logp_b = ws + 5 * TB, *partials = ws + 6 * TB;
    int rc = categorical_forward(target_output, action, logp_t, ent, (long)TB, N, st);
    if (rc) return rc;
    rc = categorical_forward(behaviour_output, action, logp_b, nullptr, (long)TB, N, st);
    if (rc) return rc;
    const ScanCfg c = scan_cfg(T, B, false);  // V=1: the 7-array row payload would spill at V=2
    VtraceOp op{value, reward, weight, logp_t, logp_b, ent, coef_pg, coef_ent, gv_unit, T, B,
                gamma, gamma * lambda, rho_clip, c_clip, rho_pg_clip, scale};
    const float sc[3] = {scale, scale, scale};
    return scan_and_finalize<VtraceOp, false>(op, c, T, B, partials, 3, sc, losses, st);
}

extern "C" int hpc_rll_vtrace_backward(const float* g_pg, const float* g_value, const float* g_ent,
                                       const float* target_output, const int64_t* action, const float* ws,
                                       float* grad_target_output, float* grad_value, int T, int B, int N,
                                       void* stream) {
    if (T < 0 || B < 0 || N <= 0) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t TB = (size_t)T * B;
    int rc = HPC_RLL_OK;
    if (grad_value) {
        if (!g_value || (TB && !ws)) return HPC_RLL_EINVAL;
        rc = scale_rows(g_value, ws + 2 * TB, grad_value, (long)TB, (long)TB + B, st);
        if (rc) return rc;
    }
    if (grad_target_output && TB) {
        if (!target_output || !action || !ws) return HPC_RLL_EINVAL;
        rc = categorical_backward(target_output, action, ws, g_pg, ws + TB, g_ent, grad_target_output, (long)TB, N,
                                  st);
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------ UPGO
// ws layout (floats): [coef TB | logp TB | partials]
extern "C" int64_t hpc_rll_upgo_workspace_floats(int T, int B) {
    return 2 * (int64_t)T * B + 8 * (((int64_t)B + 63) / 64 + 1);
}

extern "C" int hpc_rll_upgo_forward(const float* target_output, const float* rho, const int64_t* action,
                                    const float* reward, const float* value, float* loss, float* ws, int T, int B,
                                    int N, float scale, void* stream) {
    if (T < 0 || B < 0 || N <= 0 || !loss) return HPC_RLL_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (T == 0 || B == 0) return (int)hipMemsetAsync(loss, 0, sizeof(float), st);
    if (!target_output || !rho || !action || !reward || !value || !ws) return HPC_RLL_EINVAL;
    const size_t TB = (size_t)T * B;
    float *coef = ws, *logp = ws + TB, *partials = ws + 2 * TB;
    int rc = categorical_forward(target_output, action, logp, nullptr, (long)TB, N, st);
    if (rc) return rc;
    const ScanCfg c = scan_cfg(T, B, false);  // V=1 (register budget, see VtraceOp)
    UpgoOp op{value, reward, rho, logp, coef, T, B, scale};
    return scan_and_finalize<UpgoOp, false>(op, c, T, B, partials, 1, &scale, loss, st);
}

extern "C" int hpc_rll_upgo_backward(const float* g, const float* target_output, const int64_t* action,
                                     const float* ws, float* grad_target_output, int T, int B, int N,
                                     void* stream) {
    if (T < 0 || B < 0 || N <= 0) return HPC_RLL_EINVAL;
    if ((size_t)T * B == 0) return HPC_RLL_OK;
    if (!g || !target_output || !action || !ws || !grad_target_output) return HPC_RLL_EINVAL;
    return categorical_backward(target_output, action, ws, g, nullptr, nullptr, grad_target_output, (long)T * B, N,
                                (hipStream_t)stream);
}
