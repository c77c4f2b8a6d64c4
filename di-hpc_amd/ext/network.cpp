// network.cpp -- the compiled `hpc_torch_utils_network` extension module
// (reference: src/torch_utils/network/entry.cpp:8-13, include/hpc/rll/cuda/torch_utils/network/entry.h:11-29).
//
// L2 functions LstmForward/Backward, ScatterConnectionForward/Backward with the reference's names and list convention
// (native short lists AND the reference's positional lists, told apart by their lengths), plus the fused autograd ops
// `lstm` and `scatter_connection` that hpc_rll.torch_utils.network.* call.  Host-only C++ over the C ABI.
#include "common.hpp"

#include <atomic>

namespace hpc_rll_ext {
namespace {

// ========================================================================================================== LSTM
struct LstmDims { int64_t S, B, I, H, L; at::Device dev; };

LstmDims lstm_dims(const Tensor& x, const Tensor& h0, const Tensor& wx, const Tensor& wh) {
    req(x, "x");
    TORCH_CHECK(x.dim() == 3, "x: expected (S,B,I), got ", x.sizes());
    const at::Device dev = x.device();
    req(h0, "h0", dev);
    TORCH_CHECK(h0.dim() == 3 && h0.size(1) == x.size(1), "h0: expected (L,", x.size(1), ",H), got ", h0.sizes());
    const int64_t S = x.size(0), B = x.size(1), I = x.size(2), L = h0.size(0), H = h0.size(2), G = 4 * H;
    req(wx, "wx", dev);
    req(wh, "wh", dev);
    TORCH_CHECK(wx.numel() == (I + (L - 1) * H) * G && wh.numel() == L * H * G, "wx/wh: ", wx.numel(), "/", wh.numel(),
                " elements do not match I=", I, " H=", H, " L=", L);
    return {S, B, I, H, L, dev};
}

int64_t lstm_ws_floats(const LstmDims& d, double dropout) {
    const int64_t n = hpc_rll_lstm_workspace_floats(to_int(d.S, "S"), to_int(d.B, "B"), to_int(d.I, "I"), to_int(d.H, "H"),
                                                    to_int(d.L, "L"), (float)dropout);
    TORCH_CHECK(n >= 0, "lstm_workspace: invalid sizes");
    return n;
}

void lstm_check_params(const LstmDims& d, const Tensor& c0, const Tensor& bias, const Tensor& gamma, const Tensor& beta) {
    const int64_t G = 4 * d.H;
    req(c0, "c0", d.dev, {d.L, d.B, d.H});
    req(bias, "bias", d.dev);
    req(gamma, "ln_gamma", d.dev);
    req(beta, "ln_beta", d.dev);
    TORCH_CHECK(bias.numel() == d.L * G && gamma.numel() == d.L * 2 * G && beta.numel() == d.L * 2 * G,
                "bias / ln_gamma / ln_beta: wrong number of elements");
}

// The persistent small-batch kernels (B <= 4) report a co-residency timeout ASYNCHRONOUSLY (a pinned status word,
// include/hpc_rll_hip.h): the launch that timed out has already returned OK.  What the extension does about it
// (ADVICE r02): the first LSTM entry point that sees the status clears it -- which retires the persistent path for the
// rest of the process -- warns that the LSTM results produced since the last synchronisation are invalid, bumps an
// epoch and RE-RUNS itself on the step kernels; a backward whose forward ran on the persistent path in an older epoch
// refuses to run (its saved activations may be those of a launch that gave up).  hpc_rll.torch_utils.network.rnn.LSTM
// (check_persistent=True) closes the remaining window by synchronising after every persistent-path forward.
std::atomic<int64_t> g_persist_epoch{0};
constexpr int64_t kPersistMaxB = 4;

bool lstm_recover_from_timeout(const char* what) {
    if (hpc_rll_async_error() != HPC_RLL_ETIMEOUT) return false;
    check(hpc_rll_clear_async_error(), "hpc_rll_clear_async_error");
    g_persist_epoch.fetch_add(1);
    TORCH_WARN("hpc_rll LSTM: a persistent small-batch kernel gave up waiting for its co-resident workgroups (another process "
               "is holding the GPU's compute units).  LSTM results produced by this process since its last synchronisation "
               "are INVALID; ", what, " is re-run on the step kernels, which are used from now on "
               "(HPC_RLL_LSTM_PERSIST=0 selects them from the start on GPUs shared between processes).");
    return true;
}

// y_hseq: y is also the last layer's saved h sequence (hpc_rll_lstm_forward_y / _backward_y): written by the cells, no copy
// Returns the recurrence path THIS call ran on (hpc_rll_lstm_last_forward_path codes).  The C ABI records it per host
// thread inside the call (csrc/lstm.hip: t_lstm_last_path), and it is read here before this thread can run anything
// else -- another thread's forward cannot be mistaken for this one.
int lstm_forward_launch(const LstmDims& d, const Tensor& x, const Tensor& h0, const Tensor& c0, const Tensor& wx,
                         const Tensor& wh, const Tensor& bias, const Tensor& gamma, const Tensor& beta, const Tensor& y,
                         const Tensor& hn, const Tensor& cn, const Tensor& ws, double dropout, uint64_t seed,
                         bool y_hseq = false) {
    for (int attempt = 0; attempt < 2; ++attempt) {
        const int rc = (y_hseq ? hpc_rll_lstm_forward_y : hpc_rll_lstm_forward)(
            fptr(x), fptr(h0), fptr(c0), fptr(wx), fptr(wh), fptr(bias), fptr(gamma), fptr(beta), fmut(y), fmut(hn), fmut(cn),
            fmut(ws), (int)d.S, (int)d.B, (int)d.I, (int)d.H, (int)d.L, (float)dropout, seed, stream_of(d.dev));
        if (rc == HPC_RLL_ETIMEOUT && attempt == 0 && lstm_recover_from_timeout("this forward")) continue;
        check(rc, "hpc_rll_lstm_forward");
        return hpc_rll_lstm_last_forward_path();
    }
    return -1;
}

struct LstmGrads { Tensor dx, dh0, dc0, dwx, dwh, dbias, dgamma, dbeta; };

void lstm_backward_launch(const LstmDims& d, const Tensor& dy, const Tensor& dhn, const Tensor& dcn, const Tensor& x,
                          const Tensor& h0, const Tensor& c0, const Tensor& wx, const Tensor& wh, const Tensor& gamma,
                          const Tensor& ws, const LstmGrads& g, double dropout, uint64_t seed, int64_t fwd_epoch = -1,
                          const Tensor& y_hseq = Tensor(), bool fwd_persistent = true) {
    // a timeout seen now may be this graph's own forward: its saved activations cannot be trusted.  (fwd_persistent: the
    // forward ran on kernels that can time out at all -- the small-batch persistent / wavefront kernels or, round 4, the
    // large-batch row-block kernel; callers that do not know pass true and the B <= 4 rule of round 2 decides)
    const bool recovered = lstm_recover_from_timeout("nothing");
    const bool could_time_out = fwd_epoch >= 0 ? fwd_persistent : d.B <= kPersistMaxB;
    TORCH_CHECK(!(could_time_out && (recovered || (fwd_epoch >= 0 && fwd_epoch != g_persist_epoch.load()))),
                "hpc_rll LSTM backward: the forward pass of this graph ran on a persistent kernel around the time "
                "one of them timed out; its saved activations may be invalid.  Run the forward pass again (it now uses the "
                "step kernels).");
    if (y_hseq.defined()) {
        req(y_hseq, "y", d.dev, {d.S, d.B, d.H});
        check(hpc_rll_lstm_backward_y(fptr(dy), fptr(dhn), fptr(dcn), fptr(x), fptr(h0), fptr(c0), fptr(wx), fptr(wh),
                                      fptr(gamma), fptr(y_hseq), fmut(ws), fmut(g.dx), fmut(g.dh0), fmut(g.dc0), fmut(g.dwx),
                                      fmut(g.dwh), fmut(g.dbias), fmut(g.dgamma), fmut(g.dbeta), (int)d.S, (int)d.B, (int)d.I,
                                      (int)d.H, (int)d.L, (float)dropout, seed, stream_of(d.dev)),
              "hpc_rll_lstm_backward_y");
        return;
    }
    check(hpc_rll_lstm_backward(fptr(dy), fptr(dhn), fptr(dcn), fptr(x), fptr(h0), fptr(c0), fptr(wx), fptr(wh),
                                fptr(gamma), fmut(ws), fmut(g.dx), fmut(g.dh0), fmut(g.dc0), fmut(g.dwx), fmut(g.dwh),
                                fmut(g.dbias), fmut(g.dgamma), fmut(g.dbeta), (int)d.S, (int)d.B, (int)d.I, (int)d.H,
                                (int)d.L, (float)dropout, seed, stream_of(d.dev)),
          "hpc_rll_lstm_backward");
}

SavedByBuffer& saved() {
    static SavedByBuffer* s = new SavedByBuffer();
    return *s;
}
std::atomic<uint64_t> g_ref_seed{0x9E3779B97F4A7C15ull};

// native:    inputs = [x (S,B,I), h0 (L,B,H), c0 (L,B,H), wx (flat), wh (flat), bias (L*4H), ln_gamma (L,8H),
//            ln_beta (L,8H)]; outputs = [y (S,B,H), hn (L,B,H), cn (L,B,H), ws = lstm_workspace(...)]; (dropout, seed)
// reference: same inputs (rnn.py:16); outputs = [xbuf, hbuf, hn (S,L,B,H), cn (S,L,B,H), ifog, ym (L,S,B,H), ln_in,
//            ln_mean, ln_rstd, dropout_mask] (rnn.py:17): y is written to ym[L-1], the final states to hn[S-1] /
//            cn[S-1] -- exactly the views the reference L1 returns (rnn.py:27-31); the other scratch buffers are not
//            touched, the workspace is parked under `ifog`.  Reference: src/torch_utils/network/lstm.cu:29-186.
void LstmForward(const TensorList& in, const TensorList& out, double dropout, std::optional<uint64_t> seed_opt) {
    expect_len(in, 8, "LstmForward inputs");
    TORCH_CHECK(out.size() == 4 || out.size() == 10, "LstmForward outputs: expected 4 (native) or 10 (reference)");
    const LstmDims d = lstm_dims(in[0], in[1], in[3], in[4]);
    lstm_check_params(d, in[2], in[5], in[6], in[7]);
    c10::DeviceGuard g(d.dev);
    if (out.size() == 4) {
        req(out[0], "y", d.dev, {d.S, d.B, d.H});
        req(out[1], "hn", d.dev, {d.L, d.B, d.H});
        req(out[2], "cn", d.dev, {d.L, d.B, d.H});
        req(out[3], "ws", d.dev, {lstm_ws_floats(d, dropout)});
        lstm_forward_launch(d, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], out[0], out[1], out[2], out[3],
                            dropout, seed_opt.value_or(0));
        return;
    }
    const Tensor &hn_all = out[2], &cn_all = out[3], &ifog = out[4], &ym = out[5];
    req(hn_all, "hn", d.dev, {d.S, d.L, d.B, d.H});
    req(cn_all, "cn", d.dev, {d.S, d.L, d.B, d.H});
    req(ym, "ym", d.dev, {d.L, d.S, d.B, d.H});
    req(ifog, "ifog", d.dev);
    const uint64_t seed = dropout > 0.0 ? seed_opt.value_or(g_ref_seed.fetch_add(0x2545F4914F6CDD1Dull)) : 0;
    Tensor ws = new_f32({lstm_ws_floats(d, dropout)}, d.dev);
    lstm_forward_launch(d, in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], ym.select(0, d.L - 1),
                        hn_all.select(0, d.S - 1), cn_all.select(0, d.S - 1), ws, dropout, seed);
    SavedByBuffer::Entry e{{ws}};
    e.seed = seed;
    saved().put(ifog, std::move(e));
}

// native:    inputs = [dy (S,B,H)|None, dhn (L,B,H)|None, dcn (L,B,H)|None, x, h0, c0, wx, wh, ln_gamma, ws];
//            outputs = [dx|None, dh0, dc0, dwx, dwh, dbias, d_ln_gamma, d_ln_beta]; (dropout, seed)
// reference: inputs = [x, h0, c0, wx, wh, hn, cn, ifog, ym, ln_in, ln_mean, ln_rstd, ln_gamma, dropout_mask];
//            outputs = [dgate, xbuf, hbuf, dx, dwx, dwh, dbias, d_ln_gamma, d_ln_beta, dy, dh, dc] (rnn.py:20-21,35-42).
// The reference zeroes the incoming dhn/dcn (lstm.cu:309-310); here they are honoured in both conventions.
// dx = None skips the input-gradient product of layer 0 (x needs no grad).  Reference: lstm.cu:188-379.
void LstmBackward(const OptList& in, const OptList& out, double dropout, std::optional<uint64_t> seed_opt) {
    const bool native = in.size() == 10 && out.size() == 8;
    TORCH_CHECK(native || (in.size() == 14 && out.size() == 12),
                "LstmBackward: expected 10 inputs / 8 outputs (native) or 14 / 12 (reference), got ", in.size(), " / ",
                out.size());
    auto T = [](const OptTensor& t) { return has(t) ? *t : undef(); };
    Tensor dy, dhn, dcn, x, h0, c0, wx, wh, gamma, ws;
    LstmGrads gr;
    uint64_t seed = seed_opt.value_or(0);
    if (native) {
        dy = T(in[0]); dhn = T(in[1]); dcn = T(in[2]); x = T(in[3]); h0 = T(in[4]); c0 = T(in[5]); wx = T(in[6]);
        wh = T(in[7]); gamma = T(in[8]); ws = T(in[9]);
        gr = {T(out[0]), T(out[1]), T(out[2]), T(out[3]), T(out[4]), T(out[5]), T(out[6]), T(out[7])};
    } else {
        x = T(in[0]); h0 = T(in[1]); c0 = T(in[2]); wx = T(in[3]); wh = T(in[4]); gamma = T(in[12]);
        TORCH_CHECK(has(in[7]), "LstmBackward: ifog is None");
        auto e = saved().get(*in[7], "LstmBackward");
        ws = e.tensors[0];
        seed = e.seed;
        dy = T(out[9]); dhn = T(out[10]); dcn = T(out[11]);
        gr.dx = T(out[3]); gr.dwx = T(out[4]); gr.dwh = T(out[5]); gr.dbias = T(out[6]); gr.dgamma = T(out[7]);
        gr.dbeta = T(out[8]);
    }
    const LstmDims d = lstm_dims(x, h0, wx, wh);
    req(c0, "c0", d.dev, {d.L, d.B, d.H});
    req(gamma, "ln_gamma", d.dev);
    req(ws, "ws", d.dev, {lstm_ws_floats(d, dropout)});
    c10::DeviceGuard g(d.dev);
    if (!native) {   // the reference has no slots for dh0 / dc0
        gr.dh0 = at::empty_like(h0);
        gr.dc0 = at::empty_like(c0);
        if (dy.defined() && !dy.is_contiguous()) dy = dy.contiguous();
        if (dhn.defined() && !dhn.is_contiguous()) dhn = dhn.contiguous();
        if (dcn.defined() && !dcn.is_contiguous()) dcn = dcn.contiguous();
    }
    if (dy.defined()) req(dy, "dy", d.dev, {d.S, d.B, d.H});
    if (dhn.defined()) req(dhn, "dhn", d.dev, {d.L, d.B, d.H});
    if (dcn.defined()) req(dcn, "dcn", d.dev, {d.L, d.B, d.H});
    if (gr.dx.defined()) req(gr.dx, "dx", d.dev, {d.S, d.B, d.I});
    req(gr.dh0, "dh0", d.dev, {d.L, d.B, d.H});
    req(gr.dc0, "dc0", d.dev, {d.L, d.B, d.H});
    const std::pair<const Tensor*, const Tensor*> same[] = {{&gr.dwx, &wx}, {&gr.dwh, &wh}, {&gr.dgamma, &gamma},
                                                            {&gr.dbeta, &gamma}};
    const char* names[] = {"dwx", "dwh", "d_ln_gamma", "d_ln_beta"};
    for (int i = 0; i < 4; ++i) {
        req(*same[i].first, names[i], d.dev);
        TORCH_CHECK(same[i].first->numel() == same[i].second->numel(), names[i], ": ", same[i].first->numel(),
                    " elements, expected ", same[i].second->numel());
    }
    req(gr.dbias, "dbias", d.dev);
    TORCH_CHECK(gr.dbias.numel() == d.L * 4 * d.H, "dbias: wrong number of elements");
    lstm_backward_launch(d, dy, dhn, dcn, x, h0, c0, wx, wh, gamma, ws, gr, dropout, seed);
}

struct LstmFn : public ag::Function<LstmFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& x_in, const Tensor& wx_in, const Tensor& wh_in,
                                   const Tensor& bias_in, const Tensor& gamma_in, const Tensor& beta_in, const Tensor& h0_in,
                                   const Tensor& c0_in, double dropout, int64_t seed, bool y_saved) {
        const LstmDims d = lstm_dims(x_in, h0_in, wx_in, wh_in);
        lstm_check_params(d, c0_in, bias_in, gamma_in, beta_in);
        c10::DeviceGuard g(d.dev);
        // the large-batch kernels use 16-byte accesses everywhere (the C ABI returns HPC_RLL_EALIGN otherwise): an operand
        // that is contiguous but starts off a 16-byte boundary (a view into a larger buffer) is copied once
        auto al16 = [](const Tensor& t) { return (reinterpret_cast<uintptr_t>(t.data_ptr()) & 15) ? t.clone() : t; };
        const Tensor x = al16(x_in), wx = al16(wx_in), wh = al16(wh_in), bias = al16(bias_in), gamma = al16(gamma_in),
                     beta = al16(beta_in), h0 = al16(h0_in), c0 = al16(c0_in);
        Tensor hn = new_f32({d.L, d.B, d.H}, d.dev), cn = new_f32({d.L, d.B, d.H}, d.dev);
        Tensor ws = new_f32({lstm_ws_floats(d, dropout)}, d.dev);
        // y is its OWN (S,B,H) tensor in every mode (as a view of the saved workspace it pinned the whole
        // workspace for any holder of y.detach() and made in-place ops on y raise).  With a graph it is at the same time the
        // last layer's saved h sequence: the cells write it directly (no copy: 0.8 ms at C4), it is saved for the
        // backward like any op output that its own derivative needs (exp, sigmoid, ...) -- an in-place update of y is
        // allowed and caught by the version counter if (and only if) a backward through this node follows.
        Tensor y = new_f32({d.S, d.B, d.H}, d.dev);
        const bool y_hseq = y_saved && d.S > 0;
        const int fwd_path = lstm_forward_launch(d, x, h0, c0, wx, wh, bias, gamma, beta, y, hn, cn, ws, dropout, (uint64_t)seed, y_hseq);
        if (y_hseq) ctx->save_for_backward({x, h0, c0, wx, wh, gamma, ws, y});
        else ctx->save_for_backward({x, h0, c0, wx, wh, gamma, ws});
        ctx->saved_data["dropout"] = dropout;
        ctx->saved_data["seed"] = seed;
        ctx->saved_data["bias_shape"] = bias.sizes().vec();
        ctx->saved_data["beta_shape"] = beta.sizes().vec();
        ctx->saved_data["persist_epoch"] = g_persist_epoch.load();
        ctx->saved_data["persistent_fwd"] = fwd_path != 0 && fwd_path != 3;   // the path of THIS node's own launch
        return {y, hn, cn};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        const auto s = ctx->get_saved_variables();
        const Tensor &x = s[0], &h0 = s[1], &c0 = s[2], &wx = s[3], &wh = s[4], &gamma = s[5], &ws = s[6];
        const LstmDims d = lstm_dims(x, h0, wx, wh);
        c10::DeviceGuard g(d.dev);
        auto cont = [](const Tensor& t) {   // contiguous and 16-byte aligned (see forward)
            if (!t.defined()) return t;
            Tensor c = t.contiguous();
            return (reinterpret_cast<uintptr_t>(c.data_ptr()) & 15) ? c.clone() : c;
        };
        LstmGrads gr;
        gr.dx = ctx->needs_input_grad(0) ? at::empty_like(x) : undef();   // x without grad: layer-0 dx GEMM is skipped
        gr.dh0 = at::empty_like(h0);
        gr.dc0 = at::empty_like(c0);
        gr.dwx = at::empty_like(wx);
        gr.dwh = at::empty_like(wh);
        gr.dbias = new_f32(ctx->saved_data["bias_shape"].toIntVector(), d.dev);
        gr.dgamma = at::empty_like(gamma);
        gr.dbeta = new_f32(ctx->saved_data["beta_shape"].toIntVector(), d.dev);
        lstm_backward_launch(d, cont(grads[0]), cont(grads[1]), cont(grads[2]), x, h0, c0, wx, wh, gamma, ws, gr,
                             ctx->saved_data["dropout"].toDouble(), (uint64_t)ctx->saved_data["seed"].toInt(),
                             ctx->saved_data["persist_epoch"].toInt(), s.size() > 7 ? s[7] : Tensor(),
                             ctx->saved_data["persistent_fwd"].toBool());
        return {gr.dx, gr.dwx, gr.dwh, gr.dbias, gr.dgamma, gr.dbeta, gr.dh0, gr.dc0, undef(), undef(), undef()};
    }
};

// out (M,N) (+)= a (M,K) @ b (K,N) in exact fp32 on the matrix cores; a and b may be arbitrary 2-D strided views (e.g.
// w.t()), which is how the NN / NT / TN layouts of the LSTM are expressed.
Tensor gemm_f32(const Tensor& a, const Tensor& b, const OptTensor& out_opt, bool accumulate) {
    TORCH_CHECK(a.defined() && b.defined() && a.is_cuda() && b.is_cuda() && a.dim() == 2 && b.dim() == 2 &&
                a.scalar_type() == at::kFloat && b.scalar_type() == at::kFloat && a.size(1) == b.size(0) &&
                a.device() == b.device(), "gemm_f32: a (M,K), b (K,N) fp32 on one GPU expected");
    const int64_t M = a.size(0), K = a.size(1), N = b.size(1);
    const at::Device dev = a.device();
    c10::DeviceGuard g(dev);
    Tensor out = has(out_opt) ? *out_opt : new_f32({M, N}, dev);
    req(out, "out", dev, {M, N});
    check(hpc_rll_gemm_f32(fptr(a), fptr(b), fmut(out), to_int(M, "M"), to_int(N, "N"), to_int(K, "K"), a.stride(0),
                           a.stride(1), b.stride(0), b.stride(1), out.stride(0), accumulate ? 1 : 0, stream_of(dev)),
          "hpc_rll_gemm_f32");
    return out;
}

// ============================================================================================= ScatterConnection
struct ScatterDims { int64_t B, M, N, H, W; at::Device dev; };

ScatterDims scatter_check(const Tensor& x, const Tensor& location, int64_t H, int64_t W) {
    req(x, "x");
    TORCH_CHECK(x.dim() == 3, "x: expected (B,M,N), got ", x.sizes());
    const int64_t B = x.size(0), M = x.size(1), N = x.size(2);
    req(location, "location", x.device(), {B, M, 2}, at::kLong);
    return {B, M, N, H, W, x.device()};
}
int scatter_mode(const std::string& t) {
    TORCH_CHECK(t == "cover" || t == "add", "scatter_type: '", t, "'");
    return t == "add" ? 1 : 0;
}
void scatter_forward_launch(const ScatterDims& d, const Tensor& x, const Tensor& location, const Tensor& out, int add) {
    const int B = to_int(d.B, "B"), M = to_int(d.M, "M"), N = to_int(d.N, "N"), H = to_int(d.H, "H"), W = to_int(d.W, "W");
    Tensor ws = at::empty({hpc_rll_scatter_workspace_ints(B, M, H, W)}, at::TensorOptions().dtype(at::kInt).device(d.dev));
    check(hpc_rll_scatter_connection_forward(fptr(x), iptr(location), fmut(out), ws.data_ptr<int32_t>(), B, M, N, H, W, add,
                                             stream_of(d.dev)),
          "hpc_rll_scatter_connection_forward");
}
void scatter_backward_launch(const Tensor& grad_out, const Tensor& location, const Tensor& grad_x) {
    check(hpc_rll_scatter_connection_backward(fptr(grad_out), iptr(location), fmut(grad_x), (int)grad_out.size(0),
                                              (int)location.size(1), (int)grad_out.size(1), (int)grad_out.size(2),
                                              (int)grad_out.size(3), stream_of(grad_out.device())),
          "hpc_rll_scatter_connection_backward");
}

// inputs = [x (B,M,N) fp32, location (B,M,2) int64 (y,x)], outputs = [out (B,N,H,W)] (fully overwritten).
// Reference: src/torch_utils/network/scatter_connection.cu:8-49.
void ScatterConnectionForward(const TensorList& in, const TensorList& out, const std::string& scatter_type) {
    expect_len(in, 2, "ScatterConnectionForward inputs");
    expect_len(out, 1, "ScatterConnectionForward outputs");
    req(in[0], "x");
    const Tensor& o = req(out[0], "output", in[0].device());
    TORCH_CHECK(in[0].dim() == 3 && o.dim() == 4 && o.size(0) == in[0].size(0) && o.size(1) == in[0].size(2),
                "output: expected (B,N,H,W) matching x (B,M,N), got ", o.sizes(), " for x ", in[0].sizes());
    const ScatterDims d = scatter_check(in[0], in[1], o.size(2), o.size(3));
    c10::DeviceGuard g(d.dev);
    scatter_forward_launch(d, in[0], in[1], o, scatter_mode(scatter_type));
}

// inputs = [grad_out (B,N,H,W), location (B,M,2)], outputs = [grad_x (B,M,N)].  scatter_connection.cu:51-73.
void ScatterConnectionBackward(const TensorList& in, const TensorList& out) {
    expect_len(in, 2, "ScatterConnectionBackward inputs");
    expect_len(out, 1, "ScatterConnectionBackward outputs");
    const Tensor& go = req(in[0], "grad_out");
    TORCH_CHECK(go.dim() == 4, "grad_out: expected (B,N,H,W), got ", go.sizes());
    const at::Device dev = go.device();
    TORCH_CHECK(in[1].defined() && in[1].dim() == 3, "location: expected (B,M,2)");
    const int64_t M = in[1].size(1);
    req(in[1], "location", dev, {go.size(0), M, 2}, at::kLong);
    req(out[0], "grad_x", dev, {go.size(0), M, go.size(1)});
    c10::DeviceGuard g(dev);
    scatter_backward_launch(go, in[1], out[0]);
}

struct ScatterFn : public ag::Function<ScatterFn> {
    static Tensor forward(ag::AutogradContext* ctx, const Tensor& x, const Tensor& location, int64_t H, int64_t W,
                          int64_t add) {
        const ScatterDims d = scatter_check(x, location, H, W);
        c10::DeviceGuard g(d.dev);
        Tensor out = new_f32({d.B, d.N, H, W}, d.dev);
        scatter_forward_launch(d, x, location, out, (int)add);
        ctx->save_for_backward({location});
        ctx->saved_data["N"] = d.N;
        return out;
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        if (!ctx->needs_input_grad(0)) return {undef(), undef(), undef(), undef(), undef()};
        const Tensor location = ctx->get_saved_variables()[0];
        Tensor go = grads[0].contiguous();
        req(go, "grad_out", location.device());
        c10::DeviceGuard g(go.device());
        Tensor gx = new_f32({go.size(0), location.size(1), go.size(1)}, go.device());
        scatter_backward_launch(go, location, gx);
        return {gx, undef(), undef(), undef(), undef()};
    }
};

}  // namespace
}  // namespace hpc_rll_ext

PYBIND11_MODULE(hpc_torch_utils_network, m) {
    using namespace hpc_rll_ext;
    namespace py = pybind11;
    m.doc() = "hpc_torch_utils_network: LayerNorm-LSTM and ScatterConnection for MI355X (gfx950) -- compiled "
              "PyTorch-ROCm extension over the C ABI of libhpc_rll_hip.so (reference: src/torch_utils/network/entry.cpp:8-13)";
    bind_common(m);
    m.def("LstmForward", &LstmForward, py::arg("inputs"), py::arg("outputs"), py::arg("dropout"),
          py::arg("seed") = py::none(), "lstm forward (HIP)");
    m.def("LstmBackward", &LstmBackward, py::arg("inputs"), py::arg("outputs"), py::arg("dropout"),
          py::arg("seed") = py::none(), "lstm backward (HIP)");
    m.def("ScatterConnectionForward", &ScatterConnectionForward, "scatter_connection forward (HIP)");
    m.def("ScatterConnectionBackward", &ScatterConnectionBackward, "scatter_connection backward (HIP)");
    m.def("lstm_workspace", [](int64_t S, int64_t B, int64_t I, int64_t H, int64_t L, double dropout, const at::Device& dev) {
        return new_f32({lstm_ws_floats(LstmDims{S, B, I, H, L, dev}, dropout)}, dev);
    });
    m.def("lstm_last_forward_path", []() { return hpc_rll_lstm_last_forward_path(); },
          "kernels the last LSTM forward ran on: 0 step, 1 per-layer persistent, 2 wavefront, 3 interleaved step, 4 row-block");
    m.def("lstm_last_backward_path", []() { return hpc_rll_lstm_last_backward_path(); }, "the same for the last LSTM backward");
    m.def("async_error", []() { return hpc_rll_async_error(); },
          "sticky status of the persistent small-batch LSTM kernels: 0, or HPC_RLL_ETIMEOUT (-4) once one gave up waiting");
    m.def("clear_async_error", []() { check(hpc_rll_clear_async_error(), "hpc_rll_clear_async_error"); },
          "acknowledge a timeout; the process continues on the step kernels");
    m.def("_test_occupy_device", [](int ms, const at::Device& dev, int blocks) {
        c10::DeviceGuard g(dev);
        check(hpc_rll_test_occupy_device(ms, blocks, stream_of(dev)), "hpc_rll_test_occupy_device");
    }, pybind11::arg("ms"), pybind11::arg("device"), pybind11::arg("blocks") = 0,
          "test hook: keep `blocks` CUs (0 = all) of `device` busy for ~ms milliseconds on torch's current stream");
    m.def("_test_set_persist_spin_limit", [](int64_t polls, const at::Device& dev) {
        c10::DeviceGuard g(dev);
        check(hpc_rll_test_set_persist_spin_limit(polls), "hpc_rll_test_set_persist_spin_limit");
    }, "test hook: polls a persistent LSTM kernel waits before giving up (0 = shipped value)");
    m.def("gemm_f32", &gemm_f32, py::arg("a"), py::arg("b"), py::arg("out") = py::none(), py::arg("accumulate") = false);

    m.def("lstm", [](const Tensor& x, const Tensor& wx, const Tensor& wh, const Tensor& bias, const Tensor& gamma,
                     const Tensor& beta, const Tensor& h0, const Tensor& c0, double dropout, int64_t seed) {
        const bool graph = at::GradMode::is_enabled() && (x.requires_grad() || wx.requires_grad() || wh.requires_grad() ||
                                                          bias.requires_grad() || gamma.requires_grad() ||
                                                          beta.requires_grad() || h0.requires_grad() || c0.requires_grad());
        return LstmFn::apply(x, wx, wh, bias, gamma, beta, h0, c0, dropout, seed, graph);
    }, py::arg("x"), py::arg("wx"), py::arg("wh"), py::arg("bias"), py::arg("ln_gamma"), py::arg("ln_beta"), py::arg("h0"),
          py::arg("c0"), py::arg("dropout") = 0.0, py::arg("seed") = 0,
          "(y, hn, cn) = LayerNorm-LSTM(x (S,B,I), h0, c0 (L,B,H)); differentiable wrt x, the parameters, h0 and c0");
    m.def("scatter_connection", [](const Tensor& x, const Tensor& location, int64_t H, int64_t W,
                                   const std::string& scatter_type) {
        return ScatterFn::apply(x, location, H, W, (int64_t)scatter_mode(scatter_type));
    }, py::arg("x"), py::arg("location"), py::arg("H"), py::arg("W"), py::arg("scatter_type"));
}
