// rl_utils.cpp -- the compiled `hpc_rl_utils` extension module (reference: src/rl_utils/entry.cpp:8-39).
//
// Two layers in one module:
//   * the reference's L2 functions `XxxForward(inputs, outputs, scalars...)` / `XxxBackward(inputs, outputs)`
//     (declared in include/hpc/rll/cuda/rl_utils/entry.h:10-165) -- validated, launched on torch's current stream;
//   * fused autograd ops (`gae`, `td_lambda`, `vtrace`, `upgo`, `ppo`, `q_nstep_td`, `dist_nstep_td`, `iqn_nstep_td`,
//     `qrdqn_nstep_td`) -- torch::autograd::Function nodes that allocate outputs, launch and register backward in ONE
//     pybind call; these are what hpc_rll.rl_utils.* modules use.
// Host-only C++: every kernel lives behind the C ABI of libhpc_rll_hip.so (include/hpc_rll_hip.h).
#include "common.hpp"
#include "rl_utils_ops.hpp"

#include <map>
#include <mutex>
#include <tuple>

namespace hpc_rll_ext {

// =========================================================================================================== GAE
// coef table c_t = gamma*lambda*D_{t+1}/D_t: depends only on (T, gamma, lambda).  Cached per device and NEVER freed
// (a captured hipGraph or another stream may hold the pointer for the life of the process; the cache is capped, past
// the cap tables are per call).  While a stream is capturing, a miss fills a per-call table inside the capture and
// does not publish it: its contents only exist once the graph is replayed.
struct CoefCache {
    std::mutex mu;
    std::map<std::tuple<int, int, float, float>, Tensor> tab;
};
CoefCache& coef_cache() {
    static CoefCache* c = new CoefCache();   // leaked on purpose: tensors must not be destroyed after torch shuts down
    return *c;
}
constexpr size_t kCoefCacheCap = 256;

Tensor gae_coef(int64_t T, double gamma, double lambda, const at::Device& dev) {
    const auto key = std::make_tuple((int)dev.index(), (int)T, (float)gamma, (float)lambda);
    CoefCache& cc = coef_cache();
    {
        std::lock_guard<std::mutex> lk(cc.mu);
        auto it = cc.tab.find(key);
        if (it != cc.tab.end()) return it->second;
    }
    void* st = stream_of(dev);
    const int capturing = hpc_rll_stream_is_capturing(st);
    TORCH_CHECK(capturing >= 0, "gae_coef: hipStreamIsCapturing failed");
    Tensor c = new_f32({std::max<int64_t>(T, 1)}, dev);
    check(hpc_rll_gae_coef(c.data_ptr<float>(), to_int(T, "T"), (float)gamma, (float)lambda, st), "hpc_rll_gae_coef");
    if (!capturing) {
        // published to every stream: make the fill globally visible first (one ~20 us host wait per new key)
        check(hpc_rll_stream_synchronize(st), "gae_coef: stream synchronize");
        std::lock_guard<std::mutex> lk(cc.mu);
        if (cc.tab.size() < kCoefCacheCap) cc.tab.emplace(key, c);
    }
    return c;
}

void gae_forward_launch(const Tensor& value, const Tensor& reward, const Tensor& adv, const Tensor& coef, double gamma) {
    const int64_t T = reward.size(0), B = reward.size(1);
    check(hpc_rll_gae_forward(fptr(value), fptr(reward), fmut(adv), fptr(coef), to_int(T, "T"), to_int(B, "B"),
                              (float)gamma, stream_of(reward.device())),
          "hpc_rll_gae_forward");
}

void check_gae_inputs(const Tensor& value, const Tensor& reward) {
    req(reward, "reward");
    TORCH_CHECK(reward.dim() == 2, "reward: expected (T,B), got ", reward.sizes());
    req(value, "value", reward.device(), {reward.size(0) + 1, reward.size(1)});
}

// inputs = [value (T+1,B), reward (T,B)], outputs = [adv (T,B)].  Reference: src/rl_utils/gae.cu:8-28.
void GaeForward(const TensorList& inputs, const TensorList& outputs, double gamma, double lambda) {
    expect_len(inputs, 2, "GaeForward inputs");
    expect_len(outputs, 1, "GaeForward outputs");
    const Tensor &value = inputs[0], &reward = inputs[1], &adv = outputs[0];
    check_gae_inputs(value, reward);
    req(adv, "adv", reward.device(), {reward.size(0), reward.size(1)});
    c10::DeviceGuard g(reward.device());
    gae_forward_launch(value, reward, adv, gae_coef(reward.size(0), gamma, lambda, reward.device()), gamma);
}

void gae_backward_launch(const Tensor& grad_adv, const Tensor& gv, const Tensor& gr, const Tensor& coef, double gamma) {
    const int64_t T = grad_adv.size(0), B = grad_adv.size(1);
    check(hpc_rll_gae_backward(fptr(grad_adv), fmut(gv), fmut(gr), fptr(coef), to_int(T, "T"), to_int(B, "B"),
                               (float)gamma, stream_of(grad_adv.device())),
          "hpc_rll_gae_backward");
}

// inputs = [grad_adv (T,B)], outputs = [grad_value (T+1,B) | None, grad_reward (T,B) | None].  New entry (the
// reference registers no GaeBackward, entry.cpp:22): the analytic adjoint of hpc_rll.origin.gae.
void GaeBackward(const TensorList& inputs, const OptList& outputs, double gamma, double lambda) {
    expect_len(inputs, 1, "GaeBackward inputs");
    expect_len(outputs, 2, "GaeBackward outputs");
    const Tensor& ga = req(inputs[0], "grad_adv");
    TORCH_CHECK(ga.dim() == 2, "grad_adv: expected (T,B), got ", ga.sizes());
    const int64_t T = ga.size(0), B = ga.size(1);
    req_opt(outputs[0], "grad_value", ga.device(), {T + 1, B});
    req_opt(outputs[1], "grad_reward", ga.device(), {T, B});
    c10::DeviceGuard g(ga.device());
    gae_backward_launch(ga, has(outputs[0]) ? *outputs[0] : undef(), has(outputs[1]) ? *outputs[1] : undef(),
                        gae_coef(T, gamma, lambda, ga.device()), gamma);
}

struct GaeFn : public ag::Function<GaeFn> {
    static Tensor forward(ag::AutogradContext* ctx, const Tensor& value, const Tensor& reward, double gamma,
                          double lambda) {
        check_gae_inputs(value, reward);
        c10::DeviceGuard g(reward.device());
        Tensor coef = gae_coef(reward.size(0), gamma, lambda, reward.device());
        Tensor adv = at::empty_like(reward);
        gae_forward_launch(value, reward, adv, coef, gamma);
        ctx->saved_data["coef"] = coef;   // keeps a per-call (uncached) table alive until backward
        ctx->saved_data["gamma"] = gamma;
        return adv;
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        const bool need_v = ctx->needs_input_grad(0), need_r = ctx->needs_input_grad(1);
        if (!(need_v || need_r)) return {undef(), undef(), undef(), undef()};
        Tensor ga = grads[0].contiguous();
        req(ga, "grad_adv");
        const int64_t T = ga.size(0), B = ga.size(1);
        c10::DeviceGuard g(ga.device());
        Tensor gv = need_v ? new_f32({T + 1, B}, ga.device()) : undef();
        Tensor gr = need_r ? at::empty_like(ga) : undef();
        gae_backward_launch(ga, gv, gr, ctx->saved_data["coef"].toTensor(), ctx->saved_data["gamma"].toDouble());
        return {gv, gr, undef(), undef()};
    }
};

// ==================================================================================================== TD(lambda)
int td_weight_mode(const OptTensor& weight, int64_t T, int64_t B, const at::Device& dev) {
    if (!has(weight)) return 0;
    req(*weight, "weight", dev);
    if (weight->dim() == 2 && weight->size(0) == T && weight->size(1) == B) return 2;
    if (weight->dim() == 1 && weight->size(0) == B) return 1;
    TORCH_CHECK(false, "weight: shape ", weight->sizes(), ", expected (", T, ",", B, ") or (", B, ",)");
}

void td_lambda_forward_impl(const Tensor& value, const Tensor& reward, const OptTensor& weight, const Tensor& loss,
                            const Tensor& grad_buf, double gamma, double lambda, std::optional<double> scale) {
    req(reward, "reward");
    TORCH_CHECK(reward.dim() == 2, "reward: expected (T,B), got ", reward.sizes());
    const int64_t T = reward.size(0), B = reward.size(1);
    const at::Device dev = reward.device();
    req(value, "value", dev, {T + 1, B});
    req(loss, "loss", dev, {1});
    req(grad_buf, "grad_buf", dev, {T, B});
    const int mode = td_weight_mode(weight, T, B, dev);
    c10::DeviceGuard g(dev);
    Tensor partials = new_f32({hpc_rll_partials_floats(B)}, dev);
    check(hpc_rll_td_lambda_forward(fptr(value), fptr(reward), fptr(weight), mode, fmut(loss), fmut(grad_buf),
                                    fmut(partials), to_int(T, "T"), to_int(B, "B"), (float)gamma, (float)lambda,
                                    loss_scale(scale, T * B), stream_of(dev)),
          "hpc_rll_td_lambda_forward");
}

// inputs = [value (T+1,B), reward (T,B), weight (None | (B,) | (T,B))], outputs = [loss (1,), grad_buf (T,B)].
// Reference: src/rl_utils/td_lambda.cu:8-33 (which reads weight as (T,B) whatever its shape: SURVEY.md A.2).
void TdLambdaForward(const OptList& inputs, const TensorList& outputs, double gamma, double lambda,
                     std::optional<double> scale) {
    expect_len(inputs, 3, "TdLambdaForward inputs");
    expect_len(outputs, 2, "TdLambdaForward outputs");
    TORCH_CHECK(has(inputs[0]) && has(inputs[1]), "TdLambdaForward: value / reward must be tensors");
    td_lambda_forward_impl(*inputs[0], *inputs[1], inputs[2], outputs[0], outputs[1], gamma, lambda, scale);
}

// inputs = [grad_loss (scalar tensor), grad_buf (T,B)], outputs = [grad_value (T+1,B)].  td_lambda.cu:35-52.
void TdLambdaBackward(const TensorList& inputs, const TensorList& outputs) {
    expect_len(inputs, 2, "TdLambdaBackward inputs");
    expect_len(outputs, 1, "TdLambdaBackward outputs");
    const Tensor& gb = req(inputs[1], "grad_buf");
    TORCH_CHECK(gb.dim() == 2, "grad_buf: expected (T,B), got ", gb.sizes());
    const int64_t T = gb.size(0), B = gb.size(1);
    Tensor gl = grad1(inputs[0], gb.device(), "grad_loss");
    req(outputs[0], "grad_value", gb.device(), {T + 1, B});
    c10::DeviceGuard g(gb.device());
    check(hpc_rll_td_lambda_backward(fptr(gl), fptr(gb), fmut(outputs[0]), to_int(T, "T"), to_int(B, "B"),
                                     stream_of(gb.device())),
          "hpc_rll_td_lambda_backward");
}

struct TdLambdaFn : public ag::Function<TdLambdaFn> {
    static Tensor forward(ag::AutogradContext* ctx, const Tensor& value, const Tensor& reward, const OptTensor& weight,
                          double gamma, double lambda, std::optional<double> scale) {
        req(reward, "reward");
        TORCH_CHECK(reward.dim() == 2, "reward: expected (T,B), got ", reward.sizes());
        c10::DeviceGuard g(reward.device());
        Tensor loss = new_f32({1}, reward.device());
        Tensor grad_buf = at::empty_like(reward);
        td_lambda_forward_impl(value, reward, weight, loss, grad_buf, gamma, lambda, scale);
        ctx->save_for_backward({grad_buf});
        return loss;
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        const Tensor grad_buf = ctx->get_saved_variables()[0];
        const int64_t T = grad_buf.size(0), B = grad_buf.size(1);
        const at::Device dev = grad_buf.device();
        c10::DeviceGuard g(dev);
        Tensor gl = grad1(grads[0], dev, "grad_loss");
        Tensor gv = new_f32({T + 1, B}, dev);
        check(hpc_rll_td_lambda_backward(fptr(gl), fptr(grad_buf), fmut(gv), (int)T, (int)B, stream_of(dev)),
              "hpc_rll_td_lambda_backward");
        return {gv, undef(), undef(), undef(), undef(), undef()};
    }
};

// ======================================================================================================= V-trace

VtraceDims vtrace_check(const Tensor& target, const Tensor& behaviour, const Tensor& action, const Tensor& value,
                        const Tensor& reward, const OptTensor& weight) {
    req(target, "target_output");
    TORCH_CHECK(target.dim() == 3, "target_output: expected (T,B,N), got ", target.sizes());
    const int64_t T = target.size(0), B = target.size(1), N = target.size(2);
    const at::Device dev = target.device();
    req(behaviour, "behaviour_output", dev, {T, B, N});
    req(action, "action", dev, {T, B}, at::kLong);
    req(value, "value", dev, {T + 1, B});
    req(reward, "reward", dev, {T, B});
    req_opt(weight, "weight", dev, {T, B});
    return {T, B, N, dev};
}

void vtrace_forward_launch(const VtraceDims& d, const Tensor& target, const Tensor& behaviour, const Tensor& action,
                           const Tensor& value, const Tensor& reward, const OptTensor& weight, const Tensor& losses,
                           const Tensor& ws, double gamma, double lambda, double rho_clip, double c_clip,
                           double rho_pg_clip, std::optional<double> scale) {
    check(hpc_rll_vtrace_forward(fptr(target), fptr(behaviour), iptr(action), fptr(value), fptr(reward), fptr(weight),
                                 fmut(losses), fmut(ws), to_int(d.T, "T"), to_int(d.B, "B"), to_int(d.N, "N"),
                                 (float)gamma, (float)lambda, (float)rho_clip, (float)c_clip, (float)rho_pg_clip,
                                 loss_scale(scale, d.T * d.B), stream_of(d.dev)),
          "hpc_rll_vtrace_forward");
}

Tensor vtrace_workspace(int64_t T, int64_t B, const at::Device& dev) {
    return new_f32({hpc_rll_vtrace_workspace_floats(to_int(T, "T"), to_int(B, "B"))}, dev);
}

void vtrace_backward_launch(const Tensor& g_pg, const Tensor& g_v, const Tensor& g_ent, const Tensor& target,
                            const Tensor& action, const Tensor& ws, const Tensor& grad_target, const Tensor& grad_value) {
    const at::Device dev = target.device();
    check(hpc_rll_vtrace_backward(fptr(g_pg), fptr(g_v), fptr(g_ent), fptr(target), iptr(action), fptr(ws),
                                  fmut(grad_target), fmut(grad_value), (int)target.size(0), (int)target.size(1),
                                  (int)target.size(2), stream_of(dev)),
          "hpc_rll_vtrace_backward");
}

struct VtraceFn : public ag::Function<VtraceFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& target, const Tensor& behaviour,
                                   const Tensor& action, const Tensor& value, const Tensor& reward,
                                   const OptTensor& weight, double gamma, double lambda, double rho_clip, double c_clip,
                                   double rho_pg_clip, std::optional<double> scale) {
        const VtraceDims d = vtrace_check(target, behaviour, action, value, reward, weight);
        c10::DeviceGuard g(d.dev);
        Tensor losses = new_f32({3}, d.dev);
        Tensor ws = vtrace_workspace(d.T, d.B, d.dev);
        vtrace_forward_launch(d, target, behaviour, action, value, reward, weight, losses, ws, gamma, lambda, rho_clip,
                              c_clip, rho_pg_clip, scale);
        ctx->save_for_backward({target, action, ws});
        return {alias_of(losses, 0, 1), alias_of(losses, 1, 1), alias_of(losses, 2, 1)};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(12);
        const bool need_t = ctx->needs_input_grad(0), need_v = ctx->needs_input_grad(3);
        if (!(need_t || need_v)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &target = saved[0], &action = saved[1], &ws = saved[2];
        const at::Device dev = target.device();
        c10::DeviceGuard g(dev);
        const int64_t T = target.size(0), B = target.size(1);
        Tensor g_pg = grad1(grads[0], dev, "grad_policy_loss"), g_v = grad1(grads[1], dev, "grad_value_loss"),
               g_e = grad1(grads[2], dev, "grad_entropy_loss");
        Tensor grad_target = need_t ? at::empty_like(target) : undef();
        Tensor grad_value = need_v ? new_f32({T + 1, B}, dev) : undef();
        vtrace_backward_launch(g_pg, g_v, g_e, target, action, ws, grad_target, grad_value);
        out[0] = grad_target;
        out[3] = grad_value;
        return out;
    }
};

// ========================================================================================================== UPGO
struct UpgoFn : public ag::Function<UpgoFn> {
    static Tensor forward(ag::AutogradContext* ctx, const Tensor& target, const Tensor& rho, const Tensor& action,
                          const Tensor& reward, const Tensor& value, std::optional<double> scale);
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads);
};

UpgoDims upgo_check(const Tensor& target, const Tensor& rho, const Tensor& action, const Tensor& reward,
                    const Tensor& value) {
    req(target, "target_output");
    TORCH_CHECK(target.dim() == 3, "target_output: expected (T,B,N), got ", target.sizes());
    const int64_t T = target.size(0), B = target.size(1), N = target.size(2);
    const at::Device dev = target.device();
    req(rho, "rhos", dev, {T, B});
    req(action, "action", dev, {T, B}, at::kLong);
    req(reward, "rewards", dev, {T, B});
    req(value, "bootstrap_values", dev, {T + 1, B});
    return {T, B, N, dev};
}
Tensor upgo_workspace(int64_t T, int64_t B, const at::Device& dev) {
    return new_f32({hpc_rll_upgo_workspace_floats(to_int(T, "T"), to_int(B, "B"))}, dev);
}
void upgo_forward_launch(const UpgoDims& d, const Tensor& target, const Tensor& rho, const Tensor& action,
                         const Tensor& reward, const Tensor& value, const Tensor& loss, const Tensor& ws,
                         std::optional<double> scale) {
    check(hpc_rll_upgo_forward(fptr(target), fptr(rho), iptr(action), fptr(reward), fptr(value), fmut(loss), fmut(ws),
                               to_int(d.T, "T"), to_int(d.B, "B"), to_int(d.N, "N"), loss_scale(scale, d.T * d.B),
                               stream_of(d.dev)),
          "hpc_rll_upgo_forward");
}
void upgo_backward_launch(const Tensor& g, const Tensor& target, const Tensor& action, const Tensor& ws,
                          const Tensor& grad_target) {
    check(hpc_rll_upgo_backward(g.defined() ? fptr(g) : nullptr, fptr(target), iptr(action), fptr(ws),
                                fmut(grad_target), (int)target.size(0), (int)target.size(1), (int)target.size(2),
                                stream_of(target.device())),
          "hpc_rll_upgo_backward");
}

Tensor UpgoFn::forward(ag::AutogradContext* ctx, const Tensor& target, const Tensor& rho, const Tensor& action,
                       const Tensor& reward, const Tensor& value, std::optional<double> scale) {
    const UpgoDims d = upgo_check(target, rho, action, reward, value);
    c10::DeviceGuard g(d.dev);
    Tensor loss = new_f32({1}, d.dev);
    Tensor ws = upgo_workspace(d.T, d.B, d.dev);
    upgo_forward_launch(d, target, rho, action, reward, value, loss, ws, scale);
    ctx->save_for_backward({target, action, ws});
    return loss;
}
ag::tensor_list UpgoFn::backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
    ag::tensor_list out(6);
    if (!ctx->needs_input_grad(0)) return out;
    const auto saved = ctx->get_saved_variables();
    const Tensor& target = saved[0];
    c10::DeviceGuard g(target.device());
    Tensor gl = grad1(grads[0], target.device(), "grad_loss");
    Tensor grad_target = at::empty_like(target);
    upgo_backward_launch(gl, target, saved[1], saved[2], grad_target);
    out[0] = grad_target;
    return out;
}

// =========================================================================================================== PPO
PpoDims ppo_check(const Tensor& ln, const Tensor& lo, const Tensor& action, const Tensor& vn, const Tensor& vo,
                  const Tensor& adv, const Tensor& ret, const OptTensor& weight) {
    req(ln, "logits_new");
    TORCH_CHECK(ln.dim() == 2, "logits_new: expected (B,N), got ", ln.sizes());
    const int64_t B = ln.size(0), N = ln.size(1);
    const at::Device dev = ln.device();
    req(lo, "logits_old", dev, {B, N});
    req(action, "action", dev, {B}, at::kLong);
    req(vn, "value_new", dev, {B});
    req(vo, "value_old", dev, {B});
    req(adv, "adv", dev, {B});
    req(ret, "return_", dev, {B});
    req_opt(weight, "weight", dev, {B});
    return {B, N, dev};
}
Tensor ppo_workspace(int64_t B, const at::Device& dev) {
    return new_f32({hpc_rll_ppo_workspace_floats(to_int(B, "B"))}, dev);
}
void ppo_forward_launch(const PpoDims& d, const Tensor& ln, const Tensor& lo, const Tensor& action, const Tensor& vn,
                        const Tensor& vo, const Tensor& adv, const Tensor& ret, const OptTensor& weight,
                        const Tensor& out5, const Tensor& ws, bool use_value_clip, double clip_ratio, double dual_clip,
                        std::optional<double> scale) {
    check(hpc_rll_ppo_forward(fptr(ln), fptr(lo), iptr(action), fptr(vn), fptr(vo), fptr(adv), fptr(ret), fptr(weight),
                              fmut(out5), fmut(ws), to_int(d.B, "B"), to_int(d.N, "N"), (float)clip_ratio,
                              use_value_clip ? 1 : 0, (float)dual_clip, loss_scale(scale, d.B), stream_of(d.dev)),
          "hpc_rll_ppo_forward");
}
void ppo_backward_launch(const Tensor& g_p, const Tensor& g_v, const Tensor& g_e, const Tensor& ln, const Tensor& action,
                         const Tensor& ws, const Tensor& grad_logits, const Tensor& grad_value) {
    check(hpc_rll_ppo_backward(fptr(g_p), fptr(g_v), fptr(g_e), fptr(ln), iptr(action), fptr(ws), fmut(grad_logits),
                               fmut(grad_value), (int)ln.size(0), (int)ln.size(1), stream_of(ln.device())),
          "hpc_rll_ppo_backward");
}

struct PpoFn : public ag::Function<PpoFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& ln, const Tensor& lo, const Tensor& action,
                                   const Tensor& vn, const Tensor& vo, const Tensor& adv, const Tensor& ret,
                                   const OptTensor& weight, double clip_ratio, bool use_value_clip, double dual_clip,
                                   std::optional<double> scale) {
        const PpoDims d = ppo_check(ln, lo, action, vn, vo, adv, ret, weight);
        c10::DeviceGuard g(d.dev);
        Tensor out5 = new_f32({5}, d.dev);
        Tensor ws = ppo_workspace(d.B, d.dev);
        ppo_forward_launch(d, ln, lo, action, vn, vo, adv, ret, weight, out5, ws, use_value_clip, clip_ratio, dual_clip,
                           scale);
        ctx->save_for_backward({ln, action, ws});
        Tensor info = alias_of(out5, 3, 2);
        ctx->mark_non_differentiable({info});
        return {alias_of(out5, 0, 1), alias_of(out5, 1, 1), alias_of(out5, 2, 1), info};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(12);
        const bool need_l = ctx->needs_input_grad(0), need_v = ctx->needs_input_grad(3);
        if (!(need_l || need_v)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &ln = saved[0], &action = saved[1], &ws = saved[2];
        const at::Device dev = ln.device();
        c10::DeviceGuard g(dev);
        Tensor g_p = grad1(grads[0], dev, "grad_policy_loss"), g_v = grad1(grads[1], dev, "grad_value_loss"),
               g_e = grad1(grads[2], dev, "grad_entropy_loss");
        Tensor grad_logits = need_l ? at::empty_like(ln) : undef();
        Tensor grad_value = need_v ? new_f32({ln.size(0)}, dev) : undef();
        ppo_backward_launch(g_p, g_v, g_e, ln, action, ws, grad_logits, grad_value);
        out[0] = grad_logits;
        out[3] = grad_value;
        return out;
    }
};

// ==================================================================================================== q n-step TD
struct QDims { int64_t B, N, nstep; at::Device dev; };
int64_t check_nstep_reward(const Tensor& reward, int64_t B, const at::Device& dev) {
    req(reward, "reward", dev);
    TORCH_CHECK(reward.dim() == 2 && reward.size(1) == B, "reward: expected (nstep,", B, "), got ", reward.sizes());
    return reward.size(0);
}
QDims q_check(const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction, const Tensor& reward,
              const Tensor& done, const OptTensor& weight) {
    req(q, "q");
    TORCH_CHECK(q.dim() == 2, "q: expected (B,N), got ", q.sizes());
    const int64_t B = q.size(0), N = q.size(1);
    const at::Device dev = q.device();
    req(nq, "next_n_q", dev, {B, N});
    req(action, "action", dev, {B}, at::kLong);
    req(naction, "next_n_action", dev, {B}, at::kLong);
    const int64_t nstep = check_nstep_reward(reward, B, dev);
    req(done, "done", dev, {B});
    req_opt(weight, "weight", dev, {B});
    return {B, N, nstep, dev};
}
void q_forward_launch(const QDims& d, const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                      const Tensor& reward, const Tensor& done, const OptTensor& weight, const Tensor& td_err,
                      const Tensor& loss, const Tensor& grad_buf, double gamma, int rescale, std::optional<double> scale) {
    Tensor partials = new_f32({hpc_rll_partials_floats(d.B)}, d.dev);
    check(hpc_rll_q_nstep_td_forward(fptr(q), fptr(nq), iptr(action), iptr(naction), fptr(reward), fptr(done),
                                     fptr(weight), fmut(loss), fmut(td_err), fmut(grad_buf), fmut(partials),
                                     to_int(d.nstep, "nstep"), to_int(d.B, "B"), to_int(d.N, "N"), (float)gamma, rescale,
                                     loss_scale(scale, d.B), stream_of(d.dev)),
          "hpc_rll_q_nstep_td_forward");
}

// inputs = [q, next_n_q (B,N), action, next_n_action (B,) int64, reward (nstep,B), done (B,), weight (B,)|None];
// outputs = [td_err (B,), loss (1,), grad_buf (B,)].  Reference: src/rl_utils/q_nstep_td.cu:8-39,
// q_nstep_td_rescale.cu:8-39.
void q_forward_l2(const OptList& in, const TensorList& out, double gamma, int rescale, std::optional<double> scale) {
    expect_len(in, 7, "QNStepTdForward inputs");
    expect_len(out, 3, "QNStepTdForward outputs");
    for (int i = 0; i < 6; ++i) TORCH_CHECK(has(in[i]), "QNStepTdForward: inputs[", i, "] is None");
    const QDims d = q_check(*in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6]);
    req(out[0], "td_err", d.dev, {d.B});
    req(out[1], "loss", d.dev, {1});
    req(out[2], "grad_buf", d.dev, {d.B});
    c10::DeviceGuard g(d.dev);
    q_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6], out[0], out[1], out[2], gamma, rescale,
                     scale);
}
void q_backward_launch(const Tensor& gl, const Tensor& grad_buf, const Tensor& action, const Tensor& grad_q) {
    check(hpc_rll_q_nstep_td_backward(fptr(gl), fptr(grad_buf), iptr(action), fmut(grad_q), (int)grad_q.size(0),
                                      (int)grad_q.size(1), stream_of(grad_q.device())),
          "hpc_rll_q_nstep_td_backward");
}
// inputs = [grad_loss, grad_buf (B,), action]; outputs = [grad_q (B,N)].  q_nstep_td.cu:41-63.
void q_backward_l2(const TensorList& in, const TensorList& out) {
    expect_len(in, 3, "QNStepTdBackward inputs");
    expect_len(out, 1, "QNStepTdBackward outputs");
    const Tensor& gq = req(out[0], "grad_q");
    TORCH_CHECK(gq.dim() == 2, "grad_q: expected (B,N), got ", gq.sizes());
    const at::Device dev = gq.device();
    req(in[1], "grad_buf", dev, {gq.size(0)});
    req(in[2], "action", dev, {gq.size(0)}, at::kLong);
    c10::DeviceGuard g(dev);
    q_backward_launch(grad1(in[0], dev, "grad_loss"), in[1], in[2], gq);
}

template <int RESCALE> struct QNStepFn : public ag::Function<QNStepFn<RESCALE>> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& q, const Tensor& nq, const Tensor& action,
                                   const Tensor& naction, const Tensor& reward, const Tensor& done,
                                   const OptTensor& weight, double gamma, std::optional<double> scale) {
        const QDims d = q_check(q, nq, action, naction, reward, done, weight);
        c10::DeviceGuard g(d.dev);
        Tensor td_err = new_f32({d.B}, d.dev), loss = new_f32({1}, d.dev), grad_buf = new_f32({d.B}, d.dev);
        q_forward_launch(d, q, nq, action, naction, reward, done, weight, td_err, loss, grad_buf, gamma, RESCALE, scale);
        ctx->save_for_backward({grad_buf, action});
        ctx->saved_data["N"] = d.N;
        ctx->mark_non_differentiable({td_err});
        return {loss, td_err};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(9);
        if (!ctx->needs_input_grad(0)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &grad_buf = saved[0], &action = saved[1];
        const at::Device dev = grad_buf.device();
        c10::DeviceGuard g(dev);
        Tensor gq = new_f32({grad_buf.size(0), ctx->saved_data["N"].toInt()}, dev);
        q_backward_launch(grad1(grads[0], dev, "grad_loss"), grad_buf, action, gq);
        out[0] = gq;
        return out;
    }
};

// ===================================================================================================== dist (C51)
struct DistDims { int64_t B, N, A, nstep; at::Device dev; };
DistDims dist_check(const Tensor& dist, const Tensor& ndist, const Tensor& action, const Tensor& naction,
                    const Tensor& reward, const Tensor& done, const OptTensor& weight) {
    req(dist, "dist");
    TORCH_CHECK(dist.dim() == 3, "dist: expected (B,N,n_atom), got ", dist.sizes());
    const int64_t B = dist.size(0), N = dist.size(1), A = dist.size(2);
    const at::Device dev = dist.device();
    req(ndist, "next_n_dist", dev, {B, N, A});
    req(action, "action", dev, {B}, at::kLong);
    req(naction, "next_n_action", dev, {B}, at::kLong);
    const int64_t nstep = check_nstep_reward(reward, B, dev);
    req(done, "done", dev, {B});
    req_opt(weight, "weight", dev, {B});
    return {B, N, A, nstep, dev};
}
void dist_forward_launch(const DistDims& d, const Tensor& dist, const Tensor& ndist, const Tensor& action,
                         const Tensor& naction, const Tensor& reward, const Tensor& done, const OptTensor& weight,
                         const Tensor& td_err, const Tensor& loss, const Tensor& buf, double gamma, double v_min,
                         double v_max, std::optional<double> scale) {
    Tensor partials = new_f32({hpc_rll_partials_floats(d.B)}, d.dev);
    check(hpc_rll_dist_nstep_td_forward(fptr(dist), fptr(ndist), iptr(action), iptr(naction), fptr(reward), fptr(done),
                                        fptr(weight), fmut(loss), fmut(td_err), fmut(buf), fmut(partials),
                                        to_int(d.nstep, "nstep"), to_int(d.B, "B"), to_int(d.N, "N"), to_int(d.A, "n_atom"),
                                        (float)gamma, (float)v_min, (float)v_max, loss_scale(scale, d.B),
                                        stream_of(d.dev)),
          "hpc_rll_dist_nstep_td_forward");
}
// inputs = [dist, next_n_dist (B,N,n_atom), action, next_n_action (B,), reward (nstep,B), done (B,), weight (B,)|None];
// outputs = [td_err (B,), loss (1,), buf].  Reference: src/rl_utils/dist_nstep_td.cu:8-72, whose buf is
// (B + B*n_atom,) (hpc_rll/rl_utils/td.py:60): any contiguous buf with >= B*n_atom floats is accepted and its first
// B*n_atom floats receive the unit gradient wrt dist[b,a_b,:].
void DistNStepTdForward(const OptList& in, const TensorList& out, double gamma, double v_min, double v_max,
                        std::optional<double> scale) {
    expect_len(in, 7, "DistNStepTdForward inputs");
    expect_len(out, 3, "DistNStepTdForward outputs");
    for (int i = 0; i < 6; ++i) TORCH_CHECK(has(in[i]), "DistNStepTdForward: inputs[", i, "] is None");
    const DistDims d = dist_check(*in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6]);
    req(out[0], "td_err", d.dev, {d.B});
    req(out[1], "loss", d.dev, {1});
    req(out[2], "buf", d.dev);
    TORCH_CHECK(out[2].numel() >= d.B * d.A, "buf: ", out[2].numel(), " floats, need at least B*n_atom = ", d.B * d.A);
    c10::DeviceGuard g(d.dev);
    dist_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6], out[0], out[1], out[2], gamma, v_min,
                        v_max, scale);
}
void dist_backward_launch(const Tensor& gl, const Tensor& buf, const Tensor& action, const Tensor& grad_dist) {
    check(hpc_rll_dist_nstep_td_backward(fptr(gl), fptr(buf), iptr(action), fmut(grad_dist), (int)grad_dist.size(0),
                                         (int)grad_dist.size(1), (int)grad_dist.size(2), stream_of(grad_dist.device())),
          "hpc_rll_dist_nstep_td_backward");
}
// inputs = [grad_loss, buf, action]; outputs = [grad_dist (B,N,n_atom)].  dist_nstep_td.cu:74-98.
void DistNStepTdBackward(const TensorList& in, const TensorList& out) {
    expect_len(in, 3, "DistNStepTdBackward inputs");
    expect_len(out, 1, "DistNStepTdBackward outputs");
    const Tensor& gd = req(out[0], "grad_dist");
    TORCH_CHECK(gd.dim() == 3, "grad_dist: expected (B,N,n_atom), got ", gd.sizes());
    const at::Device dev = gd.device();
    req(in[1], "buf", dev);
    TORCH_CHECK(in[1].numel() >= gd.size(0) * gd.size(2), "buf: too small");
    req(in[2], "action", dev, {gd.size(0)}, at::kLong);
    c10::DeviceGuard g(dev);
    dist_backward_launch(grad1(in[0], dev, "grad_loss"), in[1], in[2], gd);
}

struct DistFn : public ag::Function<DistFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& dist, const Tensor& ndist,
                                   const Tensor& action, const Tensor& naction, const Tensor& reward, const Tensor& done,
                                   const OptTensor& weight, double gamma, double v_min, double v_max,
                                   std::optional<double> scale) {
        const DistDims d = dist_check(dist, ndist, action, naction, reward, done, weight);
        c10::DeviceGuard g(d.dev);
        Tensor td_err = new_f32({d.B}, d.dev), loss = new_f32({1}, d.dev), buf = new_f32({d.B, d.A}, d.dev);
        dist_forward_launch(d, dist, ndist, action, naction, reward, done, weight, td_err, loss, buf, gamma, v_min, v_max,
                            scale);
        ctx->save_for_backward({buf, action});
        ctx->saved_data["N"] = d.N;
        ctx->mark_non_differentiable({td_err});
        return {loss, td_err};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(11);
        if (!ctx->needs_input_grad(0)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &buf = saved[0], &action = saved[1];
        const at::Device dev = buf.device();
        c10::DeviceGuard g(dev);
        Tensor gd = new_f32({buf.size(0), ctx->saved_data["N"].toInt(), buf.size(1)}, dev);
        dist_backward_launch(grad1(grads[0], dev, "grad_loss"), buf, action, gd);
        out[0] = gd;
        return out;
    }
};

// =========================================================================================================== IQN
struct IqnDims { int64_t tau, tau_p, B, N, nstep; at::Device dev; };
// bnt (round 6, not in the reference): q (B,N,tau), next_n_q (B,N,tau') -- the quantile axis innermost, QR-DQN's layout
IqnDims iqn_check(const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction, const Tensor& reward,
                  const Tensor& done, const Tensor& rq, const OptTensor& weight, const OptTensor& vg, bool bnt = false) {
    req(q, "q");
    TORCH_CHECK(q.dim() == 3, "q: expected ", bnt ? "(B,N,tau)" : "(tau,B,N)", ", got ", q.sizes());
    const int64_t tau = bnt ? q.size(2) : q.size(0), B = bnt ? q.size(0) : q.size(1), N = bnt ? q.size(1) : q.size(2);
    const at::Device dev = q.device();
    req(nq, "next_n_q", dev);
    if (bnt) {
        TORCH_CHECK(nq.dim() == 3 && nq.size(0) == B && nq.size(1) == N, "next_n_q: shape ", nq.sizes(), ", expected (", B, ",",
                    N, ",tau')");
    } else {
        TORCH_CHECK(nq.dim() == 3 && nq.size(1) == B && nq.size(2) == N, "next_n_q: shape ", nq.sizes(), ", expected (tau',",
                    B, ",", N, ")");
    }
    req(action, "action", dev, {B}, at::kLong);
    req(naction, "next_n_action", dev, {B}, at::kLong);
    const int64_t nstep = check_nstep_reward(reward, B, dev);
    req(done, "done", dev, {B});
    req(rq, "replay_quantiles", dev);
    TORCH_CHECK(rq.numel() == tau * B, "replay_quantiles: ", rq.sizes(), " does not hold tau*B = ", tau * B, " values");
    req_opt(weight, "weight", dev, {B});
    req_opt(vg, "value_gamma", dev, {B});
    return {tau, bnt ? nq.size(2) : nq.size(0), B, N, nstep, dev};
}
void iqn_forward_launch(const IqnDims& d, const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                        const Tensor& reward, const Tensor& done, const Tensor& rq, const OptTensor& weight,
                        const OptTensor& vg, const Tensor& loss, const Tensor& td_err, const Tensor& grad_buf,
                        double gamma, double kappa, std::optional<double> scale, bool bnt = false) {
    Tensor partials = new_f32({hpc_rll_partials_floats(d.B)}, d.dev);
    check((bnt ? hpc_rll_iqn_nstep_td_forward_bnt : hpc_rll_iqn_nstep_td_forward)(fptr(q), fptr(nq), iptr(action), iptr(naction), fptr(reward), fptr(done), fptr(rq),
                                       fptr(weight), fptr(vg), fmut(loss), fmut(td_err), fmut(grad_buf), fmut(partials),
                                       to_int(d.tau, "tau"), to_int(d.tau_p, "tau'"), to_int(d.nstep, "nstep"),
                                       to_int(d.B, "B"), to_int(d.N, "N"), (float)gamma, (float)kappa,
                                       loss_scale(scale, d.B), stream_of(d.dev)),
          "hpc_rll_iqn_nstep_td_forward");
}
// inputs = [q (tau,B,N), next_n_q (tau',B,N), action, next_n_action (B,), reward (nstep,B), done (B,),
// replay_quantiles (tau,B), weight (B,)|None, value_gamma (B,)|None].
// outputs: native [loss (1,), td_err (B,), grad_buf (B,tau)], or the reference's five
// [loss, td_err, bellman_err_buf, quantile_huber_loss_buf, grad_buf (B,tau',tau)] (hpc_rll/rl_utils/td.py:378-379):
// the two (B,tau',tau) scratch outputs are ignored and the first B*tau floats of grad_buf receive the unit gradient
// wrt q[:,b,a_b].  Reference: src/rl_utils/iqn_nstep_td_error.cu:8-72.
void IQNNStepTDErrorForward(const OptList& in, const TensorList& out, double gamma, double kappa,
                            std::optional<double> scale) {
    expect_len(in, 9, "IQNNStepTDErrorForward inputs");
    TORCH_CHECK(out.size() == 3 || out.size() == 5, "IQNNStepTDErrorForward outputs: expected 3 or 5 tensors, got ",
                out.size());
    for (int i = 0; i < 7; ++i) TORCH_CHECK(has(in[i]), "IQNNStepTDErrorForward: inputs[", i, "] is None");
    const IqnDims d = iqn_check(*in[0], *in[1], *in[2], *in[3], *in[4], *in[5], *in[6], in[7], in[8]);
    const Tensor& gb = out.back();
    req(out[0], "loss", d.dev, {1});
    req(out[1], "td_err", d.dev, {d.B});
    req(gb, "grad_buf", d.dev);
    TORCH_CHECK(gb.numel() >= d.B * d.tau, "grad_buf: ", gb.numel(), " floats, need at least B*tau = ", d.B * d.tau);
    c10::DeviceGuard g(d.dev);
    iqn_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], *in[6], in[7], in[8], out[0], out[1], gb, gamma,
                       kappa, scale);
}
void iqn_backward_launch(const Tensor& gl, const Tensor& grad_buf, const Tensor& action, const Tensor& grad_q) {
    check(hpc_rll_iqn_nstep_td_backward(fptr(gl), fptr(grad_buf), iptr(action), fmut(grad_q), (int)grad_q.size(0),
                                        (int)grad_q.size(1), (int)grad_q.size(2), stream_of(grad_q.device())),
          "hpc_rll_iqn_nstep_td_backward");
}
// inputs = [grad_loss, grad_buf, action] or the reference's [grad_loss, grad_buf, weight, action] (td.py:382; the
// weight is already folded into grad_buf); outputs = [grad_q (tau,B,N)].  iqn_nstep_td_error.cu:74-104.
void IQNNStepTDErrorBackward(const OptList& in, const TensorList& out) {
    TORCH_CHECK(in.size() == 3 || in.size() == 4, "IQNNStepTDErrorBackward inputs: expected 3 or 4 tensors");
    expect_len(out, 1, "IQNNStepTDErrorBackward outputs");
    const Tensor& gq = req(out[0], "grad_q");
    TORCH_CHECK(gq.dim() == 3, "grad_q: expected (tau,B,N), got ", gq.sizes());
    const at::Device dev = gq.device();
    TORCH_CHECK(has(in[0]) && has(in[1]) && has(in.back()), "IQNNStepTDErrorBackward: None input");
    req(*in[1], "grad_buf", dev);
    TORCH_CHECK(in[1]->numel() >= gq.size(0) * gq.size(1), "grad_buf: too small");
    req(*in.back(), "action", dev, {gq.size(1)}, at::kLong);
    c10::DeviceGuard g(dev);
    iqn_backward_launch(grad1(*in[0], dev, "grad_loss"), *in[1], *in.back(), gq);
}

struct IqnFn : public ag::Function<IqnFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& q, const Tensor& nq, const Tensor& action,
                                   const Tensor& naction, const Tensor& reward, const Tensor& done, const Tensor& rq,
                                   const OptTensor& weight, const OptTensor& vg, double gamma, double kappa,
                                   std::optional<double> scale, bool bnt) {
        const IqnDims d = iqn_check(q, nq, action, naction, reward, done, rq, weight, vg, bnt);
        c10::DeviceGuard g(d.dev);
        Tensor loss = new_f32({1}, d.dev), td_err = new_f32({d.B}, d.dev), grad_buf = new_f32({d.B, d.tau}, d.dev);
        iqn_forward_launch(d, q, nq, action, naction, reward, done, rq, weight, vg, loss, td_err, grad_buf, gamma, kappa,
                           scale, bnt);
        ctx->save_for_backward({grad_buf, action});
        ctx->saved_data["N"] = d.N;
        ctx->saved_data["bnt"] = bnt;
        ctx->mark_non_differentiable({td_err});
        return {loss, td_err};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(13);
        if (!ctx->needs_input_grad(0)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &grad_buf = saved[0], &action = saved[1];
        const at::Device dev = grad_buf.device();
        c10::DeviceGuard g(dev);
        if (ctx->saved_data["bnt"].toBool()) {   // grad_q (B,N,tau): one-hot rows of tau values
            const int64_t B = grad_buf.size(0), tau = grad_buf.size(1), N = ctx->saved_data["N"].toInt();
            Tensor gq = new_f32({B, N, tau}, dev);
            check(hpc_rll_iqn_nstep_td_backward_bnt(fptr(grad1(grads[0], dev, "grad_loss")), fptr(grad_buf), iptr(action), fmut(gq),
                                                    (int)tau, (int)B, (int)N, stream_of(dev)),
                  "hpc_rll_iqn_nstep_td_backward_bnt");
            out[0] = gq;
            return out;
        }
        Tensor gq = new_f32({grad_buf.size(1), grad_buf.size(0), ctx->saved_data["N"].toInt()}, dev);
        iqn_backward_launch(grad1(grads[0], dev, "grad_loss"), grad_buf, action, gq);
        out[0] = gq;
        return out;
    }
};

// ======================================================================================================== QR-DQN
struct QrDims { int64_t B, N, tau, nstep; at::Device dev; };
QrDims qr_check(const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction, const Tensor& reward,
                const Tensor& done, const OptTensor& weight, const OptTensor& vg) {
    req(q, "q");
    TORCH_CHECK(q.dim() == 3, "q: expected (B,N,tau), got ", q.sizes());
    const int64_t B = q.size(0), N = q.size(1), tau = q.size(2);
    const at::Device dev = q.device();
    req(nq, "next_n_q", dev, {B, N, tau});
    req(action, "action", dev, {B}, at::kLong);
    req(naction, "next_n_action", dev, {B}, at::kLong);
    const int64_t nstep = check_nstep_reward(reward, B, dev);
    req(done, "done", dev, {B});
    req_opt(weight, "weight", dev, {B});
    req_opt(vg, "value_gamma", dev, {B});
    return {B, N, tau, nstep, dev};
}
void qr_forward_launch(const QrDims& d, const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                       const Tensor& reward, const Tensor& done, const OptTensor& weight, const OptTensor& vg,
                       const Tensor& loss, const Tensor& td_err, const Tensor& grad_buf, double gamma,
                       std::optional<double> tau_value, std::optional<double> scale) {
    Tensor partials = new_f32({hpc_rll_partials_floats(d.B)}, d.dev);
    check(hpc_rll_qrdqn_nstep_td_forward(fptr(q), fptr(nq), iptr(action), iptr(naction), fptr(reward), fptr(done),
                                         fptr(weight), fptr(vg), fmut(loss), fmut(td_err), fmut(grad_buf), fmut(partials),
                                         to_int(d.tau, "tau"), to_int(d.nstep, "nstep"), to_int(d.B, "B"), to_int(d.N, "N"),
                                         (float)gamma, (float)(tau_value.has_value() ? *tau_value : (double)d.tau),
                                         loss_scale(scale, d.B), stream_of(d.dev)),
          "hpc_rll_qrdqn_nstep_td_forward");
}
// inputs = [q, next_n_q (B,N,tau), action, next_n_action (B,), reward (nstep,B), done (B,), weight (B,)|None,
// value_gamma (B,)|None]; outputs: native [loss (1,), td_err (B,), grad_buf (B,tau)] or the reference's five
// [loss, td_err, bellman_err_buf, quantile_huber_loss_buf, grad_buf (B,tau)] (td.py:492-493; the two (B,tau,tau)
// scratch outputs are ignored).  `tau_value` = the `tau` the oracle is called with; default = the integer count the
// reference kernel hard-codes (qrdqn_nstep_td_error_kernel.h:60).  Reference: src/rl_utils/qrdqn_nstep_td_error.cu:8-68.
void QRDQNNStepTDErrorForward(const OptList& in, const TensorList& out, double gamma, std::optional<double> tau_value,
                              std::optional<double> scale) {
    expect_len(in, 8, "QRDQNNStepTDErrorForward inputs");
    TORCH_CHECK(out.size() == 3 || out.size() == 5, "QRDQNNStepTDErrorForward outputs: expected 3 or 5 tensors, got ",
                out.size());
    for (int i = 0; i < 6; ++i) TORCH_CHECK(has(in[i]), "QRDQNNStepTDErrorForward: inputs[", i, "] is None");
    const QrDims d = qr_check(*in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6], in[7]);
    const Tensor& gb = out.back();
    req(out[0], "loss", d.dev, {1});
    req(out[1], "td_err", d.dev, {d.B});
    req(gb, "grad_buf", d.dev);
    TORCH_CHECK(gb.numel() >= d.B * d.tau, "grad_buf: ", gb.numel(), " floats, need at least B*tau = ", d.B * d.tau);
    c10::DeviceGuard g(d.dev);
    qr_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], in[6], in[7], out[0], out[1], gb, gamma,
                      tau_value, scale);
}
void qr_backward_launch(const Tensor& gl, const Tensor& grad_buf, const Tensor& action, const Tensor& grad_q) {
    check(hpc_rll_qrdqn_nstep_td_backward(fptr(gl), fptr(grad_buf), iptr(action), fmut(grad_q), (int)grad_q.size(2),
                                          (int)grad_q.size(0), (int)grad_q.size(1), stream_of(grad_q.device())),
          "hpc_rll_qrdqn_nstep_td_backward");
}
// inputs = [grad_loss, grad_buf (B,tau), action] or the reference's [grad_loss, grad_buf, weight, action] (td.py:496);
// outputs = [grad_q (B,N,tau)].  qrdqn_nstep_td_error.cu:70-99.
void QRDQNNStepTDErrorBackward(const OptList& in, const TensorList& out) {
    TORCH_CHECK(in.size() == 3 || in.size() == 4, "QRDQNNStepTDErrorBackward inputs: expected 3 or 4 tensors");
    expect_len(out, 1, "QRDQNNStepTDErrorBackward outputs");
    const Tensor& gq = req(out[0], "grad_q");
    TORCH_CHECK(gq.dim() == 3, "grad_q: expected (B,N,tau), got ", gq.sizes());
    const at::Device dev = gq.device();
    TORCH_CHECK(has(in[0]) && has(in[1]) && has(in.back()), "QRDQNNStepTDErrorBackward: None input");
    req(*in[1], "grad_buf", dev);
    TORCH_CHECK(in[1]->numel() >= gq.size(0) * gq.size(2), "grad_buf: too small");
    req(*in.back(), "action", dev, {gq.size(0)}, at::kLong);
    c10::DeviceGuard g(dev);
    qr_backward_launch(grad1(*in[0], dev, "grad_loss"), *in[1], *in.back(), gq);
}

struct QrFn : public ag::Function<QrFn> {
    static ag::tensor_list forward(ag::AutogradContext* ctx, const Tensor& q, const Tensor& nq, const Tensor& action,
                                   const Tensor& naction, const Tensor& reward, const Tensor& done,
                                   const OptTensor& weight, const OptTensor& vg, double gamma,
                                   std::optional<double> tau_value, std::optional<double> scale) {
        const QrDims d = qr_check(q, nq, action, naction, reward, done, weight, vg);
        c10::DeviceGuard g(d.dev);
        Tensor loss = new_f32({1}, d.dev), td_err = new_f32({d.B}, d.dev), grad_buf = new_f32({d.B, d.tau}, d.dev);
        qr_forward_launch(d, q, nq, action, naction, reward, done, weight, vg, loss, td_err, grad_buf, gamma, tau_value,
                          scale);
        ctx->save_for_backward({grad_buf, action});
        ctx->saved_data["N"] = d.N;
        ctx->mark_non_differentiable({td_err});
        return {loss, td_err};
    }
    static ag::tensor_list backward(ag::AutogradContext* ctx, ag::tensor_list grads) {
        ag::tensor_list out(11);
        if (!ctx->needs_input_grad(0)) return out;
        const auto saved = ctx->get_saved_variables();
        const Tensor &grad_buf = saved[0], &action = saved[1];
        const at::Device dev = grad_buf.device();
        c10::DeviceGuard g(dev);
        Tensor gq = new_f32({grad_buf.size(0), ctx->saved_data["N"].toInt(), grad_buf.size(1)}, dev);
        qr_backward_launch(grad1(grads[0], dev, "grad_loss"), grad_buf, action, gq);
        out[0] = gq;
        return out;
    }
};

// defined in rl_utils_lists.cpp (V-trace / UPGO / PPO L2 functions incl. the reference's long positional lists) and
// rl_utils_pad.cpp (Pad / Unpad / group policies)
void bind_loss_lists(pybind11::module_& m);
void bind_padding(pybind11::module_& m);

}  // namespace hpc_rll_ext

PYBIND11_MODULE(hpc_rl_utils, m) {
    using namespace hpc_rll_ext;
    namespace py = pybind11;
    m.doc() = "hpc_rl_utils: MI355X (gfx950) operator library behind hpc_rll.rl_utils -- compiled PyTorch-ROCm "
              "extension over the C ABI of libhpc_rll_hip.so (reference: src/rl_utils/entry.cpp:8-39)";
    bind_common(m);

    // ---- L2: the reference's function names and list convention
    m.def("GaeForward", &GaeForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"), py::arg("lambda_"),
          "gae forward (HIP)");
    m.def("GaeBackward", &GaeBackward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"), py::arg("lambda_"),
          "gae backward (HIP)");
    m.def("TdLambdaForward", &TdLambdaForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"),
          py::arg("lambda_"), py::arg("scale") = py::none(), "td_lambda forward (HIP)");
    m.def("TdLambdaBackward", &TdLambdaBackward, "td_lambda backward (HIP)");
    m.def("QNStepTdForward",
          [](const OptList& in, const TensorList& out, double gamma, std::optional<double> scale) {
              q_forward_l2(in, out, gamma, 0, scale);
          },
          py::arg("inputs"), py::arg("outputs"), py::arg("gamma"), py::arg("scale") = py::none(),
          "q_nstep_td forward (HIP)");
    m.def("QNStepTdBackward", &q_backward_l2, "q_nstep_td backward (HIP)");
    m.def("QNStepTdRescaleForward",
          [](const OptList& in, const TensorList& out, double gamma, std::optional<double> scale) {
              q_forward_l2(in, out, gamma, 1, scale);
          },
          py::arg("inputs"), py::arg("outputs"), py::arg("gamma"), py::arg("scale") = py::none(),
          "q_nstep_td_with_rescale forward (HIP)");
    m.def("QNStepTdRescaleBackward", &q_backward_l2, "q_nstep_td_with_rescale backward (HIP)");
    m.def("DistNStepTdForward", &DistNStepTdForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"),
          py::arg("v_min"), py::arg("v_max"), py::arg("scale") = py::none(), "dist_nstep_td forward (HIP)");
    m.def("DistNStepTdBackward", &DistNStepTdBackward, "dist_nstep_td backward (HIP)");
    m.def("IQNNStepTDErrorForward", &IQNNStepTDErrorForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"),
          py::arg("kappa"), py::arg("scale") = py::none(), "iqn_nstep_td_error forward (HIP)");
    m.def("IQNNStepTDErrorBackward", &IQNNStepTDErrorBackward, "iqn_nstep_td_error backward (HIP)");
    m.def("QRDQNNStepTDErrorForward", &QRDQNNStepTDErrorForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"),
          py::arg("tau_value") = py::none(), py::arg("scale") = py::none(), "qrdqn_nstep_td_error forward (HIP)");
    m.def("QRDQNNStepTDErrorBackward", &QRDQNNStepTDErrorBackward, "qrdqn_nstep_td_error backward (HIP)");
    bind_loss_lists(m);   // VTrace / Upgo / PPO Forward+Backward
    bind_padding(m);      // Pad / GroupPad / Unpad {1,2,3}D, split policies, packed variants

    // ---- fused autograd ops (what hpc_rll.rl_utils.* calls)
    m.def("gae", [](const Tensor& value, const Tensor& reward, double gamma, double lambda) {
        return GaeFn::apply(value, reward, gamma, lambda);
    }, py::arg("value"), py::arg("reward"), py::arg("gamma") = 0.99, py::arg("lambda_") = 0.97,
          "adv = GAE(value (T+1,B), reward (T,B)); differentiable wrt value and reward");
    m.def("gae_coef", [](int64_t T, double gamma, double lambda, const at::Device& dev) {
        TORCH_CHECK(dev.is_cuda(), "gae_coef: device must be a GPU");
        c10::DeviceGuard g(dev);
        return gae_coef(T, gamma, lambda, dev);
    });
    m.def("td_lambda", [](const Tensor& value, const Tensor& reward, const OptTensor& weight, double gamma, double lambda,
                          std::optional<double> scale) {
        return TdLambdaFn::apply(value, reward, weight, gamma, lambda, scale);
    }, py::arg("value"), py::arg("reward"), py::arg("weight") = py::none(), py::arg("gamma") = 0.9,
          py::arg("lambda_") = 0.8, py::arg("scale") = py::none());
    m.def("vtrace", [](const Tensor& target, const Tensor& behaviour, const Tensor& action, const Tensor& value,
                       const Tensor& reward, const OptTensor& weight, double gamma, double lambda, double rho_clip,
                       double c_clip, double rho_pg_clip, std::optional<double> scale) {
        return VtraceFn::apply(target, behaviour, action, value, reward, weight, gamma, lambda, rho_clip, c_clip,
                               rho_pg_clip, scale);
    }, py::arg("target_output"), py::arg("behaviour_output"), py::arg("action"), py::arg("value"), py::arg("reward"),
          py::arg("weight") = py::none(), py::arg("gamma") = 0.99, py::arg("lambda_") = 0.95,
          py::arg("rho_clip_ratio") = 1.0, py::arg("c_clip_ratio") = 1.0, py::arg("rho_pg_clip_ratio") = 1.0,
          py::arg("scale") = py::none());
    m.def("upgo", [](const Tensor& target, const Tensor& rho, const Tensor& action, const Tensor& reward,
                     const Tensor& value, std::optional<double> scale) {
        return UpgoFn::apply(target, rho, action, reward, value, scale);
    }, py::arg("target_output"), py::arg("rhos"), py::arg("action"), py::arg("rewards"), py::arg("bootstrap_values"),
          py::arg("scale") = py::none());
    m.def("ppo", [](const Tensor& ln, const Tensor& lo, const Tensor& action, const Tensor& vn, const Tensor& vo,
                    const Tensor& adv, const Tensor& ret, const OptTensor& weight, double clip_ratio, bool use_value_clip,
                    double dual_clip, std::optional<double> scale) {
        return PpoFn::apply(ln, lo, action, vn, vo, adv, ret, weight, clip_ratio, use_value_clip, dual_clip, scale);
    }, py::arg("logits_new"), py::arg("logits_old"), py::arg("action"), py::arg("value_new"), py::arg("value_old"),
          py::arg("adv"), py::arg("return_"), py::arg("weight") = py::none(), py::arg("clip_ratio") = 0.2,
          py::arg("use_value_clip") = true, py::arg("dual_clip") = 0.0, py::arg("scale") = py::none());
    m.def("q_nstep_td", [](const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                           const Tensor& reward, const Tensor& done, const OptTensor& weight, double gamma, bool rescale,
                           std::optional<double> scale) {
        return rescale ? QNStepFn<1>::apply(q, nq, action, naction, reward, done, weight, gamma, scale)
                       : QNStepFn<0>::apply(q, nq, action, naction, reward, done, weight, gamma, scale);
    }, py::arg("q"), py::arg("next_n_q"), py::arg("action"), py::arg("next_n_action"), py::arg("reward"), py::arg("done"),
          py::arg("weight"), py::arg("gamma"), py::arg("rescale") = false, py::arg("scale") = py::none());
    m.def("dist_nstep_td", [](const Tensor& dist, const Tensor& ndist, const Tensor& action, const Tensor& naction,
                              const Tensor& reward, const Tensor& done, const OptTensor& weight, double gamma,
                              double v_min, double v_max, std::optional<double> scale) {
        return DistFn::apply(dist, ndist, action, naction, reward, done, weight, gamma, v_min, v_max, scale);
    }, py::arg("dist"), py::arg("next_n_dist"), py::arg("action"), py::arg("next_n_action"), py::arg("reward"),
          py::arg("done"), py::arg("weight"), py::arg("gamma"), py::arg("v_min"), py::arg("v_max"),
          py::arg("scale") = py::none());
    m.def("iqn_nstep_td", [](const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                             const Tensor& reward, const Tensor& done, const Tensor& rq, const OptTensor& weight,
                             const OptTensor& vg, double gamma, double kappa, std::optional<double> scale, bool bnt) {
        return IqnFn::apply(q, nq, action, naction, reward, done, rq, weight, vg, gamma, kappa, scale, bnt);
    }, py::arg("q"), py::arg("next_n_q"), py::arg("action"), py::arg("next_n_action"), py::arg("reward"), py::arg("done"),
          py::arg("replay_quantiles"), py::arg("weight") = py::none(), py::arg("value_gamma") = py::none(),
          py::arg("gamma") = 0.99, py::arg("kappa") = 1.0, py::arg("scale") = py::none(), py::arg("bnt") = false);
    m.def("qrdqn_nstep_td", [](const Tensor& q, const Tensor& nq, const Tensor& action, const Tensor& naction,
                               const Tensor& reward, const Tensor& done, const OptTensor& weight, const OptTensor& vg,
                               double gamma, std::optional<double> tau_value, std::optional<double> scale) {
        return QrFn::apply(q, nq, action, naction, reward, done, weight, vg, gamma, tau_value, scale);
    }, py::arg("q"), py::arg("next_n_q"), py::arg("action"), py::arg("next_n_action"), py::arg("reward"), py::arg("done"),
          py::arg("weight") = py::none(), py::arg("value_gamma") = py::none(), py::arg("gamma") = 0.99,
          py::arg("tau_value") = py::none(), py::arg("scale") = py::none());
}
