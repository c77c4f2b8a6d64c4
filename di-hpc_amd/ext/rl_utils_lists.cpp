// rl_utils_lists.cpp -- L2 list functions of the categorical-head losses: VTraceForward/Backward,
// UpgoForward/Backward, PPOForward/Backward (reference: include/hpc/rll/cuda/rl_utils/entry.h:131-165).
//
// Each accepts TWO positional conventions, told apart by the list lengths:
//   * native  -- short lists: backward recomputes the softmax from the logits, so forward emits the loss scalars and
//                ONE workspace tensor (24 B per (t,b) instead of the reference's 12*N B of saved (T,B,N) buffers);
//   * reference -- exactly the lists hpc_rll/rl_utils/{vtrace,upgo,ppo}.py of the reference build
//                (vtrace.py:17-27, upgo.py:10-15, ppo.py:20-30), so an unmodified reference L1 module runs on this
//                L2.  The reference's scratch outputs are left untouched except for the loss scalars; what backward
//                needs is parked with SavedByBuffer under one of the module's own scratch buffers.  Same kernels,
//                same results as the native convention.
#include "common.hpp"
#include "rl_utils_ops.hpp"

namespace hpc_rll_ext {
namespace {

SavedByBuffer& saved() {
    static SavedByBuffer* s = new SavedByBuffer();   // leaked on purpose (tensor destruction order at exit)
    return *s;
}

void need_all(const OptList& in, size_t n, const char* what) {
    for (size_t i = 0; i < n; ++i) TORCH_CHECK(has(in[i]), what, ": inputs[", i, "] is None");
}

// loss scalars computed into one contiguous tensor -> the reference's separate (1,) module buffers
void scatter_scalars(const Tensor& packed, std::initializer_list<const Tensor*> dst) {
    int64_t i = 0;
    for (const Tensor* d : dst) {
        req(*d, "loss output", packed.device());
        TORCH_CHECK(d->numel() == 1, "loss output: expected one element, got ", d->sizes());
        d->view({1}).copy_(packed.narrow(0, i++, 1), /*non_blocking=*/true);
    }
}

// ------------------------------------------------------------------------------------------------------- V-trace
// native:    inputs = [target_output (T,B,N), behaviour_output (T,B,N), action (T,B) int64, value (T+1,B), reward (T,B),
//                      weight (T,B)|None]; outputs = [losses (3,) = policy/value/entropy, ws]
// reference: same inputs; outputs = [target_output_prob, target_output_entropy, target_output_grad_logits,
//            target_output_grad_prob, target_output_grad_entropy, behaviour_output_prob, importance_weights, returns,
//            advantages, pg_loss, value_loss, entropy_loss]          (src/rl_utils/vtrace.cu:8-86)
void VTraceForward(const OptList& in, const TensorList& out, double gamma, double lambda, double rho_clip,
                   double c_clip, double rho_pg_clip, std::optional<double> scale) {
    expect_len(in, 6, "VTraceForward inputs");
    TORCH_CHECK(out.size() == 2 || out.size() == 12, "VTraceForward outputs: expected 2 (native) or 12 (reference) "
                "tensors, got ", out.size());
    need_all(in, 5, "VTraceForward");
    const VtraceDims d = vtrace_check(*in[0], *in[1], *in[2], *in[3], *in[4], in[5]);
    c10::DeviceGuard g(d.dev);
    if (out.size() == 2) {
        req(out[0], "losses", d.dev, {3});
        req(out[1], "ws", d.dev, {hpc_rll_vtrace_workspace_floats((int)d.T, (int)d.B)});
        vtrace_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], in[5], out[0], out[1], gamma, lambda, rho_clip,
                              c_clip, rho_pg_clip, scale);
        return;
    }
    Tensor losses = new_f32({3}, d.dev), ws = vtrace_workspace(d.T, d.B, d.dev);
    vtrace_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], in[5], losses, ws, gamma, lambda, rho_clip, c_clip,
                          rho_pg_clip, scale);
    scatter_scalars(losses, {&out[9], &out[10], &out[11]});
    req(out[2], "target_output_grad_logits", d.dev);
    saved().put(out[2], {{*in[0], *in[2], ws}});
}

// native:    inputs = [g_policy, g_value, g_entropy (scalar tensors), target_output, action, ws];
//            outputs = [grad_target_output (T,B,N)|None, grad_value (T+1,B)|None]
// reference: inputs = [g_policy, g_value, g_entropy, value, action, weight, returns, advantages,
//            target_output_grad_logits, target_output_grad_prob, target_output_grad_entropy];
//            outputs = [grad_value, grad_target_output]                (vtrace.py:27-45, vtrace.cu:88-130)
void VTraceBackward(const OptList& in, const OptList& out) {
    TORCH_CHECK(in.size() == 6 || in.size() == 11, "VTraceBackward inputs: expected 6 (native) or 11 (reference)");
    expect_len(out, 2, "VTraceBackward outputs");
    need_all(in, 3, "VTraceBackward");
    Tensor target, action, ws, grad_target, grad_value;
    if (in.size() == 6) {
        need_all(in, 6, "VTraceBackward");
        target = *in[3]; action = *in[4]; ws = *in[5];
        if (has(out[0])) grad_target = *out[0];
        if (has(out[1])) grad_value = *out[1];
    } else {
        TORCH_CHECK(has(in[8]), "VTraceBackward: target_output_grad_logits is None");
        auto e = saved().get(*in[8], "VTraceBackward");
        target = e.tensors[0]; action = e.tensors[1]; ws = e.tensors[2];
        if (has(out[0])) grad_value = *out[0];
        if (has(out[1])) grad_target = *out[1];
    }
    req(target, "target_output");
    TORCH_CHECK(target.dim() == 3, "target_output: expected (T,B,N)");
    const at::Device dev = target.device();
    const int64_t T = target.size(0), B = target.size(1), N = target.size(2);
    req(action, "action", dev, {T, B}, at::kLong);
    req(ws, "ws", dev, {hpc_rll_vtrace_workspace_floats((int)T, (int)B)});
    if (grad_target.defined()) req(grad_target, "grad_target_output", dev, {T, B, N});
    if (grad_value.defined()) req(grad_value, "grad_value", dev, {T + 1, B});
    c10::DeviceGuard g(dev);
    vtrace_backward_launch(grad1(*in[0], dev, "grad_policy_loss"), grad1(*in[1], dev, "grad_value_loss"),
                           grad1(*in[2], dev, "grad_entropy_loss"), target, action, ws, grad_target, grad_value);
}

// ---------------------------------------------------------------------------------------------------------- UPGO
// native:    inputs = [target_output (T,B,N), rho (T,B), action (T,B) int64, reward (T,B), value (T+1,B)];
//            outputs = [loss (1,), ws]
// reference: same inputs; outputs = [advantage, metric, loss, grad_buf (T,B,N)]   (upgo.py:10-11, upgo.cu:8-48)
void UpgoForward(const TensorList& in, const TensorList& out, std::optional<double> scale) {
    expect_len(in, 5, "UpgoForward inputs");
    TORCH_CHECK(out.size() == 2 || out.size() == 4, "UpgoForward outputs: expected 2 (native) or 4 (reference)");
    const UpgoDims d = upgo_check(in[0], in[1], in[2], in[3], in[4]);
    c10::DeviceGuard g(d.dev);
    if (out.size() == 2) {
        req(out[0], "loss", d.dev, {1});
        req(out[1], "ws", d.dev, {hpc_rll_upgo_workspace_floats((int)d.T, (int)d.B)});
        upgo_forward_launch(d, in[0], in[1], in[2], in[3], in[4], out[0], out[1], scale);
        return;
    }
    req(out[2], "loss", d.dev, {1});
    req(out[3], "grad_buf", d.dev);
    Tensor ws = upgo_workspace(d.T, d.B, d.dev);
    upgo_forward_launch(d, in[0], in[1], in[2], in[3], in[4], out[2], ws, scale);
    saved().put(out[3], {{in[0], in[2], ws}});
}

// native:    inputs = [grad_loss, target_output, action, ws]; outputs = [grad_target_output]
// reference: inputs = [grad_loss, grad_buf, advantage];       outputs = [grad_target_output]   (upgo.py:14-26)
void UpgoBackward(const TensorList& in, const TensorList& out) {
    TORCH_CHECK(in.size() == 4 || in.size() == 3, "UpgoBackward inputs: expected 4 (native) or 3 (reference)");
    expect_len(out, 1, "UpgoBackward outputs");
    Tensor target, action, ws;
    if (in.size() == 4) {
        target = in[1]; action = in[2]; ws = in[3];
    } else {
        auto e = saved().get(in[1], "UpgoBackward");
        target = e.tensors[0]; action = e.tensors[1]; ws = e.tensors[2];
    }
    req(target, "target_output");
    TORCH_CHECK(target.dim() == 3, "target_output: expected (T,B,N)");
    const at::Device dev = target.device();
    const int64_t T = target.size(0), B = target.size(1), N = target.size(2);
    req(action, "action", dev, {T, B}, at::kLong);
    req(ws, "ws", dev, {hpc_rll_upgo_workspace_floats((int)T, (int)B)});
    req(out[0], "grad_target_output", dev, {T, B, N});
    c10::DeviceGuard g(dev);
    upgo_backward_launch(grad1(in[0], dev, "grad_loss"), target, action, ws, out[0]);
}

// ----------------------------------------------------------------------------------------------------------- PPO
// native:    inputs = [logits_new (B,N), logits_old (B,N), action (B,) int64, value_new, value_old, adv, return_ (B,),
//            weight (B,)|None]; outputs = [out5 (5,) = policy, value, entropy, approx_kl, clipfrac; ws]
// reference: same inputs; outputs = [logits_new_prob, logits_new_entropy, logits_new_grad_logits,
//            logits_new_grad_prob, logits_new_grad_entropy, logit_old_prob, grad_policy_loss_buf, grad_value_loss_buf,
//            grad_entropy_loss_buf, policy_loss, value_loss, entropy_loss, approx_kl, clipfrac]  (ppo.py:20-24)
// `dual_clip` < 1 (the reference passes 0.0 for None, ppo.py:136-137) disables dual clipping.  src/rl_utils/ppo.cu:8-75.
void PPOForward(const OptList& in, const TensorList& out, bool use_value_clip, double clip_ratio, double dual_clip,
                std::optional<double> scale) {
    expect_len(in, 8, "PPOForward inputs");
    TORCH_CHECK(out.size() == 2 || out.size() == 14, "PPOForward outputs: expected 2 (native) or 14 (reference)");
    need_all(in, 7, "PPOForward");
    const PpoDims d = ppo_check(*in[0], *in[1], *in[2], *in[3], *in[4], *in[5], *in[6], in[7]);
    c10::DeviceGuard g(d.dev);
    if (out.size() == 2) {
        req(out[0], "out5", d.dev, {5});
        req(out[1], "ws", d.dev, {hpc_rll_ppo_workspace_floats((int)d.B)});
        ppo_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], *in[6], in[7], out[0], out[1],
                           use_value_clip, clip_ratio, dual_clip, scale);
        return;
    }
    Tensor out5 = new_f32({5}, d.dev), ws = ppo_workspace(d.B, d.dev);
    ppo_forward_launch(d, *in[0], *in[1], *in[2], *in[3], *in[4], *in[5], *in[6], in[7], out5, ws, use_value_clip,
                       clip_ratio, dual_clip, scale);
    scatter_scalars(out5, {&out[9], &out[10], &out[11], &out[12], &out[13]});
    req(out[2], "logits_new_grad_logits", d.dev);
    saved().put(out[2], {{*in[0], *in[2], ws}});
}

// native:    inputs = [g_policy, g_value, g_entropy, logits_new, action, ws];
//            outputs = [grad_logits_new (B,N)|None, grad_value_new (B,)|None]
// reference: inputs = [g_policy, g_value, g_entropy, grad_policy_loss_buf, grad_value_loss_buf, grad_entropy_loss_buf,
//            logits_new_grad_logits, logits_new_grad_prob, logits_new_grad_entropy];
//            outputs = [grad_value, grad_logits_new]                  (ppo.py:28-46, ppo.cu:77-111)
void PPOBackward(const OptList& in, const OptList& out) {
    TORCH_CHECK(in.size() == 6 || in.size() == 9, "PPOBackward inputs: expected 6 (native) or 9 (reference)");
    expect_len(out, 2, "PPOBackward outputs");
    need_all(in, 3, "PPOBackward");
    Tensor ln, action, ws, grad_logits, grad_value;
    if (in.size() == 6) {
        need_all(in, 6, "PPOBackward");
        ln = *in[3]; action = *in[4]; ws = *in[5];
        if (has(out[0])) grad_logits = *out[0];
        if (has(out[1])) grad_value = *out[1];
    } else {
        TORCH_CHECK(has(in[6]), "PPOBackward: logits_new_grad_logits is None");
        auto e = saved().get(*in[6], "PPOBackward");
        ln = e.tensors[0]; action = e.tensors[1]; ws = e.tensors[2];
        if (has(out[0])) grad_value = *out[0];
        if (has(out[1])) grad_logits = *out[1];
    }
    req(ln, "logits_new");
    TORCH_CHECK(ln.dim() == 2, "logits_new: expected (B,N)");
    const at::Device dev = ln.device();
    const int64_t B = ln.size(0), N = ln.size(1);
    req(action, "action", dev, {B}, at::kLong);
    req(ws, "ws", dev, {hpc_rll_ppo_workspace_floats((int)B)});
    if (grad_logits.defined()) req(grad_logits, "grad_logits_new", dev, {B, N});
    if (grad_value.defined()) req(grad_value, "grad_value_new", dev, {B});
    c10::DeviceGuard g(dev);
    ppo_backward_launch(grad1(*in[0], dev, "grad_policy_loss"), grad1(*in[1], dev, "grad_value_loss"),
                        grad1(*in[2], dev, "grad_entropy_loss"), ln, action, ws, grad_logits, grad_value);
}

}  // namespace

void bind_loss_lists(pybind11::module_& m) {
    namespace py = pybind11;
    m.def("VTraceForward", &VTraceForward, py::arg("inputs"), py::arg("outputs"), py::arg("gamma"), py::arg("lambda_"),
          py::arg("rho_clip_ratio"), py::arg("c_clip_ratio"), py::arg("rho_pg_clip_ratio"),
          py::arg("scale") = py::none(), "vtrace forward (HIP)");
    m.def("VTraceBackward", &VTraceBackward, "vtrace backward (HIP)");
    m.def("UpgoForward", &UpgoForward, py::arg("inputs"), py::arg("outputs"), py::arg("scale") = py::none(),
          "upgo forward (HIP)");
    m.def("UpgoBackward", &UpgoBackward, "upgo backward (HIP)");
    m.def("PPOForward", &PPOForward, py::arg("inputs"), py::arg("outputs"), py::arg("use_value_clip"),
          py::arg("clip_ratio"), py::arg("dual_clip"), py::arg("scale") = py::none(), "ppo forward (HIP)");
    m.def("PPOBackward", &PPOBackward, "ppo backward (HIP)");
    m.def("vtrace_workspace", [](int64_t T, int64_t B, const at::Device& dev) { return vtrace_workspace(T, B, dev); });
    m.def("upgo_workspace", [](int64_t T, int64_t B, const at::Device& dev) { return upgo_workspace(T, B, dev); });
    m.def("ppo_workspace", [](int64_t B, const at::Device& dev) { return ppo_workspace(B, dev); });
}

}  // namespace hpc_rll_ext
