// rl_utils_ops.hpp -- launch helpers of the categorical-head losses shared by rl_utils.cpp (fused autograd ops) and
// rl_utils_lists.cpp (the reference's L2 list functions).
#pragma once
#include "common.hpp"

namespace hpc_rll_ext {

struct VtraceDims { int64_t T, B, N; at::Device dev; };
VtraceDims vtrace_check(const Tensor& target, const Tensor& behaviour, const Tensor& action, const Tensor& value,
                        const Tensor& reward, const OptTensor& weight);
Tensor vtrace_workspace(int64_t T, int64_t B, const at::Device& dev);
void vtrace_forward_launch(const VtraceDims& d, const Tensor& target, const Tensor& behaviour, const Tensor& action,
                           const Tensor& value, const Tensor& reward, const OptTensor& weight, const Tensor& losses,
                           const Tensor& ws, double gamma, double lambda, double rho_clip, double c_clip,
                           double rho_pg_clip, std::optional<double> scale);
void vtrace_backward_launch(const Tensor& g_pg, const Tensor& g_v, const Tensor& g_ent, const Tensor& target,
                            const Tensor& action, const Tensor& ws, const Tensor& grad_target, const Tensor& grad_value);

struct UpgoDims { int64_t T, B, N; at::Device dev; };
UpgoDims upgo_check(const Tensor& target, const Tensor& rho, const Tensor& action, const Tensor& reward,
                    const Tensor& value);
Tensor upgo_workspace(int64_t T, int64_t B, const at::Device& dev);
void upgo_forward_launch(const UpgoDims& d, const Tensor& target, const Tensor& rho, const Tensor& action,
                         const Tensor& reward, const Tensor& value, const Tensor& loss, const Tensor& ws,
                         std::optional<double> scale);
void upgo_backward_launch(const Tensor& g, const Tensor& target, const Tensor& action, const Tensor& ws,
                          const Tensor& grad_target);

struct PpoDims { int64_t B, N; at::Device dev; };
PpoDims ppo_check(const Tensor& ln, const Tensor& lo, const Tensor& action, const Tensor& vn, const Tensor& vo,
                  const Tensor& adv, const Tensor& ret, const OptTensor& weight);
Tensor ppo_workspace(int64_t B, const at::Device& dev);
void ppo_forward_launch(const PpoDims& d, const Tensor& ln, const Tensor& lo, const Tensor& action, const Tensor& vn,
                        const Tensor& vo, const Tensor& adv, const Tensor& ret, const OptTensor& weight,
                        const Tensor& out5, const Tensor& ws, bool use_value_clip, double clip_ratio, double dual_clip,
                        std::optional<double> scale);
void ppo_backward_launch(const Tensor& g_p, const Tensor& g_v, const Tensor& g_e, const Tensor& ln, const Tensor& action,
                         const Tensor& ws, const Tensor& grad_logits, const Tensor& grad_value);

}  // namespace hpc_rll_ext
