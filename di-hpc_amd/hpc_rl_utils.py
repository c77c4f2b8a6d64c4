"""``hpc_rl_utils`` -- the reference's native extension module name (src/rl_utils/entry.cpp:8-39),
re-implemented as a thin binding over the C ABI of libhpc_rll_hip.so (include/hpc_rll_hip.h).

Every function keeps the reference's calling convention
``Fn(inputs: list[Tensor], outputs: list[Tensor], scalars...)``.  For GAE, TD-lambda, the q / dist / IQN /
QR-DQN n-step TD ops the positional tensor order is the reference's.  For V-trace, UPGO and PPO the lists are
shorter than the reference's: this library recomputes the softmax in backward instead of saving three
(rows,N) buffers, so backward takes the logits and ONE workspace tensor (documented per function).
Launches go to torch's CURRENT stream of the tensors' device (the reference uses legacy stream 0:
SURVEY.md A.10).  Unlike the reference, arguments are validated and HIP errors surface as RuntimeError.

``scale``: every scalar-loss forward takes an optional ``scale`` (default 1/local element count).  A
data-parallel caller passes 1/GLOBAL count and sums the per-rank losses with one all-reduce (hpc_rll.dist).
"""
import operator

import numpy as np
import torch

from hpc_rll import _native as N

_lib = N.lib
F32, I64 = torch.float32, torch.int64

# (device index, T, gamma, lambda) -> coef tensor.  The table depends only on these.
_gae_coef_cache = {}


def _scratch(n, dev):
    return torch.empty(int(n), dtype=F32, device=dev)


def _opt(t, name, shape, dev):
    return None if t is None else N.require(t, name, shape=shape, device=dev)


def gae_coef(T: int, gamma: float, lambda_: float, device: torch.device) -> torch.Tensor:
    key = (device.index, int(T), float(gamma), float(lambda_))
    c = _gae_coef_cache.get(key)
    if c is None:
        c = torch.empty(max(int(T), 1), dtype=F32, device=device)
        N.call("hpc_rll_gae_coef", device, c.data_ptr(), int(T), float(gamma), float(lambda_))
        # the table is cached and may later be consumed on ANY stream: make it globally visible once (a one-off
        # ~20 us host wait on a cache miss; skipped while a HIP graph is being captured, where the fill is captured too)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(device).synchronize()
        if len(_gae_coef_cache) > 64:
            _gae_coef_cache.clear()
        _gae_coef_cache[key] = c
    return c


# ------------------------------------------------------------------------------------------------ GAE
def GaeForward(inputs, outputs, gamma: float, lambda_: float) -> None:
    """inputs = [value (T+1,B), reward (T,B)], outputs = [adv (T,B)].  Reference: src/rl_utils/gae.cu:8-28."""
    value, reward = inputs
    (adv,) = outputs
    N.require(reward, "reward")
    if reward.dim() != 2:
        raise RuntimeError(f"reward: expected (T,B), got {tuple(reward.shape)}")
    T, B = reward.shape
    dev = reward.device
    N.require(value, "value", shape=(T + 1, B), device=dev)
    N.require(adv, "adv", shape=(T, B), device=dev)
    coef = gae_coef(T, gamma, lambda_, dev)
    N.call("hpc_rll_gae_forward", dev, value.data_ptr(), reward.data_ptr(), adv.data_ptr(), coef.data_ptr(), T, B,
           float(gamma))


def GaeBackward(inputs, outputs, gamma: float, lambda_: float) -> None:
    """inputs = [grad_adv (T,B)], outputs = [grad_value (T+1,B) or None, grad_reward (T,B) or None].

    New entry (the reference registers no GaeBackward: src/rl_utils/entry.cpp:22): the analytic adjoint of
    hpc_rll.origin.gae."""
    (grad_adv,) = inputs
    grad_value, grad_reward = outputs
    N.require(grad_adv, "grad_adv")
    T, B = grad_adv.shape
    dev = grad_adv.device
    _opt(grad_value, "grad_value", (T + 1, B), dev)
    _opt(grad_reward, "grad_reward", (T, B), dev)
    coef = gae_coef(T, gamma, lambda_, dev)
    N.call("hpc_rll_gae_backward", dev, grad_adv.data_ptr(), N.ptr(grad_value), N.ptr(grad_reward), coef.data_ptr(),
           T, B, float(gamma))


# ------------------------------------------------------------------------------------------------ TD(lambda)
def TdLambdaForward(inputs, outputs, gamma: float, lambda_: float, scale=None) -> None:
    """inputs = [value (T+1,B), reward (T,B), weight (None | (B,) | (T,B))], outputs = [loss (1,), grad_buf (T,B)].
    Reference: src/rl_utils/td_lambda.cu:8-33 (which reads weight as (T,B) whatever its shape: SURVEY.md A.2)."""
    value, reward, weight = inputs
    loss, grad_buf = outputs
    N.require(reward, "reward")
    T, B = reward.shape
    dev = reward.device
    N.require(value, "value", shape=(T + 1, B), device=dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(grad_buf, "grad_buf", shape=(T, B), device=dev)
    mode = 0
    if weight is not None:
        N.require(weight, "weight", device=dev)
        if tuple(weight.shape) == (T, B):
            mode = 2
        elif tuple(weight.shape) == (B,):
            mode = 1
        else:
            raise RuntimeError(f"weight: shape {tuple(weight.shape)}, expected {(T, B)} or {(B,)}")
    partials = _scratch(_lib.hpc_rll_partials_floats(B), dev)
    N.call("hpc_rll_td_lambda_forward", dev, value.data_ptr(), reward.data_ptr(), N.ptr(weight), mode,
           loss.data_ptr(), grad_buf.data_ptr(), partials.data_ptr(), T, B, float(gamma), float(lambda_),
           float(1.0 / max(T * B, 1) if scale is None else scale))


def TdLambdaBackward(inputs, outputs) -> None:
    """inputs = [grad_loss (scalar tensor), grad_buf (T,B)], outputs = [grad_value (T+1,B)].  td_lambda.cu:35-52."""
    grad_loss, grad_buf = inputs
    (grad_value,) = outputs
    N.require(grad_buf, "grad_buf")
    T, B = grad_buf.shape
    dev = grad_buf.device
    g = N.require(grad_loss.reshape(1), "grad_loss", device=dev)
    N.require(grad_value, "grad_value", shape=(T + 1, B), device=dev)
    N.call("hpc_rll_td_lambda_backward", dev, g.data_ptr(), grad_buf.data_ptr(), grad_value.data_ptr(), T, B)


# ------------------------------------------------------------------------------------------------ V-trace
def vtrace_workspace(T, B, dev):
    return _scratch(_lib.hpc_rll_vtrace_workspace_floats(int(T), int(B)), dev)


def VTraceForward(inputs, outputs, gamma, lambda_, rho_clip_ratio, c_clip_ratio, rho_pg_clip_ratio, scale=None) -> None:
    """inputs = [target_output (T,B,N), behaviour_output (T,B,N), action (T,B) int64, value (T+1,B), reward (T,B),
    weight (T,B) or None]; outputs = [losses (3,) = policy/value/entropy, ws = vtrace_workspace(T,B)].
    Reference: src/rl_utils/vtrace.cu:8-86 (6 inputs / 12 outputs; the 9 scratch outputs collapse into ``ws``)."""
    target, behaviour, action, value, reward, weight = inputs
    losses, ws = outputs
    N.require(target, "target_output")
    if target.dim() != 3:
        raise RuntimeError(f"target_output: expected (T,B,N), got {tuple(target.shape)}")
    T, B, NA = target.shape
    dev = target.device
    N.require(behaviour, "behaviour_output", shape=(T, B, NA), device=dev)
    N.require(action, "action", dtype=I64, shape=(T, B), device=dev)
    N.require(value, "value", shape=(T + 1, B), device=dev)
    N.require(reward, "reward", shape=(T, B), device=dev)
    _opt(weight, "weight", (T, B), dev)
    N.require(losses, "losses", shape=(3,), device=dev)
    N.require(ws, "ws", shape=(_lib.hpc_rll_vtrace_workspace_floats(T, B),), device=dev)
    N.call("hpc_rll_vtrace_forward", dev, target.data_ptr(), behaviour.data_ptr(), action.data_ptr(), value.data_ptr(),
           reward.data_ptr(), N.ptr(weight), losses.data_ptr(), ws.data_ptr(), T, B, NA, float(gamma), float(lambda_),
           float(rho_clip_ratio), float(c_clip_ratio), float(rho_pg_clip_ratio),
           float(1.0 / max(T * B, 1) if scale is None else scale))


def VTraceBackward(inputs, outputs) -> None:
    """inputs = [g_policy, g_value, g_entropy (scalar tensors), target_output, action, ws];
    outputs = [grad_target_output (T,B,N) or None, grad_value (T+1,B) or None].  vtrace.cu:88-130."""
    g_pg, g_v, g_ent, target, action, ws = inputs
    grad_target, grad_value = outputs
    T, B, NA = target.shape
    dev = target.device
    gs = [N.require(g.reshape(1), "grad", device=dev) for g in (g_pg, g_v, g_ent)]
    _opt(grad_target, "grad_target_output", (T, B, NA), dev)
    _opt(grad_value, "grad_value", (T + 1, B), dev)
    N.call("hpc_rll_vtrace_backward", dev, gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), target.data_ptr(),
           action.data_ptr(), ws.data_ptr(), N.ptr(grad_target), N.ptr(grad_value), T, B, NA)


# ------------------------------------------------------------------------------------------------ UPGO
def upgo_workspace(T, B, dev):
    return _scratch(_lib.hpc_rll_upgo_workspace_floats(int(T), int(B)), dev)


def UpgoForward(inputs, outputs, scale=None) -> None:
    """inputs = [target_output (T,B,N), rho (T,B), action (T,B) int64, reward (T,B), value (T+1,B)];
    outputs = [loss (1,), ws = upgo_workspace(T,B)].  Reference: src/rl_utils/upgo.cu:8-48."""
    target, rho, action, reward, value = inputs
    loss, ws = outputs
    N.require(target, "target_output")
    T, B, NA = target.shape
    dev = target.device
    N.require(rho, "rhos", shape=(T, B), device=dev)
    N.require(action, "action", dtype=I64, shape=(T, B), device=dev)
    N.require(reward, "rewards", shape=(T, B), device=dev)
    N.require(value, "bootstrap_values", shape=(T + 1, B), device=dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(ws, "ws", shape=(_lib.hpc_rll_upgo_workspace_floats(T, B),), device=dev)
    N.call("hpc_rll_upgo_forward", dev, target.data_ptr(), rho.data_ptr(), action.data_ptr(), reward.data_ptr(),
           value.data_ptr(), loss.data_ptr(), ws.data_ptr(), T, B, NA,
           float(1.0 / max(T * B, 1) if scale is None else scale))


def UpgoBackward(inputs, outputs) -> None:
    """inputs = [grad_loss, target_output, action, ws]; outputs = [grad_target_output].  upgo.cu:50-70."""
    g, target, action, ws = inputs
    (grad_target,) = outputs
    T, B, NA = target.shape
    dev = target.device
    g = N.require(g.reshape(1), "grad_loss", device=dev)
    N.require(grad_target, "grad_target_output", shape=(T, B, NA), device=dev)
    N.call("hpc_rll_upgo_backward", dev, g.data_ptr(), target.data_ptr(), action.data_ptr(), ws.data_ptr(),
           grad_target.data_ptr(), T, B, NA)


# ------------------------------------------------------------------------------------------------ PPO
def ppo_workspace(B, dev):
    return _scratch(_lib.hpc_rll_ppo_workspace_floats(int(B)), dev)


def PPOForward(inputs, outputs, use_value_clip: bool, clip_ratio: float, dual_clip: float, scale=None) -> None:
    """inputs = [logits_new (B,N), logits_old (B,N), action (B,) int64, value_new, value_old, adv, return_ (B,),
    weight (B,) or None]; outputs = [out5 (5,) = policy, value, entropy, approx_kl, clipfrac; ws = ppo_workspace(B)].
    ``dual_clip`` < 1 (the reference passes 0.0 for None) disables dual clipping.  Reference: src/rl_utils/ppo.cu:8-75."""
    ln, lo, action, vn, vo, adv, ret, weight = inputs
    out5, ws = outputs
    N.require(ln, "logits_new")
    B, NA = ln.shape
    dev = ln.device
    N.require(lo, "logits_old", shape=(B, NA), device=dev)
    N.require(action, "action", dtype=I64, shape=(B,), device=dev)
    for t, nm in ((vn, "value_new"), (vo, "value_old"), (adv, "adv"), (ret, "return_")):
        N.require(t, nm, shape=(B,), device=dev)
    _opt(weight, "weight", (B,), dev)
    N.require(out5, "out5", shape=(5,), device=dev)
    N.require(ws, "ws", shape=(_lib.hpc_rll_ppo_workspace_floats(B),), device=dev)
    N.call("hpc_rll_ppo_forward", dev, ln.data_ptr(), lo.data_ptr(), action.data_ptr(), vn.data_ptr(), vo.data_ptr(),
           adv.data_ptr(), ret.data_ptr(), N.ptr(weight), out5.data_ptr(), ws.data_ptr(), B, NA, float(clip_ratio),
           int(bool(use_value_clip)), float(dual_clip), float(1.0 / max(B, 1) if scale is None else scale))


def PPOBackward(inputs, outputs) -> None:
    """inputs = [g_policy, g_value, g_entropy, logits_new, action, ws]; outputs = [grad_logits_new or None,
    grad_value_new or None].  Reference: src/rl_utils/ppo.cu:77-111."""
    g_p, g_v, g_e, ln, action, ws = inputs
    grad_logits, grad_value = outputs
    B, NA = ln.shape
    dev = ln.device
    gs = [N.require(g.reshape(1), "grad", device=dev) for g in (g_p, g_v, g_e)]
    _opt(grad_logits, "grad_logits_new", (B, NA), dev)
    _opt(grad_value, "grad_value_new", (B,), dev)
    N.call("hpc_rll_ppo_backward", dev, gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), ln.data_ptr(),
           action.data_ptr(), ws.data_ptr(), N.ptr(grad_logits), N.ptr(grad_value), B, NA)


# ------------------------------------------------------------------------------------------------ q n-step TD
def _q_nstep_forward(inputs, outputs, gamma, rescale, scale):
    q, nq, action, naction, reward, done, weight = inputs
    td_err, loss, grad_buf = outputs
    N.require(q, "q")
    B, NA = q.shape
    dev = q.device
    N.require(nq, "next_n_q", shape=(B, NA), device=dev)
    N.require(action, "action", dtype=I64, shape=(B,), device=dev)
    N.require(naction, "next_n_action", dtype=I64, shape=(B,), device=dev)
    N.require(reward, "reward", device=dev)
    if reward.dim() != 2 or reward.shape[1] != B:
        raise RuntimeError(f"reward: expected (nstep,{B}), got {tuple(reward.shape)}")
    nstep = reward.shape[0]
    N.require(done, "done", shape=(B,), device=dev)
    _opt(weight, "weight", (B,), dev)
    N.require(td_err, "td_err", shape=(B,), device=dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(grad_buf, "grad_buf", shape=(B,), device=dev)
    partials = _scratch(_lib.hpc_rll_partials_floats(B), dev)
    N.call("hpc_rll_q_nstep_td_forward", dev, q.data_ptr(), nq.data_ptr(), action.data_ptr(), naction.data_ptr(),
           reward.data_ptr(), done.data_ptr(), N.ptr(weight), loss.data_ptr(), td_err.data_ptr(), grad_buf.data_ptr(),
           partials.data_ptr(), nstep, B, NA, float(gamma), int(rescale),
           float(1.0 / max(B, 1) if scale is None else scale))


def _q_nstep_backward(inputs, outputs):
    grad_loss, grad_buf, action = inputs
    (grad_q,) = outputs
    N.require(grad_q, "grad_q")
    B, NA = grad_q.shape
    dev = grad_q.device
    g = N.require(grad_loss.reshape(1), "grad_loss", device=dev)
    N.call("hpc_rll_q_nstep_td_backward", dev, g.data_ptr(), grad_buf.data_ptr(), action.data_ptr(), grad_q.data_ptr(),
           B, NA)


def QNStepTdForward(inputs, outputs, gamma: float, scale=None) -> None:
    """inputs = [q, next_n_q (B,N), action, next_n_action (B,) int64, reward (nstep,B), done (B,), weight (B,)|None];
    outputs = [td_err (B,), loss (1,), grad_buf (B,)].  Reference: src/rl_utils/q_nstep_td.cu:8-39."""
    _q_nstep_forward(inputs, outputs, gamma, 0, scale)


def QNStepTdBackward(inputs, outputs) -> None:
    """inputs = [grad_loss, grad_buf (B,), action]; outputs = [grad_q (B,N)].  q_nstep_td.cu:41-63."""
    _q_nstep_backward(inputs, outputs)


def QNStepTdRescaleForward(inputs, outputs, gamma: float, scale=None) -> None:
    """Same lists as QNStepTdForward, with the h / h^-1 value rescaling.  src/rl_utils/q_nstep_td_rescale.cu:8-39."""
    _q_nstep_forward(inputs, outputs, gamma, 1, scale)


def QNStepTdRescaleBackward(inputs, outputs) -> None:
    _q_nstep_backward(inputs, outputs)


# ------------------------------------------------------------------------------------------------ dist (C51)
def DistNStepTdForward(inputs, outputs, gamma: float, v_min: float, v_max: float, scale=None) -> None:
    """inputs = [dist, next_n_dist (B,N,n_atom), action, next_n_action (B,), reward (nstep,B), done (B,),
    weight (B,)|None]; outputs = [td_err (B,), loss (1,), buf (B,n_atom)].  src/rl_utils/dist_nstep_td.cu:8-72
    (whose buf is (B + B*n_atom,): td target + projected distribution; here buf is the unit gradient)."""
    dist, ndist, action, naction, reward, done, weight = inputs
    td_err, loss, buf = outputs
    N.require(dist, "dist")
    B, NA, n_atom = dist.shape
    dev = dist.device
    N.require(ndist, "next_n_dist", shape=(B, NA, n_atom), device=dev)
    N.require(action, "action", dtype=I64, shape=(B,), device=dev)
    N.require(naction, "next_n_action", dtype=I64, shape=(B,), device=dev)
    N.require(reward, "reward", device=dev)
    nstep = reward.shape[0]
    N.require(done, "done", shape=(B,), device=dev)
    _opt(weight, "weight", (B,), dev)
    N.require(td_err, "td_err", shape=(B,), device=dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(buf, "buf", shape=(B, n_atom), device=dev)
    partials = _scratch(_lib.hpc_rll_partials_floats(B), dev)
    N.call("hpc_rll_dist_nstep_td_forward", dev, dist.data_ptr(), ndist.data_ptr(), action.data_ptr(),
           naction.data_ptr(), reward.data_ptr(), done.data_ptr(), N.ptr(weight), loss.data_ptr(), td_err.data_ptr(),
           buf.data_ptr(), partials.data_ptr(), nstep, B, NA, n_atom, float(gamma), float(v_min), float(v_max),
           float(1.0 / max(B, 1) if scale is None else scale))


def DistNStepTdBackward(inputs, outputs) -> None:
    """inputs = [grad_loss, buf (B,n_atom), action]; outputs = [grad_dist (B,N,n_atom)].  dist_nstep_td.cu:74-98."""
    grad_loss, buf, action = inputs
    (grad_dist,) = outputs
    N.require(grad_dist, "grad_dist")
    B, NA, n_atom = grad_dist.shape
    dev = grad_dist.device
    g = N.require(grad_loss.reshape(1), "grad_loss", device=dev)
    N.call("hpc_rll_dist_nstep_td_backward", dev, g.data_ptr(), buf.data_ptr(), action.data_ptr(),
           grad_dist.data_ptr(), B, NA, n_atom)


# ------------------------------------------------------------------------------------------------ IQN
def IQNNStepTDErrorForward(inputs, outputs, gamma: float, kappa: float, scale=None) -> None:
    """inputs = [q (tau,B,N), next_n_q (tau',B,N), action, next_n_action (B,), reward (nstep,B), done (B,),
    replay_quantiles (tau,B), weight (B,)|None, value_gamma (B,)|None]; outputs = [loss (1,), td_err (B,),
    grad_buf (B,tau)].  Reference: src/rl_utils/iqn_nstep_td_error.cu:8-72 (3 (B,tau',tau) scratch outputs dropped)."""
    q, nq, action, naction, reward, done, rq, weight, vg = inputs
    loss, td_err, grad_buf = outputs
    N.require(q, "q")
    tau, B, NA = q.shape
    dev = q.device
    N.require(nq, "next_n_q", device=dev)
    tau_p = nq.shape[0]
    if tuple(nq.shape[1:]) != (B, NA):
        raise RuntimeError(f"next_n_q: shape {tuple(nq.shape)}, expected (tau',{B},{NA})")
    N.require(action, "action", dtype=I64, shape=(B,), device=dev)
    N.require(naction, "next_n_action", dtype=I64, shape=(B,), device=dev)
    N.require(reward, "reward", device=dev)
    nstep = reward.shape[0]
    N.require(done, "done", shape=(B,), device=dev)
    N.require(rq, "replay_quantiles", device=dev)
    if rq.numel() != tau * B:
        raise RuntimeError(f"replay_quantiles: {tuple(rq.shape)} does not hold tau*B = {tau * B} values")
    _opt(weight, "weight", (B,), dev)
    _opt(vg, "value_gamma", (B,), dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(td_err, "td_err", shape=(B,), device=dev)
    N.require(grad_buf, "grad_buf", shape=(B, tau), device=dev)
    partials = _scratch(_lib.hpc_rll_partials_floats(B), dev)
    N.call("hpc_rll_iqn_nstep_td_forward", dev, q.data_ptr(), nq.data_ptr(), action.data_ptr(), naction.data_ptr(),
           reward.data_ptr(), done.data_ptr(), rq.data_ptr(), N.ptr(weight), N.ptr(vg), loss.data_ptr(),
           td_err.data_ptr(), grad_buf.data_ptr(), partials.data_ptr(), tau, tau_p, nstep, B, NA, float(gamma),
           float(kappa), float(1.0 / max(B, 1) if scale is None else scale))


def IQNNStepTDErrorBackward(inputs, outputs) -> None:
    """inputs = [grad_loss, grad_buf (B,tau), action]; outputs = [grad_q (tau,B,N)].  iqn_nstep_td_error.cu:74-104."""
    grad_loss, grad_buf, action = inputs
    (grad_q,) = outputs
    N.require(grad_q, "grad_q")
    tau, B, NA = grad_q.shape
    dev = grad_q.device
    g = N.require(grad_loss.reshape(1), "grad_loss", device=dev)
    N.call("hpc_rll_iqn_nstep_td_backward", dev, g.data_ptr(), grad_buf.data_ptr(), action.data_ptr(),
           grad_q.data_ptr(), tau, B, NA)


# ------------------------------------------------------------------------------------------------ QR-DQN
def QRDQNNStepTDErrorForward(inputs, outputs, gamma: float, tau_value=None, scale=None) -> None:
    """inputs = [q, next_n_q (B,N,tau), action, next_n_action (B,), reward (nstep,B), done (B,), weight (B,)|None,
    value_gamma (B,)|None]; outputs = [loss (1,), td_err (B,), grad_buf (B,tau)].  ``tau_value`` is the ``tau`` the
    oracle is called with; default = the integer count, which is what the reference kernel hard-codes
    (qrdqn_nstep_td_error_kernel.h:60) and its test passes.  Reference: src/rl_utils/qrdqn_nstep_td_error.cu:8-68."""
    q, nq, action, naction, reward, done, weight, vg = inputs
    loss, td_err, grad_buf = outputs
    N.require(q, "q")
    B, NA, tau = q.shape
    dev = q.device
    N.require(nq, "next_n_q", shape=(B, NA, tau), device=dev)
    N.require(action, "action", dtype=I64, shape=(B,), device=dev)
    N.require(naction, "next_n_action", dtype=I64, shape=(B,), device=dev)
    N.require(reward, "reward", device=dev)
    nstep = reward.shape[0]
    N.require(done, "done", shape=(B,), device=dev)
    _opt(weight, "weight", (B,), dev)
    _opt(vg, "value_gamma", (B,), dev)
    N.require(loss, "loss", shape=(1,), device=dev)
    N.require(td_err, "td_err", shape=(B,), device=dev)
    N.require(grad_buf, "grad_buf", shape=(B, tau), device=dev)
    partials = _scratch(_lib.hpc_rll_partials_floats(B), dev)
    N.call("hpc_rll_qrdqn_nstep_td_forward", dev, q.data_ptr(), nq.data_ptr(), action.data_ptr(), naction.data_ptr(),
           reward.data_ptr(), done.data_ptr(), N.ptr(weight), N.ptr(vg), loss.data_ptr(), td_err.data_ptr(),
           grad_buf.data_ptr(), partials.data_ptr(), tau, nstep, B, NA, float(gamma),
           float(tau if tau_value is None else tau_value), float(1.0 / max(B, 1) if scale is None else scale))


def QRDQNNStepTDErrorBackward(inputs, outputs) -> None:
    """inputs = [grad_loss, grad_buf (B,tau), action]; outputs = [grad_q (B,N,tau)].  qrdqn_nstep_td_error.cu:70-99."""
    grad_loss, grad_buf, action = inputs
    (grad_q,) = outputs
    N.require(grad_q, "grad_q")
    B, NA, tau = grad_q.shape
    dev = grad_q.device
    g = N.require(grad_loss.reshape(1), "grad_loss", device=dev)
    N.call("hpc_rll_qrdqn_nstep_td_backward", dev, g.data_ptr(), grad_buf.data_ptr(), action.data_ptr(),
           grad_q.data_ptr(), tau, B, NA)


# ------------------------------------------------------------------------------------------------ Pad / Unpad
_get_dtype = operator.attrgetter("dtype")
_get_device = operator.attrgetter("device")
_get_shape = operator.attrgetter("shape")


def _dims3(shape):
    """(d0,d1,d2) with the tensor's own axes right-aligned: a rank-1 tensor is (1,1,L)."""
    s = tuple(int(v) for v in shape)
    return (1,) * (3 - len(s)) + s


def _device_table(rows, dev):
    """n x 4 int64 host table -> device (pinned staging, async copy on the current stream)."""
    t = torch.tensor(rows, dtype=I64).reshape(-1, 4)
    if t.numel() == 0:
        return t.to(dev)
    return t.pin_memory().to(dev, non_blocking=True)


def _pad_table(inputs, rank):
    """(n,4) int64 host table [data_ptr, d0, d1, d2] (own axes right-aligned, leading ones) and the (n,rank) shapes of a
    list of rank-`rank` tensors.  Pure host logic (works on CPU tensors: covered by the CPU test tier).
    t.shape builds a torch.Size per tensor (~0.9 us); numel()/size(d) return plain ints (0.12 / 0.3 us)."""
    n = len(inputs)
    if rank == 1:
        shapes = np.fromiter(map(torch.Tensor.numel, inputs), dtype=np.int64, count=n).reshape(n, 1)
    else:
        shapes = np.stack([np.fromiter(map(operator.methodcaller("size", d), inputs), dtype=np.int64, count=n)
                           for d in range(rank)], axis=1)
    table = np.ones((n, 4), dtype=np.int64)
    table[:, 0] = np.fromiter(map(torch.Tensor.data_ptr, inputs), dtype=np.int64, count=n)
    table[:, 4 - rank:] = shapes
    return table, shapes


def _unpad_table(shapes, padded_shape, rank):
    """Host table of the inverse: (n,4) int64 [flat offset, d0, d1, d2], per-tensor element counts, offsets (n+1) and the
    (n,rank) shape array, from the flat `shapes` list; raises if a shape does not fit the padded tensor."""
    n = len(shapes) // rank
    sh = np.asarray(shapes, dtype=np.int64).reshape(n, rank)
    lim = np.asarray(padded_shape, dtype=np.int64)
    if n and ((sh < 0).any() or (sh > lim).any()):
        bad = int(np.argmax(((sh < 0) | (sh > lim)).any(axis=1)))
        raise RuntimeError(f"shapes: {tuple(int(v) for v in sh[bad])} does not fit the padded tensor {tuple(padded_shape)}")
    numel = sh.prod(axis=1) if n else np.zeros(0, dtype=np.int64)
    offs = np.concatenate([[0], np.cumsum(numel)]).astype(np.int64)
    table = np.ones((n, 4), dtype=np.int64)
    table[:, 0] = offs[:-1]
    table[:, 4 - rank:] = sh
    return table, numel, offs, sh


def _pad_forward(inputs, value, rank, max_shape=None):
    """Host side of the list-of-tensors API.  A python loop over n tensors costs ~2 us per tensor (0.25 s at n = 131k,
    against a 45 us kernel), so validation and the (pointer, shape) table are built with C-level iteration (map /
    attrgetter / numpy) -- ~3x less host time; the packed entry points in hpc_rll.rl_utils.padding avoid it entirely."""
    n = len(inputs)
    if n == 0:
        raise RuntimeError("Padding: empty input list")
    first = inputs[0]
    if not isinstance(first, torch.Tensor) or not first.is_cuda:
        N.require(first, "inputs[0]")
    dev = first.device
    ok = (set(map(_get_dtype, inputs)) == {F32} and set(map(_get_device, inputs)) == {dev}
          and all(map(torch.Tensor.is_contiguous, inputs)) and set(map(torch.Tensor.dim, inputs)) == {rank})
    if not ok:   # slow path only to produce the precise error message
        for i, t in enumerate(inputs):
            N.require(t, f"inputs[{i}]", device=dev)
            if t.dim() != rank:
                raise RuntimeError(f"inputs[{i}]: rank {t.dim()}, expected {rank}")
    table, shapes = _pad_table(inputs, rank)
    if max_shape is None:
        max_shape = [int(v) for v in shapes.max(axis=0)]
    m = _dims3(max_shape)
    table = torch.from_numpy(table).pin_memory().to(dev, non_blocking=True)
    new_x = torch.empty([n] + list(max_shape), dtype=F32, device=dev)
    mask = torch.empty([n] + list(max_shape), dtype=torch.int32, device=dev)
    N.call("hpc_rll_pad_forward", dev, table.data_ptr(), new_x.data_ptr(), mask.data_ptr(), n, m[0], m[1], m[2],
           int(value))
    return [new_x, mask]


def Pad1DForward(inputs, value: int):
    """list of n (L_i,) tensors -> [new_x (n,maxL) fp32, mask (n,maxL) int32].  Reference: padding.cu:111-140."""
    return _pad_forward(inputs, value, 1)


def Pad2DForward(inputs, value: int):
    """Reference: padding.cu:262-297."""
    return _pad_forward(inputs, value, 2)


def Pad3DForward(inputs, value: int):
    """Reference: padding.cu:417-456."""
    return _pad_forward(inputs, value, 3)


def _group_pad_forward(inputs, group_cnt, max_shape, group_id, group_idx, value, rank):
    """inputs sorted by numel; group g = inputs[group_idx[g]:group_idx[g+1]], padded to max_shape[g*rank:(g+1)*rank].
    One pad launch per group.  Returns [list of new_x, list of mask]."""
    xs, ms = [], []
    for g in range(len(group_cnt)):
        sub = inputs[group_idx[g]:group_idx[g + 1]]
        shape = [int(v) for v in max_shape[g * rank:(g + 1) * rank]]
        x, m = _pad_forward(sub, value, rank, shape)
        xs.append(x)
        ms.append(m)
    return [xs, ms]


def GroupPad1DForward(inputs, group_cnt, max_shape, group_id, group_idx, value: int):
    """Reference: padding.cu:142-226."""
    return _group_pad_forward(inputs, group_cnt, max_shape, group_id, group_idx, value, 1)


def GroupPad2DForward(inputs, group_cnt, max_shape, group_id, group_idx, value: int):
    """Reference: padding.cu:299-379."""
    return _group_pad_forward(inputs, group_cnt, max_shape, group_id, group_idx, value, 2)


def GroupPad3DForward(inputs, group_cnt, max_shape, group_id, group_idx, value: int):
    """Reference: padding.cu:458-541."""
    return _group_pad_forward(inputs, group_cnt, max_shape, group_id, group_idx, value, 3)


def _unpad_forward(x, shapes, rank):
    """x (n, m...) padded; shapes = flat int list (rank ints per tensor, the hpc convention,
    rl_utils/padding.py:101-104).  Returns n tensors that are views of ONE flat buffer."""
    N.require(x, "x")
    if x.dim() != rank + 1:
        raise RuntimeError(f"x: rank {x.dim()}, expected {rank + 1}")
    n = x.shape[0]
    shapes = [int(v) for v in shapes]
    if len(shapes) != n * rank:
        raise RuntimeError(f"shapes: {len(shapes)} ints, expected {n}*{rank}")
    dev = x.device
    table, numel, offs, sh = _unpad_table(shapes, tuple(x.shape[1:]), rank)
    total = int(offs[-1])
    flat = torch.empty(total, dtype=F32, device=dev)
    if n and total:
        m = _dims3(x.shape[1:])
        table = torch.from_numpy(table).pin_memory().to(dev, non_blocking=True)
        N.call("hpc_rll_unpad_forward", dev, x.data_ptr(), table.data_ptr(), flat.data_ptr(), n, total, m[0], m[1], m[2])
    if rank == 1:   # split is one C++ call; the generic path builds n views from python
        return list(torch.split(flat, numel.tolist())) if n else []
    out = []
    for i in range(n):
        out.append(flat[int(offs[i]):int(offs[i + 1])].view(*[int(v) for v in sh[i]]))
    return out


def Unpad1DForward(x, shapes):
    """Reference: padding.cu:228-260."""
    return _unpad_forward(x, shapes, 1)


def Unpad2DForward(x, shapes):
    """Reference: padding.cu:381-415."""
    return _unpad_forward(x, shapes, 2)


def Unpad3DForward(x, shapes):
    """Reference: padding.cu:543-582."""
    return _unpad_forward(x, shapes, 3)


def _split_group(inputs, group, fn, *extra):
    import ctypes
    n = len(inputs)
    rank = inputs[0].dim()
    sizes = (ctypes.c_int32 * (n * rank))(*[int(v) for t in inputs for v in t.shape])
    shapes = (ctypes.c_int32 * (max(group, 1) * rank))()
    pos = (ctypes.c_int32 * (max(group, 1) + 1))()
    ng = fn(ctypes.cast(sizes, ctypes.c_void_p), n, rank, int(group), *extra, ctypes.cast(shapes, ctypes.c_void_p),
            ctypes.cast(pos, ctypes.c_void_p))
    if ng < 0:
        N.check(ng, "split_group")
    res = [[int(shapes[g * rank + d]) for d in range(rank)] for g in range(ng)]
    res.append([int(pos[g]) for g in range(ng + 1)])
    return res


def oracle_split_group(inputs, group: int):
    """Inputs sorted by numel -> [shape_0, ..., shape_{g-1}, positions].  DP minimising the padded element count
    (padding.cu:44-108; same result as hpc_rll/origin/padding.py:11-50 for 1-D lists)."""
    return _split_group(inputs, group, _lib.hpc_rll_oracle_split_group)


def sample_split_group(inputs, group: int, seed=None):
    """Random cuts (padding.cu:8-43).  The reference uses C rand(); here a splitmix64 stream seeded from python's
    ``random`` (or ``seed``), so a run is reproducible under ``random.seed``."""
    import random
    s = random.getrandbits(63) if seed is None else int(seed)
    return _split_group(inputs, group, _lib.hpc_rll_sample_split_group, s)
