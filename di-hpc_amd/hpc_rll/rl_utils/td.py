"""``hpc_rll.rl_utils.td`` -- drop-in for /root/reference/hpc_rll/rl_utils/td.py: ``DistNStepTD``, ``TDLambda``,
``QNStepTD``, ``QNStepTDRescale``, ``IQNNStepTDError``, ``QRDQNNStepTDError`` with the reference's constructor and
``forward`` signatures and return values (ctor lines td.py:41,149,224,314,410,522; forward lines
td.py:63,165,242,332,439,549).

Differences: outputs and scratch are allocated per call (the reference returns the same module
buffer every call); inputs are validated; ``weight=None`` works for TD-lambda (the reference reads its (B,) default
buffer as (T,B): SURVEY.md A.2); optional batch-axis data parallelism (``sharded=True``, see hpc_rll.dist).
"""
from typing import Optional

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp


def _new(shape, ref):
    return torch.empty(shape, dtype=torch.float32, device=ref.device)


# ------------------------------------------------------------------------------------------------ TD(lambda)
class TDLambdaFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, value, reward, weight, gamma, lambda_, sharded, group):
        T, B = reward.shape
        loss, grad_buf = _new((1,), reward), _new((T, B), reward)
        hpc_rl_utils.TdLambdaForward([value, reward, weight], [loss, grad_buf], gamma, lambda_,
                                     _dp.loss_scale(T * B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.grad_buf = grad_buf
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        grad_buf = ctx.grad_buf
        T, B = grad_buf.shape
        grad_value = _new((T + 1, B), grad_buf)
        hpc_rl_utils.TdLambdaBackward([grad_loss.contiguous(), grad_buf], [grad_value])
        return grad_value, None, None, None, None, None, None


class TDLambda(torch.nn.Module):
    """TD(lambda) loss: 0.5 * mean(weight * (lambda_return - value[:-1])^2), gradient w.r.t. ``value`` only."""

    def __init__(self, T, B, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.sharded, self.group = T, B, sharded, group

    def forward(self, value, reward, weight=None, gamma: float = 0.9, lambda_: float = 0.8) -> torch.Tensor:
        """value (T+1,B), reward (T,B), weight None | (B,) | (T,B) -> loss (1,)."""
        assert value.is_cuda
        assert reward.is_cuda
        if weight is not None:
            assert weight.is_cuda
        return TDLambdaFunction.apply(value, reward, weight, gamma, lambda_, self.sharded, self.group)


# ------------------------------------------------------------------------------------------------ q n-step TD
class _QNStepFunctionBase(torch.autograd.Function):
    FWD = BWD = None

    @classmethod
    def _forward(cls, ctx, q, next_n_q, action, next_n_action, reward, done, weight, gamma, sharded, group):
        B, N = q.shape
        td_err, loss, grad_buf = _new((B,), q), _new((1,), q), _new((B,), q)
        cls.FWD([q, next_n_q, action, next_n_action, reward, done, weight], [td_err, loss, grad_buf], gamma,
                _dp.loss_scale(B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.saved = (grad_buf, action, (B, N))
        ctx.mark_non_differentiable(td_err)
        return loss, td_err

    @classmethod
    def _backward(cls, ctx, grad_loss):
        grad_buf, action, shape = ctx.saved
        grad_q = _new(shape, grad_buf)
        cls.BWD([grad_loss.contiguous(), grad_buf, action], [grad_q])
        return grad_q


class QNStepTDFunction(_QNStepFunctionBase):
    FWD = staticmethod(hpc_rl_utils.QNStepTdForward)
    BWD = staticmethod(hpc_rl_utils.QNStepTdBackward)

    @staticmethod
    def forward(ctx, q, next_n_q, action, next_n_action, reward, done, weight, gamma, sharded, group):
        return QNStepTDFunction._forward(ctx, q, next_n_q, action, next_n_action, reward, done, weight, gamma, sharded,
                                         group)

    @staticmethod
    def backward(ctx, grad_loss, grad_td_err):
        return (QNStepTDFunction._backward(ctx, grad_loss),) + (None,) * 9


class QNStepTDRescaleFunction(_QNStepFunctionBase):
    FWD = staticmethod(hpc_rl_utils.QNStepTdRescaleForward)
    BWD = staticmethod(hpc_rl_utils.QNStepTdRescaleBackward)

    @staticmethod
    def forward(ctx, q, next_n_q, action, next_n_action, reward, done, weight, gamma, sharded, group):
        return QNStepTDRescaleFunction._forward(ctx, q, next_n_q, action, next_n_action, reward, done, weight, gamma,
                                                sharded, group)

    @staticmethod
    def backward(ctx, grad_loss, grad_td_err):
        return (QNStepTDRescaleFunction._backward(ctx, grad_loss),) + (None,) * 9


def _assert_cuda(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda


class QNStepTD(torch.nn.Module):
    """n-step TD error for q-learning: (mean(weight*(q[b,a]-target)^2), per-sample (q-target)^2)."""

    def __init__(self, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.sharded, self.group = T, B, N, sharded, group

    def forward(self, q, next_n_q, action, next_n_action, reward, done, weight, gamma: float):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, weight)
        return QNStepTDFunction.apply(q, next_n_q, action, next_n_action, reward, done, weight, gamma, self.sharded,
                                      self.group)


class QNStepTDRescale(torch.nn.Module):
    """n-step TD error with value rescaling h / h^-1 (eps = 1e-2)."""

    def __init__(self, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.sharded, self.group = T, B, N, sharded, group

    def forward(self, q, next_n_q, action, next_n_action, reward, done, weight, gamma: float):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, weight)
        return QNStepTDRescaleFunction.apply(q, next_n_q, action, next_n_action, reward, done, weight, gamma,
                                             self.sharded, self.group)


# ------------------------------------------------------------------------------------------------ dist (C51)
class DistNStepTDFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, dist, next_n_dist, action, next_n_action, reward, done, weight, gamma, v_min, v_max, sharded,
                group):
        B, N, n_atom = dist.shape
        td_err, loss, buf = _new((B,), dist), _new((1,), dist), _new((B, n_atom), dist)
        hpc_rl_utils.DistNStepTdForward([dist, next_n_dist, action, next_n_action, reward, done, weight],
                                        [td_err, loss, buf], gamma, v_min, v_max, _dp.loss_scale(B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.saved = (buf, action, (B, N, n_atom))
        ctx.mark_non_differentiable(td_err)
        return loss, td_err

    @staticmethod
    def backward(ctx, grad_loss, grad_td_err):
        buf, action, shape = ctx.saved
        grad_dist = _new(shape, buf)
        hpc_rl_utils.DistNStepTdBackward([grad_loss.contiguous(), buf, action], [grad_dist])
        return (grad_dist,) + (None,) * 11


class DistNStepTD(torch.nn.Module):
    """C51 distributional n-step TD error (categorical projection + cross entropy)."""

    def __init__(self, T, B, N, n_atom, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.n_atom, self.sharded, self.group = T, B, N, n_atom, sharded, group

    def forward(self, dist, next_n_dist, action, next_n_action, reward, done, weight, gamma: float, v_min: float,
                v_max: float):
        _assert_cuda(dist, next_n_dist, action, next_n_action, reward, done, weight)
        # (the reference additionally asserts dist[b,a] > 0 with a host sync, rl_utils/td.py:101-103; a log of a
        #  non-positive probability shows up as nan/inf in the loss instead of stalling the stream here)
        return DistNStepTDFunction.apply(dist, next_n_dist, action, next_n_action, reward, done, weight, gamma, v_min,
                                         v_max, self.sharded, self.group)


# ------------------------------------------------------------------------------------------------ IQN
class IQNNStepTDErrorFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma, gamma,
                kappa, sharded, group):
        tau, B, N = q.shape
        loss, td_err, grad_buf = _new((1,), q), _new((B,), q), _new((B, tau), q)
        hpc_rl_utils.IQNNStepTDErrorForward(
            [q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma],
            [loss, td_err, grad_buf], gamma, kappa, _dp.loss_scale(B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.saved = (grad_buf, action, (tau, B, N))
        ctx.mark_non_differentiable(td_err)
        return loss, td_err

    @staticmethod
    def backward(ctx, grad_loss, grad_td_err):
        grad_buf, action, shape = ctx.saved
        grad_q = _new(shape, grad_buf)
        hpc_rl_utils.IQNNStepTDErrorBackward([grad_loss.contiguous(), grad_buf, action], [grad_q])
        return (grad_q,) + (None,) * 12


class IQNNStepTDError(torch.nn.Module):
    """IQN n-step TD error (quantile Huber loss over tau x tau' pairs)."""

    def __init__(self, tau, tauPrime, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.tau, self.tauPrime, self.T, self.B, self.N = tau, tauPrime, T, B, N
        self.sharded, self.group = sharded, group

    def forward(self, q, next_n_q, action, next_n_action, reward, done, replay_quantiles, gamma: float,
                kappa: float = 1.0, weight: Optional[torch.Tensor] = None,
                value_gamma: Optional[torch.Tensor] = None):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma)
        return IQNNStepTDErrorFunction.apply(q, next_n_q, action, next_n_action, reward, done, replay_quantiles,
                                             weight, value_gamma, gamma, kappa, self.sharded, self.group)


# ------------------------------------------------------------------------------------------------ QR-DQN
class QRDQNNStepTDErrorFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, q, next_n_q, action, next_n_action, reward, done, weight, value_gamma, gamma, tau_value, sharded,
                group):
        B, N, tau = q.shape
        loss, td_err, grad_buf = _new((1,), q), _new((B,), q), _new((B, tau), q)
        hpc_rl_utils.QRDQNNStepTDErrorForward(
            [q, next_n_q, action, next_n_action, reward, done, weight, value_gamma], [loss, td_err, grad_buf], gamma,
            tau_value, _dp.loss_scale(B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.saved = (grad_buf, action, (B, N, tau))
        ctx.mark_non_differentiable(td_err)
        return loss, td_err

    @staticmethod
    def backward(ctx, grad_loss, grad_td_err):
        grad_buf, action, shape = ctx.saved
        grad_q = _new(shape, grad_buf)
        hpc_rl_utils.QRDQNNStepTDErrorBackward([grad_loss.contiguous(), grad_buf, action], [grad_q])
        return (grad_q,) + (None,) * 11


class QRDQNNStepTDError(torch.nn.Module):
    """QR-DQN n-step TD error.  Like the reference kernel (and its test, which passes ``tau`` = the integer count
    to the oracle) the quantile weight is |tau - 1[err <= 0]| with tau = the number of quantiles; pass
    ``tau_value`` to use something else (e.g. a true fraction)."""

    def __init__(self, tau, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.tau, self.T, self.B, self.N, self.sharded, self.group = tau, T, B, N, sharded, group

    def forward(self, q, next_n_q, action, next_n_action, reward, done, gamma: float,
                weight: Optional[torch.Tensor] = None, value_gamma: Optional[torch.Tensor] = None,
                tau_value: Optional[float] = None):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, weight, value_gamma)
        return QRDQNNStepTDErrorFunction.apply(q, next_n_q, action, next_n_action, reward, done, weight, value_gamma,
                                               gamma, tau_value, self.sharded, self.group)
