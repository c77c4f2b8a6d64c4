"""``hpc_rll.rl_utils.td`` -- drop-in for /root/reference/hpc_rll/rl_utils/td.py: ``DistNStepTD``, ``TDLambda``,
``QNStepTD``, ``QNStepTDRescale``, ``IQNNStepTDError``, ``QRDQNNStepTDError`` with the reference's constructor and
``forward`` signatures and return values (ctor lines td.py:41,149,224,314,410,522; forward lines
td.py:63,165,242,332,439,549).

Differences: outputs and scratch are allocated per call (the reference returns the same module
buffer every call); inputs are validated; ``weight=None`` works for TD-lambda (the reference reads its (B,) default
buffer as (T,B): SURVEY.md A.2); optional batch-axis data parallelism (``sharded=True``, see hpc_rll.dist).
The autograd nodes are compiled torch::autograd::Functions in ``hpc_rl_utils`` (``td_lambda``, ``q_nstep_td``,
``dist_nstep_td``, ``iqn_nstep_td``, ``qrdqn_nstep_td``): one pybind call per forward.
"""
from typing import Optional

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp


def _assert_cuda(*ts):
    for t in ts:
        if t is not None:
            assert t.is_cuda


class _ShardedLoss(torch.nn.Module):
    """Shared data-parallel plumbing: 1/(GLOBAL count) scale in, ONE all-reduce of the loss scalar out."""

    def _scale(self, local_count):
        return _dp.loss_scale(local_count, self.group, True) if self.sharded else None

    def _reduce(self, loss):
        if self.sharded:
            _dp.all_reduce_losses_(loss.detach(), self.group, True)
        return loss


class TDLambda(_ShardedLoss):
    """TD(lambda) loss: 0.5 * mean(weight * (lambda_return - value[:-1])^2), gradient w.r.t. ``value`` only."""

    def __init__(self, T, B, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.sharded, self.group = T, B, sharded, group

    def forward(self, value, reward, weight=None, gamma: float = 0.9, lambda_: float = 0.8) -> torch.Tensor:
        """value (T+1,B), reward (T,B), weight None | (B,) | (T,B) -> loss (1,)."""
        _assert_cuda(value, reward, weight)
        return self._reduce(hpc_rl_utils.td_lambda(value, reward, weight, gamma, lambda_, self._scale(reward.numel())))


class QNStepTD(_ShardedLoss):
    """n-step TD error for q-learning: (mean(weight*(q[b,a]-target)^2), per-sample (q-target)^2)."""
    _RESCALE = False

    def __init__(self, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.sharded, self.group = T, B, N, sharded, group

    def forward(self, q, next_n_q, action, next_n_action, reward, done, weight, gamma: float):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, weight)
        loss, td_err = hpc_rl_utils.q_nstep_td(q, next_n_q, action, next_n_action, reward, done, weight, gamma,
                                               self._RESCALE, self._scale(q.shape[0]))
        return self._reduce(loss), td_err


class QNStepTDRescale(QNStepTD):
    """n-step TD error with value rescaling h / h^-1 (eps = 1e-2)."""
    _RESCALE = True


class DistNStepTD(_ShardedLoss):
    """C51 distributional n-step TD error (categorical projection + cross entropy)."""

    def __init__(self, T, B, N, n_atom, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.n_atom, self.sharded, self.group = T, B, N, n_atom, sharded, group

    def forward(self, dist, next_n_dist, action, next_n_action, reward, done, weight, gamma: float, v_min: float,
                v_max: float):
        _assert_cuda(dist, next_n_dist, action, next_n_action, reward, done, weight)
        # (the reference additionally asserts dist[b,a] > 0 with a host sync, rl_utils/td.py:101-103; a log of a
        #  non-positive probability shows up as nan/inf in the loss instead of stalling the stream here)
        loss, td_err = hpc_rl_utils.dist_nstep_td(dist, next_n_dist, action, next_n_action, reward, done, weight, gamma,
                                                  v_min, v_max, self._scale(dist.shape[0]))
        return self._reduce(loss), td_err


class IQNNStepTDError(_ShardedLoss):
    """IQN n-step TD error (quantile Huber loss over tau x tau' pairs).

    ``layout`` (keyword-only, not in the reference): ``'tbn'`` (default) is the reference's ``q (tau,B,N)``, ``next_n_q
    (tau',B,N)``; ``'bnt'`` takes ``q (B,N,tau)``, ``next_n_q (B,N,tau')`` -- the quantile axis innermost, the layout of
    ``QRDQNNStepTDError`` -- and returns the gradient in that layout.  Same loss (``q_bnt = q.permute(1, 2, 0)``), but a
    sample's quantiles are one contiguous row instead of tau values a whole (B,N) plane apart: 2 cache lines per sample
    instead of 2 tau (forward 128 -> 25 us at tau = 32, B = 65536, N = 64).  ``replay_quantiles`` stays ``(tau,B)``."""

    def __init__(self, tau, tauPrime, T, B, N, sharded: bool = False, group=None, *, layout: str = 'tbn'):
        super().__init__()
        assert layout in ('tbn', 'bnt'), layout
        self.tau, self.tauPrime, self.T, self.B, self.N = tau, tauPrime, T, B, N
        self.sharded, self.group, self.layout = sharded, group, layout

    def forward(self, q, next_n_q, action, next_n_action, reward, done, replay_quantiles, gamma: float,
                kappa: float = 1.0, weight: Optional[torch.Tensor] = None,
                value_gamma: Optional[torch.Tensor] = None):
        _assert_cuda(q, next_n_q, action, next_n_action, reward, done, replay_quantiles, weight, value_gamma)
        bnt = self.layout == 'bnt'
        loss, td_err = hpc_rl_utils.iqn_nstep_td(q, next_n_q, action, next_n_action, reward, done, replay_quantiles,
                                                 weight, value_gamma, gamma, kappa, self._scale(q.shape[0 if bnt else 1]), bnt)
        return self._reduce(loss), td_err


class QRDQNNStepTDError(_ShardedLoss):
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
        loss, td_err = hpc_rl_utils.qrdqn_nstep_td(q, next_n_q, action, next_n_action, reward, done, weight, value_gamma,
                                                   gamma, tau_value, self._scale(q.shape[0]))
        return self._reduce(loss), td_err
