"""``hpc_rll.rl_utils.vtrace`` -- drop-in for /root/reference/hpc_rll/rl_utils/vtrace.py (``VTrace(T,B,N)``,
forward signature vtrace.py:83, returns the ``hpc_vtrace_loss`` namedtuple of three (1,) tensors).

Backward recomputes the softmax from ``target_output`` instead of saving three (T,B,N) gradient buffers
(reference vtrace.py:70-72): forward writes 24 B per (t,b) of scratch instead of 12*N B.  The autograd node is
``hpc_rl_utils.vtrace`` (compiled torch::autograd::Function)."""
from collections import namedtuple

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp

hpc_vtrace_loss = namedtuple('hpc_vtrace_loss', ['policy_loss', 'value_loss', 'entropy_loss'])


class VTrace(torch.nn.Module):
    """IMPALA V-trace actor-critic losses (arXiv:1802.01561)."""

    def __init__(self, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.sharded, self.group = T, B, N, sharded, group

    def forward(self, target_output, behaviour_output, action, value, reward, weight=None, gamma: float = 0.99,
                lambda_: float = 0.95, rho_clip_ratio: float = 1.0, c_clip_ratio: float = 1.0,
                rho_pg_clip_ratio: float = 1.0):
        """target/behaviour_output (T,B,N), action (T,B) int64, value (T+1,B), reward (T,B), weight (T,B) or None."""
        assert target_output.is_cuda
        assert behaviour_output.is_cuda
        assert action.is_cuda
        assert value.is_cuda
        assert reward.is_cuda
        if weight is not None:
            assert weight.is_cuda
        scale = _dp.loss_scale(reward.numel(), self.group, True) if self.sharded else None
        pg, v, e = hpc_rl_utils.vtrace(target_output, behaviour_output, action, value, reward, weight, gamma, lambda_,
                                       rho_clip_ratio, c_clip_ratio, rho_pg_clip_ratio, scale)
        if self.sharded:
            pg, v, e = _dp.all_reduce_sum((pg, v, e), self.group)     # the three scalars in ONE all-reduce
        return hpc_vtrace_loss(pg, v, e)
