"""``hpc_rll.rl_utils.upgo`` -- drop-in for /root/reference/hpc_rll/rl_utils/upgo.py (``UPGO(T,B,N)``, forward
signature upgo.py:58, returns the (1,) loss).  Backward recomputes the softmax instead of saving a (T,B,N) buffer.
The autograd node is ``hpc_rl_utils.upgo`` (compiled torch::autograd::Function)."""
import torch

import hpc_rl_utils
from hpc_rll import dist as _dp


class UPGO(torch.nn.Module):
    """Importance-sampled UPGO loss: -mean(rho * (upgo_return - V) * log pi(a))."""

    def __init__(self, T, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.T, self.B, self.N, self.sharded, self.group = T, B, N, sharded, group

    def forward(self, target_output, rhos, action, rewards, bootstrap_values):
        """target_output (T,B,N), rhos (T,B), action (T,B) int64, rewards (T,B), bootstrap_values (T+1,B)."""
        assert target_output.is_cuda
        assert rhos.is_cuda
        assert action.is_cuda
        assert rewards.is_cuda
        assert bootstrap_values.is_cuda
        scale = _dp.loss_scale(rewards.numel(), self.group, True) if self.sharded else None
        loss = hpc_rl_utils.upgo(target_output, rhos, action, rewards, bootstrap_values, scale)
        if self.sharded:
            _dp.all_reduce_losses_(loss.detach(), self.group, True)
        return loss
