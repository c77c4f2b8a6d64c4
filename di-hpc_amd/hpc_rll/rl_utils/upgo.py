"""``hpc_rll.rl_utils.upgo`` -- drop-in for /root/reference/hpc_rll/rl_utils/upgo.py (``UPGO(T,B,N)``, forward
signature upgo.py:58, returns the (1,) loss).  Backward recomputes the softmax instead of saving a (T,B,N) buffer."""
import torch

import hpc_rl_utils
from hpc_rll import dist as _dp


class UpgoFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, target_output, rho, action, reward, value, sharded, group):
        T, B, N = target_output.shape
        dev = target_output.device
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        ws = hpc_rl_utils.upgo_workspace(T, B, dev)
        hpc_rl_utils.UpgoForward([target_output, rho, action, reward, value], [loss, ws],
                                 _dp.loss_scale(T * B, group, sharded))
        _dp.all_reduce_losses_(loss, group, sharded)
        ctx.saved = (target_output, action, ws)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        target_output, action, ws = ctx.saved
        grad_target = torch.empty_like(target_output)
        hpc_rl_utils.UpgoBackward([grad_loss.contiguous(), target_output, action, ws], [grad_target])
        return grad_target, None, None, None, None, None, None


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
        return UpgoFunction.apply(target_output, rhos, action, rewards, bootstrap_values, self.sharded, self.group)
