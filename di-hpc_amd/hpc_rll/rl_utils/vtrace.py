"""``hpc_rll.rl_utils.vtrace`` -- drop-in for /root/reference/hpc_rll/rl_utils/vtrace.py (``VTrace(T,B,N)``,
forward signature vtrace.py:83, returns the ``hpc_vtrace_loss`` namedtuple of three (1,) tensors).

Backward recomputes the softmax from ``target_output`` instead of saving three (T,B,N) gradient buffers
(reference vtrace.py:70-72): forward writes 24 B per (t,b) of scratch instead of 12*N B."""
from collections import namedtuple

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp

hpc_vtrace_loss = namedtuple('hpc_vtrace_loss', ['policy_loss', 'value_loss', 'entropy_loss'])


class VtraceFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, target_output, behaviour_output, action, value, reward, weight, gamma, lambda_, rho_clip_ratio,
                c_clip_ratio, rho_pg_clip_ratio, sharded, group):
        T, B, N = target_output.shape
        dev = target_output.device
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        ws = hpc_rl_utils.vtrace_workspace(T, B, dev)
        hpc_rl_utils.VTraceForward([target_output, behaviour_output, action, value, reward, weight], [losses, ws],
                                   gamma, lambda_, rho_clip_ratio, c_clip_ratio, rho_pg_clip_ratio,
                                   _dp.loss_scale(T * B, group, sharded))
        _dp.all_reduce_losses_(losses, group, sharded)
        ctx.saved = (target_output, action, ws)
        return losses[0:1], losses[1:2], losses[2:3]

    @staticmethod
    def backward(ctx, grad_pg_loss, grad_value_loss, grad_entropy_loss):
        target_output, action, ws = ctx.saved
        T, B, N = target_output.shape
        need_t, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        grad_target = torch.empty_like(target_output) if need_t else None
        grad_value = torch.empty(T + 1, B, dtype=torch.float32, device=ws.device) if need_v else None
        if need_t or need_v:
            hpc_rl_utils.VTraceBackward([grad_pg_loss.contiguous(), grad_value_loss.contiguous(),
                                         grad_entropy_loss.contiguous(), target_output, action, ws],
                                        [grad_target, grad_value])
        return (grad_target, None, None, grad_value) + (None,) * 9


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
        pg, v, e = VtraceFunction.apply(target_output, behaviour_output, action, value, reward, weight, gamma, lambda_,
                                        rho_clip_ratio, c_clip_ratio, rho_pg_clip_ratio, self.sharded, self.group)
        return hpc_vtrace_loss(pg, v, e)
