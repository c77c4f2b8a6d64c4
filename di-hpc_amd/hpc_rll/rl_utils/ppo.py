"""``hpc_rll.rl_utils.ppo`` -- drop-in for /root/reference/hpc_rll/rl_utils/ppo.py (``PPO(B,N)``, forward signature
ppo.py:89, returns ``(hpc_ppo_loss, hpc_ppo_info)`` with python-float info like the reference, ppo.py:148)."""
from collections import namedtuple
from typing import Optional

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp

hpc_ppo_loss = namedtuple('hpc_ppo_loss', ['policy_loss', 'value_loss', 'entropy_loss'])
hpc_ppo_info = namedtuple('hpc_ppo_info', ['approx_kl', 'clipfrac'])


class PPOFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits_new, logits_old, action, value_new, value_old, adv, return_, weight, clip_ratio,
                use_value_clip, dual_clip, sharded, group):
        B, N = logits_new.shape
        dev = logits_new.device
        out5 = torch.empty(5, dtype=torch.float32, device=dev)
        ws = hpc_rl_utils.ppo_workspace(B, dev)
        hpc_rl_utils.PPOForward([logits_new, logits_old, action, value_new, value_old, adv, return_, weight],
                                [out5, ws], use_value_clip, clip_ratio, dual_clip, _dp.loss_scale(B, group, sharded))
        _dp.all_reduce_losses_(out5, group, sharded, mean_slots=(3, 4))
        ctx.saved = (logits_new, action, ws)
        info = out5[3:5]
        ctx.mark_non_differentiable(info)
        return out5[0:1], out5[1:2], out5[2:3], info

    @staticmethod
    def backward(ctx, grad_policy_loss, grad_value_loss, grad_entropy_loss, grad_info):
        logits_new, action, ws = ctx.saved
        B, N = logits_new.shape
        need_l, need_v = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        grad_logits = torch.empty_like(logits_new) if need_l else None
        grad_value = torch.empty(B, dtype=torch.float32, device=ws.device) if need_v else None
        if need_l or need_v:
            hpc_rl_utils.PPOBackward([grad_policy_loss.contiguous(), grad_value_loss.contiguous(),
                                      grad_entropy_loss.contiguous(), logits_new, action, ws], [grad_logits, grad_value])
        return (grad_logits, None, None, grad_value) + (None,) * 9


class PPO(torch.nn.Module):
    """PPO clipped surrogate (+ optional dual clip), clipped value loss and entropy (arXiv:1707.06347)."""

    def __init__(self, B, N, sharded: bool = False, group=None):
        super().__init__()
        self.B, self.N, self.sharded, self.group = B, N, sharded, group

    def forward(self, logits_new, logits_old, action, value_new, value_old, adv, return_, weight=None,
                clip_ratio: float = 0.2, use_value_clip: bool = True, dual_clip: Optional[float] = None):
        assert logits_new.is_cuda
        assert logits_old.is_cuda
        assert action.is_cuda
        assert value_new.is_cuda
        assert value_old.is_cuda
        assert adv.is_cuda
        assert return_.is_cuda
        if weight is not None:
            assert weight.is_cuda
        assert dual_clip is None or dual_clip > 1.0, \
            "dual_clip value must be greater than 1.0, but get value: {}".format(dual_clip)
        policy_loss, value_loss, entropy_loss, info = PPOFunction.apply(
            logits_new, logits_old, action, value_new, value_old, adv, return_, weight, clip_ratio, use_value_clip,
            0.0 if dual_clip is None else dual_clip, self.sharded, self.group)
        approx_kl, clipfrac = info.tolist()  # one host sync for both monitors (the reference does two .item())
        return hpc_ppo_loss(policy_loss, value_loss, entropy_loss), hpc_ppo_info(approx_kl, clipfrac)
