"""``hpc_rll.rl_utils.ppo`` -- drop-in for /root/reference/hpc_rll/rl_utils/ppo.py (``PPO(B,N)``, forward signature
ppo.py:89, returns ``(hpc_ppo_loss, hpc_ppo_info)`` with python-float info like the reference, ppo.py:148).
The autograd node is ``hpc_rl_utils.ppo`` (compiled torch::autograd::Function).

``PPO(B, N, sync_info=False)`` (keyword-only addition) returns the two monitors as 0-d device tensors instead of python
floats: no host synchronisation in ``forward``, which makes the module capturable by ``hpc_rll.graphed`` (one hipGraph
per training step) and keeps an eager training loop asynchronous."""
from collections import namedtuple
from typing import Optional

import torch

import hpc_rl_utils
from hpc_rll import dist as _dp

hpc_ppo_loss = namedtuple('hpc_ppo_loss', ['policy_loss', 'value_loss', 'entropy_loss'])
hpc_ppo_info = namedtuple('hpc_ppo_info', ['approx_kl', 'clipfrac'])


class PPO(torch.nn.Module):
    """PPO clipped surrogate (+ optional dual clip), clipped value loss and entropy (arXiv:1707.06347)."""

    def __init__(self, B, N, sharded: bool = False, group=None, *, sync_info: bool = True):
        super().__init__()
        self.B, self.N, self.sharded, self.group, self.sync_info = B, N, sharded, group, sync_info

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
        scale = _dp.loss_scale(adv.numel(), self.group, True) if self.sharded else None
        policy_loss, value_loss, entropy_loss, info = hpc_rl_utils.ppo(
            logits_new, logits_old, action, value_new, value_old, adv, return_, weight, clip_ratio, use_value_clip,
            0.0 if dual_clip is None else dual_clip, scale)
        if self.sharded:   # the five scalars in ONE all-reduce; the two monitors are per-rank means -> averaged
            policy_loss, value_loss, entropy_loss, info = _dp.all_reduce_sum(
                (policy_loss, value_loss, entropy_loss, info), self.group, mean_slots=(3, 4))
        if self.sync_info:
            approx_kl, clipfrac = info.tolist()  # one host sync for both monitors (the reference does two .item())
        else:
            approx_kl, clipfrac = info.detach().unbind(0)
        return hpc_ppo_loss(policy_loss, value_loss, entropy_loss), hpc_ppo_info(approx_kl, clipfrac)
