"""``hpc_rll.rl_utils.gae`` -- drop-in for the reference module of the same path
(/root/reference/hpc_rll/rl_utils/gae.py:6-61): same class name, constructor ``GAE(T, B)`` and
``forward(value, reward, gamma=0.99, lambda_=0.97)``.

Differences, all deliberate:
  * outputs are allocated per call from torch's caching allocator instead of being a module buffer
    that every call overwrites (reference gae.py:39);
  * backward exists: the reference returns None for every input (gae.py:17-18), here
    ``adv.backward(g)`` yields d/dvalue and d/dreward (analytic adjoint of hpc_rll.origin.gae).
"""
import torch

import hpc_rl_utils


class GAEFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, value, reward, gamma, lambda_):
        adv = torch.empty_like(reward)
        hpc_rl_utils.GaeForward([value, reward], [adv], gamma, lambda_)
        ctx.gamma, ctx.lambda_ = gamma, lambda_
        return adv

    @staticmethod
    def backward(ctx, grad_adv):
        need_v, need_r = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (need_v or need_r):
            return None, None, None, None
        grad_adv = grad_adv.contiguous()
        T, B = grad_adv.shape
        gv = torch.empty(T + 1, B, dtype=grad_adv.dtype, device=grad_adv.device) if need_v else None
        gr = torch.empty_like(grad_adv) if need_r else None
        hpc_rl_utils.GaeBackward([grad_adv], [gv, gr], ctx.gamma, ctx.lambda_)
        return gv, gr, None, None


class GAE(torch.nn.Module):
    """Generalized Advantage Estimator (arXiv:1506.02438), truncation-normalised variant of the reference.

    Arguments of the constructor are kept for API compatibility (trajectory length T, batch size B);
    the kernels take the sizes from the tensors, so any (T, B) works with one instance.
    """

    def __init__(self, T, B):
        super().__init__()
        self.T, self.B = T, B

    def forward(self, value, reward, gamma: float = 0.99, lambda_: float = 0.97) -> torch.FloatTensor:
        """value (T+1,B), reward (T,B) -> adv (T,B); all fp32 contiguous on the GPU."""
        assert value.is_cuda
        assert reward.is_cuda
        return GAEFunction.apply(value, reward, gamma, lambda_)
