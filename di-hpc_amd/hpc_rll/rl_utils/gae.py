"""``hpc_rll.rl_utils.gae`` -- drop-in for the reference module of the same path
(/root/reference/hpc_rll/rl_utils/gae.py:6-61): same class name, constructor ``GAE(T, B)`` and
``forward(value, reward, gamma=0.99, lambda_=0.97)``.

Differences, all deliberate:
  * outputs are allocated per call from torch's caching allocator instead of being a module buffer
    that every call overwrites (reference gae.py:39);
  * backward exists: the reference returns None for every input (gae.py:17-18), here
    ``adv.backward(g)`` yields d/dvalue and d/dreward (analytic adjoint of hpc_rll.origin.gae).

The autograd node lives in the compiled extension (``hpc_rl_utils.gae``, a torch::autograd::Function): one pybind
call per forward, backward runs entirely inside the autograd engine.
"""
import torch

import hpc_rl_utils


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
        return hpc_rl_utils.gae(value, reward, gamma, lambda_)
