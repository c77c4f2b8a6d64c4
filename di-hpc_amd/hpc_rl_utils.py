"""``hpc_rl_utils`` -- the reference's native extension module name (src/rl_utils/entry.cpp:8-39),
re-implemented as a thin binding over the C ABI of libhpc_rll_hip.so.

Every function keeps the reference's calling convention
``Fn(inputs: list[Tensor], outputs: list[Tensor], scalars...)`` with the same positional tensor order,
so the reference's own L1 wrappers (hpc_rll/rl_utils/*.py) run unchanged on top of it.  Launches go to
torch's CURRENT stream of the tensors' device (the reference uses legacy stream 0: SURVEY.md A.10).
Unlike the reference, arguments are validated and HIP errors surface as RuntimeError.
"""
import torch

from hpc_rll import _native as N

_lib = N.lib

# (device index, T, gamma, lambda) -> coef tensor.  The table depends only on these.
_gae_coef_cache = {}


def gae_coef(T: int, gamma: float, lambda_: float, device: torch.device) -> torch.Tensor:
    key = (device.index, int(T), float(gamma), float(lambda_))
    c = _gae_coef_cache.get(key)
    if c is None:
        c = torch.empty(max(int(T), 1), dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            N.check(_lib.hpc_rll_gae_coef(c.data_ptr(), int(T), float(gamma), float(lambda_), N.stream_ptr(device)),
                    "gae_coef")
        if len(_gae_coef_cache) > 64:
            _gae_coef_cache.clear()
        _gae_coef_cache[key] = c
    return c


def GaeForward(inputs, outputs, gamma: float, lambda_: float) -> None:
    """inputs = [value (T+1,B), reward (T,B)], outputs = [adv (T,B)].  Reference: src/rl_utils/gae.cu:8-28."""
    value, reward = inputs
    (adv,) = outputs
    N.require(reward, "reward")
    if reward.dim() != 2:
        raise RuntimeError(f"reward: expected (T,B), got {tuple(reward.shape)}")
    T, B = reward.shape
    dev = reward.device
    N.require(value, "value", shape=(T + 1, B), device=dev)
    N.require(adv, "adv", shape=(T, B), device=dev)
    coef = gae_coef(T, gamma, lambda_, dev)
    with torch.cuda.device(dev):
        N.check(_lib.hpc_rll_gae_forward(value.data_ptr(), reward.data_ptr(), adv.data_ptr(), coef.data_ptr(),
                                         T, B, float(gamma), N.stream_ptr(dev)), "GaeForward")


def GaeBackward(inputs, outputs, gamma: float, lambda_: float) -> None:
    """inputs = [grad_adv (T,B)], outputs = [grad_value (T+1,B) or None, grad_reward (T,B) or None].

    New entry (the reference registers no GaeBackward: src/rl_utils/entry.cpp:22): the analytic adjoint of
    hpc_rll.origin.gae."""
    (grad_adv,) = inputs
    grad_value, grad_reward = outputs
    N.require(grad_adv, "grad_adv")
    T, B = grad_adv.shape
    dev = grad_adv.device
    if grad_value is not None:
        N.require(grad_value, "grad_value", shape=(T + 1, B), device=dev)
    if grad_reward is not None:
        N.require(grad_reward, "grad_reward", shape=(T, B), device=dev)
    coef = gae_coef(T, gamma, lambda_, dev)
    with torch.cuda.device(dev):
        N.check(_lib.hpc_rll_gae_backward(grad_adv.data_ptr(), N.ptr(grad_value), N.ptr(grad_reward), coef.data_ptr(),
                                          T, B, float(gamma), N.stream_ptr(dev)), "GaeBackward")
