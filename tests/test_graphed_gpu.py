"""hpc_rll.graphed: a forward+backward step captured into ONE hipGraph replays the eager kernels bit for bit (the launch
path is the only thing that changes), follows in-place updates of its static inputs.  (PPO's forward synchronises with the host to return python floats,
reference rl_utils/ppo.py:148, and cannot be captured: documented in hpc_rll/graph.py.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _randn(g, *s):
    return torch.randn(*s, device=DEV, generator=g)


def test_graphed_gae_equals_eager_and_follows_new_data():
    import hpc_rll
    from hpc_rll.rl_utils.gae import GAE
    g = torch.Generator(device=DEV).manual_seed(3)
    for T, B in ((1024, 8192), (96, 200), (1024, 64)):
        value, reward, ga = _randn(g, T + 1, B).requires_grad_(True), _randn(g, T, B).requires_grad_(True), _randn(g, T, B)
        m = GAE(T, B)
        step = hpc_rll.graphed(m, value, reward, 0.99, 0.97, grad_outputs=ga)
        assert step.wrt[0] is value and step.wrt[1] is reward
        for trial in range(3):
            with torch.no_grad():      # new batch written INTO the static buffers
                value.copy_(_randn(g, T + 1, B))
                reward.copy_(_randn(g, T, B))
                ga.copy_(_randn(g, T, B))
            adv, (dv, dr) = step()
            v2, r2 = value.detach().clone().requires_grad_(True), reward.detach().clone().requires_grad_(True)
            ref = m(v2, r2, 0.99, 0.97)
            ref.backward(ga)
            assert torch.equal(adv, ref.detach()) and torch.equal(dv, v2.grad) and torch.equal(dr, r2.grad), (T, B, trial)
        assert value.grad is None                # gradients are returned, not accumulated


def test_graphed_vtrace_three_losses_and_td_lambda():
    import hpc_rll
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.vtrace import VTrace
    g = torch.Generator(device=DEV).manual_seed(4)
    T, B, N = 40, 300, 18
    to, bo = _randn(g, T, B, N).requires_grad_(True), _randn(g, T, B, N)
    a = torch.randint(0, N, (T, B), device=DEV, generator=g)
    v, r = _randn(g, T + 1, B).requires_grad_(True), _randn(g, T, B)
    m = VTrace(T, B, N)
    co = [torch.tensor([c], device=DEV) for c in (1.0, 0.5, -0.01)]
    step = hpc_rll.graphed(m, to, bo, a, v, r, grad_outputs=co)
    for _ in range(2):
        with torch.no_grad():
            to.copy_(_randn(g, T, B, N))
            v.copy_(_randn(g, T + 1, B))
        losses, (dto, dv) = step()
        to2, v2 = to.detach().clone().requires_grad_(True), v.detach().clone().requires_grad_(True)
        ref = m(to2, bo, a, v2, r)
        (co[0] * ref.policy_loss + co[1] * ref.value_loss + co[2] * ref.entropy_loss).sum().backward()
        assert all(torch.equal(x, y.detach()) for x, y in zip(losses, ref))
        assert torch.equal(dto, to2.grad) and torch.equal(dv, v2.grad)
    # forward-only capture (nothing requires grad): returns the output alone
    td = TDLambda(T, B)
    vv, rr, w = _randn(g, T + 1, B), _randn(g, T, B), torch.rand(T, B, device=DEV, generator=g)
    fstep = hpc_rll.graphed(td, vv, rr, w, 0.9, 0.8)
    assert not fstep.backward and torch.equal(fstep(), td(vv, rr, w, 0.9, 0.8))


def test_graphed_lstm_parameters_are_differentiated():
    import hpc_rll
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(0)
    S, B, I, H, L = 6, 16, 24, 32, 2
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV, requires_grad=True)
    step = hpc_rll.graphed(m, x, None)
    out, grads = step()
    assert len(grads) == 1 + len([p for p in m.parameters() if p.requires_grad])
    y, (hn, cn) = m(x, None)
    ref = torch.autograd.grad([y, hn, cn], [x] + list(m.parameters()), [torch.ones_like(y), torch.ones_like(hn), torch.ones_like(cn)])
    assert torch.equal(out[0], y.detach())
    for a, b in zip(grads, ref):
        assert torch.equal(a, b)
