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


@pytest.mark.parametrize("S,B,I,H,L", [(6, 16, 24, 32, 2),      # step kernels
                                       (6, 16, 24, 128, 2),     # persistent mid-batch forward AND backward kernels inside the graph
                                       (5, 48, 16, 256, 1)])    # mid-batch forward (two streams), step-kernel backward
def test_graphed_lstm_parameters_are_differentiated(S, B, I, H, L):
    import hpc_rll
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(0)
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
    # new data in the captured input, replayed: the persistent kernels' exchange slots are re-read with ordinary loads, and a
    # graph's kernel nodes must start with clean caches like eager launches do (no value of the first replay may survive)
    for _ in range(3):
        with torch.no_grad():
            x.normal_()
        out, grads = step()
        y, (hn, cn) = m(x, None)
        ref = torch.autograd.grad([y, hn, cn], [x] + list(m.parameters()), [torch.ones_like(y), torch.ones_like(hn), torch.ones_like(cn)])
        assert torch.equal(out[0], y.detach())
        for a, b in zip(grads, ref):
            assert torch.equal(a, b)


def test_graphed_ppo_with_device_side_monitors():
    """PPO(..., sync_info=False) keeps approx_kl / clipfrac on the device: no host sync in forward, so the whole PPO
    forward + backward is capturable; results equal the default (python-float) module bit for bit."""
    import hpc_rll
    from hpc_rll.rl_utils.ppo import PPO
    g = torch.Generator(device=DEV).manual_seed(6)
    B, N = 4096, 18
    ln = _randn(g, B, N).requires_grad_(True)
    vn = _randn(g, B).requires_grad_(True)
    args = (ln, ln.detach() + 0.3 * _randn(g, B, N), torch.randint(0, N, (B,), device=DEV, generator=g), vn,
            _randn(g, B), _randn(g, B), _randn(g, B))
    m = PPO(B, N, sync_info=False)
    step = hpc_rll.graphed(m, *args, None, 0.2, True, 3.0)
    (loss, info), (dln, dvn) = step()
    assert isinstance(info.approx_kl, torch.Tensor) and info.approx_kl.is_cuda
    ref_loss, ref_info = PPO(B, N)(*args, None, 0.2, True, 3.0)
    sum(ref_loss).sum().backward()
    assert all(torch.equal(a, b.detach()) for a, b in zip(loss, ref_loss))
    assert info.approx_kl.item() == ref_info.approx_kl and info.clipfrac.item() == ref_info.clipfrac
    assert torch.equal(dln, ln.grad) and torch.equal(dvn, vn.grad)


def test_graphed_steps_micro_batches_in_one_graph():
    """hpc_rll.graphed_steps: n steps -- micro-batches with their own static buffers, or the same batch n times -- captured
    into ONE hipGraph (one hipGraphLaunch, and one 8.5 us gap between graph launches, per n steps: profiles/r05_gae_gaps.txt).
    Every step's outputs and gradients equal the eager ones bit for bit and follow in-place updates of its buffers; a module
    with parameters (TD-lambda has none, so V-trace's logits stand in as `wrt` tensors) works the same."""
    import hpc_rll
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda
    g = torch.Generator(device=DEV).manual_seed(11)
    T, B, n = 256, 1024, 3
    m = GAE(T, B)
    sets = [(_randn(g, T + 1, B).requires_grad_(True), _randn(g, T, B).requires_grad_(True), _randn(g, T, B)) for _ in range(n)]
    steps = hpc_rll.graphed_steps(m, [(v, r, 0.99, 0.97) for v, r, _ in sets], grad_outputs=[ga for _, _, ga in sets])
    assert steps.steps == n and all(steps.wrt[i][0] is sets[i][0] for i in range(n))
    for trial in range(2):
        with torch.no_grad():
            for v, r, ga in sets:
                v.copy_(_randn(g, T + 1, B))
                r.copy_(_randn(g, T, B))
                ga.copy_(_randn(g, T, B))
        outs, grads = steps()
        for i, (v, r, ga) in enumerate(sets):
            v2, r2 = v.detach().clone().requires_grad_(True), r.detach().clone().requires_grad_(True)
            ref = m(v2, r2, 0.99, 0.97)
            ref.backward(ga)
            assert torch.equal(outs[i], ref.detach()) and torch.equal(grads[i][0], v2.grad) and torch.equal(grads[i][1], r2.grad), (trial, i)
    # the same batch twice (what bench.py's graph4 launch mode replays): both steps give the eager result
    v, r, ga = sets[0]
    two = hpc_rll.graphed_steps(m, [(v, r, 0.99, 0.97)] * 2, grad_outputs=[ga] * 2)
    outs, grads = two()
    ref = m(v, r, 0.99, 0.97)
    gv, gr = torch.autograd.grad(ref, (v, r), ga)
    for i in range(2):
        assert torch.equal(outs[i], ref.detach()) and torch.equal(grads[i][0], gv) and torch.equal(grads[i][1], gr)
    # a scalar loss with default grad_outputs (ones)
    tl = TDLambda(T, B)
    vs = [_randn(g, T + 1, B).requires_grad_(True) for _ in range(2)]
    rw, w = _randn(g, T, B), torch.rand(T, B, device=DEV, generator=g)
    st = hpc_rll.graphed_steps(tl, [(x, rw, w, 0.9, 0.8) for x in vs])
    outs, grads = st()
    for i, x in enumerate(vs):
        x2 = x.detach().clone().requires_grad_(True)
        loss = tl(x2, rw, w, 0.9, 0.8)
        loss.backward()
        assert torch.equal(outs[i], loss.detach()) and torch.equal(grads[i][0], x2.grad)
