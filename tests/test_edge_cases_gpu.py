"""Edge cases across the op set: empty batches / zero-length trajectories, single elements, non-default device
streams, non-contiguous gradients coming back from autograd."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_empty_batch_and_zero_length():
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.rl_utils.td import TDLambda, QNStepTD
    from hpc_rll.rl_utils.vtrace import VTrace
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.ppo import PPO
    for T, B in ((0, 8), (5, 0)):
        v = torch.zeros(T + 1, B, device=DEV, requires_grad=True)
        r = torch.zeros(T, B, device=DEV)
        adv = GAE(T, B)(v, r)
        assert adv.shape == (T, B)
        adv.sum().backward()
        assert v.grad.shape == (T + 1, B) and torch.equal(v.grad, torch.zeros_like(v.grad))
        loss = TDLambda(T, B)(v, r)
        assert loss.shape == (1,) and loss.item() == 0.0
        N = 4
        to = torch.zeros(T, B, N, device=DEV)
        a = torch.zeros(T, B, dtype=torch.int64, device=DEV)
        ls = VTrace(T, B, N)(to, to, a, v, r)
        assert all(x.item() == 0.0 for x in ls)
        assert UPGO(T, B, N)(to, r, a, r, v).item() == 0.0
    B, N = 0, 3
    z = torch.zeros(B, device=DEV)
    ls, info = PPO(B, N)(torch.zeros(B, N, device=DEV), torch.zeros(B, N, device=DEV), torch.zeros(B, dtype=torch.int64, device=DEV), z, z, z, z)
    assert all(x.item() == 0.0 for x in ls)
    loss, per = QNStepTD(2, B, N)(torch.zeros(B, N, device=DEV), torch.zeros(B, N, device=DEV), torch.zeros(B, dtype=torch.int64, device=DEV),
                                  torch.zeros(B, dtype=torch.int64, device=DEV), torch.zeros(2, B, device=DEV), z, None, 0.9)
    assert loss.item() == 0.0 and per.shape == (0,)


def test_expanded_upstream_gradient():
    """`loss.sum().backward()` / weighted sums hand the Function an expanded (stride-0) or 0-dim grad tensor."""
    from hpc_rll.rl_utils.td import TDLambda
    rng = np.random.default_rng(0)
    T, B = 9, 70
    v = torch.from_numpy(rng.standard_normal((T + 1, B)).astype(np.float32)).to(DEV).requires_grad_(True)
    r = torch.from_numpy(rng.standard_normal((T, B)).astype(np.float32)).to(DEV)
    m = TDLambda(T, B)
    m(v, r).sum().backward()
    g1 = v.grad.clone()
    v.grad = None
    (2.5 * m(v, r)).squeeze().backward()
    assert torch.allclose(v.grad, 2.5 * g1, rtol=1e-6, atol=0)


def test_side_stream_and_graph_capture():
    """Nothing in the GAE path allocates device memory behind torch's back or synchronises: it can be captured into a
    HIP graph and replayed."""
    import hpc_rl_utils as U
    T, B = 64, 1024
    g = torch.Generator(device=DEV).manual_seed(0)
    v = torch.randn(T + 1, B, device=DEV, generator=g)
    r = torch.randn(T, B, device=DEV, generator=g)
    adv = torch.empty_like(r)
    ref = torch.empty_like(r)
    U.GaeForward([v, r], [ref], 0.99, 0.97)          # also warms the coefficient-table cache
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        U.GaeForward([v, r], [adv], 0.99, 0.97)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        U.GaeForward([v, r], [adv], 0.99, 0.97)
    adv.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(adv, ref)


def test_hip_graph_capture_replay():
    """Launch-bound loops are meant to be replayed from a hipGraph (bench.py --graph): every entry point only
    enqueues work on the caller's stream (no host synchronisation, no allocation behind torch's back), so a captured
    step replays bit-identically -- GAE forward+backward through autograd, and the LSTM forward on both of its paths
    (B=3: persistent per-layer kernels with their exchange-buffer memset; B=12: GEMM + cell kernel per step)."""
    from hpc_rll.rl_utils.gae import GAE
    from hpc_rll.torch_utils.network.rnn import LSTM
    dev = torch.device("cuda:0")
    T, B = 64, 200
    v = torch.randn(T + 1, B, device=dev, requires_grad=True)
    r = torch.randn(T, B, device=dev)
    ga = torch.randn(T, B, device=dev)
    gae = GAE(T, B)
    lstms = [(LSTM(10, b, 24, 48, 2).to(dev), torch.randn(10, b, 24, device=dev)) for b in (3, 12)]

    def step():
        v.grad = None
        gae(v, r).backward(ga)
        return [m(x, None)[0] for m, x in lstms]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    eager = [t.clone() for t in eager]
    eager_grad = v.grad.clone()
    v.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        gae(v, r).backward(ga)
        outs = [m(x, None)[0] for m, x in lstms]
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(v.grad, eager_grad)
    for a, b in zip(outs, eager):
        assert torch.equal(a, b)


def test_graph_capture_with_cold_coefficient_cache():
    """a GAE call captured while its (T, gamma, lambda) coefficient table is NOT cached must (a) fill a
    per-call table inside the capture, (b) not publish that table (its contents only exist once the graph has been
    replayed), so an EAGER call with the same key between capture and replay still computes the right answer, and
    (c) replay correctly afterwards."""
    import hpc_rl_utils as U
    T, B = 37, 300                                   # a T no other test uses with these (gamma, lambda)
    gamma, lam = 0.9137, 0.8713
    g = torch.Generator(device=DEV).manual_seed(3)
    v = torch.randn(T + 1, B, device=DEV, generator=g)
    r = torch.randn(T, B, device=DEV, generator=g)
    ga = torch.randn(T, B, device=DEV, generator=g)
    adv_g = torch.empty_like(r)
    gv_g, gr_g = torch.empty_like(v), torch.empty_like(r)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        U.GaeForward([v, r], [torch.empty_like(r)], 0.99, 0.97)      # warm-up with ANOTHER key: the capture below misses
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        U.GaeForward([v, r], [adv_g], gamma, lam)
        U.GaeBackward([ga], [gv_g, gr_g], gamma, lam)
    # eager call with the SAME key before any replay: must not read an unfilled cached table
    adv_e = torch.empty_like(r)
    gv_e, gr_e = torch.empty_like(v), torch.empty_like(r)
    U.GaeForward([v, r], [adv_e], gamma, lam)
    U.GaeBackward([ga], [gv_e, gr_e], gamma, lam)
    torch.cuda.synchronize()
    from oracle import ref_torch as R
    ref = R.gae(v.double().cpu(), r.double().cpu(), gamma, lam)
    assert ((adv_e.double().cpu() - ref).abs() / ref.abs().clamp(min=1.0)).max().item() < 1e-5
    adv_g.fill_(float("nan"))
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(adv_g, adv_e) and torch.equal(gv_g, gv_e) and torch.equal(gr_g, gr_e)
    # many distinct keys: the cache is capped and never evicts, nothing goes stale
    outs = []
    for i in range(300):
        o = torch.empty_like(r)
        U.GaeForward([v, r], [o], 0.5 + i * 1e-3, 0.9)
        outs.append(o)
    again = torch.empty_like(r)
    U.GaeForward([v, r], [again], 0.5, 0.9)
    torch.cuda.synchronize()
    assert torch.equal(again, outs[0])


def test_inplace_change_between_forward_and_backward_is_caught():
    """backward recomputes from saved INPUTS (logits, LSTM weights); they are saved with
    save_for_backward, so autograd's version counter catches an in-place update between forward and backward."""
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.torch_utils.network.rnn import LSTM
    T, B, N = 6, 10, 5
    g = torch.Generator(device=DEV).manual_seed(0)
    logits = torch.randn(T, B, N, device=DEV, generator=g)
    to = logits.clone().requires_grad_(True)
    a = torch.randint(0, N, (T, B), device=DEV, generator=g)
    rho, r, v = (torch.rand(T, B, device=DEV, generator=g), torch.randn(T, B, device=DEV, generator=g),
                 torch.randn(T + 1, B, device=DEV, generator=g))
    h = to * 1.0                                    # non-leaf, so that an in-place op on it is legal
    loss = UPGO(T, B, N)(h, rho, a, r, v)
    h.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
    m = LSTM(4, 3, 6, 8, 1).to(DEV)
    y, _ = m(torch.randn(4, 3, 6, device=DEV), None)
    with torch.no_grad():
        m.wh.add_(0.1)                              # e.g. optimizer.step() before a second backward
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y.sum().backward()


def test_loss_ops_capture_into_a_hip_graph():
    """Strong scaling ends in the launch-latency regime, where a training step is meant to be replayed from a hipGraph.
    Every scalar-loss op (forward + backward through the C++ autograd nodes), ScatterConnection and the packed Pad take
    their outputs and scratch from torch's allocator on the current stream and never synchronise, so a captured step
    replays bit-identically.  (PPO is the exception by API: it returns python floats, i.e. a host sync.)"""
    from hpc_rll.rl_utils.padding import Padding1DPacked
    from hpc_rll.rl_utils.td import QNStepTD, TDLambda
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.vtrace import VTrace
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    g = torch.Generator(device=DEV).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    T, B, N = 40, 96, 12
    to, bo = rn(T, B, N).requires_grad_(True), rn(T, B, N)
    a = torch.randint(0, N, (T, B), device=DEV, generator=g)
    v, r, rho = rn(T + 1, B).requires_grad_(True), rn(T, B), torch.rand(T, B, device=DEV, generator=g)
    q, nq = rn(B, N).requires_grad_(True), rn(B, N)
    qa, qna = torch.randint(0, N, (B,), device=DEV, generator=g), torch.randint(0, N, (B,), device=DEV, generator=g)
    rew, done, w = rn(3, B), torch.zeros(B, device=DEV), torch.ones(B, device=DEV)
    x = rn(4, 20, 8).requires_grad_(True)
    loc = torch.stack([torch.randint(0, 6, (4, 20), device=DEV, generator=g), torch.randint(0, 5, (4, 20), device=DEV, generator=g)], -1)
    lens = torch.randint(1, 9, (50,), device=DEV, generator=g)
    flat = rn(int(lens.sum().item()))
    mods = (TDLambda(T, B), VTrace(T, B, N), UPGO(T, B, N), QNStepTD(3, B, N), ScatterConnection(4, 20, 8, 6, 5, "add"))

    def step():
        for p in (to, v, q, x):
            p.grad = None
        l1 = mods[0](v, r)
        l2 = mods[1](to, bo, a, v, r)
        l3 = mods[2](to, rho, a, r, v)
        l4, _ = mods[3](q, nq, qa, qna, rew, done, w, 0.9)
        y = mods[4](x, loc)
        (l1 + sum(l2) + l3 + l4 + (y * y).sum()).sum().backward()
        px, pm = Padding1DPacked(flat, lens, max_len=8)
        return l1, l2, l3, l4, y, px, pm

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager = step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    flat_e = [t.detach().clone() for t in (eager[0], *eager[1], eager[2], eager[3], eager[4], eager[5], eager[6])]
    grads_e = [p.grad.clone() for p in (to, v, q, x)]
    del eager          # drop the eager autograd graph (its AccumulateGrad nodes belong to the side stream)
    for p in (to, v, q, x):
        p.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    flat_g = (out[0], *out[1], out[2], out[3], out[4], out[5], out[6])
    for e, c in zip(flat_e, flat_g):
        assert torch.equal(e, c)
    for e, p in zip(grads_e, (to, v, q, x)):
        assert torch.equal(e, p.grad)


def test_loss_outputs_can_be_modified_in_place():
    """the three V-trace / PPO losses are separate autograd outputs over ONE buffer; they are independent
    tensors (not views), so `loss += ...` on a returned loss works as it does with the reference's module buffers."""
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.vtrace import VTrace
    g = torch.Generator(device=DEV).manual_seed(2)
    T, B, N = 6, 40, 5
    to = torch.randn(T, B, N, device=DEV, generator=g, requires_grad=True)
    v = torch.randn(T + 1, B, device=DEV, generator=g, requires_grad=True)
    args = (torch.randn(T, B, N, device=DEV, generator=g), torch.randint(0, N, (T, B), device=DEV, generator=g), v,
            torch.randn(T, B, device=DEV, generator=g))
    ref = VTrace(T, B, N)(to, *args)
    want = [x.item() for x in ref]
    out = VTrace(T, B, N)(to, *args)
    pl = out.policy_loss
    pl += 1.0
    pl *= 2.0
    assert abs(pl.item() - 2.0 * (want[0] + 1.0)) < 1e-5 and abs(out.value_loss.item() - want[1]) < 1e-7
    (pl + out.value_loss).sum().backward()
    assert torch.isfinite(to.grad).all() and to.grad.abs().max() > 0
    ln = torch.randn(B, N, device=DEV, generator=g, requires_grad=True)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    loss, info = PPO(B, N)(ln, r(B, N), torch.randint(0, N, (B,), device=DEV, generator=g), r(B), r(B), r(B), r(B))
    e = loss.entropy_loss
    e -= 0.5
    (loss.policy_loss + e).sum().backward()
    assert torch.isfinite(ln.grad).all()
