"""The reference's OWN L1 modules pass their module buffers to the L2 positionally (hpc_rll/rl_utils/vtrace.py:17-27,
upgo.py:10-15, ppo.py:20-30, td.py:378-383 / 492-497 / 11-16, torch_utils/network/rnn.py:16-21,35-42).  The compiled
extension accepts exactly those lists next to its own short ones, so an unmodified reference L1 runs on this L2
(VERDICT r01 item 9).  /root/reference does not exist on the GPU box, so the callers below REBUILD the reference's
positional lists (same order, same buffer shapes as the reference's register_buffer lines) and the results must equal
the native modules' bit for bit -- same kernels behind both conventions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _z(*shape, dtype=torch.float32):
    return torch.zeros(*shape, dtype=dtype, device=DEV)


def _rn(rng, *shape):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(DEV)


def test_vtrace_reference_lists():
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.vtrace import VTrace
    rng = np.random.default_rng(0)
    T, B, N = 9, 13, 6
    to, bo = _rn(rng, T, B, N), _rn(rng, T, B, N)
    a = torch.from_numpy(rng.integers(0, N, (T, B))).to(DEV)
    v, r = _rn(rng, T + 1, B), _rn(rng, T, B)
    w = torch.ones(T, B, device=DEV)                                     # vtrace.py:68 register_buffer('weight')
    # vtrace.py:69-83: the module buffers, in the order of the reference's `outputs` list (vtrace.py:19-21)
    prob, ent = _z(T, B), _z(T, B)
    g_logits, g_prob, g_ent = _z(T, B, N), _z(T, B, N), _z(T, B, N)
    b_prob, isw, ret, adv = _z(T, B), _z(T, B), _z(T, B), _z(T, B)
    pg, vl, el = _z(1), _z(1), _z(1)
    grad_value, grad_target = _z(T + 1, B), _z(T, B, N)
    U.VTraceForward([to, bo, a, v, r, w], [prob, ent, g_logits, g_prob, g_ent, b_prob, isw, ret, adv, pg, vl, el],
                    0.99, 0.95, 1.0, 1.0, 1.0)
    gp, gv, ge = (torch.tensor(x, device=DEV) for x in (1.0, 0.5, -0.25))
    # vtrace.py:27,35-40: backward inputs = 3 grads + bp_inputs; outputs = [grad_value, grad_target_output]
    U.VTraceBackward([gp, gv, ge, v, a, w, ret, adv, g_logits, g_prob, g_ent], [grad_value, grad_target])
    to2, v2 = to.clone().requires_grad_(True), v.clone().requires_grad_(True)
    l = VTrace(T, B, N)(to2, bo, a, v2, r)
    (l.policy_loss * 1.0 + l.value_loss * 0.5 + l.entropy_loss * -0.25).backward()
    assert torch.equal(pg, l.policy_loss.detach()) and torch.equal(vl, l.value_loss.detach())
    assert torch.equal(el, l.entropy_loss.detach())
    assert torch.equal(grad_target, to2.grad) and torch.equal(grad_value, v2.grad)
    # backward may run twice (retain_graph): the parked state survives
    grad_target.zero_()
    U.VTraceBackward([gp, gv, ge, v, a, w, ret, adv, g_logits, g_prob, g_ent], [grad_value, grad_target])
    assert torch.equal(grad_target, to2.grad)


def test_upgo_and_ppo_reference_lists():
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.upgo import UPGO
    rng = np.random.default_rng(1)
    T, B, N = 7, 10, 5
    to = _rn(rng, T, B, N)
    rho = torch.from_numpy(rng.random((T, B)).astype(np.float32)).to(DEV)
    a = torch.from_numpy(rng.integers(0, N, (T, B))).to(DEV)
    r, v = _rn(rng, T, B), _rn(rng, T + 1, B)
    adv, metric, loss, grad_buf, grad_target = _z(T, B), _z(T, B), _z(1), _z(T, B, N), _z(T, B, N)   # upgo.py:52-56
    U.UpgoForward([to, rho, a, r, v], [adv, metric, loss, grad_buf])                                  # upgo.py:10-12
    U.UpgoBackward([torch.tensor(2.0, device=DEV), grad_buf, adv], [grad_target])                     # upgo.py:14-26
    to2 = to.clone().requires_grad_(True)
    l = UPGO(T, B, N)(to2, rho, a, r, v)
    (2.0 * l).backward()
    assert torch.equal(loss, l.detach()) and torch.equal(grad_target, to2.grad)

    B, N = 33, 7
    ln, lo = _rn(rng, B, N), _rn(rng, B, N)
    a = torch.from_numpy(rng.integers(0, N, B)).to(DEV)
    vn, vo, ad, ret = _rn(rng, B), _rn(rng, B), _rn(rng, B), _rn(rng, B)
    w = torch.ones(B, device=DEV)                                                                     # ppo.py:69
    bufs = [_z(B), _z(B), _z(B, N), _z(B, N), _z(B, N), _z(B), _z(B), _z(B), _z(B)]                   # ppo.py:70-78
    scal = [_z(1) for _ in range(5)]                                                                  # ppo.py:80-84
    grad_value, grad_logits = _z(B), _z(B, N)                                                         # ppo.py:86-87
    U.PPOForward([ln, lo, a, vn, vo, ad, ret, w], bufs + scal, True, 0.2, 0.0)                        # ppo.py:20-26
    gs = [torch.tensor(x, device=DEV) for x in (1.0, 0.5, 0.01)]
    U.PPOBackward(gs + bufs[6:9] + bufs[2:5], [grad_value, grad_logits])                              # ppo.py:28-43
    ln2, vn2 = ln.clone().requires_grad_(True), vn.clone().requires_grad_(True)
    l, info = PPO(B, N)(ln2, lo, a, vn2, vo, ad, ret)
    (l.policy_loss + 0.5 * l.value_loss + 0.01 * l.entropy_loss).backward()
    assert [s.item() for s in scal[:3]] == [x.item() for x in l] and (scal[3].item(), scal[4].item()) == tuple(info)
    assert torch.equal(grad_logits, ln2.grad) and torch.equal(grad_value, vn2.grad)


def test_td_family_reference_lists():
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD, IQNNStepTDError, QRDQNNStepTDError
    rng = np.random.default_rng(2)
    nstep, B, N = 3, 12, 4
    a = torch.from_numpy(rng.integers(0, N, B)).to(DEV)
    na = torch.from_numpy(rng.integers(0, N, B)).to(DEV)
    r, done, w = _rn(rng, nstep, B), torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).to(DEV), torch.ones(B, device=DEV)
    g = torch.tensor(1.5, device=DEV)
    # ---- IQN: td.py:378-383, buffers td.py:430-437
    tau, taup = 5, 6
    q, nq = _rn(rng, tau, B, N), _rn(rng, taup, B, N)
    rq = torch.from_numpy(rng.random((tau, B)).astype(np.float32)).to(DEV)
    vg = torch.full((B,), 0.9 ** nstep, device=DEV)
    loss, td, bell, huber, gb, gq = _z(1), _z(B), _z(B, taup, tau), _z(B, taup, tau), _z(B, taup, tau), _z(tau, B, N)
    U.IQNNStepTDErrorForward([q, nq, a, na, r, done, rq, w, vg], [loss, td, bell, huber, gb], 0.9, 1.0)
    U.IQNNStepTDErrorBackward([g, gb, w, a], [gq])
    q2 = q.clone().requires_grad_(True)
    l2, td2 = IQNNStepTDError(tau, taup, nstep, B, N)(q2, nq, a, na, r, done, rq, 0.9, 1.0, w, vg)
    (1.5 * l2).backward()
    assert torch.equal(loss, l2.detach()) and torch.equal(td, td2) and torch.equal(gq, q2.grad)
    # ---- QR-DQN: td.py:492-497, buffers td.py:540-547
    tau = 7
    q, nq = _rn(rng, B, N, tau), _rn(rng, B, N, tau)
    loss, td, bell, huber, gb, gq = _z(1), _z(B), _z(B, tau, tau), _z(B, tau, tau), _z(B, tau), _z(B, N, tau)
    U.QRDQNNStepTDErrorForward([q, nq, a, na, r, done, w, vg], [loss, td, bell, huber, gb], 0.9)
    U.QRDQNNStepTDErrorBackward([g, gb, w, a], [gq])
    q2 = q.clone().requires_grad_(True)
    l2, td2 = QRDQNNStepTDError(tau, nstep, B, N)(q2, nq, a, na, r, done, 0.9, w, vg)
    (1.5 * l2).backward()
    assert torch.equal(loss, l2.detach()) and torch.equal(td, td2) and torch.equal(gq, q2.grad)
    # ---- C51: td.py:11-16; the reference's buf is (B + B*n_atom,) (td.py:60)
    n_atom = 11
    dist = torch.softmax(_rn(rng, B, N, n_atom), -1)
    ndist = torch.softmax(_rn(rng, B, N, n_atom), -1)
    td, loss, buf, gd = _z(B), _z(1), _z(B + B * n_atom), _z(B, N, n_atom)
    U.DistNStepTdForward([dist, ndist, a, na, r, done, w], [td, loss, buf], 0.9, -3.0, 3.0)
    U.DistNStepTdBackward([g, buf, a], [gd])
    d2 = dist.clone().requires_grad_(True)
    l2, td2 = DistNStepTD(nstep, B, N, n_atom)(d2, ndist, a, na, r, done, w, 0.9, -3.0, 3.0)
    (1.5 * l2).backward()
    assert torch.equal(loss, l2.detach()) and torch.equal(td, td2) and torch.equal(gd, d2.grad)


@pytest.mark.parametrize("S,B,I,H,L", [(5, 3, 6, 8, 2), (4, 12, 10, 16, 1)])      # wavefront path / step kernels
def test_lstm_reference_lists(S, B, I, H, L):
    import hpc_torch_utils_network as NW
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S)
    m = LSTM(S, B, I, H, L).to(DEV)
    x, h0, c0 = torch.randn(S, B, I, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    gy, gh, gc = torch.randn(S, B, H, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    G = 4 * H
    # rnn.py:122-141 module buffers in the order of rnn.py:17 (forward outputs) and rnn.py:21 (backward outputs)
    xbuf, hbuf = _z(S, B, G), _z(B, G)
    hn, cn = _z(S, L, B, H), _z(S, L, B, H)
    ifog, ym = _z(L, S, B, G), _z(L, S, B, H)
    ln_in, ln_mean, ln_rstd = _z(L, S, B, 2 * G), _z(L, S, B, B * 2), _z(L, S, B * 2)
    mask = _z(max(L - 1, 1), S, B, H, dtype=torch.int32)
    dgate, dx = _z(L, S, B, G), _z(S, B, I)
    dwx, dwh, dbias, dg, db = (torch.zeros_like(p) for p in (m.wx, m.wh, m.bias, m.ln_gamma, m.ln_beta))
    wx, wh, bias, gamma, beta = (p.detach() for p in (m.wx, m.wh, m.bias, m.ln_gamma, m.ln_beta))
    NW.LstmForward([x, h0, c0, wx, wh, bias, gamma, beta], [xbuf, hbuf, hn, cn, ifog, ym, ln_in, ln_mean, ln_rstd, mask], 0.0)
    y, h, c = ym[L - 1], hn[S - 1], cn[S - 1]                        # rnn.py:27-31
    NW.LstmBackward([x, h0, c0, wx, wh, hn, cn, ifog, ym, ln_in, ln_mean, ln_rstd, gamma, mask],
                    [dgate, xbuf, hbuf, dx, dwx, dwh, dbias, dg, db, gy, gh, gc], 0.0)   # rnn.py:35-42
    x2 = x.clone().requires_grad_(True)
    y2, (h2, c2) = m(x2, (h0, c0))
    ((y2 * gy).sum() + (h2 * gh).sum() + (c2 * gc).sum()).backward()
    assert torch.equal(y, y2.detach()) and torch.equal(h, h2.detach()) and torch.equal(c, c2.detach())
    for got, ref in ((dx, x2.grad), (dwx, m.wx.grad), (dwh, m.wh.grad), (dbias, m.bias.grad), (dg, m.ln_gamma.grad),
                     (db, m.ln_beta.grad)):
        assert torch.equal(got, ref)
