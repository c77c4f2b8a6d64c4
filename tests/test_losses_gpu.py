"""GPU parity tests for the scalar-loss ops: TD-lambda, V-trace, UPGO, PPO, q / dist / IQN / QR-DQN n-step TD.

Every op is driven through the drop-in Python API (hpc_rll.rl_utils.*), i.e. through the C ABI, and compared with
  * the golden fixtures recorded from the real reference (tests/golden/*.npz): loss, per-sample outputs, and the
    gradients of a weighted sum of the losses (distinct upstream weights per loss);
  * the fp64 oracle (oracle/ref_torch.py, autograd gradients) on seeded inputs at the reference's own test shapes.
Tolerance: max|d| <= tol * max(1,|ref|), tol = 1e-5 for returns/losses (north_star), looser where stated.
"""
import numpy as np
import pytest
import torch

from conftest import grad_err, rel_err
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = 1e-5
GTOL = 2e-5      # gradients: fp32 kernels vs fp64 autograd / fp32 reference autograd


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if grad:
        t.requires_grad_(True)
    return t


def opt(g, key):
    return G(g[key]) if key in g.files else None


def D(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).double()
    if grad:
        t.requires_grad_(True)
    return t


def f32(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


# ------------------------------------------------------------------------------------------------ TD(lambda)
def test_td_lambda_golden(golden):
    from hpc_rll.rl_utils.td import TDLambda
    g = golden("td_lambda")
    for i, (T, B, gam, lam, has_w, _) in enumerate(g["cases"]):
        v = G(g[f"c{i}_value"], True)
        loss = TDLambda(int(T), int(B))(v, G(g[f"c{i}_reward"]), opt(g, f"c{i}_weight"), float(gam), float(lam))
        assert loss.shape == (1,)
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < TOL
        assert grad_err(g[f"c{i}_grad_value"], v.grad.cpu().numpy()) < GTOL


# (1024,64), (2000,17), (513,300), (300,1000): sub-wave tiles of colscan.hpp (4 / 4 / 2 / 2 time chunks per wave)
@pytest.mark.parametrize("T,B,wmode", [(1024, 64, 2), (256, 16384, 2), (37, 4100, 1), (5, 70000, 0), (1, 1, 2), (100, 3, 1),
                                       (2000, 17, 1), (513, 300, 0), (300, 1000, 2),
                                       (1100, 96, 2), (2048, 8, 1), (1024, 100, 0)])   # 8 time chunks per wave (8-column tiles)
def test_td_lambda_oracle(T, B, wmode):
    from hpc_rll.rl_utils.td import TDLambda
    rng = np.random.default_rng(T + B)
    v, r = f32(rng, T + 1, B), f32(rng, T, B)
    w = None if wmode == 0 else rng.random((B,) if wmode == 1 else (T, B)).astype(np.float32)
    v64 = D(v, True)
    l64 = R.td_lambda_error(v64, D(r), None if w is None else D(w), 0.9, 0.8)
    l64.backward()
    dv = G(v, True)
    loss = TDLambda(T, B)(dv, G(r), None if w is None else G(w))
    (3.0 * loss).backward()
    assert rel_err(l64.item(), loss.item()) < TOL
    assert grad_err(3.0 * v64.grad.numpy(), dv.grad.cpu().numpy()) < GTOL


# ------------------------------------------------------------------------------------------------ V-trace
def test_vtrace_golden(golden):
    from hpc_rll.rl_utils.vtrace import VTrace
    g = golden("vtrace")
    co = g["coef"]
    for i, (T, B, N, gam, lam, rc, cc, pc, has_w, _) in enumerate(g["cases"]):
        to, v = G(g[f"c{i}_target_output"], True), G(g[f"c{i}_value"], True)
        ls = VTrace(int(T), int(B), int(N))(to, G(g[f"c{i}_behaviour_output"]), G(g[f"c{i}_action"]), v,
                                            G(g[f"c{i}_reward"]), opt(g, f"c{i}_weight"), float(gam), float(lam),
                                            float(rc), float(cc), float(pc))
        (co[0] * ls.policy_loss + co[1] * ls.value_loss + co[2] * ls.entropy_loss).sum().backward()
        assert rel_err(g[f"c{i}_losses"], [x.item() for x in ls]) < TOL
        assert grad_err(g[f"c{i}_grad_target_output"], to.grad.cpu().numpy()) < GTOL
        assert grad_err(g[f"c{i}_grad_value"], v.grad.cpu().numpy()) < GTOL


@pytest.mark.parametrize("T,B,N", [(128, 128, 128), (64, 300, 6), (16, 70, 1000), (9, 33, 2500), (256, 1024, 18), (3, 5, 1),
                                   (5, 7, 9), (40, 100, 4), (3, 11, 5001), (2, 3, 20000), (31, 50, 30), (7, 9, 2),
                                   (3, 7, 8192), (2, 5, 16384), (2, 9, 4100), (3, 4, 12000), (5, 3, 2052),
                                   (1024, 48, 5), (700, 200, 7), (300, 1030, 4),           # sub-wave scan tiles
                                   (1024, 96, 5), (1500, 8, 3)])                           # ... 8-column tiles
def test_vtrace_oracle(T, B, N):
    from hpc_rll.rl_utils.vtrace import VTrace
    rng = np.random.default_rng(T * 7 + N)
    to, bo = f32(rng, T, B, N), f32(rng, T, B, N)
    a = rng.integers(0, N, (T, B)).astype(np.int64)
    v, r = f32(rng, T + 1, B), f32(rng, T, B)
    to64, v64 = D(to, True), D(v, True)
    l64 = R.vtrace_error(to64, D(bo), torch.from_numpy(a), v64, D(r), None, 0.99, 0.95, 1.0, 1.0, 1.0)
    sum(l64).backward()
    dto, dv = G(to, True), G(v, True)
    ls = VTrace(T, B, N)(dto, G(bo), G(a), dv, G(r))
    sum(ls).backward()            # the reference test's pattern (tests/test_vtrace.py:44-52)
    assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < TOL
    assert grad_err(to64.grad.numpy(), dto.grad.cpu().numpy()) < GTOL
    assert grad_err(v64.grad.numpy(), dv.grad.cpu().numpy()) < GTOL


def _mask(rng, logits, action, frac=0.3):
    """-inf on a random subset of the NON-chosen actions (an action mask, as DI-engine feeds it)."""
    m = rng.random(logits.shape) < frac
    np.put_along_axis(m, action[..., None], False, axis=-1)
    out = logits.copy()
    out[m] = -np.inf
    return out


@pytest.mark.parametrize("T,B,N", [(16, 64, 128), (8, 40, 18), (5, 33, 1000), (4, 9, 6), (3, 7, 5001), (3, 5, 4096),
                                   (2, 3, 10000)])
def test_masked_actions_vtrace_ppo(T, B, N):
    """Masked actions arrive as logits = -inf: probability 0, no entropy contribution, zero gradient -- what
    torch.distributions.Categorical (hpc_rll.origin's softmax/entropy, origin/vtrace.py:76-79, origin/ppo.py:57-61)
    does by clamping log p to the most negative finite float.  Every row shape of the categorical kernels."""
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.vtrace import VTrace
    rng = np.random.default_rng(N + T)
    a = rng.integers(0, N, (T, B)).astype(np.int64)
    to, bo = _mask(rng, f32(rng, T, B, N), a), _mask(rng, f32(rng, T, B, N), a)
    v, r = f32(rng, T + 1, B), f32(rng, T, B)
    to64, v64 = D(to, True), D(v, True)
    l64 = R.vtrace_error(to64, D(bo), torch.from_numpy(a), v64, D(r), None, 0.99, 0.95, 1.0, 1.0, 1.0)
    sum(l64).backward()
    dto, dv = G(to, True), G(v, True)
    ls = VTrace(T, B, N)(dto, G(bo), G(a), dv, G(r))
    sum(ls).backward()
    assert all(np.isfinite(x.item()) for x in ls) and torch.isfinite(dto.grad).all()
    assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < TOL
    assert grad_err(to64.grad.numpy(), dto.grad.cpu().numpy()) < GTOL
    assert (dto.grad.cpu().numpy()[np.isinf(to)] == 0).all()          # masked actions get exactly zero gradient
    assert grad_err(v64.grad.numpy(), dv.grad.cpu().numpy()) < GTOL

    Bp = T * B
    ap = a.reshape(Bp)
    ln = to.reshape(Bp, N)
    lo = np.where(np.isinf(ln), ln, (ln + 0.3 * f32(rng, Bp, N))).astype(np.float32)   # the same mask on both policies
    vn, vo, adv, ret = f32(rng, Bp), f32(rng, Bp), f32(rng, Bp), f32(rng, Bp)
    ln64, vn64 = D(ln, True), D(vn, True)
    p64, i64 = R.ppo_error(ln64, D(lo), torch.from_numpy(ap), vn64, D(vo), D(adv), D(ret), None, 0.2, True, None)
    sum(p64).backward()
    dln, dvn = G(ln, True), G(vn, True)
    pl, info = PPO(Bp, N)(dln, G(lo), G(ap), dvn, G(vo), G(adv), G(ret), None, 0.2, True, None)
    sum(pl).backward()
    assert rel_err([x.item() for x in p64], [x.item() for x in pl]) < TOL
    assert rel_err(list(i64), list(info)) < 1e-4
    assert grad_err(ln64.grad.numpy(), dln.grad.cpu().numpy()) < GTOL
    assert (dln.grad.cpu().numpy()[np.isinf(ln)] == 0).all()


# ------------------------------------------------------------------------------------------------ UPGO
def test_upgo_golden(golden):
    from hpc_rll.rl_utils.upgo import UPGO
    g = golden("upgo")
    for i, (T, B, N, _) in enumerate(g["cases"]):
        to = G(g[f"c{i}_target_output"], True)
        loss = UPGO(int(T), int(B), int(N))(to, G(g[f"c{i}_rhos"]), G(g[f"c{i}_action"]), G(g[f"c{i}_reward"]),
                                            G(g[f"c{i}_value"]))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < TOL
        assert grad_err(g[f"c{i}_grad_target_output"], to.grad.cpu().numpy()) < GTOL


@pytest.mark.parametrize("T,B,N", [(256, 256, 256), (100, 70, 5), (2, 5000, 12), (1, 3, 4),
                                   (1024, 48, 5), (700, 200, 7), (300, 1030, 4),           # sub-wave scan tiles
                                   (1024, 96, 5), (1500, 8, 3)])                           # ... 8-column tiles
def test_upgo_oracle(T, B, N):
    from hpc_rll.rl_utils.upgo import UPGO
    rng = np.random.default_rng(T * 3 + N)
    to, rho = f32(rng, T, B, N), f32(rng, T, B)
    a = rng.integers(0, N, (T, B)).astype(np.int64)
    r, v = f32(rng, T, B), f32(rng, T + 1, B)
    # the data-dependent lambda is a comparison of fp32 sums: evaluate the oracle in fp32 as well so both sides see
    # the same inputs to the comparison, then compare against the fp64 oracle where no comparison is within 1e-6
    to64 = D(to, True)
    l64 = R.upgo_loss(to64, D(rho), torch.from_numpy(a), D(r), D(v))
    l64.backward()
    dto = G(to, True)
    loss = UPGO(T, B, N)(dto, G(rho), G(a), G(r), G(v))
    loss.backward()
    margin = np.abs((r[1:] + v[2:]) - v[1:-1]) if T > 1 else np.ones(1)
    if margin.min() > 1e-5:
        assert rel_err(l64.item(), loss.item()) < TOL
        assert grad_err(to64.grad.numpy(), dto.grad.cpu().numpy()) < GTOL
    else:  # a knife-edge comparison exists: fall back to the fp32 evaluation of the same oracle
        to32 = torch.from_numpy(to).requires_grad_(True)
        l32 = R.upgo_loss(to32, torch.from_numpy(rho), torch.from_numpy(a), torch.from_numpy(r), torch.from_numpy(v))
        l32.backward()
        assert rel_err(l32.item(), loss.item()) < 5e-5
        assert grad_err(to32.grad.numpy(), dto.grad.cpu().numpy()) < 5e-5


# ------------------------------------------------------------------------------------------------ PPO
def test_ppo_golden(golden):
    from hpc_rll.rl_utils.ppo import PPO
    g = golden("ppo")
    co = g["coef"]
    for i, (B, N, clip, uvc, dc, has_w, _) in enumerate(g["cases"]):
        ln, vn = G(g[f"c{i}_logit_new"], True), G(g[f"c{i}_value_new"], True)
        ls, info = PPO(int(B), int(N))(ln, G(g[f"c{i}_logit_old"]), G(g[f"c{i}_action"]), vn, G(g[f"c{i}_value_old"]),
                                       G(g[f"c{i}_adv"]), G(g[f"c{i}_return_"]), opt(g, f"c{i}_weight"), float(clip),
                                       bool(uvc), float(dc) if dc else None)
        (co[0] * ls.policy_loss + co[1] * ls.value_loss + co[2] * ls.entropy_loss).sum().backward()
        assert isinstance(info.approx_kl, float) and isinstance(info.clipfrac, float)
        assert rel_err(g[f"c{i}_losses"], [x.item() for x in ls]) < TOL
        assert rel_err(g[f"c{i}_info"], list(info)) < TOL
        assert grad_err(g[f"c{i}_grad_logit_new"], ln.grad.cpu().numpy()) < GTOL
        assert grad_err(g[f"c{i}_grad_value_new"], vn.grad.cpu().numpy()) < GTOL


@pytest.mark.parametrize("B,N,dual,uvc", [(128, 128, None, True), (4096, 18, 3.0, True), (70000, 6, None, False), (3, 1000, 1.5, True)])
def test_ppo_oracle(B, N, dual, uvc):
    from hpc_rll.rl_utils.ppo import PPO
    rng = np.random.default_rng(B + N)
    ln = f32(rng, B, N)
    lo = (ln + 0.3 * f32(rng, B, N)).astype(np.float32)
    a = rng.integers(0, N, (B,)).astype(np.int64)
    vn, vo, adv, ret = f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B)
    w = rng.random(B).astype(np.float32)
    ln64, vn64 = D(ln, True), D(vn, True)
    l64, i64 = R.ppo_error(ln64, D(lo), torch.from_numpy(a), vn64, D(vo), D(adv), D(ret), D(w), 0.2, uvc, dual)
    sum(l64).backward()
    dln, dvn = G(ln, True), G(vn, True)
    ls, info = PPO(B, N)(dln, G(lo), G(a), dvn, G(vo), G(adv), G(ret), G(w), 0.2, uvc, dual)
    sum(ls).backward()
    assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < TOL
    assert rel_err(list(i64), list(info)) < 1e-4     # clipfrac counts strict inequalities of fp32 ratios
    assert grad_err(ln64.grad.numpy(), dln.grad.cpu().numpy()) < GTOL
    assert grad_err(vn64.grad.numpy(), dvn.grad.cpu().numpy()) < GTOL


@pytest.mark.parametrize("B,N,dual,uvc", [(65536 + 19, 128, None, True), (4096, 18, 3.0, True), (70001, 6, None, False), (513, 256, 1.5, True),
                                          (40000, 64, None, True), (1000, 20, None, False), (9, 2, 2.0, True)])
def test_ppo_fused_forward_against_three_launches(B, N, dual, uvc):
    """tune key 32: ONE launch (both policy heads of a row in one lane group, the per-sample loss in its last lane, folded
    sums; csrc/categorical.hip: ppo_fwd_fused_kernel) against two categorical launches + the sample launch.  The per-sample
    coefficients the backward consumes come from the same functions: both gradients must be the same BITS; the three
    losses and the two info means add the rows in another grouping (rounding).  Row widths of every fused configuration
    (N = 2 ... 256, 4- and 16-byte loads), ragged batches, more rows than one sweep of the 512-workgroup grid, masked actions."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.ppo import PPO
    rng = np.random.default_rng(B + N)
    ln = f32(rng, B, N)
    lo = (ln + 0.3 * f32(rng, B, N)).astype(np.float32)
    a = rng.integers(0, N, (B,)).astype(np.int64)
    if N > 4:
        ln[::7, 1] = -np.inf                                  # masked actions (never the taken one)
        a[a == 1] = 2
    vn, vo, adv, ret = f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B)
    w = rng.random(B).astype(np.float32)
    res = {}
    try:
        for key in (0, 1):
            U.tune_set(32, key)
            dln, dvn = G(ln, True), G(vn, True)
            ls, info = PPO(B, N)(dln, G(lo), G(a), dvn, G(vo), G(adv), G(ret), G(w), 0.2, uvc, dual)
            (ls[0] + 0.7 * ls[1] - 0.01 * ls[2]).backward()
            res[key] = ([x.item() for x in ls], list(info), dln.grad.clone(), dvn.grad.clone())
    finally:
        U.tune_set(32, 1)
    (l0, i0, g0, v0), (l1, i1, g1, v1) = res[0], res[1]
    assert torch.equal(g0, g1) and torch.equal(v0, v1)
    assert torch.isfinite(g1).all()
    for x, y in zip(l0 + i0, l1 + i1):
        assert abs(x - y) <= 2e-6 * max(abs(x), 1e-3), (x, y)


def test_ppo_reference_inputs():
    """tests/test_ppo.py:10-27 literally: B=N=128, clip 0.2, value clip on, no dual clip, and EVERY float input an independent
    randn -- logits_old unrelated to logits_new (ratios from e^-5 to e^5, most samples clipped), negative weights, negative
    and non-binary everything else (SURVEY 4 "input quirks the build must accept")."""
    from hpc_rll.rl_utils.ppo import PPO
    B, N = 128, 128
    rng = np.random.default_rng(77)
    ln, lo = f32(rng, B, N), f32(rng, B, N)
    a = rng.integers(0, N, (B,)).astype(np.int64)
    vn, vo, adv, ret, w = f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B)
    ln64, vn64 = D(ln, True), D(vn, True)
    l64, i64 = R.ppo_error(ln64, D(lo), torch.from_numpy(a), vn64, D(vo), D(adv), D(ret), D(w), 0.2, True, None)
    sum(l64).backward()
    dln, dvn = G(ln, True), G(vn, True)
    ls, info = PPO(B, N)(dln, G(lo), G(a), dvn, G(vo), G(adv), G(ret), G(w), 0.2, True, None)
    sum(ls).backward()
    assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < TOL
    assert rel_err(list(i64), list(info)) < 1e-4
    assert grad_err(ln64.grad.numpy(), dln.grad.cpu().numpy()) < GTOL
    assert grad_err(vn64.grad.numpy(), dvn.grad.cpu().numpy()) < GTOL


# ------------------------------------------------------------------------------------------------ q n-step TD
def test_qntd_golden(golden):
    from hpc_rll.rl_utils.td import QNStepTD, QNStepTDRescale
    g = golden("qntd")
    for i, (T, B, N, gam, has_w, _) in enumerate(g["cases"]):
        for tag, cls in (("plain", QNStepTD), ("rescale", QNStepTDRescale)):
            q = G(g[f"c{i}_q"], True)
            loss, per = cls(int(T), int(B), int(N))(q, G(g[f"c{i}_next_n_q"]), G(g[f"c{i}_action"]),
                                                   G(g[f"c{i}_next_n_action"]), G(g[f"c{i}_reward"]), G(g[f"c{i}_done"]),
                                                   opt(g, f"c{i}_weight"), float(gam))
            loss.backward()
            assert rel_err(g[f"c{i}_{tag}_loss"], loss.item()) < 5e-5
            assert rel_err(g[f"c{i}_{tag}_td_err"], per.cpu().numpy()) < 5e-5
            assert grad_err(g[f"c{i}_{tag}_grad_q"], q.grad.cpu().numpy()) < 5e-5


@pytest.mark.parametrize("T", [1024, 16])
@pytest.mark.parametrize("rescale", [False, True])
def test_qntd_oracle_reference_shape(rescale, T):
    """tests/test_qntd.py:10-22: T=1024 (nstep), B=64, N=64, done / weight are randn floats -- the literal shape and input
    distributions; T=16 in addition, where gamma^n still leaves the bootstrap term visible."""
    from hpc_rll.rl_utils.td import QNStepTD, QNStepTDRescale
    B, N = 64, 64
    rng = np.random.default_rng(11 + T)
    q, nq = f32(rng, B, N), f32(rng, B, N)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, w = f32(rng, T, B), f32(rng, B), f32(rng, B)
    q64 = D(q, True)
    l64, p64 = R.q_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), 0.95, rescale)
    l64.backward()
    dq = G(q, True)
    loss, per = (QNStepTDRescale if rescale else QNStepTD)(T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), G(w), 0.95)
    loss.backward()
    assert rel_err(l64.item(), loss.item()) < 2e-5
    assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5
    assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("B", [131073, 300001, 600011])
def test_per_sample_ops_fold_at_every_batch_size(B):
    """Round 5: the one-lane-per-sample kernels (q n-step TD, PPO's three-launch form) keep their grid within the 512 workgroups
    of the in-launch loss finalisation at every batch size -- 1024-thread workgroups above 131072 samples, looping workgroups
    above 524288 -- instead of a second (finalize) launch.  Ragged batches in all three regimes against the fp64 oracle, and
    (q-TD) bit for bit against the separate finalize launch of tune key 21 = 0 on the per-sample outputs."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.ppo import PPO
    from hpc_rll.rl_utils.td import QNStepTD
    rng = np.random.default_rng(B)
    T, N = 3, 6
    q, nq = f32(rng, B, N), f32(rng, B, N)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, w = f32(rng, T, B), (rng.random(B) < 0.2).astype(np.float32), rng.random(B).astype(np.float32)
    q64 = D(q, True)
    l64, p64 = R.q_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), 0.97, False)
    l64.backward()
    res = {}
    try:
        for key in (1, 0):
            U.tune_set(21, key)
            dq = G(q, True)
            loss, per = QNStepTD(T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), G(w), 0.97)
            loss.backward()
            res[key] = (loss.detach().clone(), per.detach().clone(), dq.grad.clone())
    finally:
        U.tune_set(21, 1)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    for key in (0, 1):
        assert rel_err(l64.item(), res[key][0].item()) < 2e-5, key
    assert rel_err(p64.detach().numpy(), res[1][1].cpu().numpy()) < 2e-5
    assert grad_err(q64.grad.numpy(), res[1][2].cpu().numpy()) < 2e-5
    # PPO through its three-launch form (key 32 = 0: the per-sample kernel with five sums)
    ln = f32(rng, B, N)
    lo = (ln + 0.3 * f32(rng, B, N)).astype(np.float32)
    vn, vo, adv, ret = f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B)
    l64n, v64 = D(ln, True), D(vn, True)
    o64, _ = R.ppo_error(l64n, D(lo), torch.from_numpy(a), v64, D(vo), D(adv), D(ret), D(w), 0.2, True, None)
    (o64[0] + 0.5 * o64[1] - 0.01 * o64[2]).backward()
    try:
        U.tune_set(32, 0)
        dln, dvn = G(ln, True), G(vn, True)
        ls, info = PPO(B, N)(dln, G(lo), G(a), dvn, G(vo), G(adv), G(ret), G(w), 0.2, True, None)
        (ls[0] + 0.5 * ls[1] - 0.01 * ls[2]).backward()
    finally:
        U.tune_set(32, 1)
    for k in range(3):
        assert rel_err(o64[k].item(), ls[k].item()) < 2e-5, k
    assert grad_err(l64n.grad.numpy(), dln.grad.cpu().numpy()) < 2e-5
    assert grad_err(v64.grad.numpy(), dvn.grad.cpu().numpy()) < 2e-5


# ------------------------------------------------------------------------------------------------ dist (C51)
def test_dntd_golden(golden):
    from hpc_rll.rl_utils.td import DistNStepTD
    g = golden("dntd")
    for i, (T, B, N, na, gam, vmin, vmax, has_w, _) in enumerate(g["cases"]):
        d = G(g[f"c{i}_dist"], True)
        loss, per = DistNStepTD(int(T), int(B), int(N), int(na))(d, G(g[f"c{i}_next_n_dist"]), G(g[f"c{i}_action"]),
                                                                G(g[f"c{i}_next_n_action"]), G(g[f"c{i}_reward"]),
                                                                G(g[f"c{i}_done"]), opt(g, f"c{i}_weight"), float(gam),
                                                                float(vmin), float(vmax))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 1e-4
        assert rel_err(g[f"c{i}_td_err"], per.cpu().numpy()) < 1e-4
        assert grad_err(g[f"c{i}_grad_dist"], d.grad.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("T,quirks", [(128, True), (128, False), (4, True), (4, False)])
def test_dntd_oracle_reference_shape(T, quirks):
    """tests/test_dntd.py:10-25: T=B=N=128, n_atom=51, v in [-10,10], abs(randn) distributions; `quirks`: done / weight =
    randn exactly as the reference draws them (negative and non-binary: SURVEY 4 "input quirks the build must accept"),
    else 0/1 done and positive weights.  floor / ceil of the projected position is discontinuous, so the oracle is
    evaluated in fp32 AND fp64: a sample on which those two disagree sits on an atom boundary (its n-step return rounds to
    the other side in another summation order) and is excluded -- at most 3 % of the batch; every other sample, the loss
    restricted to them and their gradient rows must agree with the fp32 oracle to 1e-4."""
    from hpc_rll.rl_utils.td import DistNStepTD
    B, N, n_atom = 128, 128, 51
    rng = np.random.default_rng(5 + T + int(quirks))
    dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
    nd = np.abs(f32(rng, B, N, n_atom))
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = f32(rng, T, B)
    done = f32(rng, B) if quirks else (rng.random(B) < 0.3).astype(np.float32)
    w = f32(rng, B) if quirks else rng.random(B).astype(np.float32)
    d32 = torch.from_numpy(dist).requires_grad_(True)
    l32, p32 = R.dist_nstep_td_error(d32, torch.from_numpy(nd), torch.from_numpy(a), torch.from_numpy(na),
                                     torch.from_numpy(r), torch.from_numpy(done), torch.from_numpy(w), 0.95, -10., 10., n_atom)
    l32.backward()
    _, p64 = R.dist_nstep_td_error(D(dist), D(nd), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), 0.95,
                                   -10., 10., n_atom)
    dd = G(dist, True)
    loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), 0.95, -10., 10.)
    loss.backward()
    p32n, p64n, pgot = p32.detach().numpy().astype(np.float64), p64.detach().numpy(), per.cpu().numpy().astype(np.float64)
    scale = float(np.abs(p64n).max())
    edge = np.abs(p32n - p64n) > 1e-5 * scale          # the oracle itself is precision-sensitive on these samples
    assert edge.mean() <= 0.03, float(edge.mean())
    ok = ~edge
    assert float(np.abs(p32n[ok] - pgot[ok]).max()) < 1e-4 * scale
    if not edge.any():
        assert rel_err(l32.item(), loss.item()) < 1e-4
    # the gradient of sample b lives in row (b, action[b]) only and does not depend on the other samples
    g32, ggot = d32.grad.numpy()[ok], dd.grad.cpu().numpy()[ok]
    assert grad_err(g32, ggot) < 1e-4


def test_c51_run_sum_projection_against_the_gather_kernel_and_the_oracle():
    """Large batches project by run sums (csrc/dist_ops.hip: c51_project_scan): the sources that share a floor atom are one
    contiguous run of lanes, summed by a segmented doubling scan.  Pinned (a) against the small-batch kernel, which gathers
    every target's sources one by one in source order (tune key 24 = 1 forces it): equal up to the rounding of a run's few
    additions, 4e-7 of the maximum, most elements bit for bit; (b) against the fp32 oracle's two sequential scatter_add_
    passes (origin/td.py:100-103 as restated in oracle/ref_torch.py).  Inputs hold terminal samples (all sources on the same
    two atoms: one run of 51) and clamped returns (runs of 10+ sources at the ends of the support)."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD
    T, B, N, n_atom = 3, 65536, 4, 51
    v_min, v_max, gamma = -10., 10., 0.95
    rng = np.random.default_rng(11)
    dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
    nd = np.abs(f32(rng, B, N, n_atom))
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = (f32(rng, T, B) * 4.0).astype(np.float32)                       # some returns leave [v_min, v_max]: clamped atoms
    done = (rng.random(B) < 0.3).astype(np.float32)
    w = rng.random(B).astype(np.float32)
    res = {}
    try:
        for key in (1, 0):
            U.tune_set(24, key)
            dd = G(dist, True)
            loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), gamma, v_min, v_max)
            loss.backward()
            res[key] = (loss.item(), per.detach().cpu(), dd.grad.cpu())
    finally:
        U.tune_set(24, 0)
    (l1, p1, g1), (l0, p0, g0) = res[1], res[0]
    assert float((g0 - g1).abs().max()) < 4e-7 * float(g1.abs().max())
    assert float((p0 - p1).abs().max()) < 1e-6 * float(p1.abs().max())
    assert abs(l0 - l1) < 1e-6 * abs(l1)
    assert float((g0 == g1).float().mean()) > 0.9                      # the summation order only matters for runs >= 3
    d32 = torch.from_numpy(dist).requires_grad_(True)
    l32, p32 = R.dist_nstep_td_error(d32, torch.from_numpy(nd), torch.from_numpy(a), torch.from_numpy(na),
                                     torch.from_numpy(r), torch.from_numpy(done), torch.from_numpy(w), gamma, v_min, v_max, n_atom)
    l32.backward()
    # The reference drops the mass of a source whose projected position is integral (l == u: both weights 0), so a position
    # that is integral on one side and one ulp off on the other (the oracle sums the n-step return in another order) moves
    # a whole p_j: at 3.3e6 projected atoms a few hundred samples meet that discontinuity.  Everything else agrees to 1e-4.
    assert rel_err(l32.item(), l0) < 1e-4
    bad = (p32.detach() - p0).abs() > 1e-4 * float(p32.detach().abs().max())
    assert float(bad.float().mean()) < 5e-3, float(bad.float().mean())
    ok = ~bad
    assert grad_err(d32.grad[ok].numpy(), g0[ok].numpy()) < 1e-4


# ------------------------------------------------------------------------------------------------ IQN / QR-DQN
def test_iqn_golden(golden):
    from hpc_rll.rl_utils.td import IQNNStepTDError
    g = golden("iqn")
    for i, (tau, taup, T, B, N, gam, kappa, has_w, has_vg, _) in enumerate(g["cases"]):
        q = G(g[f"c{i}_q"], True)
        loss, per = IQNNStepTDError(int(tau), int(taup), int(T), int(B), int(N))(
            q, G(g[f"c{i}_next_n_q"]), G(g[f"c{i}_action"]), G(g[f"c{i}_next_n_action"]), G(g[f"c{i}_reward"]),
            G(g[f"c{i}_done"]), G(g[f"c{i}_replay_quantiles"]), float(gam), float(kappa), opt(g, f"c{i}_weight"),
            opt(g, f"c{i}_value_gamma"))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 5e-5
        assert rel_err(g[f"c{i}_td_err"], per.cpu().numpy()) < 5e-5
        assert grad_err(g[f"c{i}_grad_q"], q.grad.cpu().numpy()) < 5e-5


@pytest.mark.parametrize("quirks", [True, False])
def test_iqn_oracle_reference_shape(quirks):
    """tests/test_iqn_nstep_td_error.py:10-29: tau=33, tau'=34, T=10, B=64, N=8, kappa=0.9; `quirks`: done, replay
    quantiles, weight AND value_gamma = randn, exactly as the reference test draws them (negative, non-binary)."""
    from hpc_rll.rl_utils.td import IQNNStepTDError
    tau, taup, T, B, N, kappa = 33, 34, 10, 64, 8, 0.9
    rng = np.random.default_rng(6 + int(quirks))
    q, nq = f32(rng, tau, B, N), f32(rng, taup, B, N)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = f32(rng, T, B)
    if quirks:
        done, rq, w, vg = f32(rng, B), f32(rng, tau, B), f32(rng, B), f32(rng, B)
    else:
        done, rq, w, vg = (rng.random(B) < 0.3).astype(np.float32), rng.random((tau, B)).astype(np.float32), \
            rng.random(B).astype(np.float32), None
    q64 = D(q, True)
    l64, p64 = R.iqn_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(rq), D(w), 0.95,
                                    kappa, None if vg is None else D(vg))
    l64.backward()
    dq = G(q, True)
    loss, per = IQNNStepTDError(tau, taup, T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), G(rq), 0.95, kappa, G(w),
                                                    None if vg is None else G(vg))
    loss.backward()
    assert rel_err(l64.item(), loss.item()) < 2e-5
    assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5
    assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("tau,taup,B,N", [(32, 32, 4099, 64), (33, 34, 64, 8), (5, 7, 1000, 6), (64, 48, 513, 17), (80, 70, 130, 9), (8, 8, 9, 3),
                                         (1, 1, 70, 2), (16, 64, 257, 64)])
def test_iqn_quantile_innermost_layout(tau, taup, B, N):
    """Round 6, ``layout='bnt'`` (not in the reference): q (B,N,tau), next_n_q (B,N,tau') -- a sample's quantiles are one
    contiguous row.  Against the default (tau,B,N) layout on the permuted tensors: the same kernel arithmetic on the same
    values, so loss, per-sample errors and the gradient (compared as grad_bnt.permute(2,0,1)) are the SAME BITS; and against
    the fp64 oracle.  Every group width (tau 1 ... 64), tau != tau', the wave-per-sample kernel (tau > 64), golden cases
    of the reference through the new layout, weights / value_gamma present and absent."""
    from hpc_rll.rl_utils.td import IQNNStepTDError
    T, kappa, gamma = 4, 0.8, 0.93
    rng = np.random.default_rng(tau * 131 + taup)
    q, nq = f32(rng, tau, B, N), f32(rng, taup, B, N)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, rq = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random((tau, B)).astype(np.float32)
    w = rng.random(B).astype(np.float32) if tau % 2 == 0 else None
    vg = (0.9 + 0.1 * rng.random(B)).astype(np.float32) if taup % 2 == 0 else None
    Gw, Gvg = (None if w is None else G(w)), (None if vg is None else G(vg))
    d0 = G(q, True)
    l0, p0 = IQNNStepTDError(tau, taup, T, B, N)(d0, G(nq), G(a), G(na), G(r), G(done), G(rq), gamma, kappa, Gw, Gvg)
    l0.backward()
    d1 = G(np.ascontiguousarray(q.transpose(1, 2, 0)), True)
    l1, p1 = IQNNStepTDError(tau, taup, T, B, N, layout='bnt')(d1, G(np.ascontiguousarray(nq.transpose(1, 2, 0))), G(a), G(na), G(r),
                                                             G(done), G(rq), gamma, kappa, Gw, Gvg)
    l1.backward()
    assert d1.grad.shape == (B, N, tau)
    assert l1.item() == l0.item() and torch.equal(p1, p0)
    assert torch.equal(d1.grad.permute(2, 0, 1), d0.grad)
    q64 = D(q, True)
    l64, p64 = R.iqn_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(rq),
                                    None if w is None else D(w), gamma, kappa, None if vg is None else D(vg))
    l64.backward()
    assert rel_err(l64.item(), l1.item()) < 2e-5
    assert rel_err(p64.detach().numpy(), p1.cpu().numpy()) < 2e-5
    assert grad_err(q64.grad.numpy().transpose(1, 2, 0), d1.grad.cpu().numpy()) < 2e-5
    with pytest.raises(RuntimeError):      # the default layout's shape is rejected in the new one (B read as tau)
        IQNNStepTDError(tau, taup, T, B, N, layout='bnt')(G(q), G(nq), G(a), G(na), G(r), G(done), G(rq), gamma, kappa)


def test_qrdqn_golden(golden):
    from hpc_rll.rl_utils.td import QRDQNNStepTDError
    g = golden("qrdqn")
    for i, (tau, T, B, N, gam, has_w, has_vg, _) in enumerate(g["cases"]):
        q = G(g[f"c{i}_q"], True)
        loss, per = QRDQNNStepTDError(int(tau), int(T), int(B), int(N))(
            q, G(g[f"c{i}_next_n_q"]), G(g[f"c{i}_action"]), G(g[f"c{i}_next_n_action"]), G(g[f"c{i}_reward"]),
            G(g[f"c{i}_done"]), float(gam), opt(g, f"c{i}_weight"), opt(g, f"c{i}_value_gamma"))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 5e-5
        assert rel_err(g[f"c{i}_td_err"], per.cpu().numpy()) < 5e-5
        assert grad_err(g[f"c{i}_grad_q"], q.grad.cpu().numpy()) < 5e-5


@pytest.mark.parametrize("quirks", [True, False])
def test_qrdqn_oracle_reference_shape(quirks):
    """tests/test_qrdqn_nstep_td_error.py:10-24: tau=39, T=10, B=89, N=67; `quirks`: done, weight and value_gamma = randn
    exactly as the reference test draws them."""
    from hpc_rll.rl_utils.td import QRDQNNStepTDError
    tau, T, B, N = 39, 10, 89, 67
    rng = np.random.default_rng(8 + int(quirks))
    q, nq = f32(rng, B, N, tau), f32(rng, B, N, tau)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = f32(rng, T, B)
    if quirks:
        done, w, vg = f32(rng, B), f32(rng, B), f32(rng, B)
    else:
        done, w, vg = (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32), None
    q64 = D(q, True)
    l64, p64 = R.qrdqn_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), tau, D(w), 0.95,
                                      None if vg is None else D(vg))
    l64.backward()
    dq = G(q, True)
    loss, per = QRDQNNStepTDError(tau, T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), 0.95, G(w),
                                               None if vg is None else G(vg))
    loss.backward()
    assert rel_err(l64.item(), loss.item()) < 2e-5
    assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5
    assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5


def test_c51_integral_positions_stay_outside_the_runs():
    """Round 5: the large-batch projection leaves a source at an integral position (l == u: the reference gives it weight 0 on
    both sides, origin/td.py:100-103) out of the runs it sums.  With delta_z = 0.5, gamma = 1 and rewards that are multiples
    of 0.25 every quantity is exact in fp32 and fp64 alike: samples whose return is a multiple of 0.5 have EVERY source at an
    integral position (projection 0, error 0), the others none; returns beyond the support put a dozen clamped -- integral --
    sources at its ends, terminal samples put all 33 on the same position.  Batch kernel against the gather kernel and
    the fp64 oracle."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD
    T, B, N, n_atom = 2, 40000, 3, 33
    v_min, v_max, gamma = -8., 8., 1.0
    rng = np.random.default_rng(5)
    dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
    nd = np.abs(f32(rng, B, N, n_atom))
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = (rng.integers(-24, 25, (T, B)) * 0.25).astype(np.float32)       # n-step returns in [-12, 12]: some leave the support
    done = (rng.random(B) < 0.2).astype(np.float32)
    w = rng.random(B).astype(np.float32)
    res = {}
    try:
        for key in (1, 8, 0):
            U.tune_set(24, key)
            dd = G(dist, True)
            loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), gamma, v_min, v_max)
            loss.backward()
            res[key] = (loss.item(), per.detach().cpu(), dd.grad.cpu())
    finally:
        U.tune_set(24, 0)
    ret = r.sum(0)
    whole = (np.mod(ret * 2.0, 1.0) == 0)                               # return a multiple of delta_z: all positions integral
    assert 0.3 < whole.mean() < 0.7
    for key in (8, 0):
        l0, p0, g0 = res[key]
        assert float(p0[torch.from_numpy(whole)].abs().max()) == 0.0
        assert float((g0 - res[1][2]).abs().max()) < 4e-7 * float(res[1][2].abs().max())
        assert float((p0 - res[1][1]).abs().max()) < 1e-6 * float(res[1][1].abs().max())
    d64 = D(dist, True)
    l64, p64 = R.dist_nstep_td_error(d64, D(nd), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), gamma, v_min, v_max,
                                     n_atom)
    l64.backward()
    assert rel_err(l64.item(), res[0][0]) < 2e-5
    assert rel_err(p64.detach().numpy(), res[0][1].numpy()) < 2e-5
    assert grad_err(d64.grad.numpy(), res[0][2].numpy()) < 2e-5


@pytest.mark.parametrize("tau,B,sw", [(32, 65536 + 37, 0), (32, 40000, 64), (20, 33000, 64), (51, 33000, 64), (5, 70001, 0),
                                      (16, 33001, 32), (64, 20000, 8), (3, 9000, 64), (40, 140000, 0), (8, 33003, 16), (64, 70001, 0),
                                      (12, 40001, 8), (34, 33000, 32), (32, 33000, 16), (48, 262144 + 5, 0)])
def test_qrdqn_samples_per_wave_kernel_equals_group_kernel(tau, B, sw):
    """Large batches: a wave walks SW consecutive samples (csrc/dist_ops.hip: qrdqn_fwd_batch_kernel, and for tau a multiple
    of 4 qrdqn_fwd_quad_kernel; scalars loaded coalesced by owner lanes, targets staged through LDS).  Against the
    group-per-sample kernel (tune key 24 = 1) on ragged batches, every group width, full and partial groups: unit gradients
    bit for bit (batch kernel) or to the order of the tau-term sum (quad kernel), per-sample errors up to the order of the
    sum (DPP instead of butterfly), and both against the fp64 oracle."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import QRDQNNStepTDError
    T, N = 3, 6
    rng = np.random.default_rng(tau * 1000 + sw)
    q, nq = f32(rng, B, N, tau), f32(rng, B, N, tau)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
    vg = (0.9 + 0.1 * rng.random(B)).astype(np.float32)
    res = {}
    try:
        for key in (1, sw):
            U.tune_set(24, key)
            dq = G(q, True)
            loss, per = QRDQNNStepTDError(tau, T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), 0.95, G(w), G(vg))
            loss.backward()
            res[key] = (loss.item(), per.detach().cpu(), dq.grad.cpu())
    finally:
        U.tune_set(24, 0)
    (l1, p1, g1), (l0, p0, g0) = res[1], res[sw]
    if tau % 4 == 0 and tau >= 8:
        # four quantiles per lane (qrdqn_fwd_quad_kernel): a quantile's terms add up in target order, not even / odd targets apart
        assert float((g0 - g1).abs().max()) < 4e-6 * float(g1.abs().max())
    else:
        assert torch.equal(g0, g1), (float((g0 - g1).abs().max()), float(g1.abs().max()), int((g0 != g1).sum()), (g0 != g1).nonzero()[:4].tolist())
    assert float((p0 - p1).abs().max()) < 2e-6 * float(p1.abs().max())
    assert abs(l0 - l1) < 2e-6 * abs(l1)
    n = 4096                                                           # fp64 oracle on the ragged tail
    q64 = D(q[-n:], True)
    l64, p64 = R.qrdqn_nstep_td_error(q64, D(nq[-n:]), torch.from_numpy(a[-n:]), torch.from_numpy(na[-n:]), D(r[:, -n:]),
                                      D(done[-n:]), tau, D(w[-n:]), 0.95, D(vg[-n:]))
    l64.backward()
    assert rel_err(p64.detach().numpy(), p0[-n:].numpy()) < 2e-5
    assert grad_err(q64.grad.numpy() * (n / B), g0[-n:].numpy()) < 2e-5


@pytest.mark.parametrize("B,sw", [(70001, 0), (20011, 0), (9000, 8), (33000, 64)])
def test_c51_samples_per_wave_kernel_ragged(B, sw):
    """dist_nstep_fwd_batch_kernel on batches that are no multiple of the wave's sample count, against the gather kernel."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD
    T, N, n_atom = 2, 3, 51
    rng = np.random.default_rng(B)
    dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
    nd = np.abs(f32(rng, B, N, n_atom))
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
    res = {}
    try:
        for key in (1, sw):
            U.tune_set(24, key)
            dd = G(dist, True)
            loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), 0.97, -5., 5.)
            loss.backward()
            res[key] = (loss.item(), per.detach().cpu(), dd.grad.cpu())
    finally:
        U.tune_set(24, 0)
    (l1, p1, g1), (l0, p0, g0) = res[1], res[sw]
    assert float((g0 - g1).abs().max()) < 4e-7 * float(g1.abs().max())
    assert float((p0 - p1).abs().max()) < 1e-6 * float(p1.abs().max())
    assert abs(l0 - l1) < 1e-6 * abs(l1)


def test_c51_subnormal_probabilities_take_the_division():
    """the samples-per-wave forward computes (-w proj) / p with v_rcp_f32 + one correction; v_rcp flushes subnormal
    inputs, so p in [1e-45, 1e-38] with proj == 0 (every atom but one or two when done = 1) would give 0 * inf = NaN where
    the division of the wave-per-sample kernel (tune key 24 = 1) and the reference give 0.  Lanes with p < FLT_MIN or an
    overflowing quotient take the division: the two kernel families agree on such inputs, NaN-free where proj == 0."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD
    T, B, N, n_atom = 2, 40000, 3, 51
    rng = np.random.default_rng(77)
    dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
    tiny = rng.random((B, N, n_atom)) < 0.3
    dist[tiny] = (10.0 ** rng.uniform(-45, -38, int(tiny.sum()))).astype(np.float32)    # subnormal (some flush to 0: kept > 0 below)
    dist[tiny & (dist == 0)] = np.float32(1e-45)
    nd = np.abs(f32(rng, B, N, n_atom))
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r = (rng.integers(-8, 9, (T, B)) * 0.25).astype(np.float32)
    done = np.ones(B, np.float32)                                   # the whole target mass lands on one or two atoms
    w = rng.random(B).astype(np.float32)
    res = {}
    try:
        for key in (1, 8, 0):
            U.tune_set(24, key)
            dd = G(dist, True)
            loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), 1.0, -5., 5.)
            loss.backward()
            res[key] = (per.detach().cpu(), dd.grad.cpu())
    finally:
        U.tune_set(24, 0)
    p1, g1 = res[1]
    fin = torch.isfinite(g1)
    assert fin.float().mean() > 0.97                                # (a subnormal p under a non-zero projection overflows: in both)
    for key in (8, 0):
        p0, g0 = res[key]
        assert torch.equal(torch.isfinite(g0), fin), (int((~torch.isfinite(g0)).sum()), int((~fin).sum()))
        assert not torch.isnan(g0[fin]).any()
        assert float((g0[fin] - g1[fin]).abs().max()) <= 4e-7 * float(g1[fin].abs().max())
        zero = fin & (g1 == 0)
        assert float(g0[zero].abs().max()) == 0.0
        okp = torch.isfinite(p1)
        assert torch.equal(torch.isfinite(p0), okp)
        assert float((p0[okp] - p1[okp]).abs().max()) <= 1e-6 * float(p1[okp].abs().max())


# ------------------------------------------------------------------------------------------------ misc
@pytest.mark.parametrize("op,B,N,K", [("c51", 20011, 6, 51), ("c51", 4096, 64, 51), ("qrdqn", 33000, 5, 32), ("qrdqn", 9001, 7, 20),
                                      ("qrdqn", 5000, 3, 64), ("qrdqn", 3000, 4, 76), ("c51", 7000, 3, 18)])
def test_onehot_gradient_as_fill_plus_values_is_bit_identical(op, B, N, K):
    """Large one-hot gradients (tune key 31) are written as a fill in the store pattern that reaches the part's write rate
    plus the K values per sample as whole 128-byte lines; against the one-launch kernel (key 31 = 0) on ragged batches, rows
    that are and are not multiples of a line (N * K * 4 % 128), K above and below the 32 lanes of a sample (64, 76, 18),
    every action including 0 and N - 1 (the line rounding stops at the row's ends): the gradients must be the same bits."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD, QRDQNNStepTDError
    T = 2
    rng = np.random.default_rng(B + K)
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    a[:N] = np.arange(N)
    r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
    x = (np.abs(f32(rng, B, N, K)) + 1e-3).astype(np.float32)
    nx = np.abs(f32(rng, B, N, K))
    res = {}
    try:
        for key in (0, 1, "q256", "q4096"):     # 31 = 1: a 1 MiB threshold, every shape here takes the fill path; q*: key 35 (quads per workgroup)
            U.tune_set(31, key if key in (0, 1) else 0)
            U.tune_set(35, 0 if key in (0, 1) else int(key[1:]))
            xx = G(x, True)
            if op == "c51":
                loss, _ = DistNStepTD(T, B, N, K)(xx, G(nx), G(a), G(na), G(r), G(done), G(w), 0.97, -5., 5.)
            else:
                loss, _ = QRDQNNStepTDError(K, T, B, N)(xx, G(nx), G(a), G(na), G(r), G(done), 0.95, G(w), None)
            (loss * 3.0).backward()
            res[key] = xx.grad.clone()
    finally:
        U.tune_set(31, 3072)
        U.tune_set(35, 0)
    assert int((res[0] != 0).sum()) > B * K // 2
    for k in (1, "q256", "q4096"):
        assert torch.equal(res[0], res[k]), (k, int((res[0] != res[k]).sum()), (res[0] != res[k]).nonzero()[:4].tolist())


def test_losses_are_deterministic():
    """No float atomics: two runs give bit-identical losses and gradients."""
    from hpc_rll.rl_utils.vtrace import VTrace
    rng = np.random.default_rng(0)
    T, B, N = 40, 300, 20
    args = (f32(rng, T, B, N), f32(rng, T, B, N), rng.integers(0, N, (T, B)).astype(np.int64), f32(rng, T + 1, B), f32(rng, T, B))
    outs = []
    for _ in range(2):
        to, v = G(args[0], True), G(args[3], True)
        ls = VTrace(T, B, N)(to, G(args[1]), G(args[2]), v, G(args[4]))
        sum(ls).backward()
        outs.append([x.detach().cpu().numpy() for x in (*ls, to.grad, v.grad)])
    for x, y in zip(*outs):
        assert np.array_equal(x, y)


def test_folded_finalisation_under_concurrency():
    """Every scalar-loss forward leaves its loss sums to the LAST workgroup of its last launch (csrc/colscan.hpp:
    publish_sums / ScanFold, an arrival ticket per stream, relaxed agent-scope atomics, no release fence).  Stress it where it could fail: many
    workgroups (B = 8192 columns -> 128+ partials spread over all XCDs), four streams launching concurrently with
    different data, hundreds of back-to-back launches per stream -- every loss must equal the one the separate
    finalize launch (tune key 21 = 0) computes from the same partials, and the ticket must be back at zero (the next
    launch on the stream works)."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import QRDQNNStepTDError, TDLambda
    from hpc_rll.rl_utils.vtrace import VTrace
    T, B, N = 24, 8192, 4
    g = torch.Generator(device=DEV).manual_seed(5)
    nstream, reps = 4, 150
    data = [(torch.randn(T + 1, B, device=DEV, generator=g), torch.randn(T, B, device=DEV, generator=g),
             torch.randn(T, B, N, device=DEV, generator=g), torch.randn(T, B, N, device=DEV, generator=g),
             torch.randint(0, N, (T, B), device=DEV, generator=g)) for _ in range(nstream)]
    td, vt, qr = TDLambda(T, B), VTrace(T, B, N), QRDQNNStepTDError(N, 3, B, T)
    done = torch.zeros(B, device=DEV)

    def losses(v, r, to, bo, a):
        # (the QR-DQN call reads `to` (T,B,N) as q (B', N', tau) = (T, B, N): any dense fp32 block will do here)
        q = to.transpose(0, 1).contiguous()           # (B, T, N): B samples, T actions, N quantiles
        return torch.stack([td(v, r), *vt(to, bo, a, v, r),
                            qr(q, q.flip(0), a[0] % T, a[1] % T, r[:3].contiguous(), done, 0.9)[0]])
    try:
        U.tune_set(21, 0)
        with torch.no_grad():
            want = [losses(*d).cpu() for d in data]
        U.tune_set(21, 1)
        streams = [torch.cuda.Stream() for _ in range(nstream)]
        got = [[] for _ in range(nstream)]
        torch.cuda.synchronize()
        with torch.no_grad():
            for _ in range(reps):
                for i, s in enumerate(streams):
                    with torch.cuda.stream(s):
                        got[i].append(losses(*data[i]))
        torch.cuda.synchronize()
    finally:
        U.tune_set(21, 1)
    for i in range(nstream):
        allv = torch.stack(got[i]).cpu()
        assert torch.equal(allv, want[i].expand_as(allv)), (i, (allv - want[i]).abs().max())
