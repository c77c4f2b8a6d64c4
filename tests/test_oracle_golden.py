"""Pin the oracle: ``oracle/ref_torch.py`` (and the C restatement ``oracle/gae_ref.c``) against the golden
fixtures that tests/golden/make_golden.py recorded from the REAL reference (hpc_rll.origin, imported from
/root/reference in the build container).  Runs on CPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, grad_err, rel_err
from oracle import ref_torch as R

# Gradients of mean-reduced losses are O(1/(T*B)): compared relative to the tensor's own maximum (conftest.grad_err), not with
# rel_err's max(1,|ref|) denominator.  The fixtures are the reference's fp32 results, the oracle runs in
# fp64: measured 1e-7 ... 2.7e-7 of the maximum.
GRAD_TOL = 2e-6

T64 = lambda a: torch.from_numpy(np.asarray(a)).double()  # noqa: E731
TL = lambda a: torch.from_numpy(np.asarray(a))            # noqa: E731


def opt(g, key, conv=T64):
    return conv(g[key]) if key in g.files else None


def test_gae_restatement(golden):
    g = golden("gae")
    for i, (T, B, gam, lam, _) in enumerate(g["cases"]):
        adv = R.gae(T64(g[f"c{i}_value"]), T64(g[f"c{i}_reward"]), gam, lam)
        assert rel_err(g[f"c{i}_adv"], adv.numpy()) < 1e-5
        gv, gr = R.gae_backward(T64(g[f"c{i}_grad_adv"]), gam, lam)
        assert grad_err(g[f"c{i}_grad_value"], gv.numpy()) < GRAD_TOL
        assert grad_err(g[f"c{i}_grad_reward"], gr.numpy()) < GRAD_TOL


@pytest.fixture(scope="module")
def cref():
    so = os.path.join(ROOT, "oracle", "_build", "libgae_ref.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(so)
    fp = ctypes.POINTER(ctypes.c_float)
    lib.gae_ref_forward.argtypes = [fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    lib.gae_ref_backward.argtypes = [fp, fp, fp, fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float]
    return lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def test_gae_c_restatement(golden, cref):
    g = golden("gae")
    for i, (T, B, gam, lam, _) in enumerate(g["cases"]):
        T, B = int(T), int(B)
        v, r, ga = (np.ascontiguousarray(g[f"c{i}_{k}"]) for k in ("value", "reward", "grad_adv"))
        adv = np.empty((T, B), np.float32)
        cref.gae_ref_forward(_fp(v), _fp(r), _fp(adv), T, B, gam, lam)
        assert rel_err(g[f"c{i}_adv"], adv) < 1e-5
        gv, gr, tab = np.empty((T + 1, B), np.float32), np.empty((T, B), np.float32), np.empty(T, np.float32)
        cref.gae_ref_backward(_fp(ga), _fp(gv), _fp(gr), _fp(tab), T, B, gam, lam)
        assert grad_err(g[f"c{i}_grad_value"], gv) < 2e-5      # fp32 C scan against the reference's fp32 autograd
        assert grad_err(g[f"c{i}_grad_reward"], gr) < 2e-5


def test_gae_big_shape_statistics(golden, cref):
    """(1024,64) and (256,256): the reference test shapes; fixtures hold seeds + summary statistics."""
    g = golden("gae")
    for row in g["big"]:
        T, B, seed = int(row[0]), int(row[1]), int(row[2])
        rng = np.random.default_rng(seed)
        v = rng.standard_normal((T + 1, B)).astype(np.float32)
        r = rng.standard_normal((T, B)).astype(np.float32)
        ga = rng.standard_normal((T, B)).astype(np.float32)
        adv = np.empty((T, B), np.float32)
        cref.gae_ref_forward(_fp(v), _fp(r), _fp(adv), T, B, 0.99, 0.97)
        gv, gr, tab = np.empty((T + 1, B), np.float32), np.empty((T, B), np.float32), np.empty(T, np.float32)
        cref.gae_ref_backward(_fp(ga), _fp(gv), _fp(gr), _fp(tab), T, B, 0.99, 0.97)
        probe = np.random.default_rng(99).integers(0, T * B, 16)
        assert abs(adv.astype(np.float64).sum() - row[3]) <= 1e-5 * row[4]
        assert abs(np.abs(adv.astype(np.float64)).sum() - row[4]) <= 1e-5 * row[4]
        assert abs(np.abs(gv.astype(np.float64)).sum() - row[6]) <= 1e-5 * row[6]
        assert abs(np.abs(gr.astype(np.float64)).sum() - row[7]) <= 1e-5 * row[7]
        assert rel_err(row[8:24], adv.reshape(-1)[probe]) < 1e-5
        assert rel_err(row[24:40], gv.reshape(-1)[probe]) < 2e-5


def test_td_lambda(golden):
    g = golden("td_lambda")
    for i, (T, B, gam, lam, has_w, _) in enumerate(g["cases"]):
        v = T64(g[f"c{i}_value"]).requires_grad_(True)
        loss = R.td_lambda_error(v, T64(g[f"c{i}_reward"]), opt(g, f"c{i}_weight"), gam, lam)
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 1e-5
        assert grad_err(g[f"c{i}_grad_value"], v.grad.numpy()) < GRAD_TOL


def test_vtrace(golden):
    g = golden("vtrace")
    co = g["coef"]
    for i, (T, B, N, gam, lam, rc, cc, pc, has_w, _) in enumerate(g["cases"]):
        to = T64(g[f"c{i}_target_output"]).requires_grad_(True)
        v = T64(g[f"c{i}_value"]).requires_grad_(True)
        ls = R.vtrace_error(to, T64(g[f"c{i}_behaviour_output"]), TL(g[f"c{i}_action"]), v, T64(g[f"c{i}_reward"]),
                            opt(g, f"c{i}_weight"), gam, lam, rc, cc, pc)
        (co[0] * ls[0] + co[1] * ls[1] + co[2] * ls[2]).backward()
        assert rel_err(g[f"c{i}_losses"], [x.item() for x in ls]) < 1e-5
        assert grad_err(g[f"c{i}_grad_target_output"], to.grad.numpy()) < GRAD_TOL
        assert grad_err(g[f"c{i}_grad_value"], v.grad.numpy()) < GRAD_TOL


def test_upgo(golden):
    g = golden("upgo")
    for i in range(len(g["cases"])):
        to = T64(g[f"c{i}_target_output"]).requires_grad_(True)
        loss = R.upgo_loss(to, T64(g[f"c{i}_rhos"]), TL(g[f"c{i}_action"]), T64(g[f"c{i}_reward"]), T64(g[f"c{i}_value"]))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 1e-5
        assert grad_err(g[f"c{i}_grad_target_output"], to.grad.numpy()) < GRAD_TOL


def test_ppo(golden):
    g = golden("ppo")
    co = g["coef"]
    for i, (B, N, clip, uvc, dc, has_w, _) in enumerate(g["cases"]):
        ln = T64(g[f"c{i}_logit_new"]).requires_grad_(True)
        vn = T64(g[f"c{i}_value_new"]).requires_grad_(True)
        ls, info = R.ppo_error(ln, T64(g[f"c{i}_logit_old"]), TL(g[f"c{i}_action"]), vn, T64(g[f"c{i}_value_old"]),
                               T64(g[f"c{i}_adv"]), T64(g[f"c{i}_return_"]), opt(g, f"c{i}_weight"),
                               clip, bool(uvc), dc if dc else None)
        (co[0] * ls[0] + co[1] * ls[1] + co[2] * ls[2]).backward()
        assert rel_err(g[f"c{i}_losses"], [x.item() for x in ls]) < 1e-5
        assert rel_err(g[f"c{i}_info"], list(info)) < 1e-5
        assert grad_err(g[f"c{i}_grad_logit_new"], ln.grad.numpy()) < GRAD_TOL
        assert grad_err(g[f"c{i}_grad_value_new"], vn.grad.numpy()) < GRAD_TOL


def test_qntd(golden):
    g = golden("qntd")
    for i, (T, B, N, gam, has_w, _) in enumerate(g["cases"]):
        for tag, resc in (("plain", False), ("rescale", True)):
            q = T64(g[f"c{i}_q"]).requires_grad_(True)
            loss, per = R.q_nstep_td_error(q, T64(g[f"c{i}_next_n_q"]), TL(g[f"c{i}_action"]), TL(g[f"c{i}_next_n_action"]),
                                           T64(g[f"c{i}_reward"]), T64(g[f"c{i}_done"]), opt(g, f"c{i}_weight"), gam, resc)
            loss.backward()
            assert rel_err(g[f"c{i}_{tag}_loss"], loss.item()) < 5e-5
            assert rel_err(g[f"c{i}_{tag}_td_err"], per.detach().numpy()) < 5e-5
            assert grad_err(g[f"c{i}_{tag}_grad_q"], q.grad.numpy()) < GRAD_TOL


def test_dntd(golden):
    g = golden("dntd")
    for i, (T, B, N, na, gam, vmin, vmax, has_w, _) in enumerate(g["cases"]):
        # fp32 on purpose: floor/ceil of the projected position is discontinuous
        d = TL(g[f"c{i}_dist"]).clone().requires_grad_(True)
        loss, per = R.dist_nstep_td_error(d, TL(g[f"c{i}_next_n_dist"]), TL(g[f"c{i}_action"]), TL(g[f"c{i}_next_n_action"]),
                                          TL(g[f"c{i}_reward"]), TL(g[f"c{i}_done"]), opt(g, f"c{i}_weight", TL),
                                          gam, vmin, vmax, int(na))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 1e-4
        assert rel_err(g[f"c{i}_td_err"], per.detach().numpy()) < 1e-4
        assert grad_err(g[f"c{i}_grad_dist"], d.grad.numpy()) < 5e-6   # both sides fp32 here (floor / ceil of the projection)


def test_iqn(golden):
    g = golden("iqn")
    for i, (tau, taup, T, B, N, gam, kappa, has_w, has_vg, _) in enumerate(g["cases"]):
        q = T64(g[f"c{i}_q"]).requires_grad_(True)
        loss, per = R.iqn_nstep_td_error(q, T64(g[f"c{i}_next_n_q"]), TL(g[f"c{i}_action"]), TL(g[f"c{i}_next_n_action"]),
                                         T64(g[f"c{i}_reward"]), T64(g[f"c{i}_done"]), T64(g[f"c{i}_replay_quantiles"]),
                                         opt(g, f"c{i}_weight"), gam, kappa, opt(g, f"c{i}_value_gamma"))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 5e-5
        assert rel_err(g[f"c{i}_td_err"], per.detach().numpy()) < 5e-5
        assert grad_err(g[f"c{i}_grad_q"], q.grad.numpy()) < GRAD_TOL


def test_qrdqn(golden):
    g = golden("qrdqn")
    for i, (tau, T, B, N, gam, has_w, has_vg, _) in enumerate(g["cases"]):
        q = T64(g[f"c{i}_q"]).requires_grad_(True)
        loss, per = R.qrdqn_nstep_td_error(q, T64(g[f"c{i}_next_n_q"]), TL(g[f"c{i}_action"]), TL(g[f"c{i}_next_n_action"]),
                                           T64(g[f"c{i}_reward"]), T64(g[f"c{i}_done"]), int(tau), opt(g, f"c{i}_weight"),
                                           gam, opt(g, f"c{i}_value_gamma"))
        loss.backward()
        assert rel_err(g[f"c{i}_loss"], loss.item()) < 5e-5
        assert rel_err(g[f"c{i}_td_err"], per.detach().numpy()) < 5e-5
        assert grad_err(g[f"c{i}_grad_q"], q.grad.numpy()) < GRAD_TOL


def test_padding_bit_exact(golden):
    g = golden("padding")
    for nd in (1, 2, 3):
        xs = [TL(g[f"d{nd}_x{j}"]) for j in range(int(g[f"d{nd}_n"]))]
        for value in (0, -3):
            new_x, mask, shapes = R.pad(xs, value)
            assert np.array_equal(new_x.numpy(), g[f"d{nd}_v{value}_new_x"])
            assert np.array_equal(mask.numpy(), g[f"d{nd}_v{value}_mask"])
        back = R.unpad(new_x, shapes)
        assert all(torch.equal(a, b) for a, b in zip(xs, back))
    for j in range(int(g["split_n"])):
        pos = R.oracle_split_group([int(v) for v in g[f"split{j}_numels"]], int(g[f"split{j}_group"]))
        assert pos == [int(v) for v in g[f"split{j}_pos"]]


def test_scatter(golden):
    g = golden("scatter")
    for i, (B, M, N, H, W, _) in enumerate(g["cases"]):
        for st in ("add", "cover"):
            x = TL(g[f"c{i}_x"]).clone().requires_grad_(True)
            o = R.scatter_connection(x, TL(g[f"c{i}_location"]), int(H), int(W), st)
            (o * o).mean().backward()
            if st == "cover":
                assert np.array_equal(o.detach().numpy(), g[f"c{i}_cover_out"])
            else:
                assert rel_err(g[f"c{i}_add_out"], o.detach().numpy()) < 1e-6
            assert grad_err(g[f"c{i}_{st}_grad_x"], x.grad.numpy()) < GRAD_TOL


def test_lstm(golden):
    g = golden("lstm")
    for i, (S, B, I, H, L, _) in enumerate(g["cases"]):
        L = int(L)
        leaf = lambda a: T64(a).requires_grad_(True)  # noqa: E731
        x, h0, c0 = leaf(g[f"c{i}_x"]), leaf(g[f"c{i}_h0"]), leaf(g[f"c{i}_c0"])
        wx = [leaf(g[f"c{i}_wx{l}"]) for l in range(L)]
        wh = [leaf(g[f"c{i}_wh{l}"]) for l in range(L)]
        bias, gam, beta = leaf(g[f"c{i}_bias"]), leaf(g[f"c{i}_ln_gamma"]), leaf(g[f"c{i}_ln_beta"])
        y, hn, cn = R.lstm(x, h0, c0, wx, wh, bias, gam, beta)
        ((y * T64(g[f"c{i}_gy"])).sum() + (hn * T64(g[f"c{i}_gh"])).sum() + (cn * T64(g[f"c{i}_gc"])).sum()).backward()
        assert rel_err(g[f"c{i}_y"], y.detach().numpy()) < 1e-5
        assert rel_err(g[f"c{i}_hn"], hn.detach().numpy()) < 1e-5
        assert rel_err(g[f"c{i}_cn"], cn.detach().numpy()) < 1e-5
        tol = 2e-4  # golden grads are fp32 autograd through S*L LayerNorms
        assert grad_err(g[f"c{i}_grad_x"], x.grad.numpy()) < tol
        assert grad_err(g[f"c{i}_grad_h0"], h0.grad.numpy()) < tol
        assert grad_err(g[f"c{i}_grad_c0"], c0.grad.numpy()) < tol
        assert grad_err(g[f"c{i}_grad_bias"], bias.grad.numpy()) < tol
        assert grad_err(g[f"c{i}_grad_ln_gamma"], gam.grad.numpy()) < tol
        assert grad_err(g[f"c{i}_grad_ln_beta"], beta.grad.numpy()) < tol
        for l in range(L):
            assert grad_err(g[f"c{i}_grad_wx{l}"], wx[l].grad.numpy()) < tol
            assert grad_err(g[f"c{i}_grad_wh{l}"], wh[l].grad.numpy()) < tol
