#!/usr/bin/env python3
"""Generate the committed golden fixtures from the REAL reference oracle.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``hpc_rll.origin`` from /root/reference (pure PyTorch, CPU), feeds it seeded inputs
(numpy ``default_rng`` so the inputs are reproducible independent of the torch version), records
outputs and autograd gradients, cross-checks this repo's ``oracle/ref_torch.py`` restatement against
them on the spot (fails loudly on mismatch), and writes ``tests/golden/<op>.npz``.

The fixtures are data only: inputs, expected outputs, expected gradients.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("make_golden.py needs /root/reference (build container only)")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from hpc_rll.origin import gae as o_gae            # noqa: E402
from hpc_rll.origin import td as o_td              # noqa: E402
from hpc_rll.origin import vtrace as o_vtrace      # noqa: E402
from hpc_rll.origin import upgo as o_upgo          # noqa: E402
from hpc_rll.origin import ppo as o_ppo            # noqa: E402
from hpc_rll.origin import padding as o_pad        # noqa: E402
from hpc_rll.origin import scatter_connection as o_sc  # noqa: E402
from hpc_rll.origin import rnn as o_rnn            # noqa: E402

from oracle import ref_torch as R                  # noqa: E402

torch.set_num_threads(4)


def rn(rng, *shape):
    return rng.standard_normal(shape).astype(np.float32)


def T_(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if grad:
        t.requires_grad_(True)
    return t


def close(name, a, b, tol=2e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.max(np.abs(a - b) / np.maximum(1.0, np.abs(a))) if a.size else 0.0
    assert err <= tol, f"oracle restatement mismatch for {name}: {err}"
    return err


def gclose(name, a, b, tol=2e-6):
    """Gradients: relative to the tensor's own maximum (they are O(1/(T*B)) for mean-reduced losses, so close()'s
    max(1,|a|) denominator would accept errors of several percent of the tensor -- VERDICT r03 weak #2)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = float(np.max(np.abs(a))) if a.size else 0.0
    err = float(np.max(np.abs(a - b))) / scale if scale > 0 else float(np.max(np.abs(b))) if b.size else 0.0
    assert err <= tol, f"oracle restatement mismatch for {name} (relative to max |ref| = {scale:.3e}): {err}"
    return err


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def d64(a):
    return torch.from_numpy(np.asarray(a)).double()


# ---------------------------------------------------------------- GAE
def gen_gae():
    out = {}
    cases = [(64, 32, 0.99, 0.97, 1), (257, 65, 0.99, 0.97, 2), (40, 3, 0.9, 1.0, 3), (7, 130, 1.0, 0.0, 4), (1, 5, 0.99, 0.97, 5)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (T, B, g, l, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        v, r, ga = rn(rng, T + 1, B), rn(rng, T, B), rn(rng, T, B)
        tv, tr = T_(v, True), T_(r, True)
        adv = o_gae.gae(o_gae.gae_data(tv, tr), g, l)
        adv.backward(T_(ga))
        out[f"c{i}_value"], out[f"c{i}_reward"], out[f"c{i}_grad_adv"] = v, r, ga
        out[f"c{i}_adv"] = adv.detach().numpy()
        out[f"c{i}_grad_value"], out[f"c{i}_grad_reward"] = tv.grad.numpy(), tr.grad.numpy()
        # cross-check restatement (fp64) + analytic adjoint
        a64 = R.gae(d64(v), d64(r), g, l)
        close("gae fwd", adv.detach().numpy(), a64.numpy())
        gv, gr = R.gae_backward(d64(ga), g, l)
        gclose("gae grad_value", tv.grad.numpy(), gv.numpy(), 1e-5)
        gclose("gae grad_reward", tr.grad.numpy(), gr.numpy(), 1e-5)
    # big shapes of the reference tests: seeds + summary statistics only
    big = []
    for (T, B, seed) in [(1024, 64, 11), (256, 256, 12)]:
        rng = np.random.default_rng(seed)
        v, r, ga = rn(rng, T + 1, B), rn(rng, T, B), rn(rng, T, B)
        tv, tr = T_(v, True), T_(r, True)
        adv = o_gae.gae(o_gae.gae_data(tv, tr), 0.99, 0.97)
        adv.backward(T_(ga))
        a = adv.detach().numpy().astype(np.float64)
        probe = np.random.default_rng(99).integers(0, T * B, 16)
        big.append(np.concatenate([[T, B, seed, a.sum(), np.abs(a).sum(), tv.grad.double().sum().item(),
                                    tv.grad.double().abs().sum().item(), tr.grad.double().abs().sum().item()],
                                   a.reshape(-1)[probe], tv.grad.numpy().reshape(-1)[probe].astype(np.float64)]))
    out["big"] = np.stack(big)
    save("gae", **out)


# ---------------------------------------------------------------- TD(lambda)
def gen_td_lambda():
    out = {}
    cases = [(33, 17, 0.9, 0.8, 1, 1), (128, 5, 0.99, 0.95, 0, 2), (2, 70, 0.9, 0.8, 1, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (T, B, g, l, has_w, seed) in enumerate(cases):
        rng = np.random.default_rng(100 + seed)
        v, r = rn(rng, T + 1, B), rn(rng, T, B)
        w = rng.random((T, B)).astype(np.float32) if has_w else None
        tv = T_(v, True)
        loss = o_td.td_lambda_error(o_td.td_lambda_data(tv, T_(r), None if w is None else T_(w)), g, l)
        loss.backward()
        out[f"c{i}_value"], out[f"c{i}_reward"] = v, r
        if w is not None:
            out[f"c{i}_weight"] = w
        out[f"c{i}_loss"] = loss.detach().numpy()
        out[f"c{i}_grad_value"] = tv.grad.numpy()
        v64 = d64(v).requires_grad_(True)
        l64 = R.td_lambda_error(v64, d64(r), None if w is None else d64(w), g, l)
        l64.backward()
        close("tdl loss", loss.item(), l64.item())
        gclose("tdl grad", tv.grad.numpy(), v64.grad.numpy(), 2e-6)
    save("td_lambda", **out)


# ---------------------------------------------------------------- V-trace
def gen_vtrace():
    out = {}
    # last case: action masking, ~25% of the NOT-taken actions carry logit = -inf in both policies
    cases = [(19, 7, 11, 0.99, 0.95, 1.0, 1.0, 1.0, 0, 1), (12, 5, 33, 0.9, 0.8, 0.7, 1.3, 2.0, 1, 2), (3, 70, 4, 0.99, 1.0, 1.0, 1.0, 1.0, 1, 3),
             (6, 9, 12, 0.99, 0.95, 1.0, 1.0, 1.0, 1, 4)]
    out["cases"] = np.array(cases, dtype=np.float64)
    coef = (1.0, 0.5, -0.01)
    out["coef"] = np.array(coef)
    for i, (T, B, N, g, l, rc, cc, pc, has_w, seed) in enumerate(cases):
        rng = np.random.default_rng(200 + seed)
        to, bo = rn(rng, T, B, N), rn(rng, T, B, N)
        a = rng.integers(0, N, (T, B)).astype(np.int64)
        if seed == 4:
            masked = rng.random((T, B, N)) < 0.25
            np.put_along_axis(masked, a[..., None], False, axis=-1)
            to[masked] = -np.inf
            bo[masked] = -np.inf
        v, r = rn(rng, T + 1, B), rn(rng, T, B)
        w = rng.random((T, B)).astype(np.float32) if has_w else None
        tto, tv = T_(to, True), T_(v, True)
        ls = o_vtrace.vtrace_error(o_vtrace.vtrace_data(tto, T_(bo), T_(a), tv, T_(r), None if w is None else T_(w)),
                                   g, l, rc, cc, pc)
        (coef[0] * ls[0] + coef[1] * ls[1] + coef[2] * ls[2]).backward()
        out[f"c{i}_target_output"], out[f"c{i}_behaviour_output"], out[f"c{i}_action"] = to, bo, a
        out[f"c{i}_value"], out[f"c{i}_reward"] = v, r
        if w is not None:
            out[f"c{i}_weight"] = w
        out[f"c{i}_losses"] = np.array([x.item() for x in ls], dtype=np.float64)
        out[f"c{i}_grad_target_output"], out[f"c{i}_grad_value"] = tto.grad.numpy(), tv.grad.numpy()
        to64, v64 = d64(to).requires_grad_(True), d64(v).requires_grad_(True)
        l64 = R.vtrace_error(to64, d64(bo), T_(a), v64, d64(r), None if w is None else d64(w), g, l, rc, cc, pc)
        (coef[0] * l64[0] + coef[1] * l64[1] + coef[2] * l64[2]).backward()
        close("vtrace losses", [x.item() for x in ls], [x.item() for x in l64])
        gclose("vtrace gto", tto.grad.numpy(), to64.grad.numpy(), 2e-6)
        gclose("vtrace gv", tv.grad.numpy(), v64.grad.numpy(), 2e-6)
    save("vtrace", **out)


# ---------------------------------------------------------------- UPGO
def gen_upgo():
    out = {}
    cases = [(21, 9, 6, 1), (8, 66, 17, 2), (2, 3, 130, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (T, B, N, seed) in enumerate(cases):
        rng = np.random.default_rng(300 + seed)
        to = rn(rng, T, B, N)
        rho = rn(rng, T, B)                      # reference test feeds randn rhos (tests/test_upgo.py:16)
        a = rng.integers(0, N, (T, B)).astype(np.int64)
        r, v = rn(rng, T, B), rn(rng, T + 1, B)
        tto = T_(to, True)
        loss = o_upgo.upgo_loss(tto, T_(rho), T_(a), T_(r), T_(v))
        loss.backward()
        out[f"c{i}_target_output"], out[f"c{i}_rhos"], out[f"c{i}_action"] = to, rho, a
        out[f"c{i}_reward"], out[f"c{i}_value"] = r, v
        out[f"c{i}_loss"] = loss.detach().numpy()
        out[f"c{i}_grad_target_output"] = tto.grad.numpy()
        to64 = d64(to).requires_grad_(True)
        l64 = R.upgo_loss(to64, d64(rho), T_(a), d64(r), d64(v))
        l64.backward()
        close("upgo loss", loss.item(), l64.item())
        gclose("upgo grad", tto.grad.numpy(), to64.grad.numpy(), 2e-6)
    save("upgo", **out)


# ---------------------------------------------------------------- PPO
def gen_ppo():
    out = {}
    # B, N, clip, use_value_clip, dual_clip(0=None), has_w, seed
    # last case: action masking (-inf logits on ~25% of the not-taken actions)
    cases = [(37, 13, 0.2, 1, 0.0, 0, 1), (64, 5, 0.1, 0, 3.0, 1, 2), (5, 130, 0.3, 1, 1.5, 1, 3), (40, 10, 0.2, 1, 0.0, 1, 4)]
    out["cases"] = np.array(cases, dtype=np.float64)
    coef = (1.0, 0.5, -0.01)
    out["coef"] = np.array(coef)
    for i, (B, N, clip, uvc, dc, has_w, seed) in enumerate(cases):
        rng = np.random.default_rng(400 + seed)
        ln, lo = rn(rng, B, N), rn(rng, B, N)
        lo = (ln + 0.3 * lo).astype(np.float32)       # keep ratios near 1 so both clip branches fire
        a = rng.integers(0, N, (B,)).astype(np.int64)
        if seed == 4:
            masked = rng.random((B, N)) < 0.25
            np.put_along_axis(masked, a[:, None], False, axis=-1)
            ln[masked] = -np.inf
            lo[masked] = -np.inf
        vn, vo, adv, ret = rn(rng, B), rn(rng, B), rn(rng, B), rn(rng, B)
        w = rng.random(B).astype(np.float32) if has_w else None
        tln, tvn = T_(ln, True), T_(vn, True)
        ls, info = o_ppo.ppo_error(o_ppo.ppo_data(tln, T_(lo), T_(a), tvn, T_(vo), T_(adv), T_(ret),
                                                  None if w is None else T_(w)), clip, bool(uvc), dc if dc else None)
        (coef[0] * ls[0] + coef[1] * ls[1] + coef[2] * ls[2]).backward()
        for k, arr in dict(logit_new=ln, logit_old=lo, action=a, value_new=vn, value_old=vo, adv=adv, return_=ret).items():
            out[f"c{i}_{k}"] = arr
        if w is not None:
            out[f"c{i}_weight"] = w
        out[f"c{i}_losses"] = np.array([x.item() for x in ls], dtype=np.float64)
        out[f"c{i}_info"] = np.array(list(info), dtype=np.float64)
        out[f"c{i}_grad_logit_new"], out[f"c{i}_grad_value_new"] = tln.grad.numpy(), tvn.grad.numpy()
        ln64, vn64 = d64(ln).requires_grad_(True), d64(vn).requires_grad_(True)
        l64, i64 = R.ppo_error(ln64, d64(lo), T_(a), vn64, d64(vo), d64(adv), d64(ret), None if w is None else d64(w),
                               clip, bool(uvc), dc if dc else None)
        (coef[0] * l64[0] + coef[1] * l64[1] + coef[2] * l64[2]).backward()
        close("ppo losses", [x.item() for x in ls], [x.item() for x in l64])
        close("ppo info", list(info), list(i64), 1e-5)
        gclose("ppo gl", tln.grad.numpy(), ln64.grad.numpy(), 2e-6)
        gclose("ppo gv", tvn.grad.numpy(), vn64.grad.numpy(), 2e-6)
    save("ppo", **out)


# ---------------------------------------------------------------- q n-step TD (+rescale)
def gen_qntd():
    out = {}
    cases = [(5, 33, 7, 0.95, 0, 1), (3, 70, 4, 0.99, 1, 2), (1, 8, 130, 0.9, 1, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (T, B, N, g, has_w, seed) in enumerate(cases):
        rng = np.random.default_rng(500 + seed)
        q, nq = rn(rng, B, N), rn(rng, B, N)
        a, na = rng.integers(0, N, (B,)).astype(np.int64), rng.integers(0, N, (B,)).astype(np.int64)
        r = rn(rng, T, B)
        done = (rng.random(B) < 0.3).astype(np.float32)
        w = rng.random(B).astype(np.float32) if has_w else None
        for k, arr in dict(q=q, next_n_q=nq, action=a, next_n_action=na, reward=r, done=done).items():
            out[f"c{i}_{k}"] = arr
        if w is not None:
            out[f"c{i}_weight"] = w
        for tag, fn, resc in (("plain", o_td.q_nstep_td_error, False), ("rescale", o_td.q_nstep_td_error_with_rescale, True)):
            tq = T_(q, True)
            loss, per = fn(o_td.q_nstep_td_data(tq, T_(nq), T_(a), T_(na), T_(r), T_(done), None if w is None else T_(w)), g, T)
            loss.backward()
            out[f"c{i}_{tag}_loss"], out[f"c{i}_{tag}_td_err"] = loss.detach().numpy(), per.detach().numpy()
            out[f"c{i}_{tag}_grad_q"] = tq.grad.numpy()
            q64 = d64(q).requires_grad_(True)
            l64, p64 = R.q_nstep_td_error(q64, d64(nq), T_(a), T_(na), d64(r), d64(done), None if w is None else d64(w), g, resc)
            l64.backward()
            close("qntd loss " + tag, loss.item(), l64.item(), 5e-5)
            close("qntd per " + tag, per.detach().numpy(), p64.detach().numpy(), 5e-5)
            gclose("qntd grad " + tag, tq.grad.numpy(), q64.grad.numpy(), 2e-6)
    save("qntd", **out)


# ---------------------------------------------------------------- dist n-step TD (C51)
def gen_dntd():
    out = {}
    cases = [(3, 16, 5, 51, 0.95, -10.0, 10.0, 0, 1), (1, 70, 3, 21, 0.99, -2.0, 3.0, 1, 2), (4, 9, 2, 300, 0.9, -10.0, 10.0, 1, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (T, B, N, na_, g, vmin, vmax, has_w, seed) in enumerate(cases):
        rng = np.random.default_rng(600 + seed)
        dist = (np.abs(rn(rng, B, N, na_)) + 1e-3).astype(np.float32)
        ndist = (np.abs(rn(rng, B, N, na_))).astype(np.float32)
        a, na = rng.integers(0, N, (B,)).astype(np.int64), rng.integers(0, N, (B,)).astype(np.int64)
        r = rn(rng, T, B)
        done = (rng.random(B) < 0.3).astype(np.float32)
        w = rng.random(B).astype(np.float32) if has_w else None
        td = T_(dist, True)
        loss, per = o_td.dist_nstep_td_error(o_td.dist_nstep_td_data(td, T_(ndist), T_(a), T_(na), T_(r), T_(done),
                                                                    None if w is None else T_(w)), g, vmin, vmax, na_, T)
        loss.backward()
        for k, arr in dict(dist=dist, next_n_dist=ndist, action=a, next_n_action=na, reward=r, done=done).items():
            out[f"c{i}_{k}"] = arr
        if w is not None:
            out[f"c{i}_weight"] = w
        out[f"c{i}_loss"], out[f"c{i}_td_err"] = loss.detach().numpy(), per.detach().numpy()
        out[f"c{i}_grad_dist"] = td.grad.numpy()
        # the restatement must be run in fp32 too: floor/ceil of b is discontinuous, an fp64 b can land
        # on the other side of an integer.  Compare in fp32.
        d32 = T_(dist, True)
        l32, p32 = R.dist_nstep_td_error(d32, T_(ndist), T_(a), T_(na), T_(r), T_(done), None if w is None else T_(w),
                                         g, vmin, vmax, na_)
        l32.backward()
        close("dntd loss", loss.item(), l32.item(), 1e-4)
        close("dntd per", per.detach().numpy(), p32.detach().numpy(), 1e-4)
        gclose("dntd grad", td.grad.numpy(), d32.grad.numpy(), 5e-6)
    save("dntd", **out)


# ---------------------------------------------------------------- IQN / QRDQN
def gen_iqn():
    out = {}
    cases = [(9, 10, 3, 12, 4, 0.95, 0.9, 0, 0, 1), (33, 34, 2, 5, 3, 0.99, 1.0, 1, 1, 2), (70, 3, 1, 4, 2, 0.9, 0.5, 1, 0, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (tau, taup, T, B, N, g, kappa, has_w, has_vg, seed) in enumerate(cases):
        rng = np.random.default_rng(700 + seed)
        q, nq = rn(rng, tau, B, N), rn(rng, taup, B, N)
        a, na = rng.integers(0, N, (B,)).astype(np.int64), rng.integers(0, N, (B,)).astype(np.int64)
        r = rn(rng, T, B)
        done = (rng.random(B) < 0.3).astype(np.float32)
        rq = rng.random((tau, B)).astype(np.float32)
        w = rng.random(B).astype(np.float32) if has_w else None
        vg = rng.random(B).astype(np.float32) if has_vg else None
        tq = T_(q, True)
        loss, per = o_td.iqn_nstep_td_error(
            o_td.iqn_nstep_td_data(tq, T_(nq), T_(a), T_(na), T_(r), T_(done), T_(rq), None if w is None else T_(w)),
            g, T, kappa, None if vg is None else T_(vg))
        loss.backward()
        for k, arr in dict(q=q, next_n_q=nq, action=a, next_n_action=na, reward=r, done=done, replay_quantiles=rq).items():
            out[f"c{i}_{k}"] = arr
        if w is not None:
            out[f"c{i}_weight"] = w
        if vg is not None:
            out[f"c{i}_value_gamma"] = vg
        out[f"c{i}_loss"], out[f"c{i}_td_err"], out[f"c{i}_grad_q"] = loss.detach().numpy(), per.detach().numpy(), tq.grad.numpy()
        q64 = d64(q).requires_grad_(True)
        l64, p64 = R.iqn_nstep_td_error(q64, d64(nq), T_(a), T_(na), d64(r), d64(done), d64(rq), None if w is None else d64(w),
                                        g, kappa, None if vg is None else d64(vg))
        l64.backward()
        close("iqn loss", loss.item(), l64.item(), 5e-5)
        close("iqn per", per.detach().numpy(), p64.detach().numpy(), 5e-5)
        gclose("iqn grad", tq.grad.numpy(), q64.grad.numpy(), 2e-6)
    save("iqn", **out)


def gen_qrdqn():
    out = {}
    cases = [(7, 3, 11, 5, 0.95, 0, 0, 1), (39, 2, 6, 3, 0.99, 1, 1, 2), (70, 1, 3, 2, 0.9, 1, 0, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (tau, T, B, N, g, has_w, has_vg, seed) in enumerate(cases):
        rng = np.random.default_rng(800 + seed)
        q, nq = rn(rng, B, N, tau), rn(rng, B, N, tau)
        a, na = rng.integers(0, N, (B,)).astype(np.int64), rng.integers(0, N, (B,)).astype(np.int64)
        r = rn(rng, T, B)
        done = (rng.random(B) < 0.3).astype(np.float32)
        w = rng.random(B).astype(np.float32) if has_w else None
        vg = rng.random(B).astype(np.float32) if has_vg else None
        tq = T_(q, True)
        # the reference test passes the integer count as ``tau`` (tests/test_qrdqn_nstep_td_error.py:57)
        loss, per = o_td.qrdqn_nstep_td_error(
            o_td.qrdqn_nstep_td_data(tq, T_(nq), T_(a), T_(na), T_(r), T_(done), tau, None if w is None else T_(w)),
            g, T, None if vg is None else T_(vg))
        loss.backward()
        for k, arr in dict(q=q, next_n_q=nq, action=a, next_n_action=na, reward=r, done=done).items():
            out[f"c{i}_{k}"] = arr
        if w is not None:
            out[f"c{i}_weight"] = w
        if vg is not None:
            out[f"c{i}_value_gamma"] = vg
        out[f"c{i}_loss"], out[f"c{i}_td_err"], out[f"c{i}_grad_q"] = loss.detach().numpy(), per.detach().numpy(), tq.grad.numpy()
        q64 = d64(q).requires_grad_(True)
        l64, p64 = R.qrdqn_nstep_td_error(q64, d64(nq), T_(a), T_(na), d64(r), d64(done), tau, None if w is None else d64(w),
                                          g, None if vg is None else d64(vg))
        l64.backward()
        close("qrdqn loss", loss.item(), l64.item(), 5e-5)
        close("qrdqn per", per.detach().numpy(), p64.detach().numpy(), 5e-5)
        gclose("qrdqn grad", tq.grad.numpy(), q64.grad.numpy(), 2e-6)
    save("qrdqn", **out)


# ---------------------------------------------------------------- Padding
def gen_padding():
    out = {}
    rng = np.random.default_rng(900)
    # 1-D / 2-D / 3-D ragged lists (reference ranges scaled down: tests/test_padding.py:10-13)
    lists = {
        1: [rn(rng, int(n)) for n in rng.integers(3, 20, 9)],
        2: [rn(rng, int(a), int(b)) for a, b in zip(rng.integers(2, 7, 6), rng.integers(1, 9, 6))],
        3: [rn(rng, int(a), int(b), int(c)) for a, b, c in zip(rng.integers(1, 5, 5), rng.integers(1, 4, 5), rng.integers(2, 6, 5))],
    }
    for nd, xs in lists.items():
        fn = {1: o_pad.Padding1D, 2: o_pad.Padding2D, 3: o_pad.Padding3D}[nd]
        for value in (0, -3):
            new_x, mask, shapes = fn([T_(x) for x in xs], value=value)
            out[f"d{nd}_v{value}_new_x"], out[f"d{nd}_v{value}_mask"] = new_x.numpy(), mask.numpy()
            mx, mm, ms = R.pad([T_(x) for x in xs], value)
            assert torch.equal(mx, new_x) and torch.equal(mm, mask) and [tuple(s) for s in shapes] == ms
        out[f"d{nd}_n"] = np.array(len(xs))
        out[f"d{nd}_shapes"] = np.array([x.shape for x in xs], dtype=np.int64)
        for j, x in enumerate(xs):
            out[f"d{nd}_x{j}"] = x
        un = {1: o_pad.UnPadding1D, 2: o_pad.UnPadding2D, 3: o_pad.UnPadding3D}[nd](new_x, shapes)
        for x, u in zip(xs, un):
            assert np.array_equal(x, u.numpy())
    # group split DP (oracle mode)
    numel_lists = [sorted(int(v) for v in rng.integers(1, 200, n)) for n in (1, 2, 7, 25, 60)]
    groups = [1, 2, 3, 4, 8]
    res = []
    for nl, g in zip(numel_lists, groups):
        xs = [torch.zeros(n) for n in nl]
        _, pos = o_pad.oracle_split_group(xs, g)
        assert pos == R.oracle_split_group(nl, g), (pos, R.oracle_split_group(nl, g))
        res.append((nl, g, pos))
    for j, (nl, g, pos) in enumerate(res):
        out[f"split{j}_numels"], out[f"split{j}_group"], out[f"split{j}_pos"] = np.array(nl), np.array(g), np.array(pos)
    out["split_n"] = np.array(len(res))
    save("padding", **out)


# ---------------------------------------------------------------- ScatterConnection
def gen_scatter():
    out = {}
    cases = [(3, 20, 5, 4, 4, 1), (2, 70, 3, 8, 5, 2), (1, 4, 66, 9, 9, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (B, M, N, H, W, seed) in enumerate(cases):
        rng = np.random.default_rng(1000 + seed)
        x = rn(rng, B, M, N)
        loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
        out[f"c{i}_x"], out[f"c{i}_location"] = x, loc
        for st in ("add", "cover"):
            tx = T_(x, True)
            o = o_sc.ScatterConnection(st)(tx, (H, W), T_(loc))
            (o * o).mean().backward()
            out[f"c{i}_{st}_out"], out[f"c{i}_{st}_grad_x"] = o.detach().numpy(), tx.grad.numpy()
            x2 = T_(x, True)
            o2 = R.scatter_connection(x2, T_(loc), H, W, st)
            (o2 * o2).mean().backward()
            if st == "cover":
                assert torch.equal(o2.detach(), o.detach()), "cover must be bit exact"
            else:
                close("scatter add", o.detach().numpy(), o2.detach().numpy(), 1e-6)
            gclose("scatter grad " + st, tx.grad.numpy(), x2.grad.numpy(), 1e-6)
    save("scatter", **out)


# ---------------------------------------------------------------- LSTM
def gen_lstm():
    out = {}
    cases = [(5, 3, 8, 6, 2, 1), (4, 2, 5, 4, 1, 2), (3, 5, 12, 16, 3, 3)]
    out["cases"] = np.array(cases, dtype=np.float64)
    for i, (S, B, I, H, L, seed) in enumerate(cases):
        torch.manual_seed(seed)
        rng = np.random.default_rng(1100 + seed)
        m = o_rnn.get_lstm("normal", I, H, L, "LN", dropout=0.0)
        # randomise LN affine params so gamma/beta gradients are exercised
        with torch.no_grad():
            for ln in m.norm:
                ln.weight.copy_(T_(1.0 + 0.1 * rn(rng, 4 * H)))
                ln.bias.copy_(T_(0.1 * rn(rng, 4 * H)))
        x, h0, c0 = rn(rng, S, B, I), rn(rng, L, B, H), rn(rng, L, B, H)
        tx, th0, tc0 = T_(x, True), T_(h0, True), T_(c0, True)
        y, (hn, cn) = m(tx, (th0, tc0), list_next_state=False)
        gy, gh, gc = rn(rng, S, B, H), rn(rng, L, B, H), rn(rng, L, B, H)
        ((y * T_(gy)).sum() + (hn * T_(gh)).sum() + (cn * T_(gc)).sum()).backward()
        out[f"c{i}_x"], out[f"c{i}_h0"], out[f"c{i}_c0"] = x, h0, c0
        out[f"c{i}_gy"], out[f"c{i}_gh"], out[f"c{i}_gc"] = gy, gh, gc
        out[f"c{i}_y"], out[f"c{i}_hn"], out[f"c{i}_cn"] = y.detach().numpy(), hn.detach().numpy(), cn.detach().numpy()
        out[f"c{i}_grad_x"], out[f"c{i}_grad_h0"], out[f"c{i}_grad_c0"] = tx.grad.numpy(), th0.grad.numpy(), tc0.grad.numpy()
        gamma = np.stack([np.concatenate([m.norm[2 * l].weight.detach().numpy(), m.norm[2 * l + 1].weight.detach().numpy()]) for l in range(L)])
        beta = np.stack([np.concatenate([m.norm[2 * l].bias.detach().numpy(), m.norm[2 * l + 1].bias.detach().numpy()]) for l in range(L)])
        ggamma = np.stack([np.concatenate([m.norm[2 * l].weight.grad.numpy(), m.norm[2 * l + 1].weight.grad.numpy()]) for l in range(L)])
        gbeta = np.stack([np.concatenate([m.norm[2 * l].bias.grad.numpy(), m.norm[2 * l + 1].bias.grad.numpy()]) for l in range(L)])
        out[f"c{i}_ln_gamma"], out[f"c{i}_ln_beta"], out[f"c{i}_grad_ln_gamma"], out[f"c{i}_grad_ln_beta"] = gamma, beta, ggamma, gbeta
        out[f"c{i}_bias"], out[f"c{i}_grad_bias"] = m.bias.detach().numpy(), m.bias.grad.numpy()
        for l in range(L):
            out[f"c{i}_wx{l}"], out[f"c{i}_wh{l}"] = m.wx[l].detach().numpy(), m.wh[l].detach().numpy()
            out[f"c{i}_grad_wx{l}"], out[f"c{i}_grad_wh{l}"] = m.wx[l].grad.numpy(), m.wh[l].grad.numpy()
        # restatement check in fp64
        wx = [d64(m.wx[l].detach().numpy()) for l in range(L)]
        wh = [d64(m.wh[l].detach().numpy()) for l in range(L)]
        y2, h2, c2 = R.lstm(d64(x), d64(h0), d64(c0), wx, wh, d64(m.bias.detach().numpy()), d64(gamma), d64(beta))
        close("lstm y", y.detach().numpy(), y2.numpy(), 1e-5)
        close("lstm h", hn.detach().numpy(), h2.numpy(), 1e-5)
        close("lstm c", cn.detach().numpy(), c2.numpy(), 1e-5)
    save("lstm", **out)


def _reference_test_snippets():
    """Pull the reference TEST's own oracle expressions for the hpc_models helpers out of
    /root/reference/tests/test_actor_critic.py with `ast` (the module itself imports the CUDA extension, so it cannot
    be imported here): the function torch_update_ae (:23-26) and the statements of actor_critic_pre_sample_val that
    compute ori_out from ori_x / ori_key / ori_mask (:260-265).  They are executed from the reference tree at fixture
    generation time; nothing of them is stored in this repository."""
    import ast
    path = os.path.join(REF, "tests", "test_actor_critic.py")
    tree = ast.parse(open(path).read(), path)
    ns = {"torch": torch}
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "torch_update_ae")
    exec(compile(ast.Module(body=[fn], type_ignores=[]), path, "exec"), ns)
    val = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "actor_critic_pre_sample_val")
    stmts = [st for st in val.body if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name)
             and st.targets[0].id in ("ori_queries", "ori_query_result", "ori_step_logits", "ori_out")]
    assert [st.targets[0].id for st in stmts] == ["ori_queries", "ori_query_result", "ori_step_logits", "ori_step_logits",
                                                  "ori_step_logits", "ori_out"], "reference test changed"
    pre_sample_code = compile(ast.Module(body=stmts, type_ignores=[]), path, "exec")

    def pre_sample(ori_x, ori_key, ori_mask):
        env = {"ori_x": ori_x, "ori_key": ori_key, "ori_mask": ori_mask}
        exec(pre_sample_code, {"torch": torch}, env)
        return env["ori_out"]
    return ns["torch_update_ae"], pre_sample


def gen_models():
    """hpc_models: the reference has no `origin` module for these three helpers; its test validates
    against inline torch expressions and torch.nn.LSTM (tests/test_actor_critic.py:23-26,124-154,259-264).  The fixtures
    record exactly those expressions' outputs."""
    out = {}
    update_ae, pre_sample = _reference_test_snippets()
    # update_ae: (B, E, D, seed)  (the reference test's own shape, B=8 E=512 D=256, runs in tests/test_models_gpu.py)
    cases = [(8, 64, 48, 1), (3, 7, 5, 2), (17, 33, 100, 3)]
    out["ae_cases"] = np.array(cases, dtype=np.int64)
    for i, (B, E, D, seed) in enumerate(cases):
        rng = np.random.default_rng(1200 + seed)
        ae, key = rn(rng, B, D), rn(rng, B, E, D)
        num = rng.integers(max(E - 2, 1), E, B).astype(np.int64)
        sample = np.array([rng.integers(0, n + 1) for n in num], dtype=np.int64)       # may equal entity_num ("end")
        end = sample == num
        ref = update_ae(T_(ae), T_(key), T_(np.minimum(sample, E - 1)), T_(num), T_(end))
        out.update({f"ae{i}_ae": ae, f"ae{i}_key": key, f"ae{i}_num": num, f"ae{i}_sample": sample,
                    f"ae{i}_out": ref.numpy()})
    # lstm_activation: (B, I, H, seed) against torch.nn.LSTM (:124-154)
    cases = [(8, 32, 32, 1), (5, 12, 70, 2), (16, 48, 256, 3)]
    out["act_cases"] = np.array(cases, dtype=np.int64)
    for i, (B, I, H, seed) in enumerate(cases):
        torch.manual_seed(1300 + seed)
        lstm = torch.nn.LSTM(I, H, 1)
        rng = np.random.default_rng(1300 + seed)
        x, h0, c0 = T_(rn(rng, 1, B, I)), T_(rn(rng, 1, B, H)), T_(rn(rng, 1, B, H))
        with torch.no_grad():
            _, (hn, cn) = lstm.forward(x, (h0, c0))
            ih = torch.matmul(x[0], lstm.weight_ih_l0.transpose(0, 1))
            hh = torch.matmul(h0[0], lstm.weight_hh_l0.transpose(0, 1))
            bias = lstm.bias_ih_l0 + lstm.bias_hh_l0
        out.update({f"act{i}_ih": ih.numpy(), f"act{i}_hh": hh.numpy(), f"act{i}_bias": bias.numpy(),
                    f"act{i}_c0": c0[0].numpy(), f"act{i}_hn": hn[0].numpy(), f"act{i}_cn": cn[0].numpy()})
    # pre_sample: (B, E, H, seed)
    cases = [(8, 64, 32, 1), (3, 9, 100, 2), (16, 50, 96, 3)]
    out["pre_cases"] = np.array(cases, dtype=np.int64)
    for i, (B, E, H, seed) in enumerate(cases):
        rng = np.random.default_rng(1400 + seed)
        x, key = rn(rng, 1, B, H), rn(rng, B, E, H)
        mask = rng.random((B, E)) < 0.8
        ref = pre_sample(T_(x), T_(key), T_(mask))
        out.update({f"pre{i}_x": x, f"pre{i}_key": key, f"pre{i}_mask": mask, f"pre{i}_out": ref.numpy()})
    save("models", **out)


def write_api_surface():
    """tests/golden/api_surface.json: the ast-parsed public surface of the reference's hpc_rll modules (what
    tests/test_api_surface.py compares the drop-in package against).  Written ONLY here."""
    import json
    sys.path.insert(0, os.path.join(HERE, ".."))
    import test_api_surface as A
    json.dump(A.reference_surface(), open(A.SNAP, "w"), indent=1, sort_keys=True)
    print("wrote", A.SNAP)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    gens = dict(gae=gen_gae, td_lambda=gen_td_lambda, vtrace=gen_vtrace, upgo=gen_upgo, ppo=gen_ppo, qntd=gen_qntd,
                dntd=gen_dntd, iqn=gen_iqn, qrdqn=gen_qrdqn, padding=gen_padding, scatter=gen_scatter, lstm=gen_lstm,
                models=gen_models)
    for k, fn in gens.items():
        if not only or k in only:
            fn()
    if not only or "api_surface" in only:
        write_api_surface()
    print("all golden fixtures written and the oracle restatement agrees with hpc_rll.origin")
