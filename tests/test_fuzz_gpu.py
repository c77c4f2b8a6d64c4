"""Randomised differential test: every op on ~25 random small shapes each (seeded, so the run is reproducible), HIP vs
the fp64 oracle.  Catches tile-edge / raggedness bugs that hand-picked shapes miss (B not a multiple of the column
tile, T not a multiple of the chunk span, N straddling the row-kernel variants, ...)."""
import numpy as np
import pytest
import torch

from conftest import grad_err, rel_err
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
NCASE = 25


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.requires_grad_(True) if grad else t


def D(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).double()
    return t.requires_grad_(True) if grad else t


def shapes(rng, *ranges):
    out = []
    for _ in range(NCASE):
        out.append(tuple(int(np.exp(rng.uniform(np.log(lo), np.log(hi + 1)))) for lo, hi in ranges))
    return out


def f32(rng, *s):
    return rng.standard_normal(s).astype(np.float32)


def test_fuzz_gae():
    from hpc_rll.rl_utils.gae import GAE
    rng = np.random.default_rng(1)
    for T, B in shapes(rng, (1, 300), (1, 3000)):
        gam, lam = float(rng.uniform(0.8, 1.0)), float(rng.uniform(0.0, 1.0))
        v, r, ga = f32(rng, T + 1, B), f32(rng, T, B), f32(rng, T, B)
        ref = R.gae(D(v), D(r), gam, lam)
        gv, gr = R.gae_backward(D(ga), gam, lam)
        dv, dr = G(v, True), G(r, True)
        adv = GAE(T, B)(dv, dr, gam, lam)
        adv.backward(G(ga))
        assert rel_err(ref.numpy(), adv.detach().cpu().numpy()) < 1e-5, (T, B, gam, lam)
        assert grad_err(gv.numpy(), dv.grad.cpu().numpy()) < 2e-5, (T, B, gam, lam)
        assert grad_err(gr.numpy(), dr.grad.cpu().numpy()) < 2e-5, (T, B, gam, lam)


def test_fuzz_td_lambda():
    from hpc_rll.rl_utils.td import TDLambda
    rng = np.random.default_rng(2)
    for T, B in shapes(rng, (1, 200), (1, 2000)):
        v, r = f32(rng, T + 1, B), f32(rng, T, B)
        mode = int(rng.integers(0, 3))
        w = None if mode == 0 else rng.random((B,) if mode == 1 else (T, B)).astype(np.float32)
        v64 = D(v, True)
        l64 = R.td_lambda_error(v64, D(r), None if w is None else D(w), 0.93, 0.85)
        l64.backward()
        dv = G(v, True)
        loss = TDLambda(T, B)(dv, G(r), None if w is None else G(w), 0.93, 0.85)
        loss.backward()
        assert rel_err(l64.item(), loss.item()) < 1e-5, (T, B, mode)
        assert grad_err(v64.grad.numpy(), dv.grad.cpu().numpy()) < 2e-5, (T, B, mode)


def test_fuzz_vtrace_upgo():
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.vtrace import VTrace
    rng = np.random.default_rng(3)
    for T, B, N in shapes(rng, (1, 60), (1, 200), (1, 300)):
        to, bo = f32(rng, T, B, N), f32(rng, T, B, N)
        a = rng.integers(0, N, (T, B)).astype(np.int64)
        v, r, w = f32(rng, T + 1, B), f32(rng, T, B), rng.random((T, B)).astype(np.float32)
        clips = [float(c) for c in rng.uniform(0.5, 2.0, 3)]
        to64, v64 = D(to, True), D(v, True)
        l64 = R.vtrace_error(to64, D(bo), torch.from_numpy(a), v64, D(r), D(w), 0.99, 0.9, *clips)
        sum(l64).backward()
        dto, dv = G(to, True), G(v, True)
        ls = VTrace(T, B, N)(dto, G(bo), G(a), dv, G(r), G(w), 0.99, 0.9, *clips)
        sum(ls).backward()
        assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < 1e-5, (T, B, N)
        assert grad_err(to64.grad.numpy(), dto.grad.cpu().numpy()) < 2e-5, (T, B, N)
        assert grad_err(v64.grad.numpy(), dv.grad.cpu().numpy()) < 2e-5, (T, B, N)
        # UPGO on the same tensors; skip shapes with a knife-edge lambda comparison (fp32 vs fp64 could disagree)
        margin = np.abs((r[1:] + v[2:]) - v[1:-1]).min() if T > 1 else 1.0
        if margin > 1e-5:
            rho = rng.random((T, B)).astype(np.float32)
            to64 = D(to, True)
            l64 = R.upgo_loss(to64, D(rho), torch.from_numpy(a), D(r), D(v))
            l64.backward()
            dto = G(to, True)
            loss = UPGO(T, B, N)(dto, G(rho), G(a), G(r), G(v))
            loss.backward()
            assert rel_err(l64.item(), loss.item()) < 1e-5, (T, B, N)
            assert grad_err(to64.grad.numpy(), dto.grad.cpu().numpy()) < 2e-5, (T, B, N)


def test_fuzz_ppo():
    from hpc_rll.rl_utils.ppo import PPO
    rng = np.random.default_rng(4)
    for B, N in shapes(rng, (1, 3000), (1, 400)):
        ln = f32(rng, B, N)
        lo = (ln + 0.3 * f32(rng, B, N)).astype(np.float32)
        a = rng.integers(0, N, (B,)).astype(np.int64)
        vn, vo, adv, ret = f32(rng, B), f32(rng, B), f32(rng, B), f32(rng, B)
        dual = [None, 2.0][int(rng.integers(0, 2))]
        uvc = bool(rng.integers(0, 2))
        ln64, vn64 = D(ln, True), D(vn, True)
        l64, i64 = R.ppo_error(ln64, D(lo), torch.from_numpy(a), vn64, D(vo), D(adv), D(ret), None, 0.2, uvc, dual)
        sum(l64).backward()
        dln, dvn = G(ln, True), G(vn, True)
        ls, info = PPO(B, N)(dln, G(lo), G(a), dvn, G(vo), G(adv), G(ret), None, 0.2, uvc, dual)
        sum(ls).backward()
        assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < 1e-5, (B, N)
        assert grad_err(ln64.grad.numpy(), dln.grad.cpu().numpy()) < 2e-5, (B, N)
        assert grad_err(vn64.grad.numpy(), dvn.grad.cpu().numpy()) < 2e-5, (B, N)


def test_fuzz_td_family():
    from hpc_rll.rl_utils.td import QNStepTD, QNStepTDRescale, IQNNStepTDError, QRDQNNStepTDError
    rng = np.random.default_rng(5)
    for T, B, N, tau in shapes(rng, (1, 8), (1, 300), (1, 40), (1, 80)):
        a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
        r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
        q, nq = f32(rng, B, N), f32(rng, B, N)
        for resc, cls in ((False, QNStepTD), (True, QNStepTDRescale)):
            q64 = D(q, True)
            l64, p64 = R.q_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), 0.97, resc)
            l64.backward()
            dq = G(q, True)
            loss, per = cls(T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), G(w), 0.97)
            loss.backward()
            assert rel_err(l64.item(), loss.item()) < 2e-5, (T, B, N, resc)
            assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5
            assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5
        taup = max(1, tau // 2 + 1)
        q3, nq3 = f32(rng, tau, B, N), f32(rng, taup, B, N)
        rq = rng.random((tau, B)).astype(np.float32)
        q64 = D(q3, True)
        l64, p64 = R.iqn_nstep_td_error(q64, D(nq3), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(rq), D(w), 0.97, 0.8)
        l64.backward()
        dq = G(q3, True)
        loss, per = IQNNStepTDError(tau, taup, T, B, N)(dq, G(nq3), G(a), G(na), G(r), G(done), G(rq), 0.97, 0.8, G(w))
        loss.backward()
        assert rel_err(l64.item(), loss.item()) < 2e-5, ("iqn", tau, taup, T, B, N)
        assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5, ("iqn td_err", tau, taup, T, B, N)
        assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5
        q4, nq4 = f32(rng, B, N, tau), f32(rng, B, N, tau)
        q64 = D(q4, True)
        l64, p64 = R.qrdqn_nstep_td_error(q64, D(nq4), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), tau, D(w), 0.97)
        l64.backward()
        dq = G(q4, True)
        loss, per = QRDQNNStepTDError(tau, T, B, N)(dq, G(nq4), G(a), G(na), G(r), G(done), 0.97, G(w))
        loss.backward()
        assert rel_err(l64.item(), loss.item()) < 2e-5, ("qrdqn", tau, T, B, N)
        assert rel_err(p64.detach().numpy(), per.cpu().numpy()) < 2e-5, ("qrdqn td_err", tau, T, B, N)
        assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5


def test_fuzz_dist_nstep_td():
    """C51 over random shapes: n_atom on both sides of 64 (one-pass readlane kernel / the general kernel), row lengths
    N*n_atom that do and do not allow the 16-byte gradient kernel, done = 1 samples (all mass on one atom).  Oracle in
    fp32 (floor / ceil of the projected position is discontinuous), tolerance as the reference-shape test."""
    from hpc_rll.rl_utils.td import DistNStepTD
    rng = np.random.default_rng(9)
    for T, B, N, n_atom in shapes(rng, (1, 6), (1, 200), (1, 24), (2, 90)):
        dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
        nd = np.abs(f32(rng, B, N, n_atom))
        a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
        r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
        d32 = torch.from_numpy(dist).requires_grad_(True)
        l32, p32 = R.dist_nstep_td_error(d32, torch.from_numpy(nd), torch.from_numpy(a), torch.from_numpy(na),
                                         torch.from_numpy(r), torch.from_numpy(done), torch.from_numpy(w), 0.95, -10., 10., n_atom)
        l32.backward()
        dd = G(dist, True)
        loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), 0.95, -10., 10.)
        loss.backward()
        assert rel_err(l32.item(), loss.item()) < 1e-4, (T, B, N, n_atom)
        assert rel_err(p32.detach().numpy(), per.cpu().numpy()) < 1e-4, (T, B, N, n_atom)
        assert grad_err(d32.grad.numpy(), dd.grad.cpu().numpy()) < 1e-4, (T, B, N, n_atom)


def test_fuzz_large_batch_td_kernels():
    """Round 5: the samples-per-wave forwards (QR-DQN: four quantiles per lane / the lane-per-quantile kernel by tau % 4; C51: run
    sums with integral positions outside the runs) over random shapes at the batch sizes that select them, against the small-batch
    kernels (tune key 24 = 1) and -- on a slice -- the oracle.  C51 draws gamma so that runs of one, two, three-to-seven and
    (terminal samples) all atoms occur, value ranges whose atom step is and is not a power of two, returns beyond the support."""
    import hpc_rl_utils as U
    from hpc_rll.rl_utils.td import DistNStepTD, QRDQNNStepTDError
    rng = np.random.default_rng(77)
    n = 1500
    try:
        for case in range(10):
            tau = int(rng.choice([4, 8, 12, 20, 31, 32, 33, 40, 51, 64]))
            B = int(rng.integers(32768, 70000))
            N, T = int(rng.integers(1, 9)), int(rng.integers(1, 5))
            a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
            r, done = f32(rng, T, B), (rng.random(B) < 0.2).astype(np.float32)
            w = rng.random(B).astype(np.float32) if case % 3 else None
            vg = (0.8 + 0.2 * rng.random(B)).astype(np.float32) if case % 2 else None
            q, nq = f32(rng, B, N, tau), f32(rng, B, N, tau)
            res = {}
            for key in (1, 0):
                U.tune_set(24, key)
                dq = G(q, True)
                loss, per = QRDQNNStepTDError(tau, T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), 0.93, None if w is None else G(w),
                                                            None if vg is None else G(vg))
                loss.backward()
                res[key] = (loss.item(), per.detach().cpu(), dq.grad.cpu())
            (l1, p1, g1), (l0, p0, g0) = res[1], res[0]
            assert float((g0 - g1).abs().max()) <= 4e-6 * float(g1.abs().max()), ("qrdqn grad", tau, B, N, T)
            assert float((p0 - p1).abs().max()) <= 4e-6 * float(p1.abs().max()), ("qrdqn td_err", tau, B, N, T)
            assert abs(l0 - l1) <= 4e-6 * abs(l1)
            q64 = D(q[-n:], True)
            l64, p64 = R.qrdqn_nstep_td_error(q64, D(nq[-n:]), torch.from_numpy(a[-n:]), torch.from_numpy(na[-n:]), D(r[:, -n:]),
                                              D(done[-n:]), tau, None if w is None else D(w[-n:]), 0.93, None if vg is None else D(vg[-n:]))
            l64.backward()
            assert rel_err(p64.detach().numpy(), p0[-n:].numpy()) < 2e-5, ("qrdqn oracle", tau, B, N, T)
            assert grad_err(q64.grad.numpy() * (n / B), g0[-n:].numpy()) < 2e-5
        for case in range(10):
            n_atom = int(rng.choice([2, 3, 18, 33, 51, 51, 64]))
            B = int(rng.integers(16384, 50000))
            N, T = int(rng.integers(1, 6)), int(rng.integers(1, 5))
            gamma = float(rng.choice([1.0, 0.99, 0.9, 0.75, 0.6, 0.4]))
            v_min, v_max = [(-10., 10.), (-8., 8.), (0., 25.), (-3.5, 1.25)][case % 4]
            dist = (np.abs(f32(rng, B, N, n_atom)) + 1e-3).astype(np.float32)
            nd = np.abs(f32(rng, B, N, n_atom))
            a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
            r = (f32(rng, T, B) * float(rng.choice([0.5, 2.0, 6.0]))).astype(np.float32)
            done, w = (rng.random(B) < 0.2).astype(np.float32), rng.random(B).astype(np.float32)
            res = {}
            for key in (1, 0):
                U.tune_set(24, key)
                dd = G(dist, True)
                loss, per = DistNStepTD(T, B, N, n_atom)(dd, G(nd), G(a), G(na), G(r), G(done), G(w), gamma, v_min, v_max)
                loss.backward()
                res[key] = (loss.item(), per.detach().cpu(), dd.grad.cpu())
            (l1, p1, g1), (l0, p0, g0) = res[1], res[0]
            assert float((g0 - g1).abs().max()) < 1e-6 * float(g1.abs().max()), ("c51 grad", n_atom, B, N, T, gamma, v_min)
            assert float((p0 - p1).abs().max()) < 2e-6 * float(p1.abs().max()), ("c51 td_err", n_atom, B, N, T, gamma, v_min)
            assert abs(l0 - l1) < 2e-6 * abs(l1)
    finally:
        U.tune_set(24, 0)


def test_fuzz_scatter_and_padding():
    from hpc_rll.rl_utils import padding as P
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    rng = np.random.default_rng(6)
    for B, M, N, H, W in shapes(rng, (1, 20), (1, 120), (1, 70), (1, 40), (1, 40)):
        x = f32(rng, B, M, N)
        loc = np.stack([rng.integers(0, H, (B, M)), rng.integers(0, W, (B, M))], -1).astype(np.int64)
        go = f32(rng, B, N, H, W)
        for st in ("cover", "add"):
            xo = torch.from_numpy(x).requires_grad_(True)
            oo = R.scatter_connection(xo, torch.from_numpy(loc), H, W, st)
            oo.backward(torch.from_numpy(go))
            xd = G(x, True)
            od = ScatterConnection(B, M, N, H, W, st)(xd, G(loc))
            od.backward(G(go))
            assert torch.equal(od.detach().cpu(), oo.detach()), (B, M, N, H, W, st)
            assert torch.equal(xd.grad.cpu(), xo.grad), (B, M, N, H, W, st)
    for rank in (1, 2, 3):
        for _ in range(8):
            n = int(rng.integers(1, 40))
            xs = [torch.from_numpy(f32(rng, *[int(rng.integers(1, 12)) for _ in range(rank)])).to(DEV) for _ in range(n)]
            pad = {1: P.Padding1D, 2: P.Padding2D, 3: P.Padding3D}[rank]
            unpad = {1: P.UnPadding1D, 2: P.UnPadding2D, 3: P.UnPadding3D}[rank]
            value = int(rng.integers(-3, 4))
            new_x, mask, shp = pad(xs, value=value)
            ox, om, _ = R.pad([t.cpu() for t in xs], value)
            assert torch.equal(new_x.cpu(), ox) and torch.equal(mask.cpu(), om.to(torch.int32))
            assert all(torch.equal(a, b) for a, b in zip(xs, unpad(new_x, shp)))


def test_fuzz_scatter_backward_kernels():
    """Round 6: the three backward paths of ScatterConnection -- plane kernel (tune key 40 = 1), spatial-tile kernel wherever it
    applies (2), the rule (0) -- on random shapes (W a multiple of 4 or not: the tile kernel needs 16-byte rows and falls back;
    maps from 1 x 4 to 96 x 96, up to 300 entities, out-of-range locations) against torch's gather: the same bits."""
    import cabi as C
    import hpc_torch_utils_network as NW
    rng = np.random.default_rng(66)
    s = torch.cuda.current_stream().cuda_stream
    try:
        for B, M, N, H, W in shapes(rng, (1, 24), (1, 300), (1, 96), (1, 96), (1, 96)):
            if rng.random() < 0.7:
                W = max(4, W // 4 * 4)
            go = G(f32(rng, B, N, H, W))
            loc = torch.from_numpy(np.stack([rng.integers(-1, H + 1, (B, M)), rng.integers(-1, W + 1, (B, M))], -1).astype(np.int64)).to(DEV)
            ok = (loc[..., 0] >= 0) & (loc[..., 0] < H) & (loc[..., 1] >= 0) & (loc[..., 1] < W)
            want = (go.permute(0, 2, 3, 1)[torch.arange(B, device=DEV)[:, None], loc[..., 0].clamp(0, H - 1), loc[..., 1].clamp(0, W - 1)]
                    * ok[..., None]).cpu()
            for key in (1, 2, 0):
                NW.tune_set(40, key)
                gx = torch.full((B, M, N), float("nan"), device=DEV)
                assert C.lib.hpc_rll_scatter_connection_backward(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s) == 0
                assert torch.equal(gx.cpu(), want), (B, M, N, H, W, key)
    finally:
        NW.tune_set(40, 0)


def test_fuzz_lstm():
    from hpc_rll.torch_utils.network.rnn import LSTM
    rng = np.random.default_rng(7)
    for S, B, I, H, L in shapes(rng, (1, 6), (1, 70), (1, 40), (1, 70), (1, 3))[:12]:
        torch.manual_seed(S * 100 + B)
        m = LSTM(S, B, I, H, L).to(DEV)
        with torch.no_grad():
            m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
            m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
        x, h0, c0 = f32(rng, S, B, I), f32(rng, L, B, H), f32(rng, L, B, H)
        dx = G(x, True)
        y, (hn, cn) = m(dx, (G(h0), G(c0)))
        (y.sum() + hn.sum() * 0.5 - cn.sum()).backward()
        G4 = 4 * H
        off, wx = 0, []
        for l in range(L):
            k = (I if l == 0 else H) * G4
            wx.append(m.wx.detach().cpu().double()[off:off + k].reshape(-1, G4))
            off += k
        wh = [m.wh.detach().cpu().double()[l * H * G4:(l + 1) * H * G4].reshape(H, G4) for l in range(L)]
        ox = D(x, True)
        oy, oh, oc = R.lstm(ox, D(h0), D(c0), wx, wh, m.bias.detach().cpu().double().reshape(L, G4),
                            m.ln_gamma.detach().cpu().double(), m.ln_beta.detach().cpu().double())
        (oy.sum() + oh.sum() * 0.5 - oc.sum()).backward()
        assert rel_err(oy.detach().numpy(), y.detach().cpu().numpy()) < 2e-5, (S, B, I, H, L)
        # relative to the tensor's own maximum.  Shapes with ONE input feature are degenerate: x W_x is rank one, LayerNorm
        # removes its scale, so dx is analytically ~0 and what is left is rounding noise of the cancellation -- the SAME
        # fp64 oracle evaluated in fp32 differs from itself by 8.3e-4 on (2,2,1,2,2); every other
        # shape is held to 3e-5 (fp32-vs-fp64 oracle: <= 3e-5 on all eleven).
        tol = 3e-5 if I >= 2 else 4e-4   # (I = 1: measured 2.05e-4 on the kernel, r03)
        assert grad_err(ox.grad.numpy(), dx.grad.cpu().numpy()) < tol, (S, B, I, H, L)


def test_fuzz_lstm_mid_batch_kernels():
    """Random shapes in the range of the persistent mid-batch kernels (csrc/lstm_mid.hpp: 5 <= B <= 256, H a multiple of 16
    from 64; the backward kernel for B <= 32 by default) against the fp64 oracle: ragged batches, one to three layers, H / 16
    odd and even (k slices), one or two batch streams.  EVERY output and EVERY gradient (dx, dh0, dc0, dWx, dWh, dbias,
    dgamma, dbeta) is compared with the oracle directly (round 5, VERDICT r04 weak #2: most gradients of lstm_mid_bwd_kernel
    were only compared with the step kernels).  The path that ran is asserted, so a silent fallback to the step kernels fails."""
    import hpc_torch_utils_network as NW
    from hpc_rll.torch_utils.network.rnn import LSTM
    rng = np.random.default_rng(11)
    n_mid_bwd = 0
    for it in range(12):
        S, B, I = int(rng.integers(1, 8)), int(rng.integers(5, 201)), int(rng.integers(2, 41))
        if it % 2:
            B = int(rng.integers(5, 33))                      # every other shape inside the backward kernel's default range
        H, L = int(rng.choice([64, 80, 96, 128, 208, 256, 384, 512])), int(rng.integers(1, 4))
        torch.manual_seed(S * 1000 + B)
        m = LSTM(S, B, I, H, L).to(DEV)
        with torch.no_grad():
            m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
            m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
            m.bias.add_(0.1 * torch.randn_like(m.bias))
        x, h0, c0 = f32(rng, S, B, I), f32(rng, L, B, H), f32(rng, L, B, H)
        gy, gh, gc = f32(rng, S, B, H), f32(rng, L, B, H), f32(rng, L, B, H)
        dx, dh0, dc0 = G(x, True), G(h0, True), G(c0, True)
        y, (hn, cn) = m(dx, (dh0, dc0))
        assert NW.lstm_last_forward_path() == 5, (S, B, I, H, L)
        ((y * G(gy)).sum() + (hn * G(gh)).sum() + (cn * G(gc)).sum()).backward()
        assert NW.lstm_last_backward_path() == (5 if B <= 32 else 0), (S, B, I, H, L)
        n_mid_bwd += B <= 32
        G4 = 4 * H
        leaf = lambda t: t.detach().cpu().double().clone().requires_grad_(True)  # noqa: E731
        off, wx = 0, []
        for l in range(L):
            k = (I if l == 0 else H) * G4
            wx.append(leaf(m.wx.detach()[off:off + k].reshape(-1, G4)))
            off += k
        wh = [leaf(m.wh.detach()[l * H * G4:(l + 1) * H * G4].reshape(H, G4)) for l in range(L)]
        ob, og, obe = leaf(m.bias.detach().reshape(L, G4)), leaf(m.ln_gamma), leaf(m.ln_beta)
        ox, oh0, oc0 = D(x, True), D(h0, True), D(c0, True)
        oy, oh, oc = R.lstm(ox, oh0, oc0, wx, wh, ob, og, obe)
        ((oy * D(gy)).sum() + (oh * D(gh)).sum() + (oc * D(gc)).sum()).backward()
        shape = (S, B, I, H, L)
        assert rel_err(oy.detach().numpy(), y.detach().cpu().numpy()) < 2e-5, shape
        assert rel_err(oh.detach().numpy(), hn.detach().cpu().numpy()) < 2e-5, shape
        assert rel_err(oc.detach().numpy(), cn.detach().cpu().numpy()) < 2e-5, shape
        pairs = [("dx", ox.grad, dx.grad), ("dh0", oh0.grad, dh0.grad), ("dc0", oc0.grad, dc0.grad),
                 ("dwx", torch.cat([w.grad.reshape(-1) for w in wx]), m.wx.grad), ("dwh", torch.cat([w.grad.reshape(-1) for w in wh]), m.wh.grad),
                 ("dbias", ob.grad.reshape(-1), m.bias.grad.reshape(-1)), ("dgamma", og.grad, m.ln_gamma.grad), ("dbeta", obe.grad, m.ln_beta.grad)]
        for name, ref, got in pairs:
            # per-tensor scale (see test_lstm_oracle): max |ref - got| / max |ref|
            ref, got = ref.detach().numpy().reshape(-1), got.detach().cpu().double().numpy().reshape(-1)
            err = float(np.max(np.abs(ref - got))) / max(float(np.max(np.abs(ref))), 1e-3)
            assert err < 3e-5, (name, err, shape)
    assert n_mid_bwd >= 6


@pytest.mark.parametrize("B,N,K", [(7, 4, 1), (300, 4, 1), (5, 256, 64), (3, 8, 2), (9, 2, 2), (4, 1024, 16), (6, 6, 2)])
def test_onehot_gradient_row_shapes(B, N, K):
    """The one-hot-shaped gradients (q-TD K = 1, QR-DQN K = tau, IQN tau planes) at the corners of the 16-byte kernel's
    index arithmetic: rows of ONE quad (N*K = 4), rows longer than a workgroup's 4096 quads (one row per workgroup, no
    division), rows that do not allow 16-byte stores (the 4-byte kernel)."""
    from hpc_rll.rl_utils.td import IQNNStepTDError, QNStepTD, QRDQNNStepTDError
    rng = np.random.default_rng(B * 131 + N * 7 + K)
    T = 3
    a, na = rng.integers(0, N, B).astype(np.int64), rng.integers(0, N, B).astype(np.int64)
    r, done, w = f32(rng, T, B), (rng.random(B) < 0.3).astype(np.float32), rng.random(B).astype(np.float32)
    if K == 1:
        q, nq = f32(rng, B, N), f32(rng, B, N)
        q64 = D(q, True)
        l64, _ = R.q_nstep_td_error(q64, D(nq), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(w), 0.97, False)
        l64.backward()
        dq = G(q, True)
        QNStepTD(T, B, N)(dq, G(nq), G(a), G(na), G(r), G(done), G(w), 0.97)[0].backward()
        assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5
    tau = max(K, 2)
    q4, nq4 = f32(rng, B, N, tau), f32(rng, B, N, tau)
    q64 = D(q4, True)
    l64, _ = R.qrdqn_nstep_td_error(q64, D(nq4), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), tau, D(w), 0.97)
    l64.backward()
    dq = G(q4, True)
    QRDQNNStepTDError(tau, T, B, N)(dq, G(nq4), G(a), G(na), G(r), G(done), 0.97, G(w))[0].backward()
    assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5
    q3, nq3, rq = f32(rng, tau, B, N), f32(rng, tau, B, N), rng.random((tau, B)).astype(np.float32)
    q64 = D(q3, True)
    l64, _ = R.iqn_nstep_td_error(q64, D(nq3), torch.from_numpy(a), torch.from_numpy(na), D(r), D(done), D(rq), D(w), 0.97, 0.8)
    l64.backward()
    dq = G(q3, True)
    IQNNStepTDError(tau, tau, T, B, N)(dq, G(nq3), G(a), G(na), G(r), G(done), G(rq), 0.97, 0.8, G(w))[0].backward()
    assert grad_err(q64.grad.numpy(), dq.grad.cpu().numpy()) < 2e-5
