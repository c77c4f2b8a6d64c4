"""Parity at BASELINE.json's FULL configuration sizes (configs[2..4]); configs[1] (GAE T=1024,B=65536) is in
tests/test_gae_gpu.py.  At these sizes the CPU oracle would take minutes, so the same oracle code
(oracle/ref_torch.py, fp64) is evaluated with PyTorch-ROCm eager ops on the GPU -- an implementation independent of
the HIP kernels -- plus size-independent properties (shard additivity, winner maps, round trips)."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ref_torch as R

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(ref, got):
    ref, got = ref.double(), got.double()
    return ((ref - got).abs() / ref.abs().clamp(min=1.0)).max().item()


def test_c3_return_suite_full_size():
    """configs[2]: V-trace + UPGO + TD-lambda, T=256, B=16384 (N=128 per tests/test_vtrace.py:13)."""
    from hpc_rll.rl_utils.td import TDLambda
    from hpc_rll.rl_utils.upgo import UPGO
    from hpc_rll.rl_utils.vtrace import VTrace
    T, B, N = 256, 16384, 128
    g = torch.Generator(device=DEV).manual_seed(0)
    target = torch.randn(T, B, N, device=DEV, generator=g)
    behaviour = torch.randn(T, B, N, device=DEV, generator=g)
    action = torch.randint(0, N, (T, B), device=DEV, generator=g)
    value = torch.randn(T + 1, B, device=DEV, generator=g)
    reward = torch.randn(T, B, device=DEV, generator=g)
    weight = torch.rand(T, B, device=DEV, generator=g)
    rho = torch.rand(T, B, device=DEV, generator=g)

    # ---- TD-lambda
    v = value.clone().requires_grad_(True)
    loss = TDLambda(T, B)(v, reward, weight, 0.9, 0.8)
    loss.backward()
    v64 = value.double().requires_grad_(True)
    l64 = R.td_lambda_error(v64, reward.double(), weight.double(), 0.9, 0.8)
    l64.backward()
    assert rel_err(l64.item(), loss.item()) < 1e-5
    assert _rel(v64.grad, v.grad) < 2e-5

    # ---- V-trace (losses + both gradients)
    to = target.clone().requires_grad_(True)
    v = value.clone().requires_grad_(True)
    ls = VTrace(T, B, N)(to, behaviour, action, v, reward)
    sum(ls).backward()
    to64 = target.double().requires_grad_(True)
    v64 = value.double().requires_grad_(True)
    l64 = R.vtrace_error(to64, behaviour.double(), action, v64, reward.double(), None)
    sum(l64).backward()
    assert rel_err([x.item() for x in l64], [x.item() for x in ls]) < 1e-5
    assert _rel(v64.grad, v.grad) < 2e-5
    assert _rel(to64.grad, to.grad) < 2e-5
    del to64, l64

    # ---- UPGO
    to = target.clone().requires_grad_(True)
    loss = UPGO(T, B, N)(to, rho, action, reward, value)
    loss.backward()
    to64 = target.double().requires_grad_(True)
    l64 = R.upgo_loss(to64, rho.double(), action, reward.double(), value.double())
    l64.backward()
    # the data-dependent lambda compares fp32 sums; with 4M comparisons a handful can flip between fp32 and fp64, each
    # moving one return by O(1): compare the loss at 1e-4 and the gradient on the 99.99% of rows that agree
    assert rel_err(l64.item(), loss.item()) < 1e-4
    row_err = ((to64.grad - to.grad.double()).abs().amax(-1))
    assert (row_err > 1e-9).double().mean().item() < 1e-3


def test_c3_shard_additivity():
    """Size-independent property: per-shard losses scaled by the global count add up to the full-batch loss, and
    per-shard gradients are the corresponding columns (what the 8-GPU data-parallel run relies on)."""
    import hpc_rl_utils as U
    T, B, N = 256, 16384, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    target = torch.randn(T, B, N, device=DEV, generator=g)
    behaviour = torch.randn(T, B, N, device=DEV, generator=g)
    action = torch.randint(0, N, (T, B), device=DEV, generator=g)
    value = torch.randn(T + 1, B, device=DEV, generator=g)
    reward = torch.randn(T, B, device=DEV, generator=g)
    full, ws = torch.empty(3, device=DEV), U.vtrace_workspace(T, B, DEV)
    U.VTraceForward([target, behaviour, action, value, reward, None], [full, ws], 0.99, 0.95, 1.0, 1.0, 1.0)
    parts = torch.zeros(3, device=DEV, dtype=torch.float64)
    R_ = 8
    k = B // R_
    for r in range(R_):
        sl = slice(r * k, (r + 1) * k)
        out, w2 = torch.empty(3, device=DEV), U.vtrace_workspace(T, k, DEV)
        U.VTraceForward([target[:, sl].contiguous(), behaviour[:, sl].contiguous(), action[:, sl].contiguous(),
                         value[:, sl].contiguous(), reward[:, sl].contiguous(), None], [out, w2],
                        0.99, 0.95, 1.0, 1.0, 1.0, scale=1.0 / (T * B))
        parts += out.double()
    assert rel_err(full.cpu().numpy(), parts.cpu().numpy()) < 1e-6


def test_c4_lstm_full_size_forward():
    """configs[3]: LSTM S=128, B=4096, H=1024 (input 1024, L=1).  Forward against the fp64 oracle evaluated on the GPU;
    the full-size backward is covered by the transpose property <y, gy> consistency below at reduced S."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, B, I, H, L = 128, 4096, 1024, 1024, 1
    torch.manual_seed(0)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV)
    h0 = torch.randn(L, B, H, device=DEV)
    c0 = torch.randn(L, B, H, device=DEV)
    with torch.no_grad():
        y, (hn, cn) = m(x, (h0, c0))
        wx = [m.wx.double().reshape(I, 4 * H)]
        wh = [m.wh.double().reshape(H, 4 * H)]
        oy, oh, oc = R.lstm(x.double(), h0.double(), c0.double(), wx, wh, m.bias.double().reshape(L, 4 * H),
                            m.ln_gamma.double(), m.ln_beta.double())
        oy32, _, _ = R.lstm(x, h0, c0, [w.float() for w in wx], [w.float() for w in wh], m.bias.reshape(L, 4 * H),
                            m.ln_gamma, m.ln_beta)
    tol = max(1e-5, 3.0 * _rel(oy, oy32))      # fp32 drift through 128 LayerNorm-recurrent steps (see test_lstm_gpu.py)
    assert _rel(oy, y) < tol
    assert _rel(oh, hn) < tol and _rel(oc, cn) < tol


def test_c4_lstm_large_batch_gradients():
    """Same widths as configs[3] (B=4096, I=H=1024) at S=4 so that fp64 autograd fits comfortably: every gradient."""
    from hpc_rll.torch_utils.network.rnn import LSTM
    S, B, I, H, L = 4, 4096, 1024, 1024, 1
    torch.manual_seed(1)
    m = LSTM(S, B, I, H, L).to(DEV)
    x = torch.randn(S, B, I, device=DEV, requires_grad=True)
    h0 = torch.randn(L, B, H, device=DEV, requires_grad=True)
    c0 = torch.randn(L, B, H, device=DEV, requires_grad=True)
    gy = torch.randn(S, B, H, device=DEV)
    y, (hn, cn) = m(x, (h0, c0))
    ((y * gy).sum() + hn.sum() - cn.sum()).backward()
    leaf = lambda t: t.detach().double().requires_grad_(True)  # noqa: E731
    ox, oh0, oc0 = leaf(x), leaf(h0), leaf(c0)
    owx, owh = leaf(m.wx.reshape(I, 4 * H)), leaf(m.wh.reshape(H, 4 * H))
    ob, og, obe = leaf(m.bias.reshape(L, 4 * H)), leaf(m.ln_gamma), leaf(m.ln_beta)
    oy, ohn, ocn = R.lstm(ox, oh0, oc0, [owx], [owh], ob, og, obe)
    ((oy * gy.double()).sum() + ohn.sum() - ocn.sum()).backward()
    assert _rel(oy.detach(), y) < 1e-5
    for ref, got in ((ox.grad, x.grad), (oh0.grad, h0.grad), (oc0.grad, c0.grad), (owx.grad.reshape(-1), m.wx.grad),
                     (owh.grad.reshape(-1), m.wh.grad), (ob.grad.reshape(-1), m.bias.grad), (og.grad, m.ln_gamma.grad),
                     (obe.grad, m.ln_beta.grad)):
        # weight gradients sum 16384 products of O(1) terms: compare relative to the gradient's own scale
        scale = ref.abs().max().clamp(min=1.0)
        assert ((ref - got.double()).abs().max() / scale).item() < 2e-4


def test_c5_scatter_full_size():
    """configs[4]: ScatterConnection B=4096, M=256 (1,048,576 entities), N=64, H=W=64, both modes.  cover: every cell
    holds exactly the row of the largest entity index located there (winner map built with torch amax), empty cells are
    0; add: equals an fp64 index_add within fp32 rounding; backward is the gather."""
    from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
    B, M, N, H, W = 4096, 256, 64, 64, 64
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, M, N, device=DEV, generator=g, requires_grad=True)
    loc = torch.stack([torch.randint(0, H, (B, M), device=DEV, generator=g),
                       torch.randint(0, W, (B, M), device=DEV, generator=g)], -1)
    cell = loc[..., 0] * W + loc[..., 1]                                   # (B,M)
    ar = torch.arange(M, device=DEV).expand(B, M)
    win = torch.full((B, H * W), -1, dtype=torch.long, device=DEV).scatter_reduce(1, cell, ar, "amax")
    out = ScatterConnection(B, M, N, H, W, "cover")(x, loc)
    got = out.permute(0, 2, 3, 1).reshape(B, H * W, N)
    exp = torch.where((win >= 0).unsqueeze(-1), x.detach()[torch.arange(B, device=DEV).unsqueeze(1), win.clamp(min=0)],
                      torch.zeros((), device=DEV))
    assert torch.equal(got, exp)
    go = torch.randn(B, N, H, W, device=DEV, generator=g)
    out.backward(go)
    gexp = go.permute(0, 2, 3, 1).reshape(B, H * W, N)[torch.arange(B, device=DEV).unsqueeze(1), cell]
    assert torch.equal(x.grad, gexp)
    out_add = ScatterConnection(B, M, N, H, W, "add")(x.detach(), loc)
    ref = torch.zeros(B, H * W, N, device=DEV, dtype=torch.float64).scatter_add_(
        1, cell.unsqueeze(-1).expand(B, M, N), x.detach().double())
    assert _rel(ref, out_add.permute(0, 2, 3, 1).reshape(B, H * W, N)) < 1e-6


def test_c5_pad_round_trip_one_million_elements():
    """configs[4]: Pad1D/Unpad over a large ragged set (n = 262,144 tensors, len ~ U[32,128)): round trip bit exact,
    mask population = total length."""
    from hpc_rll.rl_utils import padding as P
    n = 1 << 18
    lens = np.random.default_rng(0).integers(32, 128, n)
    flat = torch.randn(int(lens.sum()), device=DEV)
    xs = list(torch.split(flat, [int(v) for v in lens]))
    new_x, mask, shapes = P.Padding1D(xs)
    assert new_x.shape == (n, int(lens.max())) and int(mask.sum()) == int(lens.sum())
    assert torch.equal(torch.cat(P.UnPadding1D(new_x, shapes)), flat)
