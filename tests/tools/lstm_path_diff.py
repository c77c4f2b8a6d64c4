#!/usr/bin/env python3
"""Print the per-tensor differences between the persistent and the step-kernel LSTM paths, and of each against the
fp64 oracle (which path is closer?)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cabi as N  # noqa: E402
from hpc_rll.torch_utils.network.rnn import LSTM  # noqa: E402
from oracle import ref_torch as R  # noqa: E402
from conftest import rel_err  # noqa: E402

DEV = torch.device("cuda:0")
for (S, B, I, H, L) in [(64, 3, 1792, 384, 3), (9, 8, 40, 1024, 2), (7, 5, 12, 1000, 2), (12, 2, 16, 257, 2), (3, 4, 8, 512, 1)]:
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
        return [t.detach().double().cpu().numpy() for t in (y, hn, cn, xs.grad, hs.grad, cs.grad, m.wx.grad, m.wh.grad,
                                                            m.bias.grad, m.ln_gamma.grad, m.ln_beta.grad)]

    N.check(N.lib.hpc_rll_tune_set(3, 0))
    ref = run()
    N.check(N.lib.hpc_rll_tune_set(3, 1))
    got = run()
    # fp64 oracle
    dims = [I] + [H] * L
    offs = np.cumsum([0] + [d * 4 * H for d in dims[:-1]] + [dims[-1] * 4 * H])
    D = lambda t: t.detach().double().cpu().requires_grad_(True)  # noqa: E731
    ox, oh, oc = D(x), D(h0), D(c0)
    wxf = m.wx.detach().double().cpu()
    owx = [wxf[offs[l]:offs[l + 1]].reshape(dims[l], 4 * H).clone().requires_grad_(True) for l in range(L)]
    owh = [w.clone().requires_grad_(True) for w in m.wh.detach().double().cpu().reshape(L, H, 4 * H)]
    ob, og, obe = D(m.bias.reshape(L, 4 * H)), D(m.ln_gamma), D(m.ln_beta)
    oy, ohn, ocn = R.lstm(ox, oh, oc, owx, owh, ob, og, obe)
    ((oy * gy.double().cpu()).sum() + (ohn * gh.double().cpu()).sum() + (ocn * gc.double().cpu()).sum()).backward()
    orc = [oy, ohn, ocn, ox.grad, oh.grad, oc.grad, torch.cat([w.grad.reshape(-1) for w in owx]),
           torch.cat([w.grad.reshape(-1) for w in owh]), ob.grad.reshape(-1), og.grad, obe.grad]
    orc = [t.detach().numpy().reshape(r.shape) for t, r in zip(orc, ref)]
    names = "y hn cn dx dh0 dc0 dwx dwh dbias dgamma dbeta".split()
    print(f"S={S} B={B} I={I} H={H} L={L}")
    for k, a, b, o in zip(names, ref, got, orc):
        print(f"   {k:7s} persist-vs-step {rel_err(a, b):.2e}   step-vs-fp64 {rel_err(o, a):.2e}   persist-vs-fp64 {rel_err(o, b):.2e}")
