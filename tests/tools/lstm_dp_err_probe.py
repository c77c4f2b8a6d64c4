"""Full-batch vs two half-batch LayerNorm-LSTM weight gradients against the fp64 oracle (the shapes of
tests/test_dist.py::test_gpu_two_ranks_match_single_process), per parameter."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err
from oracle import ref_torch as R
from hpc_rll.torch_utils.network.rnn import LSTM
dev = torch.device("cuda:0")
S, LB, I, H, L = 6, 8, 10, 12, 2
torch.manual_seed(11)
m = LSTM(S, LB, I, H, L).to(dev)
x = torch.randn(S, LB, I, generator=torch.Generator().manual_seed(5))


def run(mod, xs):
    for p in mod.parameters():
        p.grad = None
    xs = xs.to(dev).requires_grad_(True)
    y, _ = mod(xs, None)
    y.sum().backward()
    return {n: p.grad.double().cpu().numpy() for n, p in mod.named_parameters()}, xs.grad.double().cpu().numpy()

full, gx = run(m, x)
half = None
for r in range(2):
    mh = LSTM(S, LB // 2, I, H, L).to(dev)
    mh.load_state_dict(m.state_dict())
    g, _ = run(mh, x[:, r * 4:(r + 1) * 4])
    half = g if half is None else {k: half[k] + g[k] for k in g}

dt = torch.float64
P = {n: p.detach().cpu().to(dt) for n, p in m.named_parameters()}
dims = [I] + [H] * L
wx, off = [], 0
for l in range(L):
    n = dims[l] * 4 * H
    wx.append(P["wx"][off:off + n].reshape(dims[l], 4 * H).clone().requires_grad_(True)); off += n
wh = [P["wh"].reshape(L, H, 4 * H)[l].clone().requires_grad_(True) for l in range(L)]
b = P["bias"].reshape(L, 4 * H).clone().requires_grad_(True)
ga = P["ln_gamma"].reshape(L, 8 * H).clone().requires_grad_(True)
be = P["ln_beta"].reshape(L, 8 * H).clone().requires_grad_(True)
z = torch.zeros(L, LB, H, dtype=dt)
oy, _, _ = R.lstm(x.to(dt), z, z.clone(), wx, wh, b, ga, be)
oy.sum().backward()
ora = {"wx": torch.cat([w.grad.reshape(-1) for w in wx]).numpy(), "wh": torch.cat([w.grad.reshape(-1) for w in wh]).numpy(),
       "bias": b.grad.reshape(-1).numpy(), "ln_gamma": ga.grad.reshape(-1).numpy(), "ln_beta": be.grad.reshape(-1).numpy()}
for k in ora:
    f, h = full[k].reshape(-1), half[k].reshape(-1)
    print(f"{k:9s} |g|max {np.abs(ora[k]).max():9.3f}  full-vs-f64 {rel_err(ora[k], f):.2e}  halves-vs-f64 {rel_err(ora[k], h):.2e}  "
          f"full-vs-halves {rel_err(f, h):.2e}")
