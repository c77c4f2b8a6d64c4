#!/usr/bin/env python3
"""Workload for PMC passes over the secondary kernels: one launch of each op's forward+backward at the C3 / C5 shapes
and one fp32 GEMM (run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` separately)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch
dev = torch.device("cuda:0")
from hpc_rll.rl_utils.td import TDLambda
from hpc_rll.rl_utils.vtrace import VTrace
from hpc_rll.rl_utils.upgo import UPGO
from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection
import hpc_torch_utils_network as U
g = torch.Generator(device=dev).manual_seed(0)
T, B, N = 256, 16384, 128
value = torch.randn(T + 1, B, device=dev, generator=g, requires_grad=True)
reward = torch.randn(T, B, device=dev, generator=g)
target = torch.randn(T, B, N, device=dev, generator=g, requires_grad=True)
behaviour = torch.randn(T, B, N, device=dev, generator=g)
action = torch.randint(0, N, (T, B), device=dev, generator=g)
rho = torch.rand(T, B, device=dev, generator=g)
copy_dst = torch.empty_like(behaviour); copy_dst.copy_(behaviour)       # calibration: 2.147 GB read + written
TDLambda(T, B)(value, reward).backward()
sum(VTrace(T, B, N)(target, behaviour, action, value, reward)).backward()
UPGO(T, B, N)(target, rho, action, reward, value.detach()).backward()
Bs, M, Nn, H, W = 4096, 256, 64, 64, 64
x = torch.randn(Bs, M, Nn, device=dev, generator=g, requires_grad=True)
loc = torch.stack([torch.randint(0, H, (Bs, M), device=dev, generator=g), torch.randint(0, W, (Bs, M), device=dev, generator=g)], -1)
for st in ("cover", "add"):
    o = ScatterConnection(Bs, M, Nn, H, W, st)(x, loc)
    o.backward(torch.ones_like(o))
a, b = torch.randn(8192, 1024, device=dev), torch.randn(1024, 4096, device=dev)
U.gemm_f32(a, b)
del o, x, target, behaviour, copy_dst, a, b
# round 2: large-batch LSTM cells (gates recomputed, row-walking backward) and the TD family at B = 262144
from hpc_rll.torch_utils.network.rnn import LSTM
S, Bl, Hl = 4, 4096, 1024
m = LSTM(S, Bl, Hl, Hl, 1).to(dev)
xl = torch.randn(S, Bl, Hl, device=dev, generator=g, requires_grad=True)
y, _ = m(xl, None)
y.backward(torch.ones_like(y))
del m, xl, y
from hpc_rll.rl_utils.td import DistNStepTD, IQNNStepTDError, QRDQNNStepTDError
Bt, Nt, nstep, n_atom, tau = 1 << 18, 64, 5, 51, 32
act = lambda: torch.randint(0, Nt, (Bt,), device=dev, generator=g)  # noqa: E731
rew, done, w = torch.randn(nstep, Bt, device=dev, generator=g), (torch.rand(Bt, device=dev, generator=g) < 0.1).float(), torch.rand(Bt, device=dev, generator=g)
at_, nat = act(), act()
d = torch.softmax(torch.randn(Bt, Nt, n_atom, device=dev, generator=g), -1).requires_grad_(True)
nd = torch.softmax(torch.randn(Bt, Nt, n_atom, device=dev, generator=g), -1)
DistNStepTD(nstep, Bt, Nt, n_atom)(d, nd, at_, nat, rew, done, w, 0.99, -10.0, 10.0)[0].backward()
del d, nd
qq = torch.randn(Bt, Nt, tau, device=dev, generator=g, requires_grad=True)
nqq = torch.randn(Bt, Nt, tau, device=dev, generator=g)
QRDQNNStepTDError(tau, nstep, Bt, Nt)(qq, nqq, at_, nat, rew, done, 0.99, w)[0].backward()
del qq, nqq
Bi = Bt // 4
qi = torch.randn(tau, Bi, Nt, device=dev, generator=g, requires_grad=True)
nqi = torch.randn(tau, Bi, Nt, device=dev, generator=g)
IQNNStepTDError(tau, tau, nstep, Bi, Nt)(qi, nqi, at_[:Bi].contiguous(), nat[:Bi].contiguous(), rew[:, :Bi].contiguous(),
                                        done[:Bi].contiguous(), torch.rand(tau, Bi, device=dev, generator=g), 0.99, 1.0,
                                        w[:Bi].contiguous())[0].backward()
del qi, nqi
# round 6: the packed Pad1D over 2^20 ragged rows (C5) and the fused PPO launch
from hpc_rll.rl_utils import padding as P
from hpc_rll.rl_utils.ppo import PPO
import numpy as np
n1m = 1 << 20
lens1m = torch.from_numpy(np.random.default_rng(1).integers(32, 128, n1m)).to(dev)
flat1m = torch.randn(int(lens1m.sum().item()), device=dev)
P.Padding1DPacked(flat1m, lens1m, max_len=127)
del flat1m, lens1m
Bp, Np = 65536, 128
ln = torch.randn(Bp, Np, device=dev, generator=g, requires_grad=True)
lo = torch.randn(Bp, Np, device=dev, generator=g)
ap = torch.randint(0, Np, (Bp,), device=dev, generator=g)
vn = torch.randn(Bp, device=dev, generator=g, requires_grad=True)
vo, advp, retp = (torch.randn(Bp, device=dev, generator=g) for _ in range(3))
sum(PPO(Bp, Np)(ln, lo, ap, vn, vo, advp, retp)[0]).backward()
torch.cuda.synchronize()
print("pmc suite done")
