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
torch.cuda.synchronize()
print("pmc suite done")
