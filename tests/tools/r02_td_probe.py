#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats`: TD-lambda / V-trace / UPGO forward+backward at configs[2] and at the
reference's narrow test shape, a few dozen launches each (kernel-level split of scan / finalize / categorical rows)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
from hpc_rll.rl_utils.td import TDLambda  # noqa: E402
from hpc_rll.rl_utils.upgo import UPGO  # noqa: E402
from hpc_rll.rl_utils.vtrace import VTrace  # noqa: E402

dev = torch.device("cuda:0")
for T, B, N in ((256, 16384, 128), (1024, 64, 16)):
    g = torch.Generator(device=dev).manual_seed(0)
    v = torch.randn(T + 1, B, device=dev, generator=g, requires_grad=True)
    r = torch.randn(T, B, device=dev, generator=g)
    w = torch.rand(T, B, device=dev, generator=g)
    to = torch.randn(T, B, N, device=dev, generator=g, requires_grad=True)
    bo = torch.randn(T, B, N, device=dev, generator=g)
    a = torch.randint(0, N, (T, B), device=dev, generator=g)
    rho = torch.rand(T, B, device=dev, generator=g)
    m1, m2, m3 = TDLambda(T, B), VTrace(T, B, N), UPGO(T, B, N)
    for _ in range(20):
        v.grad = None
        to.grad = None
        m1(v, r, w).backward()
        sum(m2(to, bo, a, v, r)).backward()
        m3(to, rho, a, r, v).backward()
    torch.cuda.synchronize()
print("done")
