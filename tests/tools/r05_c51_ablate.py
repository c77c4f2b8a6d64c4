#!/usr/bin/env python3
"""Round 5: what bounds the C51 n-step TD forward at B = 262144, N = 64, 51 atoms?

    HPC_RLL_C51_ABL=<bits> python tests/tools/r05_c51_ablate.py     (one process per setting: the switch is read once)

bits (a build with the temporary switch in csrc/dist_ops.hip): 1 = no projection, 2 = no row gathers, 4 = no buf store,
8 = no log / division, 16 = no per-sample wave sum."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402

import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils.td import DistNStepTD  # noqa: E402

dev = torch.device("cuda:0")
B, N, n_atom, nstep = 1 << 18, 64, 51, 5
g = torch.Generator(device=dev).manual_seed(0)
d = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
nd = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
a = torch.randint(0, N, (B,), device=dev, generator=g)
na = torch.randint(0, N, (B,), device=dev, generator=g)
reward = torch.randn(nstep, B, device=dev, generator=g)
done = (torch.rand(B, device=dev, generator=g) < 0.1).float()
weight = torch.rand(B, device=dev, generator=g)
m = DistNStepTD(nstep, B, N, n_atom)


def timed(fn, n=200, rounds=5):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(out)


with torch.no_grad():
    for sw in (0, 8, 16, 32, 64):
        U.tune_set(24, sw)
        t = timed(lambda: m(d, nd, a, na, reward, done, weight, 0.99, -10.0, 10.0))
        print(f"abl={os.environ.get('HPC_RLL_C51_ABL', '0')} samples/wave={sw:2d}: forward us (sorted rounds)", ["%.1f" % x for x in t])
