#!/usr/bin/env python3
"""The loss finalisation of the n-step TD forwards at the suite's shapes (grids of 1024 ... 8192 workgroups): a separate finalize
launch (tune key 21 = 1: the fold stops at 512 workgroups) against the counter tree (key 21 = 2).  API-level forward times, eager and
hipGraph replay, alternating in one process."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils.td import DistNStepTD, IQNNStepTDError, QNStepTD, QRDQNNStepTDError  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
B, N, nstep, tau, n_atom = 262144, 64, 5, 32, 51
a, na = torch.randint(0, N, (B,), device=dev, generator=g), torch.randint(0, N, (B,), device=dev, generator=g)
r, done, w = torch.randn(nstep, B, device=dev, generator=g), (torch.rand(B, device=dev, generator=g) < 0.1).float(), torch.rand(B, device=dev, generator=g)
q, nq = torch.randn(B, N, device=dev, generator=g), torch.randn(B, N, device=dev, generator=g)
d = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
nd = torch.softmax(torch.randn(B, N, n_atom, device=dev, generator=g), -1)
Bi = B // 4
qi, nqi, rq = torch.randn(tau, Bi, N, device=dev, generator=g), torch.randn(tau, Bi, N, device=dev, generator=g), torch.rand(tau, Bi, device=dev, generator=g)
ai, nai, ri, di, wi = a[:Bi].contiguous(), na[:Bi].contiguous(), r[:, :Bi].contiguous(), done[:Bi].contiguous(), w[:Bi].contiguous()
qq, nqq = torch.randn(B, N, tau, device=dev, generator=g), torch.randn(B, N, tau, device=dev, generator=g)
m1, m2, m3, m4 = QNStepTD(nstep, B, N), DistNStepTD(nstep, B, N, n_atom), IQNNStepTDError(tau, tau, nstep, Bi, N), QRDQNNStepTDError(tau, nstep, B, N)
ops = {"q_nstep_td": lambda: m1(q, nq, a, na, r, done, w, 0.99),
       "dist_nstep_td": lambda: m2(d, nd, a, na, r, done, w, 0.99, -10.0, 10.0),
       "iqn_nstep_td": lambda: m3(qi, nqi, ai, nai, ri, di, rq, 0.99, 1.0, wi),
       "qrdqn_nstep_td": lambda: m4(qq, nqq, a, na, r, done, 0.99, w)}


def timed(fn, n=20, rounds=7):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


def graphed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s), torch.no_grad():
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            fn()
    return gr


with torch.no_grad():
    for name, fn in ops.items():
        for key in (1, 2, 1, 2):
            U.tune_set(21, key)
            loss = float(fn()[0])
            e = timed(fn)
            gr = graphed(fn)
            k = timed(gr.replay)
            print(f"{name:16s} key21={key}: eager {e:7.1f} us   graph replay {k:7.1f} us   loss {loss!r}", flush=True)
U.tune_set(21, 1)
