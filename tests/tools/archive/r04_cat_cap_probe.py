#!/usr/bin/env python3
"""The categorical row kernels by grid cap (tune key 0: 24 = rounds 1-3, 1024 = round 4) over row widths that take different
kernels (small-N, one-DPP-row, whole-wave, row-per-workgroup, LDS-staged row), ~2 GB of logits each, through UPGO."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils.upgo import UPGO  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=3, rounds=3):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return statistics.median(ts)


for N in (6, 18, 64, 128, 256, 1000, 1024, 4096, 5000):
    rows = max(4096, (1 << 29) // N)               # ~2 GB of logits
    T = 64
    B = rows // T
    g = torch.Generator(device=dev).manual_seed(0)
    logits = torch.randn(T, B, N, device=dev, generator=g, requires_grad=True)
    rho = torch.rand(T, B, device=dev, generator=g)
    a = torch.randint(0, N, (T, B), device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    v = torch.randn(T + 1, B, device=dev, generator=g)
    m = UPGO(T, B, N)
    out = {}
    for key in (24, 1024, 24, 1024):
        U.tune_set(0, key)
        loss = m(logits, rho, a, r, v)
        t_f = timed(lambda: m(logits, rho, a, r, v))

        def bwd():
            logits.grad = None
            loss.backward(retain_graph=True)
        t_b = timed(bwd)
        out.setdefault(key, []).append((t_f, t_b))
    U.tune_set(0, 1024)
    gb = T * B * N * 4 / 1e9
    s = f"N={N:5d} rows={T * B:8d} ({gb:.2f} GB):"
    for key in (24, 1024):
        f = min(x[0] for x in out[key]); b = min(x[1] for x in out[key])
        s += f"  cap {key:4d}: fwd {f:.3f} ms ({gb / f:.2f} TB/s) bwd {b:.3f} ms ({2 * gb / b:.2f} TB/s)"
    print(s, flush=True)
    del logits, rho, a, r, v
    torch.cuda.empty_cache()
