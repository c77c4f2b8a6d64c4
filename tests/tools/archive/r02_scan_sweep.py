#!/usr/bin/env python3
"""In-process sweep of tune key 19 (waves a column-scan launch aims for) on the C3 return suite: forward time of
TD-lambda / V-trace / UPGO (HIP events, interleaved rounds, median)."""
import json
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_rl_utils as U  # noqa: E402
from hpc_rll.rl_utils.td import TDLambda  # noqa: E402
from hpc_rll.rl_utils.upgo import UPGO  # noqa: E402
from hpc_rll.rl_utils.vtrace import VTrace  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for T, B, N in ((256, 16384, 128), (256, 4096, 32), (1024, 1024, 16)):
    g = torch.Generator(device=dev).manual_seed(0)
    v = torch.randn(T + 1, B, device=dev, generator=g)
    r = torch.randn(T, B, device=dev, generator=g)
    w = torch.rand(T, B, device=dev, generator=g)
    to = torch.randn(T, B, N, device=dev, generator=g)
    bo = torch.randn(T, B, N, device=dev, generator=g)
    a = torch.randint(0, N, (T, B), device=dev, generator=g)
    rho = torch.rand(T, B, device=dev, generator=g)
    m1, m2, m3 = TDLambda(T, B), VTrace(T, B, N), UPGO(T, B, N)
    res = {}
    with torch.no_grad():
        for rnd in range(4):
            for target in (1024, 2048, 4096, 8192):
                U.tune_set(19, target)
                res.setdefault((target, "td"), []).append(t(lambda: m1(v, r, w)))
                res.setdefault((target, "vtrace"), []).append(t(lambda: m2(to, bo, a, v, r), 5))
                res.setdefault((target, "upgo"), []).append(t(lambda: m3(to, rho, a, r, v), 5))
    U.tune_set(19, 4096)
    print(json.dumps({"shape": f"T={T} B={B} N={N}", **{f"{k[1]}@{k[0]}": round(statistics.median(x), 1) for k, x in res.items()}}), flush=True)
