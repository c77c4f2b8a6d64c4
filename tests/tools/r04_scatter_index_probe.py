#!/usr/bin/env python3
"""Scatter-connection forward at C5 (B = 4096, M = 256, N = 64, 64 x 64 maps) and at long entity lists (M = 1024, 4096: several
chunks per workgroup in the index build), API-level, eager."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
from hpc_rll.torch_utils.network.scatter_connection import ScatterConnection  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=10, rounds=5):
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


g = torch.Generator(device=dev).manual_seed(0)
for B, M, N, H, W in ((4096, 256, 64, 64, 64), (4096, 128, 64, 32, 32), (256, 256, 256, 16, 16), (1024, 1024, 64, 64, 64), (256, 4096, 32, 64, 64)):
    x = torch.randn(B, M, N, device=dev, generator=g)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g), torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    for kind in ("cover", "add"):
        m = ScatterConnection(B, M, N, H, W, kind)
        by = 4 * x.numel() + 4 * B * N * H * W
        import hpc_torch_utils_network as NW
        for key in ((0, 1, 0, 1) if kind == 'cover' else (0, 1, 0, 1)):
            NW.tune_set(37, key)
            with torch.no_grad():
                t = timed(lambda: m(x, loc))
            print(f"B={B} M={M} N={N} {kind:5s} in-kernel index (key 37) = {key}: forward {t:8.1f} us  {by / t / 1e3:6.0f} GB/s", flush=True)
