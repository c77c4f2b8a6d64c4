#!/usr/bin/env python3
"""Scatter backward at C5 by LDS per workgroup (tune key 9: planes staged per workgroup) now that the channel groups of a batch
element share an XCD (key 38): was 64 KB still the optimum?"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cabi  # noqa: E402
import hpc_torch_utils_network as NW  # noqa: E402

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
s = torch.cuda.current_stream().cuda_stream
B, M, N, H, W = 4096, 256, 64, 64, 64
go = torch.randn(B, N, H, W, device=dev, generator=g)
loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g), torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
gx = torch.empty(B, M, N, device=dev)
by = 4 * go.numel() + 4 * gx.numel()
for kb in (64, 32, 16, 96, 128, 64):
    for xcd in (1, 2):
        rc1, rc2 = NW.tune_set(9, kb), NW.tune_set(38, xcd)
        t = timed(lambda: cabi.lib.hpc_rll_scatter_connection_backward(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s))
        print(f"LDS {kb:3d} KB per workgroup, key38={xcd}: backward {t:8.1f} us  {by / t / 1e3:6.0f} GB/s", flush=True)
NW.tune_set(9, 64)
NW.tune_set(38, 1)
