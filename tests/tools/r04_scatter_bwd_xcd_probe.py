#!/usr/bin/env python3
"""Scatter-connection backward with the channel groups of a batch element on ONE XCD (tune key 38 = 1: their 16-byte pieces of the
same grad_x lines meet in one L2) against launch order (0: eight XCDs), alternating in one process."""
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
for B, M, N, H, W in ((4096, 256, 64, 64, 64), (4096, 128, 64, 32, 32), (2048, 256, 64, 64, 32), (1024, 256, 256, 16, 16), (512, 1024, 128, 64, 64)):
    go = torch.randn(B, N, H, W, device=dev, generator=g)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g), torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    gx = torch.empty(B, M, N, device=dev)
    by = 4 * go.numel() + 4 * gx.numel()
    ref = None
    for key in (0, 1, 2, 0, 1, 2):
        NW.tune_set(38, key)
        t = timed(lambda: cabi.lib.hpc_rll_scatter_connection_backward(go.data_ptr(), loc.data_ptr(), gx.data_ptr(), B, M, N, H, W, s))
        if ref is None:
            ref = gx.clone()
        same = torch.equal(ref, gx)
        print(f"B={B} M={M} N={N} {H}x{W} key38={key}: backward {t:8.1f} us  {by / t / 1e3:6.0f} GB/s  identical={same}", flush=True)
NW.tune_set(38, 1)
