#!/usr/bin/env python3
"""In-process A/B of the ScatterConnection forward kernels: the round-1 cells-per-thread kernel
(tune key 17 = 0) against the LDS-staged streaming kernel (key 17 = 1) for several channels-per-workgroup settings
(key 18), at configs[4] (B=4096, M=256, N=64, 64x64) and at the reference's test shape; results must be bit-identical.
Interleaved rounds, median of HIP-event timings on the launch stream."""
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "di-hpc_amd"))
import torch  # noqa: E402
import hpc_torch_utils_network as NW  # noqa: E402

dev = torch.device("cuda:0")


def t(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for (B, M, N, H, W) in ((4096, 256, 64, 64, 64), (256, 256, 256, 16, 16), (1024, 512, 32, 32, 32)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, M, N, device=dev, generator=g)
    loc = torch.stack([torch.randint(0, H, (B, M), device=dev, generator=g),
                       torch.randint(0, W, (B, M), device=dev, generator=g)], -1)
    out = torch.empty(B, N, H, W, device=dev)
    nbytes = 4 * B * M * N + 16 * B * M + 4 * B * N * H * W
    for typ in ("cover", "add"):
        cfgs = [(0, 0), (1, 0), (3, 0), (4, 0), (3, 64), (4, 64), (4, 16)]
        res = {c: [] for c in cfgs}
        ref = None
        for rnd in range(4):
            for c in cfgs:
                NW.tune_set(17, c[0])
                NW.tune_set(18, c[1])
                fn = lambda: NW.ScatterConnectionForward([x, loc], [out], typ)  # noqa: E731
                res[c].append(t(fn))
                if rnd == 0:
                    if ref is None:
                        ref = out.clone()
                    else:
                        assert torch.equal(out, ref), (c, typ)
        NW.tune_set(17, 1)
        NW.tune_set(18, 0)
        row = {"shape": f"B={B} M={M} N={N} {H}x{W}", "type": typ}
        for c in cfgs:
            med = statistics.median(res[c])
            row[f"lds={c[0]} npb={c[1]}"] = {"ms": med * 1e3, "GBs": nbytes / med / 1e9}
        print(json.dumps(row), flush=True)
